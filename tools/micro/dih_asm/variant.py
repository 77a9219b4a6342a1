"""python variant.py <name> <regex> [before|after] [nop] : copy dih_kernels.s with `s_nop <nop>` inserted before/after every instruction of dih_kernel<32> (and <0>) that matches the
regex; assembles <name>.hsaco"""
import re, subprocess, sys
name, rx = sys.argv[1], re.compile(sys.argv[2])
where = sys.argv[3] if len(sys.argv) > 3 else 'after'
nop = sys.argv[4] if len(sys.argv) > 4 else '3'
out, inside, n = [], False, 0
for line in open('dih_kernels.s'):
    if line.startswith('_Z10dih_kernelILi'):
        inside = True
    if inside and 's_endpgm' in line:
        inside = False
    hit = inside and line.startswith('\t') and rx.search(line.split(';')[0]) and not line.strip().startswith('.')
    ins = ('\ts_sleep %s\n' % nop[5:]) if nop.startswith('sleep') else ('\ts_nop %s\n' % nop)
    if hit and where == 'before':
        out.append(ins); n += 1
    out.append(line)
    if hit and where == 'after':
        out.append(ins); n += 1
open(name + '.s', 'w').writelines(out)
LL = '/opt/rocm/lib/llvm/bin/'
subprocess.check_call([LL + 'clang', '-x', 'assembler', '-target', 'amdgcn-amd-amdhsa', '-mcpu=gfx950', '-c', name + '.s', '-o', name + '.o'])
subprocess.check_call([LL + 'ld.lld', '-shared', name + '.o', '-o', name + '.hsaco'])
print(name, n, 'insertions')
