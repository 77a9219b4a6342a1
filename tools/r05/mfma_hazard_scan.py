"""Scan the ISA of one kernel for instructions that touch the registers of a just-issued bf16 MFMA within a few wait states:
   WAR  a non-MFMA instruction WRITES a register the MFMA reads (SrcA / SrcB / SrcC)
   RAW  a non-MFMA instruction READS the MFMA's destination (incl. a scratch store of it)
   WAW  a non-MFMA instruction WRITES the MFMA's destination
python tools/r05/mfma_hazard_scan.py file.s kernel_substring [window]"""
import re, sys, os
lines = open(sys.argv[1]).read().split('\n')
kern, win = sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 6
start = [i for i, l in enumerate(lines) if l.startswith('_ZN') and kern in l and (l.split(';')[0].rstrip().endswith(':'))][0]
end = start + [i for i, l in enumerate(lines[start:]) if 's_endpgm' in l][0]
body = [l.split(';')[0].strip() for l in lines[start:end]]
body = [l for l in body if l and not l.startswith('.') and not l.endswith(':')]
def regs(tok):
    out = set()
    for a, b in re.findall(r'\bv\[(\d+):(\d+)\]', tok): out |= set(range(int(a), int(b) + 1))
    for a in re.findall(r'\bv(\d+)\b', tok): out.add(int(a))
    return out
def split_ops(ins):
    m = re.match(r'(\S+)\s+(.*)', ins)
    if not m: return ins, []
    return m.group(1), [o.strip() for o in m.group(2).split(',')]
def writes_reads(ins):
    op, ops = split_ops(ins)
    if not ops: return set(), set()
    if op.startswith(('scratch_store', 'global_store', 'buffer_store', 'ds_write', 'ds_store', 's_', 'v_cmp', 'v_cmpx')) :
        return set(), set().union(*[regs(o) for o in ops]) if ops else set()
    w = regs(ops[0]); r = set().union(*[regs(o) for o in ops[1:]]) if len(ops) > 1 else set()
    if op.startswith(('v_fmac', 'v_mac', 'v_accvgpr')): r |= w
    return w, r
found = {'WAR': [], 'RAW': [], 'WAW': []}
for i, ins in enumerate(body):
    if not (ins.startswith('v_mfma_f32_') and (os.environ.get('ANY_MFMA') or 'bf16' in ins.split()[0])): continue
    op, ops = split_ops(ins)
    D, A, B, Cc = regs(ops[0]), regs(ops[1]), regs(ops[2]), regs(ops[3])
    states = 0
    for j in range(i + 1, min(i + 1 + win * 2, len(body))):
        nx = body[j]
        if nx.startswith('s_nop'):
            states += int(nx.split()[1]) + 1; continue
        if states >= win: break
        if nx.startswith('v_mfma'):
            states += 1; continue            # (MFMA -> MFMA dependencies are interlocked / padded by the compiler's tables)
        w, r = writes_reads(nx)
        if w & (A | B): found['WAR'].append((states, ins[:70], nx[:70], 'AB'))
        if w & Cc and not (w & D): found['WAR'].append((states, ins[:70], nx[:70], 'C'))
        if r & D: found['RAW'].append((states, ins[:70], nx[:70], ''))
        if w & D: found['WAW'].append((states, ins[:70], nx[:70], ''))
        states += 1
print('  WAR on SrcC by distance:', {d: sum(1 for x in found['WAR'] if x[0]==d and x[3]=='C') for d in range(win) if sum(1 for x in found['WAR'] if x[0]==d and x[3]=='C')})
for k, v in found.items():
    print(k, len(v), 'instances within', win, 'states; by distance:', {d: sum(1 for x in v if x[0] == d) for d in sorted(set(x[0] for x in v))})
    for x in [y for y in sorted(v) if not (k=='WAR' and y[3]=='AB')][:4]: print('   ', x)
