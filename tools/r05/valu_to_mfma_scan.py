"""Scan one kernel's ISA for an MFMA that reads (SrcA / SrcB / SrcC) a VGPR written by a VALU instruction fewer than `win` wait states earlier
(hipcc pads nothing for this pair on gfx950; tools/micro/cvt_trans.hip: v_cvt_pk_f16_f32 -> MFMA SrcB at 0 states reads the OLD value).
python tools/r05/valu_to_mfma_scan.py file.s kernel_substring [win]"""
import re, sys, collections
lines = open(sys.argv[1]).read().split('\n')
kern, win = sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 2
start = [i for i, l in enumerate(lines) if l.startswith('_ZN') and kern in l and (l.split(';')[0].rstrip().endswith(':'))][0]
end = start + [i for i, l in enumerate(lines[start:]) if 's_endpgm' in l][0]
body = [l.split(';')[0].strip() for l in lines[start:end]]
body = [l for l in body if l and not l.startswith('.') and not l.endswith(':')]
def regs(tok):
    out = set()
    for a, b in re.findall(r'\bv\[(\d+):(\d+)\]', tok): out |= set(range(int(a), int(b) + 1))
    for a in re.findall(r'\bv(\d+)\b', tok): out.add(int(a))
    return out
def ops_of(ins):
    m = re.match(r'(\S+)\s+(.*)', ins)
    return (m.group(1), [o.strip() for o in m.group(2).split(',')]) if m else (ins, [])
found = collections.Counter(); examples = {}
for i, ins in enumerate(body):
    if not ins.startswith('v_mfma'): continue
    op, ops = ops_of(ins)
    src = {'A': regs(ops[1]), 'B': regs(ops[2]), 'C': regs(ops[3]) if len(ops) > 3 else set()}
    states = 0
    for j in range(i - 1, max(i - 1 - 4 * win, -1), -1):
        pv = body[j]
        if pv.startswith('s_nop'): states += int(pv.split()[1]) + 1; continue
        if states >= win: break
        pop, pops = ops_of(pv)
        if pop.startswith('v_') and not pop.startswith(('v_mfma', 'v_cmp', 'v_readfirstlane', 'v_readlane')) and pops:
            w = regs(pops[0])
            for nm, rs in src.items():
                if w & rs:
                    key = (states, nm, pop, op.split('_')[3] if op.count('_') > 3 else op)
                    found[key] += 1; examples.setdefault(key, (pv[:60], ins[:80]))
        states += 1
for k, c in sorted(found.items()): print('states %d  Src%s  writer %-22s -> %s : %d   e.g. %s | %s' % (k[0], k[1], k[2], k[3], c, *examples[k]))
print('total', sum(found.values()))
