"""Static check of vector-memory waits in one kernel's ISA: every use (read or overwrite) of a register that an outstanding global / buffer / scratch LOAD
will write must sit behind an s_waitcnt vmcnt(N) that has retired that load (loads and stores retire in issue order on gfx9-family parts).
Path-sensitive: every path through the kernel's control-flow graph is walked, memoised on (basic block, queue of outstanding operations).
python tools/r05/vmcnt_check.py file.s kernel_substring"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
kern = sys.argv[2]
start = [i for i, l in enumerate(lines) if l.startswith('_ZN') and kern in l and (l.split(';')[0].rstrip().endswith(':'))][0]
end = start + [i for i, l in enumerate(lines[start:]) if 's_endpgm' in l][0]
prog = []
for l in lines[start + 1:end + 1]:
    t = l.split(';')[0].strip()
    if not t or t.startswith('.') and not t.endswith(':'): continue
    prog.append(t)
def regs(tok):
    out = set()
    for a, b in re.findall(r'\bv\[(\d+):(\d+)\]', tok): out |= set(range(int(a), int(b) + 1))
    for a in re.findall(r'\bv(\d+)\b', tok): out.add(int(a))
    return out
def sregs(tok):
    out = set()
    for a, b in re.findall(r'\bs\[(\d+):(\d+)\]', tok): out |= set(range(int(a), int(b) + 1))
    for a in re.findall(r'\bs(\d+)\b', tok): out.add(int(a))
    return out
def parse(ins):
    m = re.match(r'(\S+)\s*(.*)', ins)
    op, rest = m.group(1), m.group(2)
    ops = [o.strip() for o in rest.split(',')] if rest else []
    return op, ops
VM = ('global_load', 'buffer_load', 'scratch_load', 'flat_load', 'global_store', 'buffer_store', 'scratch_store', 'flat_store', 'global_atomic', 'buffer_atomic')
SM = {}       # scalar loads outstanding, keyed by the state object id (kept apart from the vector queue: a dict on the side, reset per path start)
def scan(seq, queue, tag, report):
    # scalar memory returns OUT OF ORDER: only lgkmcnt(0) retires it.  Entries ride in the same queue object, marked by a negative index
    for idx, ins in seq:
        if ins.endswith(':'): continue
        op, ops = parse(ins)
        if op == 's_waitcnt':
            m = re.search(r'vmcnt\((\d+)\)', ins)
            n = int(m.group(1)) if m else (0 if re.fullmatch(r's_waitcnt\s+0(x0+)?', ins) else None)
            if n is not None:
                vq = [e for e in queue if e[0] >= 0]
                while len(vq) > n: queue.remove(vq.pop(0))
            m2 = re.search(r'lgkmcnt\((\d+)\)', ins)
            n2 = int(m2.group(1)) if m2 else (0 if re.fullmatch(r's_waitcnt\s+0(x0+)?', ins) else None)
            if n2 == 0:
                for e in [e for e in queue if e[0] < 0]: queue.remove(e)
            continue
        sused = set().union(*[sregs(o) for o in ops]) if ops else set()
        for (qi, qins, dst) in queue:
            if qi < 0 and dst & sused:
                report.add((tag + ':smem', -qi - 1, qins[:70], idx, ins[:80]))
        if op.startswith(('s_load', 's_buffer_load')):
            queue.append((-idx - 1, ins, sregs(ops[0])))
            continue
        used = set().union(*[regs(o) for o in ops]) if ops else set()
        for (qi, qins, dst) in queue:
            if qi >= 0 and dst & used:
                report.add((tag, qi, qins[:70], idx, ins[:80]))
        if op.startswith(VM):
            dst = regs(ops[0]) if 'load' in op or ('atomic' in op and 'sc0' in ins) else set()
            queue.append((idx, ins, dst))
    return queue
# path-sensitive: every path through the control-flow graph, memoised on (block, outstanding queue)
seq = list(enumerate(prog))
labels = {ins[:-1]: i for i, ins in seq if ins.endswith(':')}
leaders = {0} | set(labels.values())
for i, ins in seq:
    if not ins.endswith(':') and (parse(ins)[0].startswith('s_cbranch') or parse(ins)[0] in ('s_branch', 's_endpgm', 's_setpc_b64')): leaders.add(i + 1)
leaders = sorted(l for l in leaders if l < len(prog))
block_of = {l: (l, (leaders[k + 1] if k + 1 < len(leaders) else len(prog))) for k, l in enumerate(leaders)}
report, seen, work = set(), set(), [(0, ())]
import sys as _s
_s.setrecursionlimit(10000)
while work:
    b, q0 = work.pop()
    key = (b, tuple(x[0] for x in q0))
    if key in seen: continue
    seen.add(key)
    lo, hi = block_of[b]
    q = scan(seq[lo:hi], list(q0), 'path', report)
    last = prog[hi - 1]
    succ = []
    if not last.endswith(':'):
        op, ops = parse(last)
        if op == 's_endpgm': continue
        if op == 's_branch': succ = [labels[ops[0]]] if ops and ops[0] in labels else []
        elif op.startswith('s_cbranch'): succ = ([labels[ops[0]]] if ops and ops[0] in labels else []) + ([hi] if hi < len(prog) else [])
        else: succ = [hi] if hi < len(prog) else []
    else: succ = [hi] if hi < len(prog) else []
    for t in succ:
        if t in block_of: work.append((t, tuple(q)))
for r in sorted(report, key=lambda r: (r[3], r[1])): print(r)
print('violations:', len(report), ' instructions:', len(prog), ' blocks:', len(leaders), ' states explored:', len(seen), ' waits:', sum(1 for _, x in seq if x.startswith('s_waitcnt')))
