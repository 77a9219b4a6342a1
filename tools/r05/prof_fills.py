"""Where do the small ATen launches of a training step come from?  torch.profiler with python stacks over two eager steps (config 5); operators
fill_ / zero_ / copy_ / cat / mm / bmm / where / index grouped by the first frames inside ab_opt_amd/:  python tools/r05/prof_fills.py [N] [L]"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
from ab_opt_amd import training
from ab_opt_amd.utils.synth import build_model, make_batch, LAYOUT_256, LAYOUT_128
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
L = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device('cuda:0')
model = build_model(100, 7, flavour='abdesign', device=dev).train()
batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in make_batch(N, LAYOUT_256 if L == 256 else LAYOUT_128).items()}
opt = training.FusedAdam(model.parameters(), lr=1e-4)
def step():
    opt.zero_grad(set_to_none=True)
    sum(model(dict(batch)).values()).backward()
    opt.step(max_grad_norm=100.0)
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
want = ('aten::fill_', 'aten::zero_', 'aten::copy_', 'aten::cat', 'aten::mm', 'aten::bmm', 'aten::where', 'aten::index', 'aten::add', 'aten::mul', 'aten::sum', 'aten::_to_copy', 'aten::add_', 'aten::clone', 'aten::contiguous')
agg = collections.defaultdict(lambda: [0, 0.0, ''])
for e in prof.events():
    if e.name in want and e.self_device_time_total > 0:
        frames = [f for f in (e.stack or []) if 'ab_opt_amd' in f or 'tools/' in f][:3]
        if not frames:
            frames = [f for f in (e.stack or [])][:3]
        key = (e.name, ' <- '.join(f.split('/')[-1] for f in frames) + ' ' + str(e.input_shapes)[:120])
        a = agg[key]; a[0] += 1; a[1] += e.self_device_time_total; a[2] = str(e.input_shapes)[:90]
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for _, v in rows)
print('ATen launches with device time in one step: %d calls, %.1f us' % (sum(v[0] for _, v in rows), tot))
for (name, where), (n, us, shp) in rows[:70]:
    print('%-16s x%-4d %8.1f us  %s  %s' % (name, n, us, where[:150], shp))
