"""Which torch operators own the non-native GPU time of a training step?  torch.profiler over two eager steps (config 5), aggregated by
operator and input shapes:  python tools/prof_train_ops.py [N] [L]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
    for _ in range(2):
        step()
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by='self_cuda_time_total', row_limit=45, max_name_column_width=42, max_shapes_column_width=70))
if len(sys.argv) > 3:          # a third argument: only the operators whose name contains it, with their shapes (e.g. copy_)
    rows = [e for e in prof.key_averages(group_by_input_shape=True) if sys.argv[3] in e.key]
    rows.sort(key=lambda e: -e.self_device_time_total)
    for e in rows[:25]:
        print('%-28s %9.1f us  x%-4d %s' % (e.key, e.self_device_time_total, e.count, str(e.input_shapes)[:150]))
if len(sys.argv) > 3 and sys.argv[3] == 'count':     # 'count': operators by number of calls (the ~500 three-microsecond launches of a step), with python source
    rows = sorted((e for e in prof.key_averages(group_by_input_shape=True) if e.self_device_time_total > 0 and e.key.startswith('aten::')), key=lambda e: -e.count)
    for e in rows[:45]:
        print('%-34s x%-5d %9.1f us  %s' % (e.key[:34], e.count, e.self_device_time_total, str(e.input_shapes)[:110]))
