"""Forward+backward of ONE training GABlock and of its IPA core alone (N=16, L=256)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import cases
from ab_opt_amd import training
from ab_opt_amd.modules import GABlock
from ab_opt_amd.utils import synth
dev = torch.device('cuda:0')
N, L = 16, 256
blk = synth.fill_module_(GABlock(128, 64), seed=1).to(dev)
R, t, x, z, mask = [a.to(dev) for a in cases.ipa_inputs(N, L, [L] * N, salt=1)]


def timed(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def block(native):
    xx, zz = x.clone().requires_grad_(True), z.clone().requires_grad_(True)
    training.ga_block(blk, R, t, xx, zz, mask, native=native).sum().backward()
    blk.zero_grad(set_to_none=True)


def block_fwd(native):
    with torch.no_grad():
        training.ga_block(blk, R, t, x, z, mask, native=native)


w_node = torch.cat([blk.proj_query.weight, blk.proj_key.weight, blk.proj_value.weight, blk.proj_query_point.weight, blk.proj_key_point.weight, blk.proj_value_point.weight], 0).detach()


def core():
    proj = (x @ w_node.t()).requires_grad_(True)
    zz = z.clone().requires_grad_(True)
    training.IpaCore.apply(proj, zz, R, t, mask, blk.proj_pair_bias.weight, blk.spatial_coef).sum().backward()
    blk.zero_grad(set_to_none=True)


def core_fwd():
    with torch.no_grad():
        training.IpaCore.apply(x @ w_node.t(), z, R, t, mask, blk.proj_pair_bias.weight, blk.spatial_coef)


print(f'block fwd: native {timed(lambda: block_fwd(True)):.2f} ms, torch {timed(lambda: block_fwd(False)):.2f} ms')
print(f'block fwd+bwd: native {timed(lambda: block(True)):.2f} ms, torch {timed(lambda: block(False)):.2f} ms')
print(f'IPA core alone: fwd {timed(core_fwd):.2f} ms, fwd+bwd {timed(core):.2f} ms')
