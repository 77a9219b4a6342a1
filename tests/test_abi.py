import math
"""CPU-side checks of the C-ABI boundary: the library loads, exports every symbol include/abopt.h declares,
and the host layer refuses to run anywhere but on a HIP device (no CPU fallback)."""
import ctypes
import os
import re
import pytest
import torch

from conftest import ROOT, build_model
from ab_opt_amd import hip

HEADER = os.path.join(ROOT, 'include', 'abopt.h')


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(abopt_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(hip.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    L = ctypes.CDLL(hip.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 13
    for s in syms:
        assert hasattr(L, s), f'{s} declared in include/abopt.h but not exported'
    assert set(hip.EXPORTS) == set(syms), 'hip.py binding list and header disagree'
    L.abopt_abi_version.restype = ctypes.c_int
    assert L.abopt_abi_version() == hip.ABI_VERSION
    assert hip.lib() is not None


def test_struct_layouts_match_header_field_order():
    """ctypes mirrors must list the header's fields in order (layout is positional)."""
    src = open(HEADER).read()
    body = src[src.index('typedef struct {', src.index('One GABlock')):src.index('} abopt_ga_weights;')]
    names = re.findall(r'const float\*\s*(\w+);', body)
    assert names == [n for n, _ in hip.GaWeights._fields_]
    body = src[src.index('typedef struct {', src.index('Schedule scalars')):src.index('} abopt_step_params;')]
    body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
    names = re.findall(r'(?:int|float)\s+([\w, ]+?)(?:\[3\])?;', body)
    flat = [x.strip() for n in names for x in n.split(',')]
    assert flat == [n for n, _ in hip.StepParams._fields_]


def test_no_cpu_fallback():
    m = build_model(10, 3)
    from ab_opt_amd.utils import synth
    batch = synth.make_batch(1, synth.LAYOUT_128, seed=1, lengths=[32])
    with pytest.raises(RuntimeError, match='HIP device only'):
        m.sample(batch)
    with pytest.raises(RuntimeError, match='HIP device only'):
        hip.so3_exp(torch.zeros(4, 3))
    # the training entry point too: model(batch) on CPU tensors raises in the binding, there is no torch restatement behind it
    m.train()
    with pytest.raises(RuntimeError, match='HIP device only'):
        m(dict(batch))
    with pytest.raises(RuntimeError, match='HIP device only'), torch.no_grad():
        m(dict(batch))                                  # (a validation pass)
    from ab_opt_amd import training, embed
    with pytest.raises(RuntimeError, match='HIP device only'):
        training._linear(torch.nn.Linear(8, 4), torch.zeros(3, 8))
    assert not hasattr(training, 'NATIVE_IPA') and not hasattr(embed, 'NATIVE_FEATURES')
    import inspect
    src = inspect.getsource(training) + inspect.getsource(embed)
    assert 'F.linear' not in src and 'einsum(\'nijh' not in src          # no second implementation of the network in the package


def test_contig_mask_and_registry():
    from ab_opt_amd.model import generate_mask_from_str, get_model, _MODEL_DICT
    t = torch.zeros(2, 10, dtype=torch.bool)
    m = generate_mask_from_str('3-5', t)
    assert m[:, 2:5].all() and m.sum() == 6
    assert {'diffab_abdock', 'diffab_abdesign'} <= set(_MODEL_DICT)
    assert type(build_model(10, 3)).__name__ == 'DiffusionAntibodyDesign'
    assert type(build_model(10, 3, flavour='abdesign')).__name__ == 'DiffusionAntibodyDesignAbDesign'


def test_bf16_three_term_split_is_exact():
    """hip.split_bf16x3 (the host statement of csrc/ipa_common.h: split3 -- since round 5 the arithmetic of the BACKWARD chains only, csrc/mlp.hip:
    tail_backward_kernel): h + m + l == w bit for bit, every term is a bf16
    number, and the six products the kernel keeps reproduce x * w to 2^-24 (the dropped m*l, l*m, l*l terms).  (Exactness needs the
    residuals to stay normal numbers, i.e. |w| > ~1e-31; below that the error is < 1e-38 absolute.)"""
    g = torch.Generator().manual_seed(5)
    w = torch.cat([torch.randn(4096, generator=g) * s for s in (1e-6, 1e-2, 1.0, 37.0, 1e6)] +
                  [torch.tensor([0.0, -0.0, 1.0, -1.0, 3.0e-30, 1.0 + 2.0 ** -23, -(2.0 - 2.0 ** -23), 65280.0, 1.9999999])])
    terms = [(t.to(torch.int32) << 16).view(torch.float32) for t in hip.split_bf16x3(w)]
    for t in terms:
        assert torch.equal(t.to(torch.bfloat16).float(), t)                      # representable in bf16
    assert torch.equal((terms[0].double() + terms[1].double() + terms[2].double()).float(), w)
    assert torch.equal(terms[0] + (terms[1] + terms[2]), w)
    x = torch.randn(w.numel(), generator=g)
    xt = [(t.to(torch.int32) << 16).view(torch.float32).double() for t in hip.split_bf16x3(x)]
    wt = [t.double() for t in terms]
    kept = xt[2] * wt[0] + xt[0] * wt[2] + xt[1] * wt[1] + xt[1] * wt[0] + xt[0] * wt[1] + xt[0] * wt[0]
    exact = x.double() * w.double()
    rel = ((kept - exact).abs() / exact.abs().clamp_min(1e-300))[exact != 0]
    assert rel.max().item() < 2.0 ** -24


def test_fp16_two_term_split_and_scale():
    """The host statement of the forward dense layers' arithmetic (csrc/ipa_common.h: split_pair2; hip.tail_weight_scale, hip._fp16_terms): the
    scale is a power of two with max |w| S in [2^14, 2^15); h and l are fp16 numbers with |S w - h - l| <= 2^-22 |S w| (+ 2^-25 absolute where l
    is subnormal); the three products the kernels keep reproduce x w to 3 x 2^-22 relative + the subnormal floor
    (2^-25 |w| for activations below 2^-3, whose low term is an fp16 subnormal: the rounding of an fp32 number of size 0.5)."""
    g = torch.Generator().manual_seed(6)
    for scale in (1e-6, 3e-3, 1.0, 900.0):
        w = torch.randn(4096, generator=g) * scale
        S = hip.tail_weight_scale(w)
        assert S == 2.0 ** round(math.log2(S)) and 2.0 ** 14 <= w.abs().max().item() * S < 2.0 ** 15
        ws = (w * S).reshape(-1, 8)
        assert torch.equal(ws / S, w.reshape(-1, 8))                                        # the scaling is exact
        hw, lw = [t.view(torch.float16).float() for t in hip._fp16_terms(ws)]
        err = (ws.double() - hw.double() - lw.double()).abs()
        assert bool((err <= 2.0 ** -22 * ws.abs().double() + 2.0 ** -25).all())
        x = (torch.randn(4096, generator=g) * torch.logspace(-4, 3, 4096)).reshape(-1, 8)      # activations are not scaled
        hx, lx = [t.view(torch.float16).float() for t in hip._fp16_terms(x)]
        assert bool(((x.double() - hx.double() - lx.double()).abs() <= 2.0 ** -22 * x.abs().double() + 2.0 ** -25).all())
        kept = (hx.double() * lw.double() + lx.double() * hw.double() + hx.double() * hw.double()) / S
        exact = x.double() * w.reshape(-1, 8).double()
        wd = w.reshape(-1, 8).abs().double()
        floor = 2.0 ** -25 * wd + 2.0 ** -25 / S * x.abs().double()          # where a low term is an fp16 subnormal: 2^-25 absolute on x, 2^-25 / S on w
        assert bool(((kept - exact).abs() <= 3 * 2.0 ** -22 * exact.abs() + 1.01 * floor).all())
    assert hip.tail_weight_scale(torch.zeros(8)) == 1.0


def test_docs_cite_existing_tests_and_files():
    """DESIGN.md / README.md / INTEGRATION.md / include/abopt.h must only cite tests that exist (the design document had drifted
    from the tree), and every profiles/ or tools/ file DESIGN.md section 5 names must be in the tree.  DESIGN_LOG.md is history: not checked."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    defs, files = set(), set()
    for f in glob.glob(os.path.join(root, 'tests', '*.py')):
        files.add(os.path.basename(f)[:-3])
        defs |= set(re.findall(r'^def (test_\w+)', open(f).read(), re.M))
    missing = []
    for doc in ('DESIGN.md', 'README.md', 'INTEGRATION.md', os.path.join('include', 'abopt.h')):
        text = open(os.path.join(root, doc)).read()
        for name in set(re.findall(r'\btest_[a-z0-9_]+\b', text)):
            if name in defs or name in files:
                continue
            if name.endswith('_') and any(d.startswith(name) for d in defs):      # an abbreviated name ("test_block_tail_...")
                continue
            missing.append((doc, name))
    assert not missing, missing
    design = open(os.path.join(root, 'DESIGN.md')).read()
    sec5 = design[design.index('## 5. Measurement'):design.index('## 6. Multi-GPU')]
    for path in set(re.findall(r'`((?:profiles|tools)/[\w./\[\]*-]+)`', sec5)):
        pat = re.sub(r'\[([^\]]*)\]', lambda m: '[' + m.group(1).replace('|', '') + ']', path)
        assert glob.glob(os.path.join(root, pat)) or glob.glob(os.path.join(root, pat) + '*'), f'DESIGN.md section 5 cites {path}, which is not in the tree'
    # ... and the ABI number: every "abopt_abi_version() == N" / "ABI number N" the documents state is the header's (INTEGRATION.md said 39)
    abi = int(re.search(r'#define ABOPT_ABI_VERSION (\d+)', open(os.path.join(root, 'include', 'abopt.h')).read()).group(1))
    for doc in ('DESIGN.md', 'README.md', 'INTEGRATION.md'):
        text = open(os.path.join(root, doc)).read()
        for n in re.findall(r'abopt_abi_version\(\)\s*==\s*(\d+)', text) + re.findall(r'ABI number (\d+)', text):
            assert int(n) == abi, f'{doc} states ABI {n}, include/abopt.h is {abi}'


def test_no_packed_fp32_instructions_in_the_library(tmp_path):
    """No v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 in any kernel of libabopt_hip.so (csrc/Makefile: NOPK).  On gfx950 these instructions return wrong results in lanes
    48-63 while another wave of the SIMD executes v_mfma_f32_16x16x32_f16 / _bf16 (round 6; tools/micro/dih_asm/partner_classes.cpp) -- and this
    library issues those MFMAs in most kernels.  The device code objects are cut out of the library's .hip_fatbin section and disassembled."""
    import re
    import shutil
    import subprocess
    llvm = '/opt/rocm/lib/llvm/bin'
    if not all(os.path.exists(os.path.join(llvm, t)) for t in ('llvm-objcopy', 'clang-offload-bundler', 'llvm-objdump')):
        pytest.skip('ROCm LLVM tools not installed')
    lib = os.path.join(ROOT, 'ab_opt_amd', 'libabopt_hip.so')
    fat = tmp_path / 'fat.bin'
    subprocess.run([os.path.join(llvm, 'llvm-objcopy'), '-O', 'binary', '--only-section=.hip_fatbin', lib, str(fat)], check=True)
    blob = fat.read_bytes()
    magic = b'__CLANG_OFFLOAD_BUNDLE__'
    starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
    assert len(starts) >= 10, 'one bundle per kernel source expected'
    packed, kernels, mfma = [], 0, 0
    for n, a in enumerate(starts):
        b = starts[n + 1] if n + 1 < len(starts) else len(blob)
        bundle, co = tmp_path / f'b{n}.bin', tmp_path / f'b{n}.co'
        bundle.write_bytes(blob[a:b])
        subprocess.run([os.path.join(llvm, 'clang-offload-bundler'), '--unbundle', '--type=o', f'--input={bundle}', '--targets=hipv4-amdgcn-amd-amdhsa--gfx950',
                        f'--output={co}'], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        dis = subprocess.run([os.path.join(llvm, 'llvm-objdump'), '-d', '--mcpu=gfx950', str(co)], check=True, capture_output=True, text=True).stdout
        kernels += len(re.findall(r'^[0-9a-f]+ <_Z\w+>:', dis, flags=re.M))
        mfma += len(re.findall(r'v_mfma_f32_16x16x32_(?:f16|bf16)', dis))
        packed += re.findall(r'v_pk_(?:mul|add|fma)_f32', dis)
    assert kernels >= 60 and mfma > 100, (kernels, mfma)            # the disassembly really is this library's
    assert not packed, f'{len(packed)} packed-FP32 instructions in the device code: the library was built without the NOPK flag of csrc/Makefile'
