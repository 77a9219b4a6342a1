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


def test_contig_mask_and_registry():
    from ab_opt_amd.model import generate_mask_from_str, get_model, _MODEL_DICT
    t = torch.zeros(2, 10, dtype=torch.bool)
    m = generate_mask_from_str('3-5', t)
    assert m[:, 2:5].all() and m.sum() == 6
    assert {'diffab_abdock', 'diffab_abdesign'} <= set(_MODEL_DICT)
    assert type(build_model(10, 3)).__name__ == 'DiffusionAntibodyDesign'
    assert type(build_model(10, 3, flavour='abdesign')).__name__ == 'DiffusionAntibodyDesignAbDesign'
