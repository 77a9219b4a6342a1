"""CPU oracle for the denoising hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A from-scratch torch-CPU (fp32) restatement of the reference algorithm for the path
named by BASELINE.json's north_star.  Every function cites the
reference file:line it follows.  Floating-point path => a torch fp32 restatement, in
the reference's own op order, is the checker (it reproduces the reference to ~1e-6).

Pinned against the reference itself: `tests/golden/make_golden.py` imports
/root/reference (in the build container only) with hash-filled weights and writes
the fixtures in `tests/golden/*.npz`; `tests/test_oracle_golden.py` holds the oracle
to those fixtures.  Parity is therefore pinned (not "unpinned").

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import
this package.  Nothing under `ab_opt_amd/` imports it; the product path raises if its
HIP library is missing instead of falling back to anything here.
"""
