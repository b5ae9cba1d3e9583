"""CPU oracle for the frankenstein hot path — TEST INFRASTRUCTURE, NOT PRODUCT.

A from-scratch restatement (plain PyTorch CPU fp32 tensor math + autograd, no GPU, no
HIP extension) of the reference's brainformer / GPT-2 forward, losses, AdamW step and LR
schedule.  Each function cites the reference file:line it follows.

Pinned: `tests/test_oracle_golden.py` checks every function here against golden vectors
produced by running the actual reference modules in the build container
(`tests/golden/make_golden.py`, fixtures under `tests/golden/*.npz`).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import
this package.  The product (`frankenstein_amd/`) never does: it fails loudly when the HIP
extension is missing.
"""
