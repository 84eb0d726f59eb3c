"""CPU oracle for the ConsistencyTTA hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain fp32 PyTorch-CPU restatement of the reference's algorithm for the
path named in BASELINE.json (U-Net -> AudioLDM VAE decoder -> HiFi-GAN, Heun solver, EMA),
each function citing the reference file:line it follows.  It is imported ONLY by `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` -- always as the checker
or the timed CPU baseline, never as the product.  Nothing under `consistencytta_amd/`
imports it, and the product path raises when the HIP library is missing.

Pinning: the reference ships no tests or golden vectors for this path, so
the oracle is pinned against outputs of the reference's own modules imported in the build
container (`tests/golden/make_golden.py` -> `tests/golden/*.npz`, checked by
`tests/test_oracle_golden.py` on every box, and directly against the live reference modules
by `tests/test_oracle_vs_reference.py` where /root/reference is mounted).
"""
