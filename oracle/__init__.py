"""CPU oracle for the HiCDiff hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

This package is a from-scratch fp32 restatement (torch CPU functional ops, the
same math library the reference itself runs on) of the reference's DDPM / DDRM
sampling path and its two noise predictors.  Every function cites the
reference file:line it follows (paths relative to the upstream checkout).

Who may import it: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- as the checker / timed CPU baseline,
never as the thing shipped.  Nothing under ``hicdiff_amd/`` imports it; the
product path raises when the HIP library is missing instead of falling back.

Parity pin: the reference ships no tests or golden vectors, so the oracle is
pinned by fixtures generated from the reference's own Python code
(``tests/golden/make_golden.py`` imports ``/root/reference`` in the build
container and writes ``tests/golden/*.npz``); ``tests/test_oracle_golden.py``
checks the oracle against every one of them, and
``tests/test_oracle_vs_reference.py`` re-checks it live against the imported
reference when that checkout is present.
"""
