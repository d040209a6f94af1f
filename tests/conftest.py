import os
import sys

import pytest

# The CPU oracle runs on torch's OpenMP pool (libgomp) with one thread per granted core.  With exactly as many workers as cores, ONE more runnable
# thread in the process parks a worker behind seven others that spin at the region's barrier for libgomp's default 300 000 iterations: every
# parallel region then costs a scheduler tick instead of microseconds -- seen here as the oracle chain tests taking 15 minutes instead of 13 s in
# two of five runs of this suite (700 % CPU, no progress).  A bounded spin keeps the fast path (31 s vs 28 s for the chain tests; 1000: 59 s) and
# lets a parked worker run.  Must be set before torch loads libgomp; the thread count itself stays (test_ddim's margin is thread-count sensitive).
os.environ.setdefault("GOMP_SPINCOUNT", "20000")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The CPU oracle runs on torch's intra-op pool.  A GPU box shows every host core but grants this job a share of 16: with the default
    # (one thread per visible core) the oracle's convolutions crawl under oversubscription.
    import torch
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(cores, int(os.environ.get("HICDIFF_CPU_THREADS", "16")))))


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
