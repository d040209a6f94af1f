"""The engine's HOST code under AddressSanitizer + UBSan, without a GPU (GPU sanitizers are not available on the pool).
tools/sanitize builds every csrc/*.hip --offload-host-only against a stand-in HIP runtime (device memory = host memory, launches = no-ops) and
tools/sanitize/run_host_asan.py drives the C ABI through its pointer arithmetic: loaders, dry-run sizing + first-fit pool, eager / captured /
replayed steps on one to four lanes inside and outside a chain bracket, precision switches, DDRM steps, both trainers.  Round 4's first run
found one real defect (an offset applied to a null base in the UNet trainer's sizing pass)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = os.path.join(ROOT, "tools", "sanitize")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.mark.skipif(not (os.path.exists(HIPCC) and os.path.exists(CLANG) and shutil.which("make")), reason="hipcc / clang / make not installed")
def test_host_code_is_clean_under_asan_and_ubsan():
    rt = subprocess.run([CLANG, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(rt) or not os.path.exists(rt):
        pytest.skip("no shared AddressSanitizer runtime in this toolchain")
    build = subprocess.run(["make", "-C", SAN, "-j4"], capture_output=True, text=True, timeout=1200)
    assert build.returncode == 0, build.stdout[-2000:] + build.stderr[-2000:]
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    run = subprocess.run([sys.executable, os.path.join(SAN, "run_host_asan.py")], capture_output=True, text=True, env=env, timeout=1200)
    out = run.stdout + run.stderr
    assert run.returncode == 0, out[-4000:]
    assert "AddressSanitizer" not in out and "runtime error:" not in out, out[-4000:]
    assert "host sanitizer drive: clean" in out
