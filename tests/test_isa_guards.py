"""Build-time guards on the generated gfx950 code of the convolution kernels (no GPU needed: hipcc cross-compiles).

1. No scratch: register arrays that hipcc cannot keep in registers silently go to scratch memory and cost 2-3x.
2. No packed fp32 instruction in the split-bf16 conv kernels broadcasts BOTH lanes of an operand from the odd register
   of a pair (``op_sel:[0,1]`` / ``[1,0]`` on a two-source op, ``op_sel:[0,1,0]``-style on v_pk_fma).  Every build of the
   LayerNorm loader that did produced exact-zero lanes on large grids; the kernel avoids the
   pattern by copying the statistics out of their load pair, and this test keeps it that way.
"""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "hicdiff_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _asm(src):
    out = os.path.join(tempfile.mkdtemp(prefix="hd_isa_"), os.path.basename(src) + ".s")
    cmd = [HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-S", "--cuda-device-only",
           os.path.join(CSRC, src), "-o", out]
    subprocess.run(cmd, check=True, cwd=CSRC, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    with open(out) as f:
        text = f.read()
    shutil.rmtree(os.path.dirname(out), ignore_errors=True)
    return text


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("src", ["conv_bf16x3_ck16.hip", "conv_bf16x3_ck32.hip", "conv_bf16_plain_ck32.hip"])
def test_conv_kernels_have_no_scratch_and_no_odd_register_broadcast(src):
    text = _asm(src)
    kernels = re.findall(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, flags=re.S)
    assert len(kernels) >= (20 if "x3" in src else 9)
    for name, body in kernels:
        m = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body)
        assert m and int(m.group(1)) == 0, f"{name} uses {m.group(1) if m else '?'} bytes of scratch per lane"
    bad = []
    for line in text.splitlines():
        line = line.strip()
        if not line.startswith("v_pk_") or "_f32" not in line.split()[0] or "op_sel:" not in line:
            continue
        sel = [int(x) for x in re.search(r"op_sel:\[([0-9,]+)\]", line).group(1).split(",")]
        hi = re.search(r"op_sel_hi:\[([0-9,]+)\]", line)
        sel_hi = [int(x) for x in hi.group(1).split(",")] if hi else [1] * len(sel)
        if any(a == 1 and b == 1 for a, b in zip(sel, sel_hi)):      # lane 0 AND lane 1 of that operand read the odd register
            bad.append(line)
    assert not bad, "packed fp32 ops broadcasting from the odd register of a pair:\n" + "\n".join(bad[:8])


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_weight_gradient_gemm_keeps_its_prefetch_in_registers():
    """train.hip: with the prefetch registers in an indexed array hipcc parked them in scratch and waited after every load
    (4.6x slower).  The GEMM's slice loop must have no scratch traffic and must issue its 13 loads of a slice back to back."""
    text = _asm("train.hip")
    kernels = dict(re.findall(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, flags=re.S))
    # <PLAIN, ONE, TMW>: <false, false, 2> = split-bf16 x3, three column taps, 128 input channels per workgroup (72 MFMAs and 13 loads per
    # slice); <true, false, 2> = plain bf16 (24 MFMAs, 7 loads); <false, true, 2> = 1x1 filter, split-bf16 x3 (24 MFMAs, 13 loads);
    # <false, false, 1> = the 64-input-channel tile (36 MFMAs, 9 loads)
    for tag, mfmas, loads in (("ILb0ELb0ELi2E", 72, 13), ("ILb1ELb0ELi2E", 24, 7), ("ILb0ELb1ELi2E", 24, 13), ("ILb0ELb0ELi1E", 36, 9)):
        name = next(k for k in kernels if "wgrad_gemm_kernel" in k and tag in k)
        body = text[text.index(name + ":"):]
        body = body[:body.index("s_endpgm")]
        # the slice loop is the innermost loop: no scratch traffic inside it (a few address registers may spill around the outer
        # row-shift loop, which runs three times)
        inner = body[body.rindex("Depth=2" if "Depth=2" in body else "Depth=1"):]      # (the 1x1 variant has no row-shift loop around it)
        inner = inner[:inner.index("s_cbranch")]
        assert "scratch_" not in inner and inner.count("v_mfma") == mfmas, (tag, inner.count("v_mfma"))
        ops = [l.split()[0] for l in body.splitlines() if l.strip() and not l.strip().startswith((";", ".")) and not l.strip().endswith(":")]
        runs, cur = [], 0
        for op in ops:
            if op.startswith("global_load_dwordx4"):
                cur += 1
            elif op.startswith("s_waitcnt") or op.startswith("scratch"):
                runs.append(cur)
                cur = 0
        runs.append(cur)
        assert max(runs) >= loads, f"{tag}: longest run of global loads without a wait: {max(runs)}"
