#!/usr/bin/env python3
"""Drive the HOST code of the engine under AddressSanitizer + UBSan, without a GPU (tools/sanitize/Makefile builds the library against a
stand-in HIP runtime: device memory is host memory, launches are no-ops).  Every code path that does pointer arithmetic on the host runs:
weight loaders (both networks, three flavours), the dry-run sizing and the first-fit pool, eps forwards of ragged batches, the fused
sampler steps eagerly / captured / replayed, one whole-batch lane and 2-4 sub-batch lanes inside and outside a chain bracket, the precision
switches (context-wide and per step), DDRM steps incl. skipped ones, the operator primitives, tile cut / stitch, and the native trainers of
both networks (create, slots, stages, loss_backward, Adam).  Run by tests/test_host_sanitizers.py:

    make -C tools/sanitize && LD_PRELOAD=$(clang++ -print-file-name=libclang_rt.asan-x86_64.so) python3 tools/sanitize/run_host_asan.py

Exit code 0 and no sanitizer report = clean.  No torch in this process (the sanitizer runtime must be the first library loaded).
"""
import ctypes as C
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def load_module(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


L = load_module("hd_lib", os.path.join(ROOT, "hicdiff_amd", "_lib.py"))         # structs + symbol table only (it imports torch in load(), not here)
SP = load_module("hd_specs", os.path.join(ROOT, "hicdiff_amd", "_specs.py"))
lib = C.CDLL(os.path.join(HERE, "_build", "libhicdiff_hip_asan.so"))
for name, (res, args) in L.SYMBOLS.items():
    fn = getattr(lib, name)
    fn.restype, fn.argtypes = res, args
lib.hipstub_launches.restype = C.c_long
lib.hipstub_graph_launches.restype = C.c_long
P = lambda a: None if a is None else C.c_void_p(a.ctypes.data)
KEEP = []


def arr(shape, fill=None, dtype=np.float32):
    a = np.zeros(shape, dtype=dtype) if fill is None else np.full(shape, fill, dtype=dtype)
    KEEP.append(a)
    return a


def check(rc, ctx=None, what=""):
    if rc != 0:
        raise SystemExit(f"{what}: rc {rc}: {(lib.hd_last_error(ctx) or b'').decode()}")


def make_ctx(kind, dim, mults=(), nres=0, cond=False, sr3=False):
    a = L.HdArchDesc()
    a.kind, a.dim, a.n_mults = kind, dim, len(mults)
    for i, m in enumerate(mults):
        a.mults[i] = m
    a.channels, a.self_condition, a.sr3, a.groups, a.number_resnet = 1, int(cond), int(sr3), 8 if kind == L.HD_ARCH_UNET else 1, nres
    ctx = C.c_void_p()
    check(lib.hd_create(C.byref(ctx), 0, C.byref(a)), None, "hd_create")
    specs = SP.unet_specs(dim, mults, 1, cond, sr3) if kind == L.HD_ARCH_UNET else SP.hicedrn_specs(1, nres, cond, sr3, dim)
    named = (L.HdNamedTensor * len(specs))()
    for i, (name, shape, _, _) in enumerate(specs):
        t = arr(shape, 0.01)
        named[i].name, named[i].data, named[i].ndim = name.encode(), t.ctypes.data, len(shape)
        for j, s in enumerate(shape):
            named[i].shape[j] = s
    KEEP.append(named)
    check(lib.hd_load_weights(ctx, named, len(specs), None), ctx, "hd_load_weights")
    check(lib.hd_load_weights(ctx, named, len(specs), None), ctx, "hd_load_weights (again: after an optimizer step)")
    # a wrong shape and a missing entry must be refused, not read
    bad = (L.HdNamedTensor * len(specs))()
    C.memmove(bad, named, C.sizeof(named))
    bad[0].shape[0] += 1
    assert lib.hd_load_weights(ctx, bad, len(specs), None) == L.HD_ENOWEIGHT
    assert lib.hd_load_weights(ctx, named, len(specs) - 1, None) == L.HD_ENOWEIGHT
    check(lib.hd_load_weights(ctx, named, len(specs), None), ctx, "hd_load_weights (restore)")
    return ctx, a


def sampler_paths(ctx, B, S, cond):
    x, y, x0, eps = arr((B, 1, S, S)), arr((B, 1, S, S)), arr((B, 1, S, S)), arr((B, 1, S, S))
    noise = arr((3, B, 1, S, S))
    t_i, t_f = arr((B,), 5, np.int64), arr((B,), 0.5)
    assert lib.hd_ddpm_step(ctx, P(x), P(y) if cond else None, None, C.byref(L.HdDdpmCoef()), None, B, S, 1, 0, 3, None) == L.HD_ENOMEM      # before hd_reserve
    need = C.c_size_t()
    check(lib.hd_workspace_bytes(ctx, B, S, C.byref(need)), ctx, "hd_workspace_bytes")
    check(lib.hd_reserve(ctx, B, S), ctx, "hd_reserve")
    for b in (1, 2, B):                                                           # ragged batches inside the reservation
        check(lib.hd_eps_forward(ctx, P(x), P(t_i), L.HD_T_INT64, P(y) if cond else None, P(eps), b, S, None), ctx, "hd_eps_forward")
    check(lib.hd_eps_forward(ctx, P(x), P(t_f), L.HD_T_FLOAT32, P(y) if cond else None, P(eps), B, S, None), ctx, "hd_eps_forward float t")
    co = L.HdDdpmCoef()
    co.sqrt_recip_alphas_cumprod, co.sqrt_recipm1_alphas_cumprod, co.posterior_mean_coef1, co.posterior_mean_coef2, co.sigma, co.time_value = 1.2, 0.6, 0.3, 0.7, 0.1, 9.0
    old = (C.c_uint32 * 8)(28)                                                    # a revision-3 struct without eps_coef / arith
    C.memmove(C.byref(old, 4), C.byref(co, 4), 24)
    junk = (C.c_float * 9)(1.5)                                                   # a struct without the size prefix: refused, never read past
    assert lib.hd_ddpm_step(ctx, P(x), P(y) if cond else None, None, C.cast(junk, C.POINTER(L.HdDdpmCoef)), None, B, S, 1, 0, 3, None) == L.HD_EINVAL
    for graphs in (0, 1):
        check(lib.hd_set_graphs(ctx, graphs), ctx, "hd_set_graphs")
        for chains in (1, 2, 3, 4, 0):
            check(lib.hd_set_chains(ctx, chains), ctx, "hd_set_chains")
            for bracket in (False, True):
                if bracket:
                    check(lib.hd_chain_begin(ctx, None), ctx, "hd_chain_begin")
                    assert lib.hd_reserve(ctx, B, S) == L.HD_ESTATE
                for k in range(5):                                                # eager, capture, replay, replay, + one with replayed noise
                    co.arith = L.HD_ARITH_F16W2 if k % 2 else L.HD_ARITH_DEFAULT
                    check(lib.hd_ddpm_step(ctx, P(x), P(y) if cond else None, P(noise) if k == 4 else None, C.byref(co), P(x0) if k == 2 else None,
                                           B, S, 7, 11, k, None), ctx, "hd_ddpm_step")
                check(lib.hd_ddpm_step(ctx, P(x), P(y) if cond else None, None, C.cast(old, C.POINTER(L.HdDdpmCoef)), None, B, S, 7, 0, 1, None), ctx, "old struct")
                if bracket:
                    check(lib.hd_chain_end(ctx, None), ctx, "hd_chain_end")
    if not cond:
        dc = L.HdDdrmCoef()
        dc.sqrt_at, dc.sqrt_1m_at, dc.sqrt_at_next, dc.sigma_next, dc.sigma_0, dc.etaA, dc.etaB, dc.etaC, dc.time_value = 0.8, 0.6, 0.85, 0.62, 0.1, 0.85, 1.0, 0.85, 300.0
        check(lib.hd_chain_begin(ctx, None), ctx, "hd_chain_begin")
        for k in range(4):
            dc.skip_network = 1 if k == 2 else 0
            check(lib.hd_ddrm_step(ctx, P(x), P(y), P(noise) if k == 3 else None, C.byref(dc), P(x0), B, S, 7, 0, k, None), ctx, "hd_ddrm_step")
        check(lib.hd_chain_end(ctx, None), ctx, "hd_chain_end")
        dc.skip_network, dc.sigma_0 = 1, 1.0
        assert lib.hd_ddrm_step(ctx, P(x), P(y), None, C.byref(dc), None, B, S, 7, 0, 0, None) == L.HD_EINVAL      # not an inert step
    for mode in (L.HD_PRECISION_F32, L.HD_PRECISION_F16W2, L.HD_PRECISION_BF16X3):
        check(lib.hd_set_precision(ctx, mode), ctx, "hd_set_precision")
        check(lib.hd_eps_forward(ctx, P(x), P(t_i), L.HD_T_INT64, P(y) if cond else None, P(eps), B, S, None), ctx, "hd_eps_forward")
        check(lib.hd_ddpm_step(ctx, P(x), P(y) if cond else None, None, C.byref(co), None, B, S, 7, 0, 1, None), ctx, "hd_ddpm_step")
    check(lib.hd_reserve(ctx, B + 3, S), ctx, "hd_reserve (grow: graphs dropped)")
    check(lib.hd_ddpm_step(ctx, P(x), P(y) if cond else None, None, C.byref(co), None, B, S, 7, 0, 1, None), ctx, "hd_ddpm_step after grow")
    check(lib.hd_q_sample(ctx, P(x), P(eps), P(t_f), P(t_f), P(x0), B, S, None), ctx, "hd_q_sample")
    check(lib.hd_loss_per_sample(ctx, P(x), P(eps), 1, P(t_f), B, S, None), ctx, "hd_loss_per_sample")
    check(lib.hd_randn(ctx, P(x), B, S, 1, 2, 3, None), ctx, "hd_randn")


def trainer_paths(arch, B, S):
    h = C.c_void_p()
    check(lib.hd_train_create(C.byref(h), 0, C.byref(arch), B, S), None, "hd_train_create")
    total = C.c_longlong()
    n = lib.hd_train_param_count(h, C.byref(total))
    assert n > 0 and total.value > 0
    for i in range(n):
        name, off, shape, nd, stage = C.c_char_p(), C.c_longlong(), (C.c_longlong * 4)(), C.c_int(), C.c_int()
        check(lib.hd_train_param_slot(h, i, C.byref(name), C.byref(off), shape, C.byref(nd)), None, "hd_train_param_slot")
        check(lib.hd_train_slot_stage(h, i, C.byref(stage)), None, "hd_train_slot_stage")
        assert 0 <= off.value < total.value and 0 <= stage.value < lib.hd_train_stage_count(h)
    assert lib.hd_train_param_slot(h, n, C.byref(name), C.byref(off), shape, C.byref(nd)) != 0      # past the end: refused
    flat, grads, m, v = arr((total.value,), 0.01), arr((total.value,)), arr((total.value,)), arr((total.value,))
    x0, cond, noise, loss = arr((B, 1, S, S)), arr((B, 1, S, S)), arr((B, 1, S, S)), arr((1,))
    a_t, s_t, lw = arr((B,), 0.8), arr((B,), 0.6), arr((B,), 0.5)
    t = arr((B,), 0.4) if arch.sr3 else arr((B,), 7, np.int64)
    kind = L.HD_T_FLOAT32 if arch.sr3 else L.HD_T_INT64
    for prec in (1, L.HD_TRAIN_PREC_BF16):
        check(lib.hd_train_set_precision(h, prec), None, "hd_train_set_precision")
        for w in ((None, P(lw)) if not arch.sr3 else (None,)):
            check(lib.hd_train_set_loss_weights(h, w), None, "hd_train_set_loss_weights")
            rc = lib.hd_train_loss_backward(h, P(flat), P(grads), P(x0), P(cond) if arch.self_condition else None, P(t), kind, P(noise), P(a_t), P(s_t), 1, P(loss), None)
            if rc != 0:
                raise SystemExit(f"hd_train_loss_backward: rc {rc}: {(lib.hd_train_last_error(h) or b'').decode()}")
            for k in range(lib.hd_train_stage_count(h)):
                check(lib.hd_train_stage_wait(h, k, None), None, "hd_train_stage_wait")
    check(lib.hd_adam_step(P(flat), P(grads), P(m), P(v), total.value, 2e-5, 0.9, 0.999, 1e-8, 1, 1.0, None), None, "hd_adam_step")
    lib.hd_train_destroy(h)


def main():
    UNET, HIC = L.HD_ARCH_UNET, L.HD_ARCH_HICEDRN
    cases = [("unet16 uncond", UNET, 16, (1, 2), 0, False, False, 5, 16), ("unet32 cond", UNET, 32, (1, 2, 4), 0, True, False, 6, 40),
             ("unet64 sr3", UNET, 64, (1, 2, 4, 8), 0, True, True, 4, 64), ("unet64 uncond 40", UNET, 64, (1, 2, 4, 8), 0, False, False, 9, 40),
             ("hicedrn2 uncond", HIC, 256, (), 2, False, False, 5, 24), ("hicedrn3 cond", HIC, 256, (), 3, True, False, 4, 64), ("hicedrn2 sr3", HIC, 256, (), 2, True, True, 3, 40)]
    for label, kind, dim, mults, nres, cond, sr3, B, S in cases:
        ctx, a = make_ctx(kind, dim, mults, nres, cond, sr3)
        sampler_paths(ctx, B, S, cond)
        lib.hd_destroy(ctx)
        if kind == HIC or dim % 64 == 0:                 # (the UNet trainer takes dim % 64 == 0)
            trainer_paths(a, min(B, 4), S)
        print(f"  {label}: ok ({lib.hipstub_launches()} launches, {lib.hipstub_graph_launches()} graph replays so far)", flush=True)
    # operator primitives and tiles
    n, S = 3, 40
    A, X, D = arr((S, S)), arr((n, S, S)), arr((n, S, S))
    check(lib.hd_sandwich_matmul(P(A), P(X), P(A), P(D), n, S, None), None, "hd_sandwich_matmul")
    assert lib.hd_sandwich_matmul(P(A), P(X), P(A), P(D), n, 65, None) == L.HD_EINVAL
    check(lib.hd_dense_matmul(P(X), P(A), P(D), n * S, S, S, None), None, "hd_dense_matmul")
    check(lib.hd_fwht(P(X), n * S, S if S & (S - 1) == 0 else 32, 1.0, None), None, "hd_fwht")
    gc = L.HdDdrmCoef()
    check(lib.hd_ddrm_general_update(P(X), P(X), P(X), P(A), 16, None, None, None, C.byref(gc), P(D), n, S * S, 1, 0, 0, None), None, "hd_ddrm_general_update")
    gc.struct_bytes = 12
    assert lib.hd_ddrm_general_update(P(X), P(X), P(X), P(A), 16, None, None, None, C.byref(gc), P(D), n, S * S, 1, 0, 0, None) == L.HD_EINVAL
    mat, org, tl = arr((200, 200)), arr((6,), 0, np.int32), arr((6, 1, 40, 40))
    check(lib.hd_split_pieces(P(mat), 200, P(org), 6, 40, P(tl), None), None, "hd_split_pieces")
    print("host sanitizer drive: clean")


if __name__ == "__main__":
    sys.exit(main())
