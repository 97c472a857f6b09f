"""Experiment: weight-gradient GEMMs on a CU-masked side stream while the dgrad chain (GEMMs + LayerNorm backward +
attention backward) runs on the complementary CUs, against the same launches back to back on one stream.
ViT-B shapes of the N=1024 / B=256 step.  python tools/cumask_bench.py [side_cus ...]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import p4p_amd                                    # noqa: E402
from p4p_amd import ops, _lib                     # noqa: E402

dev = 'cuda'
lib = _lib.lib()
NCU = _lib.call_int('p4p_device_cu_count')


def masked_stream(cus):
    words = (NCU + 31) // 32
    m = (ctypes.c_uint32 * words)()
    for c in cus:
        m[c // 32] |= 1 << (c % 32)
    out = ctypes.c_void_p()
    _lib.call('p4p_stream_create_cu_mask', m, words, ctypes.byref(out))
    return torch.cuda.ExternalStream(out.value)


def main():
    B, T, D, H = 256, 257, 768, 12
    M = B * T
    g = torch.Generator().manual_seed(0)
    rb = lambda *s: torch.randn(*s, generator=g).to(torch.bfloat16).to(dev)
    w = lambda n, k: (torch.randn(n, k, generator=g) * k ** -0.5).to(torch.bfloat16).to(dev)
    dxb, gact, ln2, att, ln1 = rb(M, D), rb(M, 4 * D), rb(M, D), rb(M, D), rb(M, D)
    hpre = rb(M, 4 * D)
    Wfc2T, Wfc1T, WprojT, WqkvT = w(4 * D, D), w(D, 4 * D), w(D, D), w(D, 3 * D)
    qkv = rb(M, 3 * D)
    out, lse = ops.attention_fwd(qkv, B, T, H)
    x = torch.randn(M, D, device=dev)
    gamma = torch.ones(D, device=dev)
    _, _, mean, rstd = ops.layernorm_fwd(x, gamma, torch.zeros(D, device=dev), 1e-6)
    dx = torch.zeros(M, D, device=dev)
    dxb_out = torch.empty(M, D, dtype=torch.bfloat16, device=dev)

    hold = {}

    def dgrad_chain():
        dh = ops.gemm_tn(dxb, Wfc2T, ops.EPI_BF16_DGELU, pre=hpre)
        dln2 = ops.gemm_tn(dh, Wfc1T, ops.EPI_BF16)
        ops.layernorm_bwd(dln2, x, mean, rstd, gamma, dx=dx, dx_bf16=dxb_out, want_dxsum=True)
        datt = ops.gemm_tn(dxb_out, WprojT, ops.EPI_BF16)
        dqkv = ops.attention_bwd(qkv, out, datt, lse, B, T, H)
        dln1 = ops.gemm_tn(dqkv, WqkvT, ops.EPI_BF16)
        ops.layernorm_bwd(dln1, x, mean, rstd, gamma, dx=dx, dx_bf16=dxb_out, want_dxsum=True)
        hold['dh'], hold['dqkv'] = dh, dqkv

    dh0, dqkv0 = rb(M, 4 * D), rb(M, 3 * D)

    def wgrads():
        ops.gemm_wgrad(dxb, gact)            # fc2
        ops.gemm_wgrad(dh0, ln2)             # fc1
        ops.gemm_wgrad(dxb, att)             # proj
        ops.gemm_wgrad(dqkv0, ln1)           # qkv

    def timeit(fn, iters=12):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    t_d, t_w = timeit(dgrad_chain), timeit(wgrads)
    t_seq = timeit(lambda: (dgrad_chain(), wgrads()))
    print(f'{NCU} CUs; one stream: dgrad chain {t_d:.3f} ms, wgrads {t_w:.3f} ms, both {t_seq:.3f} ms per block', flush=True)

    sides = [int(a) for a in sys.argv[1:]] or [32, 48, 64, 96]
    for pattern in ('interleaved', 'low'):
        for ns in sides:
            if pattern == 'interleaved':
                stride = NCU // ns
                side = [c for c in range(NCU) if c % stride == 0][:ns]
            else:
                side = list(range(ns))
            sset = set(side)
            s_main = masked_stream([c for c in range(NCU) if c not in sset])
            s_side = masked_stream(side)
            # the wgrad workspace is shared through ops._workspace: give each stream its own by running them in turn once
            def both():
                cur = torch.cuda.current_stream()
                s_main.wait_stream(cur); s_side.wait_stream(cur)
                with torch.cuda.stream(s_main):
                    dgrad_chain()
                with torch.cuda.stream(s_side):
                    wgrads()
                cur.wait_stream(s_main); cur.wait_stream(s_side)
            with torch.cuda.stream(s_main):
                tm = timeit(dgrad_chain)
            with torch.cuda.stream(s_side):
                ts = timeit(wgrads)
            tb = timeit(both)
            print(f'{pattern:11s} side {ns:3d} CUs: dgrad alone on {NCU - ns} CUs {tm:.3f} ms, wgrads alone on {ns} CUs {ts:.3f} ms, '
                  f'concurrent {tb:.3f} ms (one stream {t_seq:.3f})', flush=True)
            torch.cuda.synchronize()


if __name__ == '__main__':
    main()
