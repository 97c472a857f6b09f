"""Random-shape sweep of the kernels against their references (diagnostic, time-boxed): GEMMs on integer data (exact), weight
gradients (exact), attention / LayerNorm / FPS / kNN through the bodies of tests/test_gpu_01_kernels.py at shapes the
parametrised tests do not list.  FUZZ_SECONDS (default 120), FUZZ_SEED."""
import os, sys, time, traceback
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import p4p_amd
from p4p_amd import ops
import test_gpu_01_kernels as T

DEV = 'cuda'
rng = np.random.default_rng(0)


def logint(lo, hi):
    return int(round(np.exp(rng.uniform(np.log(lo), np.log(hi)))))


def ints(shape, lo, hi):
    return torch.from_numpy(rng.integers(lo, hi + 1, size=shape)).to(torch.bfloat16).to(DEV)


def gemm_case():
    M = logint(1, 70000)
    N = int(rng.choice([8 * logint(1, 64), 64 * logint(1, 48), 256 * logint(1, 12)]))
    K = 64 * logint(1, 48)
    a, w = ints((M, K), -2, 2), ints((N, K), -1, 1)
    want = a.float() @ w.float().t()
    bias = (torch.arange(N, device=DEV, dtype=torch.float32) % 5 - 2)
    resid = ints((M, N), -3, 3).float()
    got = ops.gemm_tn(a, w, ops.EPI_F32_RESID, bias=bias, resid=resid)
    assert torch.equal(got, want + bias + resid), ('f32_resid', M, N, K)
    got = ops.gemm_tn(a, w, ops.EPI_BF16, bias=bias)
    assert torch.equal(got, (want + bias).to(torch.bfloat16)), ('bf16', M, N, K)
    return ('gemm_tn', M, N, K)


def wgrad_case():
    M = logint(1, 70000)
    N = int(rng.choice([8 * logint(1, 96), 256 * logint(1, 12)]))
    K = int(rng.choice([8 * logint(1, 96), 256 * logint(1, 4)]))
    g, a = ints((M, N), -1, 1), ints((M, K), -2, 2)
    want = g.float().t() @ a.float()
    from p4p_amd import _lib
    variant = int(rng.choice([2, 3, 6, 10]))          # gemm_wgrad3.h / gemm_wgrad4.h / gemm_wgrad5.h (round 5), where the shape takes them
    old = _lib.call_int('p4p_set_wgrad_wide', variant)
    try:
        assert torch.equal(ops.gemm_wgrad(g, a), want), ('wgrad', M, N, K, variant)
        dw, db = ops.gemm_wgrad(g, a, want_bias=True)
        assert torch.equal(dw, want) and torch.equal(db, g.float().sum(0)), ('wgrad+bias', M, N, K, variant)
        queue = []                                     # the deferred slab sum (round 5): same bits after the flush
        dq = ops.gemm_wgrad(g, a, defer=queue)
        ops.wgrad_flush(queue)
        assert torch.equal(dq, want), ('wgrad deferred', M, N, K, variant)
    finally:
        _lib.call_int('p4p_set_wgrad_wide', old)
    return ('wgrad', M, N, K)


def attn_case():
    """fp32 reference within bf16 noise (forward and gradients); against the rounding-matched statement every output within
    half an ulp plus FOUR numerators of its row rounded the other way (2^-9 p_max |v|_max each; three before the lazy rescale of round 4, under
    which the largest numerator of a row is no longer exactly 1: one of 282 cases reached 3.0 + 1e-4, tools/attn_lazy_flips.py): the statement's own bound
    (tests/emulate.py 'att') is a statistic of fixed seeds, a sweep needs the per-row form."""
    from tests import emulate as E
    B, H = int(rng.integers(1, 5)), int(rng.integers(1, 17))
    Tn = int(rng.choice([logint(1, 2100), 64 * logint(1, 32) + 1, 64 * logint(1, 32)]))
    if B * H * Tn * Tn > 3.5e8:
        B, H = 1, min(H, 4)
    D = 64 * H
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    qkv = (torch.randn(B * Tn, 3 * D, generator=g) * 1.5).to(torch.bfloat16).to(DEV)
    out, lse = ops.attention_fwd(qkv, B, Tn, H)
    qr = qkv.float().requires_grad_(True)
    ref = T._attn_ref(qr, B, Tn, H)
    torch.testing.assert_close(out.float(), ref, rtol=2e-2, atol=2e-2)
    with torch.no_grad():
        emu = E.attention_emulated(qkv.float(), B, Tn, H, round_output=False)
        q, k, v = qkv.float().view(B, Tn, 3, H, 64).permute(2, 0, 3, 1, 4)
        pmax = torch.softmax(q @ k.transpose(-1, -2) * 0.125, -1).amax(-1)              # (B,H,T)
        vmax = v.abs().amax(2, keepdim=True)                                            # (B,H,1,64)
        slack = (4 * 2.0 ** -9) * (pmax.unsqueeze(-1) * vmax).permute(0, 2, 1, 3).reshape(B * Tn, D)
        excess = ((out.float() - emu).abs() - 0.5 * E.bf16_ulp(emu) - slack - 1e-6).max()
        assert float(excess) <= 0, ('attention vs statement', B, Tn, H, float(excess))
    dout = torch.randn(B * Tn, D, generator=g).to(torch.bfloat16).to(DEV)
    ref.backward(dout.float())
    dqkv, dbias = ops.attention_bwd(qkv, out, dout, lse, B, Tn, H, want_colsum=True)
    err, scale = (dqkv.float() - qr.grad).abs().max().item(), qr.grad.abs().max().item()
    assert err <= 3e-2 * max(scale, 1.0), ('attention bwd', B, Tn, H, err, scale)
    assert (dbias.double() - dqkv.double().sum(0)).abs().max().item() <= 1e-5 * max(dqkv.float().abs().sum(0).max().item(), 1.0)
    return ('attention', B, Tn, H)


def ln_case():
    M, D = logint(1, 40000), 64 * int(rng.integers(1, 17))
    T.test_layernorm_fwd_bwd(M, D)
    return ('layernorm', M, D)


def knn_case():
    B, N = int(rng.integers(1, 4)), logint(16, 9000)
    S, k = int(rng.integers(1, max(2, N // 2))), int(rng.integers(1, min(64, N) + 1))
    if B * S * N > 3e7:
        S = max(1, int(3e7 // (B * N)))
    T.test_knn_matches_oracle_ragged(B, N, S, k)
    return ('knn', B, N, S, k)


def fps_case():
    B, N = int(rng.integers(1, 5)), logint(4, 12000)
    T.test_fps_matches_oracle_ragged_sizes(B, N)
    return ('fps', B, N)


cases = [gemm_case, gemm_case, wgrad_case, wgrad_case, attn_case, ln_case, knn_case, fps_case]


def run(budget, seed, verbose=True):
    """Random cases for ``budget`` seconds; returns (cases run, list of failures)."""
    global rng
    rng = np.random.default_rng(seed)
    t0, n, bad, counts = time.time(), 0, [], {}
    while time.time() - t0 < budget:
        fn = cases[n % len(cases)]
        n += 1
        try:
            tag = fn()
            counts[tag[0]] = counts.get(tag[0], 0) + 1
        except Exception as e:                                  # noqa: BLE001
            bad.append((fn.__name__, repr(e)[:300]))
            if verbose:
                print('FAIL', fn.__name__, repr(e)[:300], flush=True)
                traceback.print_exc(limit=3)
        if verbose and n % 40 == 0:
            print(f'{n} cases, {len(bad)} failures, {time.time() - t0:.0f} s', counts, flush=True)
    if verbose:
        print(f'done: {n} cases, {len(bad)} failures', counts)
    return n, bad


if __name__ == '__main__':
    _, failures = run(float(os.environ.get('FUZZ_SECONDS', 120)), int(os.environ.get('FUZZ_SEED', 0)))
    sys.exit(1 if failures else 0)
