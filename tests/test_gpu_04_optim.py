"""FusedAdamW (csrc/optim.hip) against stock torch: same update rule as torch.optim.AdamW and the same clipping as
torch.nn.utils.clip_grad_norm_, checked on parameter sets with the reference's decay / no-decay groups, odd sizes and
unaligned views; state_dict round trip with torch.optim.AdamW in both directions."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _params(seed, dev):
    g = torch.Generator().manual_seed(seed)
    shapes = [(768, 768), (2304,), (5,), (3, 7, 11), (1,), (40000,), (16385,), (129, 3)]
    return [torch.nn.Parameter(torch.randn(*s, generator=g).to(dev)) for s in shapes]


def _grads(ps, seed, scale):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(p.shape, generator=g) * scale for p in ps]


@pytest.mark.parametrize('max_norm,scale', [(None, 1.0), (10.0, 1.0), (10.0, 1e-3)])
def test_matches_torch_adamw_and_clip(max_norm, scale):
    from p4p_amd.optim import FusedAdamW
    ours, ref = _params(0, DEV), _params(0, 'cpu')
    groups = lambda ps: [{'params': ps[:4], 'weight_decay': 5e-2}, {'params': ps[4:], 'weight_decay': 0.0, 'lr': 2e-3}]
    opt = FusedAdamW(groups(ours), lr=5e-4, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-2)
    tor = torch.optim.AdamW(groups(ref), lr=5e-4, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-2)
    for it in range(4):
        gs = _grads(ref, 100 + it, scale)
        for p, q, g in zip(ours, ref, gs):
            p.grad = g.to(DEV)
            q.grad = g.clone()
        want_norm = None
        if max_norm is not None:
            want_norm = torch.nn.utils.clip_grad_norm_(ref, max_norm)
        tor.step()
        got_norm = opt.step(max_norm=max_norm)
        if max_norm is not None:
            torch.testing.assert_close(got_norm.cpu(), want_norm, rtol=1e-5, atol=1e-7)
            for p, g in zip(ours, gs):
                assert torch.equal(p.grad.cpu(), g)                  # gradients themselves stay unscaled (documented)
        for p, q in zip(ours, ref):
            torch.testing.assert_close(p.detach().cpu(), q.detach(), rtol=2e-6, atol=2e-7)
    for p, q in zip(ours, ref):
        torch.testing.assert_close(opt.state[p]['exp_avg'].cpu(), tor.state[q]['exp_avg'], rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(opt.state[p]['exp_avg_sq'].cpu(), tor.state[q]['exp_avg_sq'], rtol=1e-5, atol=1e-7)


def test_state_dict_round_trip_with_torch_adamw():
    from p4p_amd.optim import FusedAdamW
    ours, ref = _params(1, DEV), _params(1, DEV)
    opt = FusedAdamW([{'params': ours}], lr=1e-3, weight_decay=5e-2)
    tor = torch.optim.AdamW([{'params': ref}], lr=1e-3, weight_decay=5e-2, fused=True)
    for it in range(2):
        for p, q, g in zip(ours, ref, _grads(ref, 7 + it, 1.0)):
            p.grad, q.grad = g.to(DEV), g.to(DEV)
        opt.step()
        tor.step()
    # torch -> ours and ours -> torch, then one more identical step each
    opt2 = FusedAdamW([{'params': ours}], lr=1e-3, weight_decay=5e-2)
    opt2.load_state_dict(tor.state_dict())
    tor2 = torch.optim.AdamW([{'params': ref}], lr=1e-3, weight_decay=5e-2)
    tor2.load_state_dict(opt.state_dict())
    for p, q, g in zip(ours, ref, _grads(ref, 99, 1.0)):
        p.grad, q.grad = g.to(DEV), g.to(DEV)
    opt2.step()
    tor2.step()
    for p, q in zip(ours, ref):
        torch.testing.assert_close(p.detach(), q.detach(), rtol=2e-6, atol=2e-7)


def test_rejects_cpu_parameters():
    from p4p_amd.optim import FusedAdamW
    p = torch.nn.Parameter(torch.zeros(4))
    p.grad = torch.ones(4)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        FusedAdamW([p]).step()


def test_step_invalidates_the_bf16_operand_caches():
    """The update kernel writes parameters through raw pointers; the GEMM operand copies cached on each weight are keyed on
    the tensor's version counter, so the step must bump it -- otherwise every GEMM would keep running on the initial weights."""
    from p4p_amd.functional import _bf16_weights
    from p4p_amd.optim import FusedAdamW
    w = torch.nn.Parameter(torch.randn(64, 64, device='cuda'))
    wb0, _ = _bf16_weights(w, False)
    assert _bf16_weights(w, False)[0] is wb0                       # cached while the weight is unchanged
    v0 = w._version
    w.grad = torch.ones_like(w)
    FusedAdamW([w], lr=0.1).step()
    assert w._version > v0
    wb1, _ = _bf16_weights(w, False)
    assert wb1 is not wb0 and torch.equal(wb1, w.detach().bfloat16()) and not torch.equal(wb1, wb0)


def test_parameters_with_different_step_counts_get_their_own_bias_correction():
    """parameters that share betas / eps but not the step count (here: two of them receive their first gradient
    at step 3) must each be updated exactly once per step with THEIR bias correction, as torch.optim.AdamW does."""
    from p4p_amd.optim import FusedAdamW
    ours, ref = _params(3, DEV), _params(3, 'cpu')
    opt = FusedAdamW([{'params': ours}], lr=1e-3, weight_decay=5e-2)
    tor = torch.optim.AdamW([{'params': ref}], lr=1e-3, weight_decay=5e-2)
    late = {1, 5}
    for it in range(5):
        gs = _grads(ref, 300 + it, 1.0)
        for j, (p, q, g) in enumerate(zip(ours, ref, gs)):
            if j in late and it < 2:
                p.grad = q.grad = None
            else:
                p.grad, q.grad = g.to(DEV), g.clone()
        want_norm = torch.nn.utils.clip_grad_norm_([q for q in ref if q.grad is not None], 5.0)
        tor.step()
        got_norm = opt.step(max_norm=5.0)
        torch.testing.assert_close(got_norm.cpu(), want_norm, rtol=1e-5, atol=1e-7)
        for p, q in zip(ours, ref):
            torch.testing.assert_close(p.detach().cpu(), q.detach(), rtol=2e-6, atol=2e-7)
    assert float(opt.state[ours[1]]['step']) == 3 and float(opt.state[ours[0]]['step']) == 5


def test_grad_scale_is_the_data_parallel_mean_and_nan_norms_propagate():
    """grad_scale (1/world of the gradient all-reduce) folded into the clip coefficient == scaling the gradients first; a NaN
    gradient norm poisons the update as torch.nn.utils.clip_grad_norm_ does (its clamp keeps NaN), it is not clipped away."""
    from p4p_amd.optim import FusedAdamW
    for max_norm in (None, 0.5):
        ours, ref = _params(4, DEV), _params(4, 'cpu')
        opt = FusedAdamW([{'params': ours}], lr=1e-3, weight_decay=5e-2)
        tor = torch.optim.AdamW([{'params': ref}], lr=1e-3, weight_decay=5e-2)
        for it in range(3):
            gs = _grads(ref, 400 + it, 1.0)
            for p, q, g in zip(ours, ref, gs):
                p.grad, q.grad = (g * 8).to(DEV), g.clone()                    # a sum over 8 ranks vs its mean
            if max_norm is not None:
                torch.nn.utils.clip_grad_norm_(ref, max_norm)
            tor.step()
            opt.step(max_norm=max_norm, grad_scale=1.0 / 8)
            for p, q in zip(ours, ref):
                torch.testing.assert_close(p.detach().cpu(), q.detach(), rtol=2e-6, atol=2e-7)
    ours = _params(5, DEV)
    opt = FusedAdamW([{'params': ours}], lr=1e-3)
    for p in ours:
        p.grad = torch.ones_like(p)
    ours[2].grad[0] = float('nan')
    norm = opt.step(max_norm=1.0)
    assert bool(torch.isnan(norm)) and all(bool(torch.isnan(p).any()) for p in ours)


def test_invalidate_operand_caches_after_a_write_that_bypasses_the_version_counter():
    """a write through ``.data`` leaves the bf16 operand copies stale (no error); the documented remedy is
    p4p_amd.invalidate_operand_caches(module)."""
    import p4p_amd
    from p4p_amd import functional as F
    lin = torch.nn.Linear(64, 64).to(DEV)
    x = torch.randn(8, 64, device=DEV)
    y0 = F.linear(x, lin.weight, lin.bias)
    lin.weight.data.mul_(2.0)                                   # bypasses the version counter
    assert torch.equal(F.linear(x, lin.weight, lin.bias), y0)   # stale operands: the documented hazard
    assert p4p_amd.invalidate_operand_caches(lin) >= 1
    y1 = F.linear(x, lin.weight, lin.bias)
    assert float((y1 - lin.bias - 2 * (y0 - lin.bias)).abs().max()) < 2e-2
    with pytest.raises(NotImplementedError, match='N // 4'):
        p4p_amd.P3Embed(in_channels=3, sample_ratio=0.5, scale=2, k=16)
