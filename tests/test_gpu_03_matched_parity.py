"""The north star's float tolerance (logits within 1e-3), asserted the only way two bf16 pipelines can be compared.

bf16 MFMA operands put the build 4e-3..7e-3 from the PURE fp32 oracle on O(1) logits (tests/test_gpu_02_model.py keeps
that comparison at 1e-2 and shows stock torch bf16 has the same error).  That window is too wide to catch a kernel-logic
error of a few 1e-3, so the whole model is compared with the oracle's arithmetic ROUNDED WHERE THE PIPELINE ROUNDS
(tests/emulate.py, tied to the oracle on the CPU by tests/test_emulation_cpu.py: with the roundings off it IS the oracle):

* link by link (``teacher_forced_links``): every tensor a stage hands on must be a correct bf16 rounding of the stated fp32
  value computed from the pipeline's OWN upstream tensors (<= half an ulp + 5 % slack), fp32 hand-offs agree to fp32
  accuracy, and the final link is the logits, bounded by the north star's 1e-3 (measured ~1e-6).  This covers the whole
  model -- tokenizer, token streams, every block, final norm, head -- and a logic error shows up in exactly one link;
* free running (both sides from the points): fp32 summation order differs at 1e-7, which flips ~1e-4 of the bf16
  roundings, and those flips grow chaotically (d -> sqrt(d * ulp) per rounding stage) to the bf16 noise floor within one
  transformer block, so this form cannot go below a few 1e-3 for ANY second implementation; it is kept with the measured
  bound as a regression guard, next to the 1e-2 bound against the pure fp32 oracle."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

import p4p_amd                                    # noqa: E402
from oracle import pix4point_oracle as O          # noqa: E402
from tests import emulate as E                    # noqa: E402
from tests.test_gpu_02_model import build_pair, rel_l2, _no_dropout      # noqa: E402

DEV = 'cuda'
NORTH_STAR_TOL = E.NORTH_STAR_TOL


def _indices(pts, start, k=16):
    N = pts.shape[1]
    idx = O.farthest_point_sampling(pts, N // 4, start)
    cen = torch.gather(pts, 1, idx.unsqueeze(-1).expand(-1, -1, 3))
    return idx, O.knn_indices(cen, pts, k)


CONFIGS = [(192, 2, 3, 8, 1024),        # C1: the tiny ViT of BASELINE configs[0]
           (768, 12, 12, 2, 1024),      # C2: ViT-B/12, N=1024 (T=257, resident attention)
           (768, 12, 12, 2, 2048),      # C3's share: ViT-B/12, N=2048 (T=513, streaming attention)
           (1024, 2, 16, 1, 8192),      # C5's width and token count (T=2049, 16 heads)
           (384, 3, 6, 3, 300)]         # ragged everything (T=76)


def _captured_forward(model, pts, start, training):
    from p4p_amd import functional as F
    F._CAPTURE = cap = {}
    try:
        if training:
            logits = model(pts, start)
        else:
            with torch.no_grad():
                logits = model(pts, start)
    finally:
        F._CAPTURE = None
    return logits, cap


_check_links = E.assert_links          # every hand-off within LINK_LIMITS, logits within NORTH_STAR_TOL (tests/emulate.py)


@pytest.mark.parametrize('D,L,H,B,N', CONFIGS)
def test_eval_forward_link_by_link(D, L, H, B, N):
    oracle, model = build_pair(40, D, L, H, seed=D + L)
    oracle.eval(), model.eval()
    pts = O.synthetic_clouds(B, N, seed=N)
    start = torch.arange(B) * 7
    idx, nb = _indices(pts, start)
    logits, cap = _captured_forward(model, pts.to(DEV), start.to(DEV), False)
    assert len(cap['blocks']) == L
    _, worst = _check_links(oracle.to(DEV), cap, pts.to(DEV), idx.to(DEV), nb.to(DEV), logits, False)
    print('worst links:', sorted(((round(v, 4), n) for n, v in worst.items()), reverse=True)[:5])


@pytest.mark.parametrize('D,L,H,B,N,gmax_tol,gmed_tol', [(192, 2, 3, 32, 256, 6e-2, 2.5e-2), (768, 12, 12, 8, 1024, 0.12, 5e-2)])
def test_train_step_link_by_link_loss_and_gradients(D, L, H, B, N, gmax_tol, gmed_tol):
    """Train mode (batch-statistics BatchNorm in the tokenizer and the head, dropout off): the same chain, ending in the
    cross-entropy, within 1e-3."""
    oracle, model = build_pair(40, D, L, H, seed=11)
    _no_dropout(oracle, model)
    oracle.train(), model.train()
    pts = O.synthetic_clouds(B, N, seed=5)
    start = torch.arange(B) * 3
    labels = torch.randint(0, 40, (B,), generator=torch.Generator().manual_seed(2)).to(DEV)
    idx, nb = _indices(pts, start)
    logits, cap = _captured_forward(model, pts.to(DEV), start.to(DEV), True)
    oracle = oracle.to(DEV)
    stated, _ = _check_links(oracle, cap, pts.to(DEV), idx.to(DEV), nb.to(DEV), logits, True, grad=True)
    lk = torch.nn.functional.cross_entropy(logits, labels)
    le = torch.nn.functional.cross_entropy(stated, labels)
    assert abs(float(lk.detach()) - float(le.detach())) <= NORTH_STAR_TOL, (float(lk.detach()), float(le.detach()))
    # gradients: the statement's autograd, evaluated AT the pipeline's forward tensors (identical ReLU masks, arg-max picks
    # and softmax numerators), against the backward kernels; what differs is the bf16 rounding of the backward operands
    lk.backward()
    le.backward()
    og = dict(oracle.named_parameters())
    gmax = max(float(p.grad.norm()) for p in og.values() if p.grad is not None)
    errs = {}
    for name, p in model.named_parameters():
        ref = og[name].grad
        if ref is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        assert p.grad is not None, name
        if float(ref.norm()) < 1e-4 * gmax:
            assert float(p.grad.norm()) < 1e-3 * gmax, name       # structurally zero (a bias in front of train-mode BN)
            continue
        errs[name] = rel_l2(p.grad, ref)
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:6]
    med = sorted(errs.values())[len(errs) // 2]
    if os.path.isdir('gpurun_out'):
        import json
        with open(f'gpurun_out/grad_errs_d{D}.json', 'w') as fh:
            json.dump({n: [e, float(og[n].grad.norm())] for n, e in errs.items()}, fh, indent=0)
    print('gradient rel-L2 vs the link-by-link statement: median %.2e, worst %s' % (med, worst[:3]))
    # measured: tiny 4.6e-2 worst / 1.4e-2 median; ViT-B/12 at batch 8: 8.9e-2 (norm1 / qkv biases of the late blocks, whose
    # key third is exactly zero in exact arithmetic) / 3.6e-2 -- the bf16 rounding of every backward operand, accumulated
    # over 12 blocks of nearly uncorrelated per-row gradients (random weights, random labels).  A wrong factor on one
    # tensor is a >= 100 % error.
    assert worst[0][1] < gmax_tol, worst
    assert med < gmed_tol, (med, worst)


def test_vit_b12_block_gradients_teacher_forced_per_block():
    """Round 4: the weight gradients of ViT-B/12, block by block.  The whole-step comparison above accumulates the bf16 rounding of
    every backward operand over 12 blocks (worst tensor 9e-2); here every block's backward is checked on its own: the statement of
    ONE block (tests/emulate.py: block_emulated) evaluated at the pipeline's forward tensors (the tape substitutes them) and driven
    by the pipeline's OWN upstream gradient of that block (functional._CAPTURE_BWD), against the weight gradients and the block-input
    gradient the kernels produced.  What is left is the rounding of the backward operands inside one block."""
    from p4p_amd import functional as F
    D, L, H, B, N = 768, 12, 12, 8, 1024
    oracle, model = build_pair(40, D, L, H, seed=11)
    _no_dropout(oracle, model)
    oracle.train(), model.train()
    pts = O.synthetic_clouds(B, N, seed=5)
    start = torch.arange(B) * 3
    labels = torch.randint(0, 40, (B,), generator=torch.Generator().manual_seed(2)).to(DEV)
    logits, cap = _captured_forward(model, pts.to(DEV), start.to(DEV), True)
    F._CAPTURE_BWD = bcap = {}
    try:
        torch.nn.functional.cross_entropy(logits, labels).backward()
    finally:
        F._CAPTURE_BWD = None
    oracle = oracle.to(DEV)
    T = N // 4 + 1
    names = ['norm1.weight', 'norm1.bias', 'attn.qkv.weight', 'attn.qkv.bias', 'attn.proj.weight', 'attn.proj.bias', 'norm2.weight',
             'norm2.bias', 'mlp.fc1.weight', 'mlp.fc1.bias', 'mlp.fc2.weight', 'mlp.fc2.bias']
    got = dict(model.named_parameters())
    res = {}
    for matched in (True, False):
        worst, worst_dx = {}, 0.0
        for l in range(L):
            blk = oracle.model.vit_blocks[l]
            for p in blk.parameters():
                p.grad = None
            c = cap['blocks'][l]
            xin = c['xin'].detach().clone().requires_grad_(True)
            tape = E.Tape({'blocks': [c]})
            out = E.block_emulated(xin, blk, B, T, H, tape, 0, matched_attention_backward=(c['att'], c['lse']) if matched else None)
            out.backward(bcap['block_dout'][l])
            for n in names:
                ref = dict(blk.named_parameters())[n].grad
                g = got[f'model.vit.blocks.{l}.{n}'].grad
                e = rel_l2(g, ref)
                if n == 'attn.qkv.bias':      # its key third is exactly zero in exact arithmetic: compare the q and v thirds
                    e = rel_l2(torch.cat([g[:D], g[2 * D:]]), torch.cat([ref[:D], ref[2 * D:]]))
                worst[n] = max(worst.get(n, 0.0), e)
            din = bcap['block_dout'][l - 1] if l > 0 else bcap['encoder_din']
            worst_dx = max(worst_dx, rel_l2(din, xin.grad))
        print(f'per-block teacher-forced gradients ({"kernel-contract" if matched else "exact fp32"} attention backward in the statement), '
              'worst block per tensor:', {k: round(v, 4) for k, v in worst.items()}, 'dx', round(worst_dx, 4))
        res[matched] = (worst, worst_dx)
    # What this shows (measured, ViT-B/12, batch 8): the whole-step figures above are NOT an accumulation over the blocks -- one block
    # on the pipeline's own upstream gradient shows the same numbers.  The block-input gradient agrees to 5e-3 and every tensor of the
    # MLP branch and the projection to 1-3e-2 (the bf16 rounding of dx / dh / dln in front of the backward GEMMs); the two BIAS-like
    # sums of the attention branch, norm1.bias and attn.qkv.bias, sit at 8e-2 and norm1.weight / attn.qkv.weight at 3.5-4e-2, with the
    # exact fp32 attention gradient in the statement AND with the kernels' own contract there (P, dS rounded to bf16: 3.4e-2 / 7.8e-2):
    # column sums over all rows of gradients whose rows nearly cancel (the key third of the qkv bias gradient is exactly zero in exact
    # arithmetic and pure rounding noise here -- it is left out of the comparison but still reaches norm1 through the qkv weight).
    # That the attention backward kernels themselves follow their contract ELEMENT BY ELEMENT is test_attention_fwd_bwd's job
    # (tests/emulate.py: attention_backward_emulated: every stored element within four flipped numerators of a correct rounding).
    for matched in (True, False):
        w, wdx = res[matched]
        assert wdx < 1e-2, (matched, wdx)
        assert max(w[n] for n in names if n.startswith(('mlp', 'norm2', 'attn.proj'))) < 3.5e-2, (matched, w)
        assert max(w['norm1.weight'], w['attn.qkv.weight']) < 5e-2, (matched, w)
        assert max(w.values()) < 0.11, (matched, w)


FREE_RUNNING_TOL = 6e-3        # measured 2.8e-3 (C1) .. see DESIGN.md section 2; the pure-fp32 bound is 1e-2


@pytest.mark.parametrize('D,L,H,B,N', CONFIGS)
def test_eval_logits_free_running_vs_rounding_matched_oracle(D, L, H, B, N):
    oracle, model = build_pair(40, D, L, H, seed=D + L)
    oracle.eval(), model.eval()
    pts = O.synthetic_clouds(B, N, seed=N)
    start = torch.arange(B) * 7
    idx, nb = _indices(pts, start)
    with torch.no_grad():
        got = model(pts.to(DEV), start.to(DEV)).float().cpu()
        want = E.pix4point_emulated(oracle.to(DEV), pts.to(DEV), idx.to(DEV), nb.to(DEV)).cpu()
        pure = oracle.cpu()(pts, start)
    err, err_pure = float((got - want).abs().max()), float((got - pure).abs().max())
    print(f'free-running: |build - matched| = {err:.2e}, |build - fp32 oracle| = {err_pure:.2e}, '
          f'|matched - fp32 oracle| = {float((want - pure).abs().max()):.2e}')
    assert err <= FREE_RUNNING_TOL * max(1.0, float(want.abs().max())), (err, float(want.abs().max()))
    # and the statement itself sits where the build sits relative to the pure fp32 oracle
    assert float((want - pure).abs().max()) <= 1e-2 * max(1.0, float(pure.abs().max()))
