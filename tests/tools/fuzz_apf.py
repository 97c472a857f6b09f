"""Random shapes through the AdaptPointFormer path against the CPU oracle (diagnostic, time-boxed): the PointNet tokenizer
(FPS / kNN group membership / Morton order bit-exact through the token values: fused path == group + encoder, tokens within
bf16 noise of the oracle), the adapter layer forward / input gradient, and small whole models (eval logits).
FUZZ_SECONDS (default 150), FUZZ_SEED."""
import os, sys, time, traceback
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import p4p_amd
from p4p_amd import apf as PA
from oracle import apf_oracle as A
from tests.test_gpu_05_apf import rel_l2, _layer_pair, _model_pair

DEV = 'cuda'


def pointnet_case(rng):
    cin = int(rng.choice([3, 4]))
    gs = int(rng.choice([8, 16, 32, 32]))
    G = int(rng.integers(1, 200))
    N = int(rng.integers(max(gs, G) + 1, 2100))
    B = int(rng.integers(1, 4))
    E = int(rng.choice([128, 192, 384, 768]))
    pn = PA.PointNet(E, G, gs, 2 * cin)
    want = A.PointNet(E, G, gs, 2 * cin)
    sd = A.seeded_state_dict(want, seed=int(rng.integers(1 << 20)))
    want.load_state_dict(sd), pn.load_state_dict(sd)
    pn = pn.to(DEV).eval()
    xyz = A.synthetic_clouds(B, N, seed=int(rng.integers(1 << 20)))
    pts = xyz if cin == 3 else torch.cat([xyz, xyz[:, :, 1:2] - xyz[:, :, 1:2].min(dim=1, keepdim=True).values], -1).contiguous()
    start = torch.from_numpy(rng.integers(0, N, size=B))
    with torch.no_grad():
        fused = pn(pts.to(DEV), start.to(DEV))
        nb, _ = pn.group(pts.to(DEV), pts[:, :, :3].contiguous().to(DEV), start.to(DEV))
        assert torch.equal(fused, pn.encoder(nb)), ('fused != group + encoder', cin, gs, G, N, B, E)
        ref = want.eval()(pts, start)
    r = rel_l2(fused, ref)
    assert r < 1.5e-2, ('pointnet tokens', cin, gs, G, N, B, E, r)
    return ('pointnet', cin, gs, G, N, B, E)


def layer_case(rng):
    D = int(rng.choice([192, 384, 768]))
    B, T = int(rng.integers(1, 5)), int(rng.integers(1, 400))
    o, m = _layer_pair(D, D // 64, seed=int(rng.integers(1 << 20)))
    o.eval(), m.eval()
    g = torch.Generator().manual_seed(int(rng.integers(1 << 20)))
    x, w = torch.randn(B, T, D, generator=g), torch.randn(B, T, D, generator=g)
    xo = x.clone().requires_grad_(True)
    yo = o(xo)
    (yo * w).sum().backward()
    xg = x.to(DEV).requires_grad_(True)
    yg = m(xg)
    (yg * w.to(DEV)).sum().backward()
    assert rel_l2(yg, yo) < 1e-2 and rel_l2(xg.grad, xo.grad) < 3e-2, ('layer', D, B, T, rel_l2(yg, yo), rel_l2(xg.grad, xo.grad))
    return ('layer', D, B, T)


def model_case(rng):
    cin = int(rng.choice([3, 4]))
    G, gs = int(rng.integers(2, 100)), int(rng.choice([8, 16, 32]))
    N = int(rng.integers(max(G, gs) + 1, 1200))
    B = int(rng.integers(1, 4))
    o, m = _model_pair(int(rng.integers(1 << 20)), in_channels=cin, npoint=G, nsample=gs, depth=int(rng.integers(1, 3)))
    o.eval(), m.eval()
    xyz = A.synthetic_clouds(B, N, seed=int(rng.integers(1 << 20)))
    pts = xyz if cin == 3 else torch.cat([xyz, xyz[:, :, 1:2] - xyz[:, :, 1:2].min(dim=1, keepdim=True).values], -1).contiguous()
    start = torch.from_numpy(rng.integers(0, N, size=B))
    with torch.no_grad():
        got = m(pts.to(DEV), start.to(DEV)).cpu()
        want = o(pts, start)
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) < 1.5e-2 * max(scale, 1.0), ('model logits', cin, G, gs, N, B, float((got - want).abs().max()), scale)
    return ('model', cin, G, gs, N, B)


def run(budget, seed, verbose=True):
    rng = np.random.default_rng(seed)
    cases = [pointnet_case, layer_case, model_case]
    t0, n, bad, counts = time.time(), 0, [], {}
    while time.time() - t0 < budget:
        fn = cases[n % 3]
        n += 1
        try:
            tag = fn(rng)
            counts[tag[0]] = counts.get(tag[0], 0) + 1
        except Exception as e:                                  # noqa: BLE001
            bad.append(repr(e)[:400])
            if verbose:
                print('FAIL', fn.__name__, repr(e)[:400], flush=True)
                traceback.print_exc(limit=3)
    if verbose:
        print(f'done: {n} cases, {len(bad)} failures in {time.time() - t0:.0f} s', counts)
    return n, bad


if __name__ == '__main__':
    _, failures = run(float(os.environ.get('FUZZ_SECONDS', 150)), int(os.environ.get('FUZZ_SEED', 0)))
    sys.exit(1 if failures else 0)
