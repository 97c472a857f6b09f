"""Random small models and cloud sizes through the whole-model link-by-link check (tests/emulate.py: every stage hand-off a
correct rounding of the oracle's arithmetic, logits within 1e-3), eval and train mode.  Diagnostic, time-boxed:
FUZZ_SECONDS (default 150), FUZZ_SEED."""
import os, sys, time, traceback
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import p4p_amd
from oracle import pix4point_oracle as O
from tests import emulate as E
from tests.test_gpu_02_model import build_pair, _no_dropout
from tests.test_gpu_03_matched_parity import _captured_forward

DEV = 'cuda'


def one(rng):
    D = int(rng.choice([192, 384, 768]))
    L, H = int(rng.integers(1, 4)), D // 64
    k = int(rng.choice([8, 16, 16, 32]))
    N = 4 * int(rng.integers(max(16, k // 4 + 1), 560))
    B = int(rng.integers(1, 6)) if N <= 1024 else int(rng.integers(1, 3))
    training = bool(rng.integers(0, 2)) and B >= 2          # the head's BatchNorm1d refuses a batch of one in train mode (torch)
    oracle, model = build_pair(40, D, L, H, seed=int(rng.integers(1 << 20)), k=k)
    _no_dropout(oracle, model)
    (oracle.train(), model.train()) if training else (oracle.eval(), model.eval())
    pts = O.synthetic_clouds(B, N, seed=int(rng.integers(1 << 20)), dup_frac=float(rng.choice([0.0, 0.0, 0.05])))
    start = torch.from_numpy(rng.integers(0, N, size=B))
    idx = O.farthest_point_sampling(pts, N // 4, start)
    cen = torch.gather(pts, 1, idx.unsqueeze(-1).expand(-1, -1, 3))
    nb = O.knn_indices(cen, pts, k)
    logits, cap = _captured_forward(model, pts.to(DEV), start.to(DEV), training)
    E.assert_links(oracle.to(DEV), cap, pts.to(DEV), idx.to(DEV), nb.to(DEV), logits, training)
    return (D, L, k, B, N, 'train' if training else 'eval')


def run(budget, seed, verbose=True):
    rng = np.random.default_rng(seed)
    t0, n, bad = time.time(), 0, []
    while time.time() - t0 < budget:
        n += 1
        try:
            tag = one(rng)
            if verbose:
                print('ok', tag, flush=True)
        except Exception as e:                                  # noqa: BLE001
            bad.append(repr(e)[:400])
            if verbose:
                print('FAIL', repr(e)[:400], flush=True)
                traceback.print_exc(limit=4)
    if verbose:
        print(f'done: {n} cases, {len(bad)} failures in {time.time() - t0:.0f} s')
    return n, bad


if __name__ == '__main__':
    _, failures = run(float(os.environ.get('FUZZ_SECONDS', 150)), int(os.environ.get('FUZZ_SEED', 0)))
    sys.exit(1 if failures else 0)
