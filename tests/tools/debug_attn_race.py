"""Stress the attention backward for run-to-run differences and localise them (b, head, row, q/k/v)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import p4p_amd
from p4p_amd import ops, _lib as L

DEV = 'cuda'
if os.environ.get('P4P_DEBUG_LIB'):          # bisecting: load an older build of the library (debug tool only)
    L.LIB_PATH = os.path.abspath(os.environ['P4P_DEBUG_LIB'])
H, D = 12, 768
B = int(os.environ.get('RB', 128))
T = int(os.environ.get('RT', 513))
RUNS = int(os.environ.get('RUNS', 24))
COLSUM = int(os.environ.get('COLSUM', 1))
torch.manual_seed(0)
qkv = torch.randn(B * T, 3 * D, device=DEV).bfloat16()
do = torch.randn(B * T, D, device=DEV).bfloat16()
o, lse = ops.attention_fwd(qkv, B, T, H)
st = ops._stream()

def run():
    dqkv = torch.full_like(qkv, float('nan'))
    delta = torch.full((L.call_int('p4p_attention_bwd_workspace', B, T, H),), float('nan'), dtype=torch.float32, device=DEV)
    part = None
    if COLSUM:
        rows = L.call_int('p4p_attention_colsum_rows', B, T)
        part = torch.zeros(rows * 3 * D, dtype=torch.float32, device=DEV)
    L.call('p4p_attention_bwd', L.ptr(qkv), L.ptr(o), L.ptr(do), L.ptr(lse), L.ptr(delta), L.ptr(dqkv), L.ptr(part),
           B, T, H, D, float((D // H) ** -0.5), st)
    torch.cuda.synchronize()
    return dqkv, delta

ref, dref = run()
print('nan in ref', int(torch.isnan(ref.float()).sum()), int(torch.isnan(dref).sum()))
nbad = 0
for r in range(RUNS):
    got, dgot = run()
    if torch.equal(got, ref) and torch.equal(dgot, dref):
        continue
    nbad += 1
    d = (got.float() - ref.float()).abs().reshape(B, T, 3, H, 64)
    nz = (d.amax(dim=4) > 0).nonzero()
    blocks = sorted({(int(b), int(h), int(w), int(t) // 16) for b, t, w, h in nz.tolist()})
    print('run', r, 'delta equal', bool(torch.equal(dgot, dref)), 'max diff', float(d.max()), 'elements', int((d > 0).sum()),
          '(b, head, q/k/v, 16-row tile):', blocks[:12], '... total', len(blocks))
print('differing runs', nbad, 'of', RUNS)
