"""Fused attention backward (attention_fused.h) against the two-kernel resident path: results, then time, in one process.
p4p_set_attention_variant(8) switches the fused kernel off."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import p4p_amd
from p4p_amd import ops, _lib
if os.environ.get('P4P_DEBUG_LIB'):          # A/B of two builds of the library in one gpurun call
    _lib.LIB_PATH = os.path.abspath(os.environ['P4P_DEBUG_LIB'])
lib = _lib.lib()


def ref_grad(qkv, dout, B, T, H):
    D = qkv.shape[1] // 3
    qr = qkv.float().requires_grad_(True)
    x = qr.reshape(B, T, 3, H, D // H).permute(2, 0, 3, 1, 4)
    a = ((x[0] @ x[1].transpose(-2, -1)) * (D // H) ** -0.5).softmax(-1)
    (a @ x[2]).transpose(1, 2).reshape(B * T, D).backward(dout.float())
    return qr.grad


ok = True
for B, T, H in [(2, 65, 3), (1, 129, 1), (3, 129, 2), (2, 257, 3), (5, 257, 12)]:
    D = 64 * H
    g = torch.Generator().manual_seed(T + B)
    qkv = (torch.randn(B * T, 3 * D, generator=g) * 1.5).to(torch.bfloat16).cuda()
    dout = torch.randn(B * T, D, generator=g).to(torch.bfloat16).cuda()
    out, lse = ops.attention_fwd(qkv, B, T, H)
    lib.p4p_set_attention_variant(8)
    old, old_cs = ops.attention_bwd(qkv, out, dout, lse, B, T, H, want_colsum=True)
    old, old_cs = old.clone(), old_cs.clone()
    lib.p4p_set_attention_variant(128)               # a workgroup per pair (the default walks the pairs: one workgroup per CU)
    per, per_cs = ops.attention_bwd(qkv, out, dout, lse, B, T, H, want_colsum=True)
    per, per_cs = per.clone(), per_cs.clone()
    lib.p4p_set_attention_variant(0)
    new, new_cs = ops.attention_bwd(qkv, out, dout, lse, B, T, H, want_colsum=True)
    torch.cuda.synchronize()
    print('workgroup per pair == persistent walk, bit for bit:', torch.equal(per, new), torch.equal(per_cs, new_cs), flush=True)
    ref = ref_grad(qkv, dout, B, T, H)
    sc = max(ref.abs().max().item(), 1.0)
    e_old, e_new = (old.float() - ref).abs().max().item() / sc, (new.float() - ref).abs().max().item() / sc
    d = (new.float() - old.float()).abs()
    Dq = d[:, :D].max().item(); Dk = d[:, D:2 * D].max().item(); Dv = d[:, 2 * D:].max().item()
    last = d.view(B, T, 3 * D)[:, T - 1].max().item()
    cs_want = new.double().sum(0)
    cs_err = (new_cs.double() - cs_want).abs().max().item() / max(new.float().abs().sum(0).max().item(), 1.0)
    good = e_new <= 3e-2 and cs_err <= 1e-5 and torch.isfinite(new.float()).all().item()
    ok &= good
    print(f'B={B} T={T} H={H}: err vs fp32 autograd old {e_old:.2e} new {e_new:.2e} | new-old dq {Dq:.2e} dk {Dk:.2e} dv {Dv:.2e} '
          f'last row {last:.2e} | colsum err {cs_err:.1e} | {"ok" if good else "FAIL"}', flush=True)
if not ok:
    sys.exit(1)

B, T, H, D = 256, int(os.environ.get('AT_T', 257)), 12, 768
g = torch.Generator().manual_seed(0)
qkv = torch.randn(B * T, 3 * D, generator=g).to(torch.bfloat16).cuda()
dout = torch.randn(B * T, D, generator=g).to(torch.bfloat16).cuda()
out, lse = ops.attention_fwd(qkv, B, T, H)


def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n


for rnd in range(3):
    for v, name in ((8, 'two-kernel'), (128, 'fused, a workgroup per pair'), (0, 'fused, persistent walk'), (32, 'persistent, no shift')):
        lib.p4p_set_attention_variant(v)
        a = timeit(lambda: ops.attention_bwd(qkv, out, dout, lse, B, T, H))
        b = timeit(lambda: ops.attention_bwd(qkv, out, dout, lse, B, T, H, want_colsum=True))
        print(f'round {rnd} {name:24s}: bwd {a:7.1f} us   with colsum {b:7.1f} us', flush=True)
lib.p4p_set_attention_variant(0)
ref = ops.attention_bwd(qkv, out, dout, lse, B, T, H, want_colsum=True)
ref = (ref[0].clone(), ref[1].clone())
bad = 0
for r in range(20):
    got = ops.attention_bwd(qkv, out, dout, lse, B, T, H, want_colsum=True)
    bad += int(not (torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])))
print('bitwise differing runs of 20:', bad)
