"""Two builds of libp4p_hip.so, one process, interleaved rounds: every ViT-B GEMM of the step (with its epilogue) timed on both and the
outputs compared bit for bit.      python tools/gemm_lib_ab.py <other.so> [M]
(the other build is loaded through ctypes next to the product library; the operands are random bf16)"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import p4p_amd
from p4p_amd import _lib
new = _lib.lib()
old = ctypes.CDLL(os.path.abspath(sys.argv[1]))
M = int(sys.argv[2]) if len(sys.argv) > 2 else 256 * 257
D = 768
ptr = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
g = torch.Generator().manual_seed(0)
shapes = [('qkv', 0, 3 * D, D, 0), ('proj (resid)', 2, D, D, 1), ('fc1 (GELU)', 1, 4 * D, D, 0), ('fc2 (resid + pos)', 2, D, 4 * D, 2), ('dfc2 (dGELU)', 3, 4 * D, D, 0),
          ('dfc1', 0, D, 4 * D, 0), ('dproj', 0, D, D, 0), ('dqkv', 0, D, 3 * D, 0)]
for name, epi, N, K, nres in shapes:
    a = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16).cuda()
    bias = torch.randn(N, generator=g).cuda()
    r1 = torch.randn(M, N, generator=g).cuda() if nres >= 1 else None
    r2 = torch.randn(M, N, generator=g).cuda() if nres >= 2 else None
    pre = torch.rand(M, N, generator=g).to(torch.bfloat16).cuda() if epi in (1, 3) else None
    cs = torch.zeros(2 * ((M + 255) // 256) * N, device='cuda') if epi == 3 else None
    outs, times = {}, {'old': [], 'new': []}
    for tag, lib in (('old', old), ('new', new)):
        out = torch.empty(M, N, dtype=torch.float32 if epi == 2 else torch.bfloat16, device='cuda')
        p2 = torch.empty_like(pre) if epi == 1 else pre
        outs[tag] = (out, p2)
    def run(tag, lib):
        out, p2 = outs[tag]
        rc = lib.p4p_gemm_tn_bf16(epi, ptr(a), ptr(w), ptr(out), ptr(None if epi == 3 else bias), ptr(r1), ptr(p2), ptr(r2), ptr(cs), M, N, K, 1, st)
        assert rc == 0
    for rnd in range(5):
        for tag, lib in (('old', old), ('new', new)):
            run(tag, lib); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): run(tag, lib)
            e1.record(); torch.cuda.synchronize()
            times[tag].append(e0.elapsed_time(e1) * 100)
    same = torch.equal(outs['old'][0], outs['new'][0]) and (epi != 1 or torch.equal(outs['old'][1], outs['new'][1]))
    fmt = lambda ts: f'{min(ts):7.1f} (median {sorted(ts)[len(ts) // 2]:7.1f})'
    print(f'{name:18s} M={M} N={N} K={K}: old {fmt(times["old"])} us   new {fmt(times["new"])} us   {"bit-identical" if same else "OUTPUTS DIFFER"}', flush=True)
    assert same, name
