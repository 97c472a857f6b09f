"""Determinism sweep of the statistics epilogues (EPI_BF16_ROWGROUP / EPI_BF16_STATS) of the 128x128 kernel (variant 1) and the automatic
choice (256x256 staggered kernel): N launches each, distinct outputs counted; also the plain bf16 epilogue of variant 1."""
import ctypes, os, sys, torch, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import p4p_amd
from p4p_amd import _lib
lib = _lib.lib()
ptr = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
M, N, K, grp = 76800, 512, 256, 16
g = torch.Generator().manual_seed(M)
a = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
w = (torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16).cuda()
rg = torch.randn(M // grp, N, generator=g).cuda()
bias = torch.randn(N, generator=g).cuda()
R = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for variant in (1, 0):
    lib.p4p_set_gemm_variant(variant)
    for epi in (4, 5, 0):
        outs = []
        out = torch.zeros(M, N, dtype=torch.bfloat16, device='cuda')
        stats = torch.zeros(M // 128 * 2 * N, device='cuda')
        ref = None
        bad = []
        for rep in range(R):
            out.fill_(7.0)
            rc = lib.p4p_gemm_tn_bf16(epi, ptr(a), ptr(w), ptr(out), ptr(bias if epi != 4 else None), ptr(None), ptr(None), ptr(rg if epi == 4 else None), ptr(stats if epi else None), M, N, K, grp if epi == 4 else 1, st)
            assert rc == 0
            torch.cuda.synchronize()
            if ref is None:
                ref = out.clone()
            else:
                d = (out.float() - ref.float()).abs()
                if float(d.max()) > 0:
                    nz = (d > 0).nonzero()
                    bad.append((rep, int((d > 0).sum()), float(d.max()), nz[0].tolist(), nz[-1].tolist()))
        print(f'variant {variant} epilogue {epi}: {len(bad)} of {R - 1} launches differ from the first', bad[:6], flush=True)
lib.p4p_set_gemm_variant(0)
