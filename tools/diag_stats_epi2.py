"""As tools/diag_stats_epi.py, but the way the test calls it: fresh output / statistics tensors per launch, the two epilogues alternating."""
import ctypes, os, sys, torch
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
R = int(sys.argv[1]) if len(sys.argv) > 1 else 30
ref, bad = {}, {}
for rep in range(R):
    for variant in (1, 0):
        lib.p4p_set_gemm_variant(variant)
        for epi in (4, 5):
            out = torch.zeros(M, N, dtype=torch.bfloat16, device='cuda')
            stats = torch.zeros(M // 128 * 2 * N, device='cuda')
            rc = lib.p4p_gemm_tn_bf16(epi, ptr(a), ptr(w), ptr(out), ptr(bias if epi == 5 else None), ptr(None), ptr(None), ptr(rg if epi == 4 else None), ptr(stats), M, N, K, grp if epi == 4 else 1, st)
            assert rc == 0
            torch.cuda.synchronize()
            if epi not in ref:
                ref[epi] = out.clone()
            else:
                d = (out.float() - ref[epi].float()).abs()
                if float(d.max()) > 0:
                    nz = (d > 0).nonzero()
                    bad.setdefault((variant, epi), []).append((rep, int((d > 0).sum()), float(d.max()), nz[0].tolist()))
lib.p4p_set_gemm_variant(0)
print('launches per (variant, epilogue):', R, 'differing from the first launch of that epilogue:', bad)
