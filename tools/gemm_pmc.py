import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import p4p_amd
from p4p_amd import ops, _lib
lib = _lib.lib()
M, D = 256 * 256, 768
g = torch.Generator().manual_seed(0)
a = torch.randn(M, 4 * D, generator=g).to(torch.bfloat16).cuda()
w = (torch.randn(D, 4 * D, generator=g) * 0.02).to(torch.bfloat16).cuda()
out = torch.empty(M, D, dtype=torch.bfloat16, device='cuda')
for v in (8, 10):
    lib.p4p_set_gemm_variant(v)
    for _ in range(3):
        ops.gemm_tn(a, w, ops.EPI_BF16, out=out)
    torch.cuda.synchronize()
gg = torch.randn(M, 4 * D, generator=g).to(torch.bfloat16).cuda()
aa = torch.randn(M, D, generator=g).to(torch.bfloat16).cuda()
for _ in range(3):
    ops.gemm_wgrad(gg, aa)
torch.cuda.synchronize()
