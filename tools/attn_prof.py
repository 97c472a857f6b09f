import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import p4p_amd
from p4p_amd import ops
B, T, H, D = 256, int(os.environ.get('AT_T', 257)), 12, 768
g = torch.Generator().manual_seed(0)
qkv = torch.randn(B * T, 3 * D, generator=g).to(torch.bfloat16).cuda()
dout = torch.randn(B * T, D, generator=g).to(torch.bfloat16).cuda()
for _ in range(8):
    out, lse = ops.attention_fwd(qkv, B, T, H)
    dqkv = ops.attention_bwd(qkv, out, dout, lse, B, T, H)
torch.cuda.synchronize()
