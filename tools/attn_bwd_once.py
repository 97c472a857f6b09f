"""A few launches of the attention backward at the benchmark's shape (for rocprofv3 --pmc passes).  AT_VARIANT as tools/attn_ab.py."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import p4p_amd
from p4p_amd import ops, _lib
_lib.lib().p4p_set_attention_variant(int(os.environ.get('AT_VARIANT', '0')))
B, T, H, D = 256, int(os.environ.get('AT_T', 257)), 12, 768
g = torch.Generator().manual_seed(0)
qkv = torch.randn(B * T, 3 * D, generator=g).to(torch.bfloat16).cuda()
dout = torch.randn(B * T, D, generator=g).to(torch.bfloat16).cuda()
out, lse = ops.attention_fwd(qkv, B, T, H)
for _ in range(4):
    ops.attention_bwd(qkv, out, dout, lse, B, T, H, want_colsum=True)
torch.cuda.synchronize()
