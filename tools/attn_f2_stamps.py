"""Where does a workgroup of the fused T = 513 attention backward (attention_fused2.h) spend its cycles?  Builds a DIAGNOSTIC copy of
the attention kernels with s_memtime stamps (-DP4P_F2_STAMPS; the product library has none), runs it and prints the phase shares.
    python tools/attn_f2_stamps.py            (GPU box; F2_B=2 for a nearly empty chip; F2_CS=1 with the bias partial rows)"""
import ctypes, os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, 'adapting-2d-vits-for-3d-point-cloud-understanding_amd', 'csrc')
so = '/tmp/libp4p_f2stamps.so'
extra = [a for a in sys.argv[1:] if a.startswith('-D')]
subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '-shared', '--offload-arch=gfx950', '-ffp-contract=off',
                       '-fno-honor-nans', '-fno-slp-vectorize', '-Wno-unused-value', '-Wno-pass-failed', '-DP4P_F2_STAMPS', '-I', CS] + extra +
                      [os.path.join(CS, 'attention.hip'), '-x', 'hip', os.path.join(CS, 'capi.cpp'), '-o', so])
lib = ctypes.CDLL(so)
B, T, H = int(os.environ.get('F2_B', 256)), 513, int(os.environ.get('F2_H', 12))
D = 64 * H
g = torch.Generator().manual_seed(0)
qkv = torch.randn(B * T, 3 * D, generator=g).to(torch.bfloat16).cuda()
dout = torch.randn(B * T, D, generator=g).to(torch.bfloat16).cuda()
out = torch.empty(B * T, D, dtype=torch.bfloat16, device='cuda')
dqkv = torch.empty_like(qkv)
lse = torch.zeros(B * H * T, dtype=torch.float32, device='cuda')
lib.p4p_attention_bwd_workspace.restype = ctypes.c_longlong
park = int(lib.p4p_attention_bwd_workspace(B, T, H))        # the kernel parks dQ^T chains here; the stamps go behind them
delta = torch.zeros(park + B * H * 32, dtype=torch.float32, device='cuda')
cs = torch.zeros(B * 9 * 4 * 3 * D, dtype=torch.float32, device='cuda') if os.environ.get('F2_CS') else None
ptr = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
assert lib.p4p_attention_fwd(ptr(qkv), ptr(out), ptr(lse), B, T, H, D, ctypes.c_float(0.125), st) == 0
for _ in range(3):
    assert lib.p4p_attention_bwd(ptr(qkv), ptr(out), ptr(dout), ptr(lse), ptr(delta), ptr(dqkv), ptr(cs), B, T, H, D,
                                 ctypes.c_float(0.125), st) == 0
torch.cuda.synchronize()
raw = delta[park:].view(torch.int64)[:B * H * 16].view(B * H, 16).cpu().double()
rel = raw - raw[:, 0:1]
names = ['start', 'Q / dO DMAs issued, odd pair', 'images landed + barrier', 'delta + odd key phase', 'half 0: K chunks, K^T frags', 'half 0: 17 query tiles',
         'half 0: dK / dV out', 'half 1: K chunks, K^T frags', 'half 1: 17 query tiles', 'half 1: dK / dV out (end)']
print(f'mean s_memtime ticks since the pair started (thread 0), over {B * H} pairs (colsum {"on" if cs is not None else "off"}):')
prev = 0.0
for k, n in enumerate(names):
    m = float(rel[:, k].mean())
    print(f'  {n:32s} {m:10.0f}   (+{m - prev:8.0f})')
    prev = m
print('kernel span', float(raw[:, 9].max() - raw[:, 0].min()), 'ticks; sum of pair lifetimes / 256 CUs', float(rel[:, 9].sum() / 256))
