"""Where does a workgroup of the K/V-resident T = 513 attention forward (attention_fwd2.h) spend its cycles?  Builds a DIAGNOSTIC copy of the
attention kernels with s_memtime stamps (-DP4P_FW2_STAMPS; the product library has none), runs it and prints the phase shares.
    python tools/attn_fwd2_stamps.py            (GPU box; FW2_B=2 for a nearly empty chip)"""
import ctypes, os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, 'adapting-2d-vits-for-3d-point-cloud-understanding_amd', 'csrc')
so = '/tmp/libp4p_fw2stamps.so'
extra = [a for a in sys.argv[1:] if a.startswith('-D')]
subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '-shared', '--offload-arch=gfx950', '-ffp-contract=off',
                       '-fno-honor-nans', '-fno-slp-vectorize', '-Wno-unused-value', '-Wno-pass-failed', '-DP4P_FW2_STAMPS', '-I', CS] + extra +
                      [os.path.join(CS, 'attention.hip'), '-x', 'hip', os.path.join(CS, 'capi.cpp'), '-o', so])
lib = ctypes.CDLL(so)
lib.p4p_set_attention_fwd_variant(1 | int(os.environ.get('FW2_VARIANT', '0')))
B, T, H = int(os.environ.get('FW2_B', 256)), 513, int(os.environ.get('FW2_H', 12))
D = 64 * H
g = torch.Generator().manual_seed(0)
qkv = torch.randn(B * T, 3 * D, generator=g).to(torch.bfloat16).cuda()
out = torch.empty(B * T, D, dtype=torch.bfloat16, device='cuda')
lse = torch.zeros(B * H * T + B * H * 32, dtype=torch.float32, device='cuda')
ptr = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(3):
    assert lib.p4p_attention_fwd(ptr(qkv), ptr(out), ptr(lse), B, T, H, D, ctypes.c_float(0.125), st) == 0
torch.cuda.synchronize()
raw = lse[B * H * T:].view(torch.int64).view(B * H, 16).cpu().double()
rel = raw - raw[:, 0:1]
names = ['top of the pair', 'odd query tiles of waves 0-3', 'key tiles of half 0 (steps 0 .. 14)', 'barrier E + DMAs of the next pair, half 0', 'key tiles of half 1 + drain',
         'Q of the next pair, outputs, odd tiles of waves 4-7', 'barrier T, DMAs half 1, combine (end)']
print(f'mean s_memtime ticks since the pair started (thread 0), over {B * H} pairs:')
prev = 0.0
for k, n in enumerate(names):
    m = float(rel[:, k].mean())
    print(f'  {n:36s} {m:10.0f}   (+{m - prev:8.0f})')
    prev = m
print('kernel span', float(raw[:, 6].max() - raw[:, 0].min()), 'ticks (flags', ' '.join(extra), '); sum of pair lifetimes / 256 CUs', float(rel[:, 6].sum() / 256))
