"""Where does a workgroup of the fused attention backward (attention_fused.h) spend its cycles?  Builds a DIAGNOSTIC copy of the
attention kernels with s_memtime stamps (-DP4P_FZ_STAMPS; the product library has none), runs it and prints the phase shares.
python tools/attn_fused_stamps.py  (GPU box)"""
import ctypes, os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, 'adapting-2d-vits-for-3d-point-cloud-understanding_amd', 'csrc')
so = os.environ.get('P4P_STAMPS_LIB', '/tmp/libp4p_fzstamps.so')       # a prebuilt diagnostic library (tools/build_stamps.sh), or build here
if 'P4P_STAMPS_LIB' not in os.environ:
    extra = [a for a in sys.argv[1:] if a.startswith('-D')]
    subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '-shared', '--offload-arch=gfx950', '-ffp-contract=off',
                           '-fno-honor-nans', '-fno-slp-vectorize', '-Wno-unused-value', '-Wno-pass-failed', '-DP4P_FZ_STAMPS', '-I', CS] + extra +
                          [os.path.join(CS, 'attention.hip'), '-x', 'hip', os.path.join(CS, 'capi.cpp'), '-o', so])
lib = ctypes.CDLL(so)
B, T, H = int(os.environ.get('FZ_B', 256)), 257, int(os.environ.get('FZ_H', 12))      # FZ_B=2: 24 workgroups, the chip to themselves
D = 64 * H
g = torch.Generator().manual_seed(0)
qkv = torch.randn(B * T, 3 * D, generator=g).to(torch.bfloat16).cuda()
dout = torch.randn(B * T, D, generator=g).to(torch.bfloat16).cuda()
out = torch.empty(B * T, D, dtype=torch.bfloat16, device='cuda')
dqkv = torch.empty_like(qkv)
lse = torch.zeros(B * H * T, dtype=torch.float32, device='cuda')
delta = torch.zeros(B * H * T, dtype=torch.float32, device='cuda')
ptr = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
assert lib.p4p_attention_fwd(ptr(qkv), ptr(out), ptr(lse), B, T, H, D, ctypes.c_float(0.125), st) == 0
for _ in range(3):
    assert lib.p4p_attention_bwd(ptr(qkv), ptr(out), ptr(dout), ptr(lse), ptr(delta), ptr(dqkv), ctypes.c_void_p(0), B, T, H, D,
                                 ctypes.c_float(0.125), st) == 0
torch.cuda.synchronize()
raw = delta.view(torch.int64)[:B * H * 16].view(B * H, 16).cpu().double()
rel = raw - raw[:, 0:1]
names = ['start', 'loads issued, stats done', 'staging landed + barrier', 'K^T fragments + odd key', 'barrier', 'query-tile loop done', 'end']
print('mean s_memtime ticks since the workgroup started (wave 0), over', B * H, 'workgroups:')
for k, n in enumerate(names):
    print(f'  {n:28s} {float(rel[:, k].mean()):10.0f}')
print('kernel span', float(raw[:, 6].max() - raw[:, 0].min()), 'ticks; sum of lifetimes / 256 CUs', float(rel[:, 6].sum() / 256))

