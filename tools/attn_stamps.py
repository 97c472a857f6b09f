"""Where does a workgroup of the resident attention forward spend its cycles?  Builds a DIAGNOSTIC copy of the attention kernels
with s_memtime stamps (-DP4P_LONE_STAMPS; the product library has none), runs it and prints the phase shares.
python tools/attn_stamps.py  (GPU box)"""
import ctypes, os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, 'adapting-2d-vits-for-3d-point-cloud-understanding_amd', 'csrc')
so = '/tmp/libp4p_stamps.so'
subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '-shared', '--offload-arch=gfx950', '-ffp-contract=off',
                       '-fno-honor-nans', '-fno-slp-vectorize', '-Wno-unused-value', '-Wno-pass-failed', '-DP4P_LONE_STAMPS', '-I', CS,
                       os.path.join(CS, 'attention.hip'), '-x', 'hip', os.path.join(CS, 'capi.cpp'), '-o', so])
lib = ctypes.CDLL(so)
B, T, H = 256, 257, 12
D = 64 * H
g = torch.Generator().manual_seed(0)
qkv = torch.randn(B * T, 3 * D, generator=g).to(torch.bfloat16).cuda()
out = torch.empty(B * T, D, dtype=torch.bfloat16, device='cuda')
nw = B * H * 4
lse = torch.zeros(B * H * T + 2 * nw * 8 + 8, dtype=torch.float32, device='cuda')
ptr = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(3):
    rc = lib.p4p_attention_fwd(ptr(qkv), ptr(out), ptr(lse), B, T, H, D, ctypes.c_float(0.125), st)
    assert rc == 0
torch.cuda.synchronize()
raw = lse[B * H * T:B * H * T + 2 * nw * 8].view(torch.int64).view(B * H, 4, 8).cpu().double()
t0 = raw[..., 0:1]
rel = raw - t0
names = ['start', 'loads issued', 'staging landed + barrier', 'first tile done', 'all tiles done', 'last-row partial', 'barrier', 'end']
print('mean cycles since the wave started (s_memtime ticks), over', B * H * 4, 'waves:')
for k, n in enumerate(names):
    print(f'  {n:28s} {float(rel[..., k].mean()):10.0f}   (wave 0: {float(rel[:, 0, k].mean()):8.0f}, wave 3: {float(rel[:, 3, k].mean()):8.0f})')
life = (raw[..., 7].max(dim=1).values - raw[..., 0].min(dim=1).values)
print('workgroup lifetime mean', float(life.mean()), 'ticks; kernel span', float(raw[..., 7].max() - raw[..., 0].min()), 'ticks')

# ---- streaming forward (T = 513): wait at the stage fence vs tile math, per wave
B, T, H = 256, 513, 12
qkv = torch.randn(B * T, 3 * D, generator=g).to(torch.bfloat16).cuda()
out = torch.empty(B * T, D, dtype=torch.bfloat16, device='cuda')
nblk = (T + 127) // 128
nwg = ((B * H + 7) // 8) * 8 * nblk
lse = torch.zeros(B * H * T + 2 * nwg * 4 * 4 + 8, dtype=torch.float32, device='cuda')
for _ in range(3):
    assert lib.p4p_attention_fwd(ptr(qkv), ptr(out), ptr(lse), B, T, H, D, ctypes.c_float(0.125), st) == 0
torch.cuda.synchronize()
raw = lse[B * H * T:B * H * T + 2 * nwg * 4 * 4].view(torch.int64).view(nwg * 4, 4).cpu().double()
act = raw[:, 3] > 0
print(f'streaming forward T={T}: active waves {int(act.sum())} of {raw.shape[0]}')
for name, sel in (('active waves', act), ('waves without a valid query', ~act & (raw[:, 2] > 0))):
    r = raw[sel]
    if len(r):
        print(f'  {name:30s} fence wait {float(r[:, 0].mean()):8.0f}  tile math {float(r[:, 1].mean()):8.0f}  whole loop+prologue {float(r[:, 2].mean()):8.0f} cycles')
