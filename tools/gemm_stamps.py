"""Where does a workgroup of the persistent GEMM (gemm_tn4p.h) spend its cycles?  Builds a DIAGNOSTIC copy of gemm.hip with s_memtime
stamps (-DP4P_T4P_STAMPS; the product library has none), runs the ViT-B shapes with their epilogues and prints, per shape, the mean
cycles of a tile's phases per wave group (group 0 = waves 0-3 leads by one barrier interval, group 1 = waves 4-7):
    tile start -> K loop start | K loop (12 K-tiles at K = 768) | catch-up barrier | epilogue (all its stores ISSUED) | seam wait
Extra -D flags go to the compiler (e.g. -DP4P_EPI_NO_STORES, -DP4P_EPI_NO_GELU: timing-only ablations of the epilogue).
    python tools/gemm_stamps.py [-D...]"""
import ctypes, os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, 'adapting-2d-vits-for-3d-point-cloud-understanding_amd', 'csrc')
so = '/tmp/libp4p_t4pstamps.so'
extra = [a for a in sys.argv[1:] if a.startswith('-D')]
subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '-shared', '--offload-arch=gfx950', '-ffp-contract=off',
                       '-Wno-unused-value', '-Wno-pass-failed', '-DP4P_T4P_STAMPS', '-I', CS] + extra +
                      [os.path.join(CS, 'gemm.hip'), os.path.join(CS, 'elementwise.hip'), '-x', 'hip', os.path.join(CS, 'capi.cpp'), '-o', so])
lib = ctypes.CDLL(so)
if os.environ.get('GS_RESERVED'):                      # fewer workgroups: is the epilogue's rate a per-CU limit or a shared (HBM) one?
    lib.p4p_set_reserved_cus(int(os.environ['GS_RESERVED']))
NWG = lib.p4p_persistent_grid_size()
M = 256 * 257 if not os.environ.get('GS_M') else int(os.environ['GS_M'])
g = torch.Generator().manual_seed(0)
ptr = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
stamps = torch.zeros(256 * 16 * 8 * 6, dtype=torch.int64, device='cuda')
print('persistent grid', NWG, 'workgroups')
assert lib.p4p_debug_set_t4p_stamps(ptr(stamps)) == 0
shapes = [('qkv  (bf16)', 0, 2304, 768), ('fc1  (GELU + gelu\')', 1, 3072, 768), ('proj (f32 + residual)', 2, 768, 768), ('dfc2 (dGELU + colsum)', 3, 3072, 768),
          ('dqkv (bf16, K=2304)', 0, 768, 2304)]
for name, epi, N, K in shapes:
    a = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16).cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).cuda()
    out = torch.empty(M, N, dtype=torch.float32 if epi == 2 else torch.bfloat16, device='cuda')
    bias = torch.randn(N, generator=g).cuda()
    resid = torch.randn(M, N, generator=g).cuda() if epi == 2 else None
    pre = (torch.rand(M, N, generator=g)).to(torch.bfloat16).cuda() if epi in (1, 3) else None
    cs = torch.zeros(2 * 257 * N, device='cuda') if epi == 3 else None
    def run():
        rc = lib.p4p_gemm_tn_bf16(epi, ptr(a), ptr(w), ptr(out), ptr(bias) if epi != 3 else ptr(None), ptr(resid), ptr(pre), ptr(None), ptr(cs), M, N, K, 1, st)
        assert rc == 0, lib.p4p_last_error()
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    stamps.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    raw = stamps.view(256, 16, 8, 6)[:NWG].cpu().double()
    tiles_per_wg = (M // 256) * (N // 256) // NWG
    use = raw[:, 1:tiles_per_wg - 1]                     # steady-state tiles of the walk (not the first: fresh prologue; not the last: no seam)
    print(f'{name}: M={M} N={N} K={K}, {tiles_per_wg} tiles per workgroup, {e0.elapsed_time(e1) * 1e3:.1f} us (stamped build); mean cycles per tile:')
    for grp, waves in (('group 0', slice(0, 4)), ('group 1', slice(4, 8))):
        u = use[:, :, waves]
        d = [float((u[..., k + 1] - u[..., k]).mean()) for k in range(5)]
        nxt = float((raw[:, 2:tiles_per_wg, waves, 0] - raw[:, 1:tiles_per_wg - 1, waves, 5]).mean())
        tot = float((raw[:, 2:tiles_per_wg, waves, 0] - raw[:, 1:tiles_per_wg - 1, waves, 0]).mean())
        print(f'   {grp}: prologue {d[0]:7.0f} | K loop {d[1]:7.0f} | catch-up {d[2]:6.0f} | epilogue {d[3]:7.0f} | seam wait {d[4]:6.0f} | to next tile {nxt:5.0f} | tile {tot:7.0f}')
    del a, w, out
