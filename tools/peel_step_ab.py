"""Whole N=1024 / N=2048 training step with the wide GEMMs' tail row panel peeled (p4p_set_gemm_peel_rounds(32)) and not (the default),
interleaved in one process on one box.   usage (GPU box): python3 tools/peel_step_ab.py"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import p4p_amd
from p4p_amd import _lib
import bench
lib = _lib.lib()
dev = torch.device('cuda:0')
for points in (1024, 2048):
    args = types.SimpleNamespace(classes=40, dim=768, vit='vit_base_patch16_384', bucket_mb=None)
    st = bench.Stepper(args, dev, False)
    batch = bench.synthetic_batch(256, points, 40, 0, dev)
    res = {4: [], 32: []}
    for rnd in range(4):
        for v in (4, 32):
            lib.p4p_set_gemm_peel_rounds(v)
            for _ in range(2):
                st.step(*batch)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                st.step(*batch)
            torch.cuda.synchronize()
            res[v].append((time.perf_counter() - t0) * 100)
    lib.p4p_set_gemm_peel_rounds(4)
    print(f'N={points}: wide tails whole ' + ' '.join(f'{x:.2f}' for x in res[4]) + '   peeled ' + ' '.join(f'{x:.2f}' for x in res[32]) + ' ms / step', flush=True)
    del st, batch
    torch.cuda.empty_cache()
