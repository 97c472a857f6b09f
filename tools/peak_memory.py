"""Peak device memory of one N=1024 / N=2048 training step (batch 256, ViT-B/12) with the encoder backward's gradient slab stack freed early
(functional._FREE_DSTACK_EARLY, the default) and held until the embedding-stage backward is done; and the step time of both.
usage (GPU box): python3 tools/peak_memory.py"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import p4p_amd
from p4p_amd import functional as F
import bench

dev = torch.device('cuda:0')
for points in (1024, 2048):
    args = types.SimpleNamespace(classes=40, dim=768, vit='vit_base_patch16_384', bucket_mb=None)
    st = bench.Stepper(args, dev, False)
    batch = bench.synthetic_batch(256, points, 40, 0, dev)
    for early in (True, False, True, False):
        F._FREE_DSTACK_EARLY = early
        for _ in range(3):
            st.step(*batch)
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        t0 = time.perf_counter()
        for _ in range(10):
            st.step(*batch)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 100
        print(f'N={points} free-early={early}: peak {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB, {ms:.2f} ms / step', flush=True)
    del st, batch
    torch.cuda.empty_cache()
