"""One-GPU rehearsal of the CU contention between this library's one-workgroup-per-CU persistent kernels and a long-lived
foreign kernel (RCCL's all-reduce workgroups in a data-parallel backward).

r stand-in workgroups (p4p_debug_occupy_cus: one wave, 96 KiB of LDS each -> one per CU, and no 160-KiB workgroup of ours fits
beside it) are parked on a second stream for the whole timed region; the training step of bench.py (N=1024, ViT-B/12, batch 256
by default) is timed with
    none     : persistent grids of ALL CUs (the world-size-1 default): r of their workgroups start late with a whole share to do
    reserve  : p4p_set_reserved_cus(r): persistent grids of CUs - r
    per-tile : p4p_set_reserved_cus(-1): a workgroup per tile / pair, the hardware dispatcher balances (build_reducer's default
               at world > 1)
for r in 0, 8, 16, 32.  Prints one JSON line per (r, mode) and a markdown table at the end.

    python tools/contention_rehearsal.py [--steps 8] [--points 1024] [--batch 256] [--rs 0,8,16,32]
"""
import argparse
import json
import os
import sys
import time

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--points', type=int, default=1024)
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--vit', default='vit_base_patch16_384')
    ap.add_argument('--rs', default='0,8,16,32')
    a = ap.parse_args()
    from p4p_amd import _lib, ops
    from p4p_amd.vit import parse_model_name
    vd, vl, vh, _, _ = parse_model_name(a.vit)
    args = argparse.Namespace(classes=40, dim=vd, vit=a.vit, bucket_mb=None, batch=a.batch, points=a.points)
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    stepper = bench.Stepper(args, dev, False)
    batch = bench.synthetic_batch(a.batch, a.points, 40, 1234, dev)
    for _ in range(3):
        stepper.step(*batch)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    ncu = _lib.call_int('p4p_device_cu_count')
    rows = []
    for r in [int(x) for x in a.rs.split(',')]:
        for mode in ('none', 'reserve', 'per-tile'):
            if r == 0 and mode == 'reserve':
                continue
            ops.set_reserved_cus({'none': 0, 'reserve': r, 'per-tile': -1}[mode])
            for _ in range(2):
                stepper.step(*batch)
            torch.cuda.synchronize()
            # one untimed step gives the duration to park the stand-ins for (bounded: they leave by themselves)
            t0 = time.perf_counter()
            stepper.step(*batch)
            torch.cuda.synchronize()
            one = time.perf_counter() - t0
            park_us = int(min(4.5e6, (a.steps + 1.5) * one * 2.2e6))
            if r > 0:
                _lib.call('p4p_debug_occupy_cus', r, 96 * 1024, _lib.i64(park_us), _lib.ctypes.c_void_p(side.cuda_stream))
                time.sleep(0.01)                       # the stand-ins are resident before the first kernel of the step arrives
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.steps):
                stepper.step(*batch)
            e1.record()
            e1.synchronize()
            ms = e0.elapsed_time(e1) / a.steps
            still = not side.query() if r > 0 else True      # the stand-ins must have outlived the timed region
            torch.cuda.synchronize()
            row = {'r': r, 'mode': mode, 'grid': ops.persistent_grid_size(), 'cus': ncu, 'ms_per_step': round(ms, 3),
                   'standins_resident_throughout': bool(still), 'points': a.points, 'batch': a.batch}
            print(json.dumps(row), flush=True)
            rows.append(row)
    ops.set_reserved_cus(0)
    print('\n| r | all CUs (no remedy) | grid = CUs - r | a workgroup per tile / pair |\n|---|---|---|---|')
    for r in sorted({x['r'] for x in rows}):
        get = lambda m: next((f"{x['ms_per_step']:.2f}{'' if x['standins_resident_throughout'] else ' (!)'}" for x in rows if x['r'] == r and x['mode'] == m), '-')
        print(f"| {r} | {get('none')} | {get('reserve')} | {get('per-tile')} |")


if __name__ == '__main__':
    main()
