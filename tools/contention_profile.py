"""Per-kernel view of the CU-contention rehearsal (tools/contention_rehearsal.py): ONE configuration (r stand-in workgroups parked,
the library on grids of CUs - r, or r = 0 and nothing parked), a few steps, meant to run under rocprofv3 --kernel-trace --stats:

    rocprofv3 --kernel-trace --stats -d gpurun_out/cp_r8 -o cp -- python3 tools/contention_profile.py --r 8
    python tools/prof_summary.py gpurun_out/cp_r8/.../cp_kernel_stats.csv <steps>

so that the per-kernel averages of r = 0 and r = 8 / 32 can be put side by side: which kernels pay for the reserved CUs?"""
import argparse
import os
import sys
import time

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--r', type=int, default=8)
    ap.add_argument('--steps', type=int, default=6)
    ap.add_argument('--no-park', action='store_true', help='grids of CUs - r but nothing parked (what the reservation alone costs)')
    a = ap.parse_args()
    from p4p_amd import _lib, ops
    from p4p_amd.vit import parse_model_name
    vd, vl, vh, _, _ = parse_model_name('vit_base_patch16_384')
    args = argparse.Namespace(classes=40, dim=vd, vit='vit_base_patch16_384', bucket_mb=None, batch=256, points=1024)
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    stepper = bench.Stepper(args, dev, False)
    batch = bench.synthetic_batch(256, 1024, 40, 1234, dev)
    ops.set_reserved_cus(a.r)
    for _ in range(3):
        stepper.step(*batch)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    if a.r > 0 and not a.no_park:
        _lib.call('p4p_debug_occupy_cus', a.r, 96 * 1024, int(min(4.5e6, (a.steps + 2) * 0.12e6)), _lib.ctypes.c_void_p(side.cuda_stream))
        time.sleep(0.01)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.steps):
        stepper.step(*batch)
    e1.record()
    e1.synchronize()
    print(f'r = {a.r}, parked = {a.r > 0 and not a.no_park}: {e0.elapsed_time(e1) / a.steps:.2f} ms per step, stand-ins still resident: {not side.query()}')
    torch.cuda.synchronize()
    ops.set_reserved_cus(0)


if __name__ == '__main__':
    main()
