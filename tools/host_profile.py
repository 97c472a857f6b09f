"""Where the HOST time of a training step goes (cProfile over a few steps; the device is drained before and after).
    python tools/host_profile.py [--points 8192 --batch 8 --vit vit_large_patch16_384] [--steps 5] [--top 45]
Prints wall / host-enqueue time per step and the profile sorted by own time."""
import argparse, cProfile, io, os, pstats, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument('--points', type=int, default=8192)
ap.add_argument('--batch', type=int, default=8)
ap.add_argument('--vit', default='vit_large_patch16_384')
ap.add_argument('--steps', type=int, default=5)
ap.add_argument('--top', type=int, default=45)
a = ap.parse_args()
from p4p_amd.vit import parse_model_name
vd, vl, vh, _, _ = parse_model_name(a.vit)
args = argparse.Namespace(classes=40, dim=vd, vit=a.vit, bucket_mb=None, batch=a.batch, points=a.points)
dev = torch.device('cuda', 0)
st = bench.Stepper(args, dev, False)
batch = bench.synthetic_batch(a.batch, a.points, 40, 1, dev)
for _ in range(3):
    st.step(*batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    st.step(*batch)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'un-profiled: host enqueue {(t1 - t0) / a.steps * 1e3:.2f} ms/step, wall {(t2 - t0) / a.steps * 1e3:.2f} ms/step')
pr = cProfile.Profile()
pr.enable()
for _ in range(a.steps):
    st.step(*batch)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(a.top)
print(s.getvalue())
