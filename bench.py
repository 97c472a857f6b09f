"""Contract benchmark: point-clouds/sec of one Pix4Point training step on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): ModelNet40-shaped synthetic clouds N=1024, Pix4Point ViT-B/12
(D=768, H=12), k=16, batch 256 per GPU, random-init weights, bf16 MFMA operands / fp32 accumulate.
A step = zero_grad + forward + cross-entropy + backward (+ RCCL gradient all-reduce when N>1) +
clip_grad_norm_ + AdamW, i.e. reference src/train/pix4point_trainer.py:162-167, inputs resident in HBM.
One JSON line on rank 0; `roofline` is measured in a separate instrumented pass after the timed region
(HIP events around every MFMA GEMM launch on the launch stream); `cpu_baseline` times the CPU oracle
(fp32 torch restatement of the reference) on the host cores for a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0        # MI355X dense bf16 MFMA (guides/MI355X_MICROARCH.md)
def flop_per_cloud(points, D, L, k=16):
    """fwd+bwd algorithmic FLOPs per cloud, SURVEY.md section 8d: 3 x (S k 920,576 + L (24 T D^2 + 4 T^2 D))."""
    S = points // 4
    T = S + 1
    return 3.0 * (S * k * 920576.0 + L * (24.0 * T * D * D + 4.0 * T * T * D))


def apf_flop_per_cloud(G=196, gs=32, E=768, L=12, bottleneck=64, cin=8):
    """fwd+bwd algorithmic FLOPs per cloud of the APF path (BASELINE.json configs[3]).  Encoder as the reference computes
    it (src/models/apf.py:129-143: every point meets the full 2E x 2E layer; cin = 8 input channels = the shipped
    in_channels=4 with the height column, train_apf.py:5), trainable => 3x forward; transformer with
    T = G tokens and the bottleneck adapter, frozen (apf.py:335-346) => forward + input gradient = 2x forward."""
    enc = G * gs * 2.0 * (cin * 256 + 256 * 512 + 512 * E + 2 * E * 2 * E + 2 * E * E)
    vit = L * (24.0 * G * E * E + 4.0 * G * G * E + 4.0 * G * E * bottleneck)
    return 3.0 * enc + 2.0 * vit


def log(msg):
    print(f'[bench] {msg}', file=sys.stderr, flush=True)


def usable_cores():
    """Cores this process may really use: the affinity mask capped by the cgroup CPU quota (a GPU box gives
    one GPU's share of the host, not the whole machine)."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            txt = open(path).read().split()
            if path.endswith('cpu.max'):
                if txt[0] != 'max':
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
                    n = min(n, max(1, q // per))
        except Exception:
            pass
    return max(1, min(n, int(os.environ.get('P4P_CPU_THREADS', '16'))))


def synthetic_batch(B, N, num_classes, seed, device):
    g = torch.Generator().manual_seed(seed)
    p = torch.rand(B, N, 3, generator=g) * 2 - 1
    p = p - p.mean(dim=1, keepdim=True)
    p = p / p.norm(dim=2).max(dim=1).values.view(B, 1, 1)
    labels = torch.randint(0, num_classes, (B,), generator=g)
    start = torch.randint(0, N, (B,), generator=g)
    return p.contiguous().to(device), labels.to(device), start.to(device)


class Stepper:
    def __init__(self, args, device, distributed):
        import p4p_amd
        from p4p_amd.ddp import broadcast_parameters, build_reducer
        torch.manual_seed(1234)
        import contextlib
        with contextlib.redirect_stdout(sys.stderr):      # the constructor prints (as the reference's does)
            self.model = p4p_amd.Pix4Point(num_classes=args.classes, embed_dim=args.dim, pretrained_model=args.vit,
                                           pretrained=False, k_neighbors=16).to(device)
        self.model.train()
        broadcast_parameters(self.model)
        self.reducer = build_reducer(self.model, bucket_mb=args.bucket_mb) if distributed else None
        from p4p_amd.optim import FusedAdamW
        self.opt = FusedAdamW(self.model.get_param_groups(), lr=5e-4, weight_decay=5e-2)
        self.fused_opt = isinstance(self.opt, FusedAdamW)
        self.crit = torch.nn.CrossEntropyLoss()

    def step(self, pts, labels, start, optim=True):
        """optim=False: the metric's literal 'fwd+bwd' (zero_grad + forward + CE + backward + gradient all-reduce),
        reference src/train/pix4point_trainer.py:162-165; optim=True adds clip_grad_norm_ + AdamW (:166-167)."""
        if self.reducer is not None:
            self.reducer.zero_grad()
        else:
            self.opt.zero_grad(set_to_none=True)
        loss = self.crit(self.model(pts, start), labels)
        loss.backward()
        scale = 1.0
        if self.reducer is not None:
            self.reducer.finish(average=not self.fused_opt)     # fused: sums stay, 1/world rides in the clip coefficient
            scale = self.reducer.grad_scale if self.fused_opt else 1.0
        if not optim:
            return loss
        if self.fused_opt:
            self.opt.step(max_norm=10, grad_scale=scale)        # clip_grad_norm_ + AdamW in three launches (p4p_amd.optim)
        else:
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), max_norm=10, foreach=True)
            self.opt.step()
        return loss


class ApfStepper:
    """One APF training step as the reference's APFTrainer runs it (src/train/apf_trainer.py:164-171,260-275): zero_grad,
    forward (train mode: DropPath, adapter / head dropout, batch-statistics BatchNorm), CrossEntropy with label smoothing
    0.3, backward, AdamW over model.parameters() (no gradient clipping on this path)."""

    def __init__(self, args, device, distributed):
        import contextlib
        import p4p_amd
        from p4p_amd.ddp import GradReducer, broadcast_parameters
        from p4p_amd.optim import FusedAdamW
        torch.manual_seed(1234)
        with contextlib.redirect_stdout(sys.stderr):
            self.model = p4p_amd.AdaptPointFormer(num_classes=args.classes, vit_name=args.vit, pretrained=False,
                                                  in_channels=4).to(device)     # reference train_apf.py:5: xyz + height
        self.model.train()
        broadcast_parameters(self.model)
        self.reducer = (GradReducer([[p for p in self.model.parameters() if p.requires_grad]], outside=list(self.model.parameters()))
                        if distributed else None)
        self.opt = FusedAdamW(self.model.parameters(), lr=5e-4, weight_decay=5e-2)
        self.crit = torch.nn.CrossEntropyLoss(label_smoothing=0.3)

    def step(self, pts, labels, start, optim=True):
        if self.reducer is not None:
            self.reducer.zero_grad()
        else:
            self.opt.zero_grad(set_to_none=True)
        loss = self.crit(self.model(pts, start), labels)
        loss.backward()
        if self.reducer is not None:
            self.reducer.finish(average=False)
        if optim:
            self.opt.step(grad_scale=self.reducer.grad_scale if self.reducer is not None else 1.0)
        return loss


PROF_KINDS = ((1, 'gemm_wgrad3_kernel'), (2, 'gemm_tn (all tile configs / epilogues)'),
              (3, 'attention forward'), (4, 'attention backward (dQ + dK/dV launches)'))


def _pmc_traffic(kernel_tag):
    """HBM-side bytes per launch of the dominant kernel.  PMC counters need rocprofv3 AROUND the process, so this is the
    committed measurement of this same command (profiles/rNN_pmc_<tag>.json: FETCH_SIZE x2 + WRITE_SIZE, separate --pmc
    passes, as MI355X_MICROARCH.md prescribes), newest round first; None when no such file exists."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', f'r*_pmc_{kernel_tag}.json')), reverse=True)
    for f in files:
        with open(f) as fh:
            j = json.load(fh)
        return (j['fetch_bytes_per_launch'] + j['write_bytes_per_launch'],
                f'bytes/launch, rocprofv3 PMC ({os.path.relpath(f, ROOT)}), not collected in this run')
    return None, None


def live_pmc_traffic(kernel_substr, timeout_s=150):
    """HBM-side bytes per launch of one kernel, collected NOW: two child runs of this script under rocprofv3 (`--pmc FETCH_SIZE`,
    then `--pmc WRITE_SIZE`: the two do not fit one pass; counters alone with --kernel-trace, as the pool's rules ask), one warm-up
    + one step each, corrected as MI355X_MICROARCH.md prescribes for gfx950 (the counters are in KB; FETCH_SIZE reports half of
    the bytes of wide coalesced reads).  Returns (bytes, note) or (None, reason)."""
    import csv, glob, shutil, subprocess, tempfile
    if shutil.which('rocprofv3') is None:
        return None, 'rocprofv3 not found'
    tmp = tempfile.mkdtemp(prefix='p4p_pmc_', dir='/tmp')
    res = {}
    try:
        for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
            out = os.path.join(tmp, ctr)
            cmd = ['rocprofv3', '--pmc', ctr, '--kernel-trace', '--output-format', 'csv', '-d', out, '--',
                   sys.executable, os.path.abspath(__file__), '--steps', '1', '--warmup', '1', '--no-cpu-baseline', '--no-roofline',
                   '--no-north-star', '--no-pmc']
            env = dict(os.environ, TMPDIR='/tmp')
            # own session: on a timeout the whole group goes (rocprofv3 AND the child python that owns the GPU), and is waited
            # for, so that nothing of it runs beside a later timed section
            proc = subprocess.Popen(cmd, cwd='/tmp', env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, start_new_session=True)
            try:
                proc.communicate(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                import signal
                try:
                    os.killpg(proc.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
                proc.communicate()
                return None, f'rocprofv3 --pmc {ctr} timed out after {timeout_s} s (process group killed)'
            r = proc
            files = glob.glob(os.path.join(out, '**', '*counter_collection.csv'), recursive=True)
            if r.returncode != 0 or not files:
                return None, f'rocprofv3 --pmc {ctr} failed (rc {r.returncode})'
            tot, n = 0.0, 0
            with open(files[0]) as fh:
                for row in csv.DictReader(fh):
                    if row.get('Counter_Name') == ctr and kernel_substr in row.get('Kernel_Name', ''):
                        tot += float(row['Counter_Value'])
                        n += 1
            if n == 0:
                return None, f'{kernel_substr} not in the {ctr} pass'
            res[ctr] = (tot / n * 1024.0, n)
    except Exception as e:                               # a profiler that is not usable here must not cost the benchmark line
        return None, f'{type(e).__name__}: {e}'
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    fetch, write = 2.0 * res['FETCH_SIZE'][0], res['WRITE_SIZE'][0]
    return fetch + write, (f'bytes/launch collected in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in two child passes of this '
                           f'script ({res["FETCH_SIZE"][1]} launches each), FETCH x2 (gfx950) = {fetch * 1e-6:.0f} MB + {write * 1e-6:.0f} MB written')


def instrumented_roofline(stepper, batch, steps, measure=True, attn_dims=None):
    """Rooflines measured live: the library brackets exactly the launches of one kind with HIP events on the launch
    stream (include/p4p_hip.h: p4p_prof_enable / p4p_prof_collect) during `steps` extra training steps per kind;
    algorithmic FLOPs = 2 M N K per GEMM launch, 4 B H T^2 64 per attention forward (x 2.5 backward).  The headline entry
    is the kernel with the largest total time (the weight-gradient GEMM main kernel, as in profiles/)."""
    import ctypes
    from p4p_amd import _lib
    lib = _lib.lib()
    res = {}
    for kind, name in PROF_KINDS:
        if measure:
            lib.p4p_prof_enable(kind)
        for _ in range(steps):                      # EVERY rank takes these steps: they contain the gradient all-reduce
            stepper.step(*batch)
        torch.cuda.synchronize()
        if not measure:
            continue
        out = (ctypes.c_double * 3)()
        rc = lib.p4p_prof_collect(out)
        assert rc == 0
        res[kind] = {'name': name, 'ms': out[0], 'launches': out[1], 'flops': out[2]}
    if not measure:
        return None
    gemms = [res[1], res[2]]
    agg_fl = sum(r['flops'] for r in gemms)
    agg_ms = sum(r['ms'] for r in gemms)
    dom = res[1]
    ach = dom['flops'] / (dom['ms'] * 1e-3) * 1e-12
    traffic, traffic_note = _pmc_traffic('wgrad')
    tf = lambda r: r['flops'] / (r['ms'] * 1e-3) * 1e-12 if r['ms'] > 0 else None
    return {
        'bound': 'mfma', 'kernel': dom['name'], 'achieved': ach, 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s',
        'frac': ach / PEAK_BF16_TFLOPS, 'traffic': traffic, 'traffic_note': traffic_note,
        'avg_launch_us': dom['ms'] / dom['launches'] * 1e3, 'launches_per_step': dom['launches'] / steps,
        'flop_per_launch': dom['flops'] / dom['launches'], 'ms_per_step': dom['ms'] / steps,
        'all_mfma_gemms': {'achieved': agg_fl / (agg_ms * 1e-3) * 1e-12,
                           'frac': agg_fl / (agg_ms * 1e-3) * 1e-12 / PEAK_BF16_TFLOPS, 'ms_per_step': agg_ms / steps},
        'attention': _attention_roofline(res[3], res[4], steps, attn_dims, tf),
        'per_kernel': {r['name']: {'TFLOP/s': tf(r), 'ms_per_step': r['ms'] / steps,
                                   'launches_per_step': r['launches'] / steps} for r in res.values()},
    }


def _attention_roofline(fwd, bwd, steps, dims, tf):
    """Attention against BOTH roofs.  At head_dim 64 the algorithmic intensity is 4 T^2 64 / (4 T 128 B) = T / 2 FLOP per byte
    forward (q, k, v read, o written once): 128 at T = 257, 256 at T = 513, 1,024 at T = 2049, against a machine balance of
    2,500 / 8 = 312 -- so the model's own N = 1024 shape is HBM bound (and the MFMA fraction says little there), N = 2048 sits on
    the ridge, N = 8192 is matrix-pipe bound.  Algorithmic bytes per launch: forward 4 x B T D x 2 B; backward 8 x B T D x 2 B
    (q, k, v, dO, O read; dq, dk, dv written), each tensor counted once."""
    out = {'fwd_TFLOPs': tf(fwd), 'bwd_TFLOPs': tf(bwd),
           'fwd_frac': (tf(fwd) or 0) / PEAK_BF16_TFLOPS, 'bwd_frac': (tf(bwd) or 0) / PEAK_BF16_TFLOPS,
           'ms_per_step': (fwd['ms'] + bwd['ms']) / steps}
    if dims and fwd['ms'] > 0 and bwd['ms'] > 0:
        B, T, D = dims
        tensor = B * T * D * 2.0
        fb, bb = 4 * tensor * fwd['launches'], 8 * tensor * bwd['launches']
        out.update({'shape': {'B': B, 'T': T, 'D': D, 'head_dim': 64},
                    'flop_per_byte_fwd': T / 2.0, 'machine_balance': PEAK_BF16_TFLOPS / 8.0,
                    'bound': 'hbm' if T / 2.0 < PEAK_BF16_TFLOPS / 8.0 else 'mfma',
                    'fwd_hbm_TBps': fb / (fwd['ms'] * 1e-3) * 1e-12, 'bwd_hbm_TBps': bb / (bwd['ms'] * 1e-3) * 1e-12,
                    'fwd_hbm_frac': fb / (fwd['ms'] * 1e-3) / 8e12, 'bwd_hbm_frac': bb / (bwd['ms'] * 1e-3) / 8e12,
                    'fwd_us_per_launch': fwd['ms'] / fwd['launches'] * 1e3, 'bwd_us_per_launch': bwd['ms'] / bwd['launches'] * 1e3})
    return out


def index_ops_roofline(B, N, S, k, device, iters=20):
    """FPS and kNN on their own (SURVEY.md section 8d asks for both figures per op): `hbm` = compulsory bytes
    ((12N+8S) B for FPS, (12N+12S+4Sk) B for kNN) / time against 8 TB/s -- tiny BY CONSTRUCTION: both kernels read the
    cloud from HBM exactly once and then work on chip; `onchip` = the operand traffic the algorithm implies (FPS: S rounds x
    16 B per point; kNN: S x N pair distances x 12 B) / time against the aggregate LDS rate 256 CUs x 128 B/clk x 2.4 GHz
    = 78.6 TB/s -- the figure that says how busy the on-chip side is (FPS keeps the points in registers, so its real
    bound is the latency of S dependent arg-max rounds)."""
    from p4p_amd import ops
    g = torch.Generator().manual_seed(5)
    pts = (torch.rand(B, N, 3, generator=g) * 2 - 1).to(device)
    start = torch.zeros(B, dtype=torch.long, device=device)

    def timeit(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / iters
    t_fps = timeit(lambda: ops.fps(pts, S, start))
    idx = ops.fps(pts, S, start)
    cen = torch.gather(pts, 1, idx.unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    t_knn = timeit(lambda: ops.knn(pts, cen, k))
    lds_peak = 256 * 128 * 2.4e9
    fps_hbm, knn_hbm = (12 * N + 8 * S) * B, (12 * N + 12 * S + 4 * S * k) * B
    fps_on, knn_on = S * 16 * N * B, S * N * 12 * B
    return {
        'shape': {'B': B, 'N': N, 'S': S, 'k': k},
        'fps': {'us': t_fps * 1e6, 'hbm_GBps': fps_hbm / t_fps * 1e-9, 'hbm_frac': fps_hbm / t_fps / 8e12,
                'onchip_TBps': fps_on / t_fps * 1e-12, 'onchip_frac': fps_on / t_fps / lds_peak,
                'bound': 'latency of S dependent arg-max rounds (%.2f us per round)' % (t_fps * 1e6 / S)},
        'knn': {'us': t_knn * 1e6, 'hbm_GBps': knn_hbm / t_knn * 1e-9, 'hbm_frac': knn_hbm / t_knn / 8e12,
                'onchip_TBps': knn_on / t_knn * 1e-12, 'onchip_frac': knn_on / t_knn / lds_peak,
                'pair_distance_TFLOPs': 8.0 * S * N * B / t_knn * 1e-12, 'bound': 'VALU / LDS (selection), not HBM'},
    }


def cpu_baseline(args, budget_s=25.0):
    """The CPU oracle (pure-torch fp32 restatement of the reference path) doing the same training step on the
    host cores, on a bounded sample: batch 8 of the same synthetic clouds, as many steps as fit the budget."""
    from oracle import pix4point_oracle as O
    from p4p_amd.vit import parse_model_name
    D, L, H, _, _ = parse_model_name(args.vit)
    cores = usable_cores()
    torch.set_num_threads(cores)
    log(f'cpu_baseline: {cores} threads')
    torch.manual_seed(0)
    model = O.Pix4Point(num_classes=args.classes, embed_dim=D, depth=L, heads=H)
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=5e-4, weight_decay=5e-2)
    B = 8
    pts, labels, start = synthetic_batch(B, args.points, args.classes, 99, 'cpu')

    def one():
        opt.zero_grad()
        loss = torch.nn.functional.cross_entropy(model(pts, start), labels)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 10)
        opt.step()
    tw = time.perf_counter()
    one()                                   # warm-up (allocator, thread pool)
    log(f'cpu_baseline: warm-up step {time.perf_counter() - tw:.1f} s')
    t0, n = time.perf_counter(), 0
    while True:
        one()
        n += 1
        el = time.perf_counter() - t0
        log(f'cpu_baseline: {n} steps in {el:.1f} s')
        if el > budget_s or n >= 8:
            break
    return {'value': B * n / el, 'unit': 'point-clouds/sec', 'cores': cores, 'kind': 'port',
            'sample': f'{n} train steps of batch {B} (N={args.points}, {args.vit}, fp32 torch CPU oracle)',
            'torch_threads': torch.get_num_threads()}


def cpu_baseline_apf(args, budget_s=25.0):
    """The APF CPU oracle (oracle/apf_oracle.py: fp32 torch restatement of src/models/apf.py + apf_utils.py) doing the same
    training step (src/train/apf_trainer.py:260-275) on the host cores, on a bounded sample."""
    from oracle import apf_oracle as A
    from p4p_amd.vit import parse_model_name
    D, L, H, _, _ = parse_model_name(args.vit)
    cores = usable_cores()
    torch.set_num_threads(cores)
    log(f'cpu_baseline: {cores} threads')
    torch.manual_seed(0)
    model = A.AdaptPointFormer(num_classes=args.classes, embedding_dim=D, depth=L, num_heads=H, in_channels=4)
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=5e-4, weight_decay=5e-2)
    B = 4
    pts, labels, start = synthetic_batch(B, args.points, args.classes, 99, 'cpu')
    pts = torch.cat([pts, pts[:, :, 1:2] - pts[:, :, 1:2].min(dim=1, keepdim=True).values], dim=-1)

    def one():
        opt.zero_grad()
        loss = torch.nn.functional.cross_entropy(model(pts, start), labels, label_smoothing=0.3)
        loss.backward()
        opt.step()
    tw = time.perf_counter()
    one()
    log(f'cpu_baseline: warm-up step {time.perf_counter() - tw:.1f} s')
    t0, n = time.perf_counter(), 0
    while True:
        one()
        n += 1
        el = time.perf_counter() - t0
        log(f'cpu_baseline: {n} steps in {el:.1f} s')
        if el > budget_s or n >= 8:
            break
    return {'value': B * n / el, 'unit': 'point-clouds/sec', 'cores': cores, 'kind': 'port',
            'sample': f'{n} train steps of batch {B} (N={args.points}, APF {args.vit} frozen, fp32 torch CPU oracle)',
            'torch_threads': torch.get_num_threads()}


def spawn_ranks(args):
    """`python bench.py --gpus N` with N > 1 outside a launcher: start one rank per GPU with torch.distributed.run as a
    CHILD process (this process has not touched the GPU yet and never will) and pass its exit code on."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log('spawning: ' + ' '.join(cmd))
    return subprocess.call(cmd, env=env)


LAST_ENQUEUE_S = [0.0]


def host_enqueue_ms(stepper, batch, reps=3):
    """Host time to ENQUEUE one training step into an EMPTY queue (device drained before and after; median of `reps`).  Measured
    over a long timed region instead, the figure saturates at the device time as soon as the launch queue is full (the host blocks in
    the launch call), so it could not tell a host-bound step from a device-bound one with the host running ahead."""
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        stepper.step(*batch)
        ts.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
    return sorted(ts)[len(ts) // 2] * 1e3


def timed_steps(stepper, batch, steps, fence, distributed, device, optim=True):
    """EXACTLY `steps` steps between two fences (barrier + device synchronize); seconds, max over ranks, and every
    rank's own time."""
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = stepper.step(*batch, optim=optim)
    LAST_ENQUEUE_S[0] = time.perf_counter() - t0          # host time to ENQUEUE the steps (no sync): ~= elapsed means host-bound
    fence()
    elapsed = time.perf_counter() - t0
    per_rank = [elapsed]
    if distributed:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        allt = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(allt, t)
        per_rank = [float(x.item()) for x in allt]
        elapsed = max(per_rank)
    return elapsed, per_rank, float(loss.detach())


def allreduce_exposed_ms(stepper, batch, steps=3):
    """Diagnostic for multi-GPU runs: how long the step waits in GradReducer.finish() after the backward has been
    drained (device-synchronised on both sides, so this pass is not part of any reported throughput)."""
    tot = 0.0
    for _ in range(steps):
        stepper.reducer.zero_grad()
        loss = stepper.crit(stepper.model(batch[0], batch[2]), batch[1])
        loss.backward()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        stepper.reducer.finish(average=False)
        torch.cuda.synchronize()
        tot += time.perf_counter() - t0
        if getattr(stepper, 'fused_opt', True):
            stepper.opt.step(max_norm=10 if hasattr(stepper, 'fused_opt') else None, grad_scale=stepper.reducer.grad_scale)
        else:
            stepper.reducer.flat.mul_(stepper.reducer.grad_scale)
            stepper.opt.step()
    return tot / steps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=None,
                    help='ranks = GPUs of this node (default: WORLD_SIZE when a launcher set it, else 1)')
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', choices=('pix4point', 'apf'), default='pix4point',
                    help='pix4point = BASELINE.json configs[1] (the contract line); apf = configs[3], the adapter path')
    ap.add_argument('--batch', type=int, default=None, help='clouds per GPU (default 256; apf: 128)')
    ap.add_argument('--points', type=int, default=None, help='points per cloud (default 1024; apf: 2048)')
    ap.add_argument('--vit', default=None)
    ap.add_argument('--dim', type=int, default=None)
    ap.add_argument('--classes', type=int, default=None)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--bucket-mb', type=float, default=None,
                    help='gradient all-reduce bucket size (default: one bucket per transformer block, 28 MB for ViT-B)')
    ap.add_argument('--no-pmc', action='store_true', help='do not collect roofline.traffic live (two short rocprofv3 --pmc child runs)')
    ap.add_argument('--no-north-star', action='store_true',
                    help='skip the extra N=2048, B=256 measurement (the point north_star states its targets at)')
    args = ap.parse_args()
    apf = args.workload == 'apf'
    default_config = args.batch is None and args.points is None and args.vit is None and not apf
    args.batch = args.batch or (128 if apf else 256)
    args.points = args.points or (2048 if apf else 1024)
    args.vit = args.vit or ('vit_base_patch16_224' if apf else 'vit_base_patch16_384')
    args.classes = args.classes or (15 if apf else 40)
    from p4p_amd.vit import parse_model_name          # imports torch only: no GPU call before the rank check below
    vd, vl, vh, _, _ = parse_model_name(args.vit)
    args.dim = args.dim or vd

    world_env = os.environ.get('WORLD_SIZE')
    if args.gpus is None:          # `torchrun --nproc-per-node N bench.py` without --gpus: the launcher's world is the answer
        args.gpus = int(world_env or '1')
    if world_env is None and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    world = int(world_env or '1')
    if world != args.gpus:
        print(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU '
              f'(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)', file=sys.stderr)
        sys.exit(2)
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    distributed = world > 1
    if distributed:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        # rehearsal switches for a one-GPU box (tests/test_gpu_09_ddp.py style): P4P_DIST_BACKEND=gloo P4P_SINGLE_DEVICE=1 runs
        # all ranks on device 0 with host-side collectives, to exercise the multi-rank control flow of this file
        backend = os.environ.get('P4P_DIST_BACKEND', 'nccl')
        if os.environ.get('P4P_SINGLE_DEVICE'):
            local = 0
        torch.cuda.set_device(local)
        if backend == 'nccl':
            from p4p_amd.ddp import configure_rccl_env
            configure_rccl_env()                        # bounds RCCL's channels (= CUs) before the communicator exists
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend)
    device = torch.device('cuda', local)
    torch.cuda.set_device(device)

    stepper = (ApfStepper if apf else Stepper)(args, device, distributed)
    batch = synthetic_batch(args.batch, args.points, args.classes, 1234 + rank, device)
    if apf:      # the height column of center_and_normalize_point_cloud (src/data/augment.py:247-248): y - min y per cloud
        xyz = batch[0]
        batch = (torch.cat([xyz, xyz[:, :, 1:2] - xyz[:, :, 1:2].min(dim=1, keepdim=True).values], dim=-1).contiguous(),) + batch[1:]

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    ddp_info = None
    if getattr(stepper, 'reducer', None) is not None:
        from p4p_amd import ops as _ops
        ddp_info = dict(stepper.reducer.describe(), backend=dist.get_backend(), persistent_grid=_ops.persistent_grid_size())
        log(f'rank {rank}: gradient buckets {ddp_info["buckets"]} ({ddp_info["total_bytes"] * 1e-6:.0f} MB in all), CUs left to the '
            f'collectives inside the backward window {ddp_info["reserved_cus"]} (persistent grid there '
            f'{ddp_info.get("persistent_grid_in_backward")}, elsewhere {ddp_info["persistent_grid"]})')
    log(f'rank {rank}/{world}: model built, warm-up {args.warmup} steps')
    for _ in range(args.warmup):
        stepper.step(*batch)
    elapsed, per_rank, loss_val = timed_steps(stepper, batch, args.steps, fence, distributed, device)
    log(f'timed region: {args.steps} steps in {elapsed:.3f} s (host returned from the last enqueue after {LAST_ENQUEUE_S[0]:.3f} s)')
    host_enq = host_enqueue_ms(stepper, batch)
    # the metric's literal "fwd+bwd" (no clip / AdamW), SURVEY.md section 8d's first figure: the same K steps again
    fb_elapsed, _, _ = timed_steps(stepper, batch, args.steps, fence, distributed, device, optim=False)

    roof = None
    if not args.no_roofline:
        # all ranks run the instrumented steps (a step holds collectives); only rank 0 brackets its kernels with events
        roof = instrumented_roofline(stepper, batch, max(2, min(args.steps, 3)), measure=rank == 0,
                                     attn_dims=(args.batch, 196 if apf else args.points // 4 + 1, vd))
    if rank == 0 and roof is not None:
        if apf:
            roof['index_ops'] = index_ops_roofline(args.batch, args.points, 196, 32, device)
        else:
            roof['index_ops'] = index_ops_roofline(args.batch, args.points, args.points // 4, 16, device)
        log('instrumented roofline pass done')
    ar_wait = allreduce_exposed_ms(stepper, batch) if distributed else None

    # ---- the point north_star states its utilisation targets at: N=2048, B=256, ViT-B/12, one GPU
    ns = None
    if default_config and world == 1 and not args.no_north_star:
        nb, npts = 256, 2048
        ns_batch = synthetic_batch(nb, npts, args.classes, 4321, device)
        for _ in range(2):
            stepper.step(*ns_batch)
        ns_steps = 10
        ns_el, _, _ = timed_steps(stepper, ns_batch, ns_steps, fence, False, device)
        ns_enq = host_enqueue_ms(stepper, ns_batch)
        ns_fb, _, _ = timed_steps(stepper, ns_batch, 5, fence, False, device, optim=False)
        ns_roof = instrumented_roofline(stepper, ns_batch, 2, attn_dims=(nb, npts // 4 + 1, vd)) if not args.no_roofline else None
        fl = flop_per_cloud(npts, vd, vl)
        v = nb * ns_steps / ns_el
        ns = {'workload': f'Pix4Point {args.vit} N={npts} k=16, batch {nb}, 1 GPU (same step as the headline line)',
              'clouds_per_sec': v, 'ms_per_step': ns_el / ns_steps * 1e3, 'steps': ns_steps,
              'algorithmic_gflop_per_cloud': fl * 1e-9, 'algorithmic_tflops': v * fl * 1e-12,
              'mfma_frac_whole_step': v * fl * 1e-12 / PEAK_BF16_TFLOPS, 'host_enqueue_ms_per_step': ns_enq,
              'fwd_bwd_only': {'clouds_per_sec': nb * 5 / ns_fb, 'ms_per_step': ns_fb / 5 * 1e3}}
        if ns_roof:
            ns['gemm_wgrad_TFLOPs'] = ns_roof['achieved']
            ns['all_mfma_gemms'] = ns_roof['all_mfma_gemms']
            ns['attention'] = ns_roof['attention']
            ns['index_ops'] = index_ops_roofline(nb, npts, npts // 4, 16, device)
        log('north-star point (N=2048, B=256) done')
        del ns_batch
    # ---- the third size north_star names: N=8192 (BASELINE configs[4]: ViT-L/24 tokens, 8 clouds per GPU), one GPU, short
    n8k = None
    if default_config and world == 1 and not args.no_north_star:
        import copy
        del stepper
        torch.cuda.empty_cache()
        a2 = copy.copy(args)
        a2.vit, a2.points, a2.batch = 'vit_large_patch16_384', 8192, 8
        ld_, ll_, lh_, _, _ = parse_model_name(a2.vit)
        a2.dim = ld_
        st2 = Stepper(a2, device, False)
        b2 = synthetic_batch(a2.batch, a2.points, a2.classes, 777, device)
        for _ in range(3):
            st2.step(*b2)
        k_steps = 20
        runs = []
        for _ in range(3):                              # three timed regions of 20 steps: the scatter between them is reported
            el2, _, _ = timed_steps(st2, b2, k_steps, fence, False, device)
            runs.append(el2 / k_steps * 1e3)
        runs.sort()
        ms2, enq2 = runs[1], host_enqueue_ms(st2, b2)   # median region
        fl2 = flop_per_cloud(a2.points, ld_, ll_)
        v2 = a2.batch / (ms2 * 1e-3)
        n8k = {'workload': f'Pix4Point {a2.vit} (D={ld_}, {ll_} blocks, {lh_} heads) N=8192 k=16, batch {a2.batch}, 1 GPU (same step as the headline line)',
               'clouds_per_sec': v2, 'ms_per_step': ms2, 'steps': k_steps, 'ms_per_step_runs': runs,
               'scatter_pct': (runs[-1] - runs[0]) / ms2 * 50.0,      # +- half of the range, in % of the median
               'host_enqueue_ms_per_step': enq2,
               'algorithmic_gflop_per_cloud': fl2 * 1e-9, 'algorithmic_tflops': v2 * fl2 * 1e-12,
               'mfma_frac_whole_step': v2 * fl2 * 1e-12 / PEAK_BF16_TFLOPS}
        if not args.no_roofline:
            r2 = instrumented_roofline(st2, b2, 2, attn_dims=(a2.batch, a2.points // 4 + 1, ld_))
            n8k['gemm_wgrad_TFLOPs'] = r2['achieved']
            n8k['all_mfma_gemms'] = r2['all_mfma_gemms']
            n8k['attention'] = r2['attention']
            n8k['per_kernel'] = r2['per_kernel']
            n8k['index_ops'] = index_ops_roofline(a2.batch, a2.points, a2.points // 4, 16, device)
        log('N=8192 ViT-L/24 point done')
        del st2, b2
        torch.cuda.empty_cache()
    if rank == 0 and roof is not None and world == 1 and default_config and not args.no_pmc:
        # after EVERY timed section (a child that outlives its timeout must not run beside one), with this process's model and
        # batches freed
        live, note = live_pmc_traffic('gemm_wgrad3_kernel')
        if live is not None:
            roof['traffic'], roof['traffic_note'] = live, note
        else:
            roof['traffic_note'] = (roof.get('traffic_note') or '') + f' (live collection skipped: {note})'
        log('PMC traffic: ' + note)
    if distributed:
        dist.barrier()
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline_apf(args) if apf else cpu_baseline(args)

    if rank == 0:
        clouds = args.batch * world * args.steps
        value = clouds / elapsed
        fl = apf_flop_per_cloud(E=vd, L=vl) if apf else flop_per_cloud(args.points, vd, vl)
        fam = {192: 'ViT-Ti', 384: 'ViT-S', 768: 'ViT-B', 1024: 'ViT-L'}.get(vd, f'ViT(D={vd})')
        out = {
            'metric': f'point-clouds/sec fwd+bwd (train step), ModelNet40-shaped N={args.points} {fam}',
            'value': value, 'unit': 'point-clouds/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': f'Pix4Point {args.vit} (D={vd}, {vl} blocks, {vh} heads) N={args.points} k=16, '
                                   f'{args.classes} classes, batch {args.batch}/GPU; step = zero_grad+fwd+CE+bwd'
                                   f'{"+RCCL all-reduce" if distributed else ""}+clip_grad_norm+AdamW(fused)',
                       'global_batch': args.batch * world, 'points': args.points, 'parallelism': f'dp{world}',
                       'final_loss': loss_val, 'algorithmic_gflop_per_cloud': fl * 1e-9},
            'algorithmic_tflops_per_gpu': (value / world * fl * 1e-12) if fl else None,
            'mfma_frac_whole_step': (value / world * fl * 1e-12 / PEAK_BF16_TFLOPS) if fl else None,
            # SURVEY.md section 8d's first figure: zero_grad + forward + CE + backward (+ all-reduce), no clip / AdamW
            'fwd_bwd_only': {'value': clouds / fb_elapsed, 'unit': 'point-clouds/sec',
                             'ms_per_step': fb_elapsed / args.steps * 1e3, 'steps': args.steps},
            'per_rank_ms_per_step': [t / args.steps * 1e3 for t in per_rank],
            'host_enqueue_ms_per_step': host_enq,      # rank 0's host time to enqueue ONE step into an empty queue; >= ms_per_step means host-bound
        }
        if ddp_info is not None:
            out['ddp'] = ddp_info
        if ar_wait is not None:
            out['allreduce_exposed_ms'] = ar_wait
        if apf:
            out['metric'] = f'point-clouds/sec fwd+bwd (train step), APF adapter path N={args.points} {fam} frozen'
            out['config']['workload'] = (f'AdaptPointFormer {args.vit} (D={vd}, {vl} frozen blocks + adapters), in_channels=4 (xyz + height), 196 groups of 32, '
                                         f'N={args.points}, {args.classes} classes, batch {args.batch}/GPU; step = zero_grad+fwd+'
                                         f'CE(ls 0.3)+bwd{"+RCCL all-reduce" if distributed else ""}+AdamW(fused), train-mode '
                                         'DropPath/dropout; FLOP model: encoder 3x fwd as the reference computes it, frozen '
                                         'transformer 2x fwd')
        if roof:
            out['roofline'] = roof
        if ns:
            out['north_star_point'] = ns
        if n8k:
            out['n8192_point'] = n8k
        if cpu:
            out['cpu_baseline'] = cpu
        print(json.dumps(out))
    if distributed:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
