"""Data-parallel gradient exchange: one process per GPU, RCCL all-reduce over xGMI.

The reference imports DistributedDataParallel but never uses it (src/train/pix4point_trainer.py:4-6);
this is new functionality.  Clouds are independent, so the only exchange per step
is the gradient all-reduce.  Design:

* every trainable parameter's ``.grad`` is a VIEW into one flat fp32 buffer, laid out in the order
  gradients become ready: [block L-1 | ... | block 0 | everything else].  No flatten / unflatten copies.
* the encoder's backward (functional._EncoderFn) asks for a block's destination views (``grad_views``) and lets the
  weight-gradient GEMMs write STRAIGHT into them (the four matrices are 99.9 % of a block's 28 MB); the eight small
  vectors are copied.  ``block_done`` then launches that bucket's all-reduce asynchronously on RCCL's stream,
  overlapping with the backward of the remaining blocks.  Nothing is added into the buffer, so only the tail bucket
  (whose gradients autograd accumulates) is zeroed per step.
* ``finish`` (after ``loss.backward()``) zeroes every block region the step's backward did not write (a loss that does
  not reach the encoder), reduces the tail bucket and waits for all handles.  The buffer -- and therefore every
  ``p.grad`` a logger or a custom clipper reads afterwards -- then holds SUMS over the ranks, not means;
  ``grad_scale`` = 1/world goes to ``FusedAdamW.step(grad_scale=...)``, which folds it into the clip coefficient (no
  pass over the 345-MB buffer); ``finish(average=True)`` scales in place for a stock optimizer.
  BatchNorm statistics stay per rank (what a per-rank reference run would compute).
* parameters that never receive a gradient (timm's unused patch_embed / pos_embed / head, SURVEY.md
  quirk 8) are simply not in any bucket.

xGMI is point-to-point (7 links per GPU): a ViT-B gradient set is 345 MB fp32; the default of one bucket per block
(28 MB) keeps each ring step well above the latency floor while leaving 12 collectives in flight behind ~40 ms of
backward compute; ``bucket_mb`` merges consecutive blocks into larger buckets (to be tuned against the measured RCCL bus
bandwidth on an 8-GPU node: bench.py prints per-rank step times and the exposed all-reduce wait).
"""
import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, bucket_params: List[List[torch.nn.Parameter]], process_group=None, blocks_per_bucket=None,
                 outside: Optional[Iterable[torch.nn.Parameter]] = None, force_collectives: bool = False):
        """bucket_params[i] = parameters of bucket i, in the order buckets become ready; the last bucket is the tail reduced
        by finish().  blocks_per_bucket[i] (optional) = how many block_done calls complete bucket i (default 1).
        outside: the model's parameters that are in NO bucket (frozen or never used at build time); finish() refuses a step in
        which one of them turned trainable or received a gradient (it would never be all-reduced).
        force_collectives: launch the all-reduces at world size 1 too (a process group must exist) -- the single-GPU
        rehearsal of the RCCL stream hand-off, tests/test_gpu_09_ddp.py."""
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.grad_scale = 1.0 / self.world
        self.force_collectives = bool(force_collectives)
        if self.force_collectives and not dist.is_initialized():
            raise RuntimeError('GradReducer(force_collectives=True) needs an initialised process group')
        self.collectives_launched = 0      # all-reduces handed to the backend so far (diagnostic; bench.py prints it)
        self.buckets = []
        self._regions = {}                 # id(first parameter of a block) -> [bucket, start, end, written this step]
        params = [p for b in bucket_params for p in b]
        if not params:
            raise ValueError('GradReducer: no parameters')
        dev, total = params[0].device, sum(p.numel() for p in params)
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        self._bucket_of = {}
        for i, bp in enumerate(bucket_params):
            start = off
            for p in bp:
                p.grad = self.flat[off:off + p.numel()].view_as(p)
                self._bucket_of[id(p)] = i
                off += p.numel()
            self.buckets.append((start, off))
        self._need = list(blocks_per_bucket) if blocks_per_bucket else [1] * len(bucket_params)
        self._have = [0] * len(bucket_params)
        self._handles = []
        self._launched = set()
        self._trainable = [(p, p.requires_grad) for p in params]
        inside = {id(p) for p in params}
        self._outside = [p for p in (outside or []) if id(p) not in inside]
        self._stale_zeroed = 0             # regions finish() had to zero (a block whose backward did not write): diagnostic
        self.reserved_cus = None           # the PLAN of reserve_cus(): CUs left to RCCL inside the backward window (None: none)
        self._cus_before = None            # inside the window: the library's value before backward_begins(), restored by finish()

    # -- compute units for the collectives ---------------------------------------------------
    def reserve_cus(self, reserved_cus=None):
        """PLAN to size this library's one-round grids to CUs - r while collectives can be in flight (planned_reserved_cus decides
        r: an explicit value, P4P_RESERVED_CUS, or RCCL's channel bound when the group's backend is RCCL and world > 1; host-side
        backends such as gloo hold no CUs and reserve nothing).  The library-global switch is only SET inside the window in which
        an all-reduce can overlap this library's kernels -- from backward_begins() (the encoder's backward calls it) to the end
        of finish(): the forward pass and the optimizer step never run next to a collective and keep all CUs (measured on one
        GPU, N=1024 step: grids of CUs - 8 alone cost 1.5 ms of 51.7, tools/contention_profile.py).  The setting changes the
        summation order of the weight-gradient M-splits (not the forward).  release_cus() drops the plan; trainers call it in
        train()'s finally.  Returns the planned value, None if there is none."""
        if reserved_cus is None:
            reserved_cus = getattr(self, '_cus_asked', None)            # a later call (train() after release_cus()) repeats the first
        self._cus_asked = reserved_cus
        self.reserved_cus = planned_reserved_cus(self.world, reserved_cus, _backend_of(self.group))
        return self.reserved_cus

    def backward_begins(self):
        """Called where the overlap window opens (functional._EncoderFn.backward): from here to finish() the buckets' all-reduces
        run beside this library's kernels."""
        if self.reserved_cus is not None and self._cus_before is None:
            from . import ops
            self._cus_before = ops.set_reserved_cus(self.reserved_cus)

    def _window_closes(self):
        if self._cus_before is not None:
            from . import ops
            ops.set_reserved_cus(self._cus_before)
            self._cus_before = None

    def release_cus(self):
        """Drop the plan (and close the window if a step was abandoned inside it): the library is where it was before."""
        self._window_closes()
        self.reserved_cus = None

    def register_block(self, params: Iterable[torch.nn.Parameter]):
        """Declare one block (a run of consecutive parameters of one bucket that the backward writes as a unit): finish()
        zeroes the region of every registered block whose block_done did not arrive in the step."""
        params = list(params)
        i = self._bucket_of[id(params[0])]
        s = params[0].grad.data_ptr() - self.flat.data_ptr()
        e = params[-1].grad.data_ptr() - self.flat.data_ptr() + params[-1].numel() * 4
        self._regions[id(params[0])] = [i, s // 4, e // 4, False]

    def check_trainable(self):
        """The bucket layout is fixed at construction: a requires_grad flag flipped afterwards (staged unfreezing) would leave
        a block without its per-step write (stale sums) or a gradient outside the buffer.  Rebuild the reducer instead.
        Both directions are refused: trainable -> frozen for a bucketed parameter, and frozen -> trainable (or a gradient
        that appeared) for a parameter that was left out of the buckets -- its gradient would stay local to the rank and the
        ranks would diverge without an error."""
        for p, was in self._trainable:
            if p.requires_grad != was:
                raise RuntimeError('GradReducer: requires_grad of a bucketed parameter changed after build_reducer(); '
                                   'build a new reducer for the new set of trainable parameters')
        for p in self._outside:
            if p.requires_grad and p.grad is not None:
                raise RuntimeError('GradReducer: a parameter outside the buckets (frozen or unused at build_reducer()) received a '
                                   'gradient: it would not be all-reduced; build a new reducer after changing what is trainable')

    # -- called from inside backward ---------------------------------------------------------
    def grad_views(self, params: Iterable[torch.nn.Parameter]):
        """The flat-buffer views of one block's parameters (what the backward kernels write into), or None when the
        block is not bucketed (e.g. frozen)."""
        params = list(params)
        if self._bucket_of.get(id(params[0])) is None:
            return None
        return [p.grad for p in params]

    def block_done(self, params: Iterable[torch.nn.Parameter]):
        """One block's gradients are in its views: count it, and launch the bucket's all-reduce when it is complete."""
        first = next(iter(params))
        i = self._bucket_of[id(first)]
        reg = self._regions.get(id(first))
        if reg is not None:
            reg[3] = True
        self._have[i] += 1
        if self._have[i] >= self._need[i]:
            self._launch(i)

    def block_ready(self, layer: int, params: Iterable[torch.nn.Parameter], grads: List[Optional[torch.Tensor]]):
        """Hand-off of freshly computed gradient tensors (copied into the views); kept for callers that do not write in
        place.  Returns what autograd should see (None: the gradients already sit in ``.grad``)."""
        params = list(params)
        views = self.grad_views(params)
        if views is None:                  # not bucketed (e.g. frozen): let autograd handle it
            return grads
        dst, src = [], []
        for v, g in zip(views, grads):
            if g is not None and g.data_ptr() != v.data_ptr():
                dst.append(v)
                src.append(g.view_as(v))
        if dst:
            torch._foreach_copy_(dst, src)
        self.block_done(params)
        return [None] * len(grads)

    def _launch(self, i):
        if (self.world > 1 or self.force_collectives) and i not in self._launched:
            s, e = self.buckets[i]
            self._handles.append(dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            self.collectives_launched += 1
        self._launched.add(i)

    # -- called after backward ---------------------------------------------------------------
    def finish(self, average=True):
        """Reduce what is left, wait for everything.  average=True divides by the world size in place (stock optimizers);
        average=False leaves SUMS and the caller passes ``grad_scale`` to FusedAdamW.step."""
        self.check_trainable()
        for reg in self._regions.values():
            # a block bucket is only correct if its backward OVERWROTE its views in this step (zero_grad() leaves them
            # alone); a block the loss did not reach keeps last step's sums, so it is zeroed here, before its all-reduce
            if not reg[3]:
                if reg[0] in self._launched:
                    raise RuntimeError('GradReducer: a bucket was reduced before all of its blocks were written')
                self.flat[reg[1]:reg[2]].zero_()
                self._stale_zeroed += 1
            reg[3] = False
        for i in range(len(self.buckets)):
            self._launch(i)
        for h in self._handles:
            h.wait()
        self._handles.clear()
        self._launched.clear()
        self._have = [0] * len(self.buckets)
        self._window_closes()              # every collective has been waited for: the optimizer and the next forward get all CUs
        if average and self.world > 1:
            self.flat.mul_(self.grad_scale)

    def describe(self):
        """Bucket layout for logs / bench.py: count, bytes per bucket (in readiness order, the tail last), the reserved CUs asked
        for and what the library made of them (it rounds r up to a multiple of 8 and never goes below a grid of 64: the
        effective figure is CUs - persistent grid)."""
        d = {'buckets': len(self.buckets), 'bucket_bytes': [(e - s) * 4 for s, e in self.buckets],
             'total_bytes': int(self.flat.numel()) * 4, 'reserved_cus': self.reserved_cus,
             'collectives_launched': self.collectives_launched}
        if self.reserved_cus is not None and self.flat.is_cuda:
            from . import ops
            old = ops.set_reserved_cus(self.reserved_cus)
            cus, grid = ops.device_cu_count(), ops.persistent_grid_size()
            ops.set_reserved_cus(old)
            d['persistent_grid_in_backward'] = grid
            d['reserved_cus_effective'] = (cus - grid) if (cus > 0 and grid > 0) else None      # None: persistent kernels off (r < 0)
        return d

    def zero_grad(self):
        """Only the tail bucket accumulates (autograd adds into its views); the block buckets are overwritten by the
        encoder's backward (finish() zeroes any registered block that was NOT written in a step)."""
        self._window_closes()              # a step abandoned between backward and finish() must not leave the next forward on CUs - r
        s, e = self.buckets[-1]
        self.flat[s:e].zero_()


RCCL_MAX_CHANNELS = 8


def configure_rccl_env(max_channels=RCCL_MAX_CHANNELS):
    """Call BEFORE init_process_group('nccl'): bounds the workgroups (= channels, one CU each) an RCCL collective may occupy, so
    that the number of CUs to leave to it is KNOWN (planned_reserved_cus).  What the gradient exchange needs is little: 345 MB
    per step (ViT-B) spread over ~35 ms of backward, and the one bucket whose all-reduce is not covered by a later block (28 MB,
    block 0) has the ~2.5 ms of the tokenizer's backward to finish in -- 20 GB/s of bus bandwidth, an eighth of what 8 channels
    move.  What the channels COST is measured (one GPU, N=1024 step, grids of CUs - r and nothing else changed,
    tools/contention_profile.py --no-park): r = 8 +1.6 ms, 16 +2.3, 24 / 32 +2.9 of 51.3 ms for the whole step, about two thirds
    of that inside the backward window where the reservation applies.  Round 5: 8 (was 32).  bench.py --gpus N prints
    allreduce_exposed_ms: if a real 8-GPU run shows the all-reduce exposed, raise it.  An NCCL_MAX_NCHANNELS the user has set is
    respected."""
    os.environ.setdefault('NCCL_MAX_NCHANNELS', str(int(max_channels)))
    return int(os.environ['NCCL_MAX_NCHANNELS'])


def _backend_of(group=None):
    if not dist.is_initialized():
        return None
    try:
        return str(dist.get_backend(group))
    except Exception:                                       # pragma: no cover - a group this rank is not part of
        return None


def planned_reserved_cus(world, reserved_cus=None, backend=None):
    """How many CUs a reducer for this world / backend asks the library to leave free, WITHOUT setting anything (tests take
    their reference pass under the same value).  RCCL's all-reduce workgroups are long-lived and cannot share a CU with this
    library's kernels that size ONE round of workgroups to the chip -- the persistent short-K GEMM tile loop, the walking
    attention backward (every byte of a CU's LDS) and the weight-gradient GEMM's M-splits: with r CUs held, r of those
    workgroups start only when others have finished, with their whole share still to do, and the launch takes twice as long.
    One-GPU rehearsal with r parked stand-in workgroups (tools/contention_rehearsal.py, N=1024 step, 51.4 ms alone): 72.2 ms
    with r = 8 and no remedy, 55.9 ms with all those grids on CUs - r (p4p_set_reserved_cus), 62.1 ms with a workgroup per
    tile / pair (r < 0: the weight gradients keep 256 splits).
    Order: the explicit argument; P4P_RESERVED_CUS; world > 1 on RCCL ('nccl'): NCCL_MAX_NCHANNELS (configure_rccl_env: 8),
    i.e. what RCCL can take at most; otherwise None -- world 1,
    and host-side backends (gloo collectives run on the CPU and hold no CU), leave the library alone."""
    if reserved_cus is not None:
        return int(reserved_cus)
    if 'P4P_RESERVED_CUS' in os.environ:
        return int(os.environ['P4P_RESERVED_CUS'])
    if world > 1 and backend is not None and 'nccl' in backend:
        return int(os.environ.get('NCCL_MAX_NCHANNELS', RCCL_MAX_CHANNELS))
    return None


def build_reducer(model, process_group=None, bucket_mb=None, force_collectives=False, reserved_cus=None) -> GradReducer:
    """Buckets for a pix4point.Pix4Point: encoder blocks last-to-first (consecutive blocks merged until a bucket holds
    at least ``bucket_mb`` MB; default: one block per bucket), then every other parameter that can receive a gradient.
    Installs itself as the encoder's grad sink.  Every parameter of the model that ends up in no bucket (frozen blocks, the
    never-used timm parameters of quirk 8) is remembered: GradReducer.check_trainable refuses a step in which one of them
    received a gradient (staged unfreezing needs a new reducer)."""
    pv = model.model
    from .functional import block_params
    blocks = list(pv.vit_blocks)
    buckets, counts, seen = [], [], set()
    want = 0 if not bucket_mb else float(bucket_mb) * 1e6
    for blk in reversed(blocks):
        bp = [p for p in block_params(blk) if p.requires_grad]
        if bp and len(bp) == 12:
            if buckets and counts[-1] and sum(p.numel() for p in buckets[-1]) * 4 < want:
                buckets[-1] += bp
                counts[-1] += 1
            else:
                buckets.append(bp)
                counts.append(1)
            seen.update(id(p) for p in bp)
    unused = {id(p) for n, p in pv.vit.named_parameters()
              if n.startswith(('patch_embed.', 'head.')) or n == 'pos_embed'}
    tail = [p for p in model.parameters() if p.requires_grad and id(p) not in seen and id(p) not in unused]
    buckets.append(tail)
    counts.append(0)
    red = GradReducer(buckets, process_group, blocks_per_bucket=counts, outside=list(model.parameters()),
                      force_collectives=force_collectives)
    red.reserve_cus(reserved_cus)
    for blk in blocks:
        bp = [p for p in block_params(blk) if p.requires_grad]
        if bp and len(bp) == 12:
            red.register_block(bp)
    pv.grad_sink = red
    return red


def broadcast_parameters(model, src=0, process_group=None):
    """Same initial weights and buffers on every rank."""
    if not dist.is_initialized() or dist.get_world_size(process_group) == 1:
        return
    ts = list(model.parameters()) + list(model.buffers())
    for t in ts:
        dist.broadcast(t.data, src, group=process_group)
    torch._C._increment_version(ts)        # `.data` writes do not bump the version the bf16 operand caches are keyed on
