"""Train-step tail of the reference trainer (src/train/pix4point_trainer.py:164-167): ``clip_grad_norm_`` followed by
``torch.optim.AdamW.step()``, as three HIP launches over all parameter tensors (csrc/optim.hip) without host syncs.

``FusedAdamW`` is a ``torch.optim.Optimizer``: same constructor arguments as ``torch.optim.AdamW`` (amsgrad / maximize
are not built), same update rule, and the same ``state_dict`` layout (``step``, ``exp_avg``, ``exp_avg_sq`` per parameter), so
checkpoints written by the reference's ``torch.optim.AdamW`` load into it and vice versa.

    opt = FusedAdamW(model.get_param_groups(), lr=5e-4, weight_decay=5e-2)
    loss.backward()
    total_norm = opt.step(max_norm=10.0)      # == clip_grad_norm_(params, 10) ; opt.step()

Difference from the two stock calls: the clipped gradients are consumed inside the update kernel and the ``.grad``
tensors keep their unclipped values (nothing on the training path reads them after the step).
"""

import torch

from . import _lib as L


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        if lr < 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1):
            raise ValueError('FusedAdamW: invalid hyper-parameter')
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._key = None           # identity of the cached device tables
        self._tables = None
        self._staging = {}         # pinned upload rings, _upload()
        self._chunk_cache = {}     # (tensor sizes, chunk, device) -> device chunk list

    # ------------------------------------------------------------------------------------------
    def _upload(self, host, dev):
        """host: a CPU tensor -> device, without a host-device sync and without pinning a fresh buffer per call (pin_memory() of
        a new tensor costs ~2 ms): a small ring of persistent pinned staging buffers per (dtype, size), each guarded by the event
        recorded behind its last copy (waited for only if the host is a whole ring ahead of the device)."""
        key = (host.dtype, host.numel())
        ring = self._staging.get(key)
        if ring is None:
            # a new table size = a new set of parameters with gradients; the rings of sizes no longer in use would otherwise stay
            # pinned for good.  Few sizes are live at a time (tensor table, chunk list, per-subset tables): keep the 6 most recent
            while len(self._staging) >= 6:
                old = self._staging.pop(next(iter(self._staging)))
                for _, ev in old['slots']:
                    if ev is not None:
                        ev.synchronize()           # its last copy has left the pinned buffer before the buffer is freed
            ring = self._staging[key] = {'slots': [], 'next': 0}
        else:
            self._staging[key] = self._staging.pop(key)      # most recently used last
        if len(ring['slots']) < 8:
            ring['slots'].append([torch.empty(host.numel(), dtype=host.dtype).pin_memory(), None])
            slot = ring['slots'][-1]
        else:
            slot = ring['slots'][ring['next']]
            ring['next'] = (ring['next'] + 1) % len(ring['slots'])
            if slot[1] is not None:
                slot[1].synchronize()
        slot[0].copy_(host.reshape(-1))
        out = slot[0].to(dev, non_blocking=True).view(host.shape)
        slot[1] = torch.cuda.Event()
        slot[1].record(torch.cuda.current_stream(dev))
        return out

    def _build_tables(self, entries, dev):
        """entries: list of (param, grad, exp_avg, exp_avg_sq, lr, wd).  Device tables for csrc/optim.hip.  Gradient tensors are
        re-allocated at new addresses by zero_grad(set_to_none=True), i.e. every step in the reference's loop
        (src/train/pix4point_trainer.py:162), so this runs per step and must be cheap and must not synchronise host and device:
        the chunk list depends on the tensor sizes only (cached on the device; built vectorised), the descriptor table is
        filled through numpy and uploaded from a pinned staging ring.  (Round 4: the per-chunk Python loop + two pin_memory()
        calls cost 34 ms per step for ViT-L -- the N = 8192, batch 8 step was HOST-bound, 74 ms against ~60 ms of kernels.)"""
        import numpy as np
        chunk = L.call_int('p4p_opt_chunk_elems')
        numels = tuple(int(e[0].numel()) for e in entries)
        ck = (numels, chunk, str(dev))
        cached = self._chunk_cache.get(ck)
        if cached is None:
            counts = (np.asarray(numels, dtype=np.int64) + chunk - 1) // chunk
            owner = np.repeat(np.arange(len(numels), dtype=np.int64), counts)
            first = np.repeat(np.cumsum(counts) - counts, counts)
            chunks = np.stack([owner, np.arange(int(counts.sum()), dtype=np.int64) - first], axis=1).astype(np.int32)
            cached = (self._upload(torch.from_numpy(chunks), dev), int(chunks.shape[0]))
            if len(self._chunk_cache) > 16:
                self._chunk_cache.clear()
            self._chunk_cache[ck] = cached
        ch, nchunks = cached
        desc = np.empty((len(entries), 6), dtype=np.int64)
        desc[:, 0] = [e[0].data_ptr() for e in entries]
        desc[:, 1] = [e[1].data_ptr() for e in entries]
        desc[:, 2] = [e[2].data_ptr() for e in entries]
        desc[:, 3] = [e[3].data_ptr() for e in entries]
        desc[:, 4] = numels
        hp = np.empty((len(entries), 2), dtype=np.float32)     # (lr, wd) as two floats in the sixth 8-byte field
        hp[:, 0] = [e[4] for e in entries]
        hp[:, 1] = [e[5] for e in entries]
        desc[:, 5] = hp.view(np.int64).reshape(-1)
        return self._upload(torch.from_numpy(desc), dev), ch, nchunks

    @torch.no_grad()
    def step(self, closure=None, max_norm=None, grad_scale=1.0):
        """One AdamW update.  With ``max_norm`` the gradients are first scaled by min(1, max_norm / (||g||_2 + 1e-6)) over
        ALL parameters of this optimizer (clip_grad_norm_ semantics); returns the total norm (0-dim device tensor).
        ``grad_scale`` multiplies every gradient first (data parallel: the all-reduce leaves SUMS in ``.grad`` and 1/world
        travels here, folded into the one factor the update kernel applies: no pass over the gradient buffer)."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        by_hyper = {}
        dev = None
        for group in self.param_groups:
            beta1, beta2 = group['betas']
            for p in group['params']:
                if p.grad is None:
                    continue
                if p.grad.is_sparse or p.dtype != torch.float32 or p.grad.dtype != torch.float32:
                    raise RuntimeError('FusedAdamW: dense fp32 parameters and gradients only')
                if not p.is_cuda:
                    raise RuntimeError('FusedAdamW: parameters must live on the GPU (no CPU fallback)')
                if not (p.is_contiguous() and p.grad.is_contiguous()):
                    raise RuntimeError('FusedAdamW: parameters and gradients must be contiguous')
                dev = p.device
                st = self.state[p]
                if len(st) == 0:
                    st['step'] = torch.zeros((), dtype=torch.float32)
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if st['step'].is_cuda:                      # checkpoints of torch's fused AdamW keep it on the device
                    st['step'] = st['step'].cpu()
                st['step'] += 1
                key = (beta1, beta2, group['eps'], float(st['step']))
                by_hyper.setdefault(key, []).append((p, p.grad, st['exp_avg'], st['exp_avg_sq'], float(group['lr']),
                                                     float(group['weight_decay'])))
        if not by_hyper:
            return loss
        # one launch set per (betas, eps, step) combination: a single one in practice
        entries = [e for es in by_hyper.values() for e in es]
        ident = tuple((e[0].data_ptr(), e[1].data_ptr(), e[2].data_ptr(), e[3].data_ptr(), e[4], e[5]) for e in entries)
        if ident != self._key:
            self._tables = {}
            for key, es in by_hyper.items():
                self._tables[key[:3] + (len(es), id(es[0][0]))] = self._build_tables(es, dev)
            # a single (betas, eps, step) subset -- the normal case -- IS the list of all entries: one table serves both
            allt = next(iter(self._tables.values())) if len(by_hyper) == 1 else self._build_tables(entries, dev)
            self._tables['all'] = allt + (torch.empty(allt[2], dtype=torch.float32, device=dev),
                                          torch.empty(2, dtype=torch.float32, device=dev))
            self._key = ident
        stream = L.ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        coef = None
        total_norm = None
        if max_norm is not None or grad_scale != 1.0:
            desc, ch, n, partial, out = self._tables['all']
            L.call('p4p_grad_sumsq', L.ptr(desc), L.ptr(ch), n, L.ptr(partial), stream)
            L.call('p4p_clip_coef', L.ptr(partial), n, float(max_norm) if max_norm is not None else float('inf'),
                   float(grad_scale), L.ptr(out), stream)
            coef, total_norm = out, out[1]
        for (beta1, beta2, eps, step), es in by_hyper.items():
            # one table per (betas, eps, STEP COUNT) subset: parameters that share the hyper-parameters but not the step
            # count (first gradient later, added group, mixed loaded state) get their own bias corrections
            desc, ch, n = self._tables[(beta1, beta2, eps, len(es), id(es[0][0]))][:3]
            L.call('p4p_adamw_step', L.ptr(desc), L.ptr(ch), n, L.ptr(coef), float(beta1), float(beta2), float(eps),
                   float(1.0 - beta1 ** step), float(1.0 - beta2 ** step), stream)
        # the update kernel wrote the parameters through raw pointers: tell torch (autograd's in-place checks, and the
        # bf16 operand caches of functional._bf16_weights, which are keyed on the version counter and MUST be re-made)
        torch._C._increment_version([e[0] for e in entries])
        return total_norm if max_norm is not None else loss
