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
        self._pinned = []

    # ------------------------------------------------------------------------------------------
    def _build_tables(self, entries, dev):
        """entries: list of (param, grad, exp_avg, exp_avg_sq, lr, wd).  Device tables for csrc/optim.hip.  The host
        copies are pinned and uploaded asynchronously on the current stream: rebuilding the tables (gradient tensors
        re-allocated by zero_grad(set_to_none=True) at new addresses) must not synchronise host and device."""
        chunk = L.call_int('p4p_opt_chunk_elems')
        desc = torch.empty(len(entries), 6, dtype=torch.int64).pin_memory()
        hp = torch.empty(len(entries), 2, dtype=torch.float32)
        chunks = []
        for i, (p, g, m, v, lr, wd) in enumerate(entries):
            desc[i, 0], desc[i, 1], desc[i, 2], desc[i, 3] = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr()
            desc[i, 4] = p.numel()
            hp[i, 0], hp[i, 1] = lr, wd
            chunks.extend((i, c) for c in range((p.numel() + chunk - 1) // chunk))
        desc[:, 5] = hp.view(torch.int64).view(-1)             # (lr, wd) as two floats in the sixth 8-byte field
        ch = torch.tensor(chunks, dtype=torch.int32).pin_memory()
        self._pinned.append((desc, ch))                        # keep the staging buffers alive until the copies have run
        if len(self._pinned) > 8:
            del self._pinned[:4]
        return desc.to(dev, non_blocking=True), ch.to(dev, non_blocking=True), len(chunks)

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
            allt = self._build_tables(entries, dev)
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
