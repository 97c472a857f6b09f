"""CPU, world_size 2, gloo: the gradient exchange (ddp.GradReducer) used for the N>1 path.
The HIP kernels are not involved (no GPU here); the reducer's bucketing, in-backward launches and
averaging are exercised with a small torch model whose 'blocks' report their gradients the way the
encoder's backward does."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from p4p_amd.ddp import GradReducer, broadcast_parameters


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)                 # different init per rank on purpose
        blocks = torch.nn.ModuleList([torch.nn.Linear(8, 8) for _ in range(3)])
        tail = torch.nn.Linear(8, 4)
        unused = torch.nn.Linear(8, 8)                # never gets a gradient (quirk 8)
        model = torch.nn.ModuleDict({'blocks': blocks, 'tail': tail, 'unused': unused})
        broadcast_parameters(model)
        w0 = [p.detach().clone() for p in model.parameters()]
        gathered = [torch.zeros_like(w0[0]) for _ in range(world)]
        dist.all_gather(gathered, w0[0])
        assert all(torch.equal(g, gathered[0]) for g in gathered), 'broadcast_parameters failed'

        buckets = [list(b.parameters()) for b in reversed(blocks)] + [list(tail.parameters())]
        red = GradReducer(buckets)
        for step in range(2):                         # two steps: zero_grad / reuse of the flat views
            red.zero_grad()
            x = torch.randn(5, 8, generator=torch.Generator().manual_seed(rank * 10 + step))
            h = x
            for b in blocks:
                h = torch.tanh(b(h))
            loss = tail(h).pow(2).mean()
            params = [p for b in blocks for p in b.parameters()]
            grads = torch.autograd.grad(loss, params + list(tail.parameters()))
            local = [g.clone() for g in grads]
            # "backward": blocks report last-to-first, the tail arrives through .grad accumulation
            for l in reversed(range(3)):
                ret = red.block_ready(l, list(blocks[l].parameters()), list(grads[2 * l:2 * l + 2]))
                assert ret == [None, None]
            for p, g in zip(tail.parameters(), grads[6:]):
                p.grad.add_(g)                        # what AccumulateGrad does with an existing .grad
            red.finish()
            # reference: average of the per-rank local gradients
            for p, g in zip(params + list(tail.parameters()), local):
                both = [torch.zeros_like(g) for _ in range(world)]
                dist.all_gather(both, g)
                want = sum(both) / world
                assert torch.allclose(p.grad, want, rtol=1e-6, atol=1e-7)
                assert p.grad.data_ptr() >= red.flat.data_ptr()          # still a view of the flat buffer
            assert all(p.grad is None for p in unused.parameters())
        # merged buckets (two blocks per all-reduce) and the sums + grad_scale hand-off the fused optimizer uses
        red2 = GradReducer([list(blocks[2].parameters()) + list(blocks[1].parameters()), list(blocks[0].parameters()),
                            list(tail.parameters())], blocks_per_bucket=[2, 1, 0])
        assert red2.grad_scale == 1.0 / world
        red2.zero_grad()
        g = torch.Generator().manual_seed(7 + rank)
        local = [torch.randn(p.shape, generator=g) for b in blocks for p in b.parameters()]
        for l in reversed(range(3)):
            views = red2.grad_views(list(blocks[l].parameters()))
            views[0].copy_(local[2 * l])                              # a kernel writing straight into the view
            ret = red2.block_ready(l, list(blocks[l].parameters()), [views[0], local[2 * l + 1]])
            assert ret == [None, None]
            assert len(red2._handles) == (0 if l == 2 else 1 if l == 1 else 2)      # bucket 0 waits for its second block
        red2.finish(average=False)
        for p, gl in zip([p for b in blocks for p in b.parameters()], local):
            both = [torch.zeros_like(gl) for _ in range(world)]
            dist.all_gather(both, gl)
            assert torch.allclose(p.grad, sum(both), rtol=1e-6, atol=1e-7)         # SUMS: 1/world travels as grad_scale
        out.put((rank, 'ok'))
    except Exception as e:                            # pragma: no cover
        out.put((rank, repr(e)))
        raise
    finally:
        dist.destroy_process_group()


def test_grad_reducer_two_ranks_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, 'ok'), (1, 'ok')], res


def test_grad_reducer_single_process_is_identity():
    lin = torch.nn.Linear(4, 4)
    red = GradReducer([list(lin.parameters())])
    lin(torch.ones(2, 4)).sum().backward()
    g = lin.weight.grad.clone()
    red.finish()
    assert torch.equal(lin.weight.grad, g)
    red.zero_grad()
    assert float(lin.weight.grad.abs().max()) == 0.0


def test_block_not_written_in_a_step_is_zeroed_not_reapplied():
    """zero_grad() leaves the block buckets alone (their backward overwrites them), so a block whose backward
    did not run in a step must not keep last step's sums: finish() zeroes every registered block without a block_done."""
    blocks = [torch.nn.Linear(4, 4) for _ in range(2)]
    tail = torch.nn.Linear(4, 2)
    red = GradReducer([list(blocks[1].parameters()) + list(blocks[0].parameters()), list(tail.parameters())],
                      blocks_per_bucket=[2, 0])
    for b in blocks:
        red.register_block(list(b.parameters()))
    ones = [torch.ones_like(p) for p in blocks[0].parameters()]
    for l in (1, 0):
        red.block_ready(l, list(blocks[l].parameters()), ones)
    red.finish(average=False)
    assert all(float(p.grad.min()) == 1.0 for b in blocks for p in b.parameters())
    # next step: the loss reaches block 1 only
    red.zero_grad()
    red.block_ready(1, list(blocks[1].parameters()), [2 * o for o in ones])
    red.finish(average=False)
    assert all(float(p.grad.min()) == 2.0 for p in blocks[1].parameters())
    assert all(float(p.grad.abs().max()) == 0.0 for p in blocks[0].parameters()), 'stale sums survived'
    assert red._stale_zeroed == 1
    # a requires_grad flag flipped after construction is refused (the layout is fixed)
    blocks[0].weight.requires_grad_(False)
    with pytest.raises(RuntimeError, match='requires_grad'):
        red.finish()


def test_parameter_unfrozen_after_build_is_refused():
    """the staged-unfreezing direction.  A block that was frozen when the reducer was built is in no bucket; once it
    is unfrozen its gradient lands in a plain local .grad that no all-reduce ever sees -- finish() must refuse the step."""
    blocks = [torch.nn.Linear(4, 4) for _ in range(2)]
    tail = torch.nn.Linear(4, 2)
    for p in blocks[0].parameters():
        p.requires_grad_(False)                       # frozen at build time: not bucketed
    every = [p for m in blocks + [tail] for p in m.parameters()]
    red = GradReducer([list(blocks[1].parameters()), list(tail.parameters())], blocks_per_bucket=[1, 0], outside=every)
    red.register_block(list(blocks[1].parameters()))
    x = torch.ones(3, 4)
    tail(blocks[1](blocks[0](x))).sum().backward()
    red.block_done(list(blocks[1].parameters()))
    red.finish()                                      # still frozen: fine
    assert all(p.grad is None for p in blocks[0].parameters())
    for p in blocks[0].parameters():
        p.requires_grad_(True)                        # staged unfreezing without a new reducer
    red.zero_grad()
    tail(blocks[1](blocks[0](x))).sum().backward()
    red.block_done(list(blocks[1].parameters()))
    with pytest.raises(RuntimeError, match='outside the buckets'):
        red.finish()


def test_force_collectives_needs_a_process_group():
    lin = torch.nn.Linear(4, 4)
    with pytest.raises(RuntimeError, match='process group'):
        GradReducer([list(lin.parameters())], force_collectives=True)


def test_reserved_cu_plan_follows_backend_and_environment(monkeypatch):
    """the CUs left to the collectives are decided in ONE place, and only a backend that runs on the GPU gets any
    (gloo's collectives run on the host: reserving 32 CUs for them only slowed the rehearsals down)."""
    from p4p_amd.ddp import planned_reserved_cus
    monkeypatch.delenv('P4P_RESERVED_CUS', raising=False)
    monkeypatch.delenv('NCCL_MAX_NCHANNELS', raising=False)
    assert planned_reserved_cus(1, None, 'nccl') is None            # world 1: the library is left alone
    assert planned_reserved_cus(8, None, 'gloo') is None            # host-side collectives hold no CU
    assert planned_reserved_cus(8, None, None) is None
    assert planned_reserved_cus(8, None, 'nccl') == 8               # RCCL's channel bound (configure_rccl_env default)
    assert planned_reserved_cus(8, None, 'cuda:nccl,cpu:gloo') == 8
    monkeypatch.setenv('NCCL_MAX_NCHANNELS', '16')
    assert planned_reserved_cus(8, None, 'nccl') == 16
    monkeypatch.setenv('P4P_RESERVED_CUS', '8')
    assert planned_reserved_cus(8, None, 'nccl') == 8 and planned_reserved_cus(1, None, 'gloo') == 8
    assert planned_reserved_cus(8, 0, 'nccl') == 0                  # the explicit argument wins


def test_reserved_cus_are_set_only_inside_the_backward_window():
    """building a reducer must not leave the process on CUs - 32 for good.  The reservation is a PLAN; the
    library-global switch is set from backward_begins() (the encoder's backward) to the end of finish() -- the only span in which
    an all-reduce runs beside this library's kernels -- and a step abandoned inside the window is repaired by zero_grad()."""
    from p4p_amd import ops
    lin = torch.nn.Linear(4, 4)
    start = ops.set_reserved_cus(8)
    try:
        red = GradReducer([list(lin.parameters())])
        assert red.reserve_cus() is None and red.reserved_cus is None       # world 1, no request: no plan
        red.backward_begins()
        assert ops.set_reserved_cus(8) == 8                                 # ... and no window
        assert red.reserve_cus(24) == 24 and red.reserved_cus == 24
        assert ops.set_reserved_cus(8) == 8                                 # a plan does not touch the library (forward: all CUs)
        lin(torch.ones(2, 4)).sum().backward()
        red.backward_begins()
        assert ops.set_reserved_cus(24) == 24                               # inside the window
        red.backward_begins()                                               # idempotent: the value to restore is not overwritten
        red.finish()
        assert ops.set_reserved_cus(8) == 8                                 # finish() puts back what was there before the window
        red.backward_begins()
        red.zero_grad()                                                     # a step abandoned before finish(): the next one starts clean
        assert ops.set_reserved_cus(8) == 8
        assert red.reserve_cus() == 24                                      # a later call repeats the first request
        red.backward_begins()
        red.release_cus()                                                   # train()'s finally: plan dropped, window closed
        assert red.reserved_cus is None and ops.set_reserved_cus(8) == 8
        red.backward_begins()
        assert ops.set_reserved_cus(8) == 8
    finally:
        ops.set_reserved_cus(start)
