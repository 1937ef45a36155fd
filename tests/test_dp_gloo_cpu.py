"""Data-parallel path (train.DataParallel) on CPU: two processes, gloo backend, 127.0.0.1 rendezvous.
Checks: gradients of a flat buffer are summed across ranks, parameters are broadcast from rank 0, the NaN flag is
collective, and the averaged-gradient AdamW update equals the single-process update on the concatenated batch."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _FakeBank:
    """Stands in for rtg.bank.WeightBank on CPU: flat parameter / gradient buffers."""

    def __init__(self, n, seed):
        g = torch.Generator().manual_seed(seed)
        self.flat = torch.randn(n, generator=g)
        self.n_params = n
        self.gflat = torch.zeros(n + 1)                    # gradients + the loss flag slot (rtg/bank.py)
        self.on_flush = None

    def flag(self):
        return self.gflat[self.n_params:]


class _FakeModel:
    def __init__(self, n, seed):
        self._b = _FakeBank(n, seed)

    def bank(self):
        return self._b


def _worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                    'transtacos-retunegan_amd'))
    from train import DataParallel
    m = _FakeModel(1000, seed=10 + rank)                  # ranks start with DIFFERENT parameters
    dp = DataParallel([m])
    assert dp.enabled and dp.world == world
    dp.broadcast_parameters()
    ref = _FakeBank(1000, 10).flat
    assert torch.equal(m.bank().flat, ref)                # everyone now holds rank 0's parameters
    # per-rank gradient = rank-dependent; all-reduce (async) sums them
    m.bank().gflat[:1000].copy_(torch.arange(1000, dtype=torch.float32) * (rank + 1))
    m.bank().flag().fill_(1.5)                             # the loss of the step rides in the last slot
    dp.reduce_async(m.bank().gflat)
    dp.wait()
    expect = torch.arange(1000, dtype=torch.float32) * sum(r + 1 for r in range(world))
    assert torch.equal(m.bank().gflat[:1000], expect)
    assert m.bank().flag().item() == pytest.approx(1.5 * world)
    # collective NaN guard without a collective of its own: only rank 1 sees a NaN loss, the summed flag is NaN on every
    # rank (rtg_adamw then skips the update everywhere), the gradients are still the plain sums
    m.bank().gflat[:1000].copy_(torch.arange(1000, dtype=torch.float32) * (rank + 1))
    m.bank().flag().fill_(float('nan') if rank == 1 else 1.0)
    dp.reduce_async(m.bank().gflat)
    dp.wait()
    assert torch.isnan(m.bank().flag()).all()
    assert torch.equal(m.bank().gflat[:1000], expect)
    # tuner picks: rank 0 holds the timed tables, the others hold stale / partial ones; after sync_tuner every rank has
    # rank 0's picks for rank 0's problems, the digest agrees, and rank 0's "met an unknown problem" flag reaches everybody
    from rtg import tune
    for name in tune._TABLES:
        getattr(tune, name).clear()
    if rank == 0:
        tune._conv.update({b'\x01\x02': 3211, b'\x03': 8108})
        tune._wgrad[b'\x09'] = 12
        tune._alt[b'gconv\x00'] = 1
        tune.MISSED = True
    else:
        tune._conv[b'\x01\x02'] = 1611          # a stale pick of its own for the same problem
        tune.MISSED = False
    missed = dp.sync_tuner()
    assert missed is True
    assert tune._conv == {b'\x01\x02': 3211, b'\x03': 8108} and tune._wgrad == {b'\x09': 12} and tune._alt == {b'gconv\x00': 1}
    digests = [None] * world
    dist.all_gather_object(digests, tune.digest())
    assert len(set(digests)) == 1
    dp.pending_probe = dp.reduce_async(m.bank().gflat)     # a collective in flight ...
    dp.drain()                                             # ... is finished and every rank has arrived when drain returns
    assert dp.pending == []
    if rank == 0:
        out.put('ok')
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_gradient_allreduce_and_broadcast():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert q.get(timeout=5) == 'ok'
