"""data.PinnedFeeder on the GPU box: the defining piece of BASELINE configs[4] ("pinned async H2D").  A slow producer
thread fills page-locked buffers, the copies run on a side stream, next() orders them before the consumer's first use
without a host sync."""
import os
import sys
import threading
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, 'transtacos-retunegan_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)


def _host_batch(step):
    g = np.random.RandomState(1000 + step)
    return (torch.from_numpy(g.rand(4, 80, 32).astype(np.float32)),
            torch.from_numpy(g.rand(4, 1, 8192).astype(np.float32)),
            torch.from_numpy(np.full((4, 1, 8192), float(step), np.float32)))


def test_pinned_feeder_delivers_every_batch_in_order_and_stops_cleanly():
    import data as D
    n_threads = threading.active_count()

    def slow(step):
        time.sleep(0.004)                                   # host DSP stand-in: slower than the consumer at times
        return _host_batch(step)

    f = D.PinnedFeeder(slow, 'cuda:0')
    busy = torch.empty(1 << 22, device='cuda')
    kept = []
    for step in range(50):
        mel, tmpl, wav = f.next()
        assert mel.is_cuda and mel.shape == (4, 80, 32) and wav.shape == (4, 1, 8192)
        # consume on the training stream only (no host sync here): a copy of what was delivered, behind a kernel that keeps
        # the stream busy so that a device slot recycled too early would be overwritten before this copy ran
        busy.normal_()
        kept.append((mel.clone(), tmpl.clone(), wav.clone()))
        if step % 7 == 0:
            time.sleep(0.01)                                # ... and a consumer that is sometimes the slow side
    torch.cuda.synchronize()
    for step, got in enumerate(kept):
        for g_, h_ in zip(got, _host_batch(step)):
            assert torch.equal(g_.cpu(), h_), step
    f.close()
    f.close()                                               # idempotent
    assert not f._thread.is_alive()
    assert threading.active_count() == n_threads
    with pytest.raises(RuntimeError):
        f.next()


def test_pinned_feeder_reports_a_failing_producer():
    import data as D

    def bad(step):
        if step == 2:
            raise ValueError('boom')
        return _host_batch(step)

    f = D.PinnedFeeder(bad, 'cuda:0')
    f.next(); f.next()
    with pytest.raises(RuntimeError, match='producer thread failed'):
        f.next()
    f.close()
    assert not f._thread.is_alive()


def test_pinned_feeder_close_without_next_and_with_unconsumed_batches():
    import data as D
    f = D.PinnedFeeder(_host_batch, 'cuda:0')
    f.close()                                               # never started
    f2 = D.PinnedFeeder(_host_batch, 'cuda:0')
    f2.next()
    time.sleep(0.05)                                        # the producer has filled the ready queue and blocks on it
    f2.close()
    assert not f2._thread.is_alive()


def test_config5_finetune_feed_drives_the_graphed_step():
    """BASELINE configs[4] the way it is stated: Griffin-Lim reference waves on the host -> pinned double-buffered async H2D
    (data.PinnedFeeder, exactly bench.py's `finetune_feeder`) -> Trainer.train_step_graphed on the full stack at 16 clips x
    22016 samples (86 frames), 12 steps.  The device batches the step consumed equal the host batches the producer made (in
    order), the losses stay finite, the generator moves."""
    import bench
    import data as D
    from train import Trainer
    torch.manual_seed(11)
    host = []
    feeder = bench.finetune_feeder(16, 22016, 5, 'cuda:0', pool=6)
    make = feeder.make_batch

    def recording(step):
        b = make(step)
        host.append(tuple(t.clone() for t in b))
        return b
    feeder.make_batch = recording
    tr = Trainer(use_mpd=True, use_mtd=True, d_train_times=2, dev='cuda:0')
    before = tr.generator.bank().flat.clone()
    seen, losses = [], []
    try:
        for step in range(12):
            x, y_tmpl, y = feeder.next()
            assert x.shape == (16, 80, 86) and y_tmpl.shape == (16, 1, 22016) and y.shape == (16, 1, 22016)
            seen.append((x.clone(), y_tmpl.clone(), y.clone()))
            dl, gl = tr.train_step_graphed(x, y_tmpl, y)
            losses.append((dl['disc_all'].clone(), gl['gen_all'].clone()))     # (static scalars: overwritten by the next replay)
        torch.cuda.synchronize()
    finally:
        feeder.close()
    assert tr._graphs is not None                                             # the steps were graph replays
    for step, got in enumerate(seen):
        for g_, h_ in zip(got, host[step]):
            assert torch.equal(g_.cpu(), h_), step
    vals = np.array([[a.item(), b.item()] for a, b in losses])
    assert np.isfinite(vals).all(), vals
    assert np.abs(np.diff(vals[:, 1])).max() > 0                              # different batches, different losses
    moved = (tr.generator.bank().flat - before).abs().max().item()
    assert 0 < moved < 12 * 4 * 1.8e-4 and torch.isfinite(tr.generator.bank().flat).all()
