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
