#!/usr/bin/env python3
"""Wall time of the phases of one train step (dev tool): G forward, each D step, the G step split into its forward part
(losses + frozen D forward) and backward + update.  Every phase is bracketed by a device synchronisation, so the sum is
a little above the step time (no overlap across phase boundaries)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'transtacos-retunegan_amd')); sys.path.insert(0, REPO)
import torch, bench
import hparam as hp
from train import Trainer
from models.loss import stft_cache
torch.manual_seed(hp.randseed)
tr = Trainer(use_mpd=True, use_mtd=False, d_train_times=2, dev='cuda')
data = bench.synthetic_batch(32, 8192, 1, 'cuda')
for _ in range(4): tr.train_step(*data)
x, y_tmpl, y = data
acc = {}
def tick(name, t0):
    torch.cuda.synchronize(); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
N = 10
for _ in range(N):
    with stft_cache():
        torch.cuda.synchronize(); t = time.perf_counter()
        y_hat = tr.generator(x, y_tmpl); tick('G forward', t)
        y_det = y_hat.detach()
        for i in range(2):
            t = time.perf_counter(); tr.d_step(y, y_det); tick('D step', t)
        t = time.perf_counter(); tr.g_step(y, y_hat); tick('G step (D fwd + D bwd-data + G bwd + update)', t)
tot = 0
for k, v in acc.items():
    print('%-50s %7.2f ms' % (k, v / N * 1e3)); tot += v
print('%-50s %7.2f ms' % ('sum', tot / N * 1e3))
