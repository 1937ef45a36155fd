#!/usr/bin/env python3
"""cProfile of the host side of one train step (dev tool; best with the empty-kernel library so that nothing waits)."""
import cProfile, pstats, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'transtacos-retunegan_amd')); sys.path.insert(0, REPO)
import torch, bench
import hparam as hp
from train import Trainer
torch.manual_seed(hp.randseed)
tr = Trainer(use_mpd=True, use_mtd=False, d_train_times=2, dev='cuda')
data = bench.synthetic_batch(32, 8192, 1, 'cuda')
for _ in range(4): tr.train_step(*data)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10): tr.train_step(*data)
host = (time.perf_counter() - t) / 10
torch.cuda.synchronize()
print('host issue time per step: %.2f ms' % (host * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(5): tr.train_step(*data)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(28)
