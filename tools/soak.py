#!/usr/bin/env python3
"""Stability soak (dev tool): N train steps on a fixed synthetic batch, losses printed every 25 steps.
usage: soak.py [fp32|bf16] [steps]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'transtacos-retunegan_amd')); sys.path.insert(0, REPO)
import torch, bench
import hparam as hp
hp.compute_dtype = sys.argv[1] if len(sys.argv) > 1 else 'fp32'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
from train import Trainer
torch.manual_seed(hp.randseed)
tr = Trainer(use_mpd=True, use_mtd=True, d_train_times=2, dev='cuda')
data = bench.synthetic_batch(8, 8192, 1, 'cuda')
for i in range(n):
    dl, gl = tr.train_step(*data)
    if i % 25 == 0 or i == n - 1:
        print(i, 'D %.4f' % dl['disc_all'].item(), 'G %.4f' % gl['gen_all'].item(), 'mstft %.4f' % gl['mstft'].item(),
              'finite', bool(torch.isfinite(tr.generator.bank().flat).all()), flush=True)
