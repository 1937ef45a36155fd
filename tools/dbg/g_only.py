#!/usr/bin/env python3
"""Wall time of the generator alone: forward, and forward + backward (+ weight-gradient flush) on synthetic input (dev tool)."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, 'transtacos-retunegan_amd'))
sys.path.insert(0, REPO)
import torch  # noqa: E402
import bench  # noqa: E402
import hparam as hp  # noqa: E402
from models import Generator_RefineGAN_small  # noqa: E402
from rtg import tune  # noqa: E402

torch.manual_seed(hp.randseed)
g = Generator_RefineGAN_small().cuda().train()
x, y_tmpl, y = bench.synthetic_batch(32, 8192, 1, 'cuda')
dy = torch.randn_like(y)
tune.ACTIVE = True
for _ in range(2):
    g.zero_grad(); g(x, y_tmpl).backward(dy)
tune.ACTIVE = False
for _ in range(3):
    g.zero_grad(); g(x, y_tmpl).backward(dy)
torch.cuda.synchronize()


def wall(fn, n=20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()          # host time to issue
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3


def fwd():
    with torch.no_grad():
        g(x, y_tmpl)


def fb():
    g.zero_grad()
    g(x, y_tmpl).backward(dy)


for name, fn in (('forward (no grad)', fwd), ('forward + backward', fb)):
    host, tot = wall(fn)
    print(f'{name:22s} host issue {host:6.3f} ms   wall {tot:6.3f} ms')
