#!/usr/bin/env python3
"""dev: rtg_weightnorm_backward of each bank after a settled config-2 step, timed back to back (what the step's five launches
cost when they have the chip to themselves) with the bytes they move"""
import ctypes as C
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, 'transtacos-retunegan_amd'))
sys.path.insert(0, REPO)
import torch  # noqa: E402
import bench  # noqa: E402
from train import Trainer  # noqa: E402
import hparam as hp  # noqa: E402
from rtg.lib import lib  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else 'config2'
desc, use_mpd, use_mtd, d_times, batch, T = bench.WORKLOADS[wl]
torch.manual_seed(hp.randseed)
tr = Trainer(use_mpd=use_mpd, use_mtd=use_mtd, d_train_times=d_times, dev='cuda')
data = bench.synthetic_batch(batch, T, 1, 'cuda')
for _ in range(3):
    tr.train_step(*data)
torch.cuda.synchronize()
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, m in (('G', tr.generator), *[(type(d).__name__, d) for d in tr.discs]):
    b = m.bank()
    tab = b._wn_table
    n = len(b.layers)
    part = sum(ly.splits * ly.rows * (ly.inner + 1) * 4 for ly in b.layers)
    par = sum(ly.rows * (ly.inner + 1) * 4 for ly in b.layers)
    f = lambda: lib.rtg_weightnorm_backward(P(tab), n, b.max_rows, b.max_inner, P(b.flat), P(b.scales), P(b.flat), P(b.gflat), st)  # noqa: E731
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f'{name:28s} {n:3d} layers  partials {part / 1e6:7.1f} MB + 4 x params {par / 1e6:5.1f} MB   {us:7.1f} us   {(part + 4 * par) / us / 1e6:5.2f} TB/s', flush=True)
