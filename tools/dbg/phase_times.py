#!/usr/bin/env python3
"""Wall time of the phases of one train step (G forward | D steps | G step), events on the main stream (dev tool)."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, 'transtacos-retunegan_amd'))
sys.path.insert(0, REPO)
import torch  # noqa: E402
import bench  # noqa: E402
import train  # noqa: E402
import hparam as hp  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else 'config2'
desc, use_mpd, use_mtd, d_times, batch, T = bench.WORKLOADS[wl]
hp.compute_dtype = os.environ.get('RTG_DTYPE', 'fp32')
torch.manual_seed(hp.randseed)
tr = train.Trainer(use_mpd=use_mpd, use_mtd=use_mtd, d_train_times=d_times, dev='cuda')
x, y_tmpl, y = bench.synthetic_batch(batch, T, 1, 'cuda')
for _ in range(4):
    tr.train_step(x, y_tmpl, y)
torch.cuda.synchronize()
N = 10
acc = None
for _ in range(N):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3 + d_times)]
    ev[0].record()
    with train.stft_cache():
        y_hat = tr.generator(x, y_tmpl)
        ev[1].record()
        y_det = y_hat.detach()
        for i in range(d_times):
            tr.d_step(y, y_det)
            ev[2 + i].record()
        tr.g_step(y, y_hat)
        ev[2 + d_times].record()
    torch.cuda.synchronize()
    t = [ev[i].elapsed_time(ev[i + 1]) for i in range(len(ev) - 1)]
    acc = t if acc is None else [a + b for a, b in zip(acc, t)]
names = ['G forward'] + [f'D step {i + 1}' for i in range(d_times)] + ['G step (D fwd/bwd, G bwd, update)']
for n, a in zip(names, acc):
    print(f'{n:36s} {a / N:7.3f} ms')
print(f'{"total":36s} {sum(acc) / N:7.3f} ms')
