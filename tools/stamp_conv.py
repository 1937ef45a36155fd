#!/usr/bin/env python3
"""In-kernel phase timing of one conv shape (dev tool; needs tools/dev_build.sh's librtg_dev.so via RTG_DEV_LIB).
usage: stamp_conv.py B Cin Cout L K stride dil pad groups
Stamps per (block, wave): 0 start, 1 first patch requested, 2 first chunk published, 3 main loop done, 4 end."""
import ctypes as C
import sys
import numpy as np
import torch
import bench_conv as bc
import rtg.lib as RL
dll = C.CDLL(RL.LIB_PATH)
args = [int(v) for v in sys.argv[1:10]]
bc.run('fwd', *args, iters=5)
dll.rtg_dev_stamp_clear()
torch.cuda.synchronize()
bc.run('fwd', *args, iters=1)
torch.cuda.synchronize()
n = 1 << 23
host = np.zeros(n, dtype=np.uint64)
dll.rtg_dev_stamp_read(host.ctypes.data_as(C.c_void_p), C.c_longlong(n))
st = host.reshape(-1, 8).astype(np.int64)
st = st[st[:, 0] != 0]
print('waves with stamps', len(st))
t0 = st[:, 0].min()
d = st[:, :5] - st[:, :1]
names = ['start->stage issued', 'stage->published', 'main loop', 'epilogue']
for i in range(4):
    seg = st[:, i + 1] - st[:, i]
    print(f'{names[i]:22s} median {np.median(seg):9.0f}  p10 {np.percentile(seg, 10):9.0f}  p90 {np.percentile(seg, 90):9.0f} shader cycles')
tot = st[:, 4] - st[:, 0]
print('block lifetime median', np.median(tot), ' kernel span', st[:, 4].max() - t0)
print('start spread (p90-p10)', np.percentile(st[:, 0] - t0, 90) - np.percentile(st[:, 0] - t0, 10), 'last start', (st[:, 0] - t0).max())

# per-step stamps of the first blocks: 0 entry, 1 prefetches issued, 2 MFMAs issued, 3 swrite done, 4 stage issued, 5 barrier passed, 7 exit
s2 = host[(4 << 20):(4 << 20) + 512 * 4 * 128].reshape(-1, 128).astype(np.int64)
s2 = s2[s2[:, 0] != 0]
ns = 16
a = s2.reshape(len(s2), ns, 8)
print('per-step medians over', len(a), 'waves: prefetch | mfma | swrite | stage | barrier | tail || total')
def med(x):
    return float(np.median(x))
for st_ in range(ns - 1):
    e = a[:, st_, :]
    pub = (e[:, 3] > 0).all()
    if pub:
        print(f'  step {st_:2d}: {med(e[:,1]-e[:,0]):6.0f} {med(e[:,2]-e[:,1]):6.0f} | {med(e[:,3]-e[:,2]):6.0f} {med(e[:,4]-e[:,3]):6.0f} {med(e[:,5]-e[:,4]):6.0f} {med(e[:,7]-e[:,5]):6.0f} || {med(a[:,st_+1,0]-e[:,0]):7.0f}')
    else:
        print(f'  step {st_:2d}: {med(e[:,1]-e[:,0]):6.0f} {med(e[:,2]-e[:,1]):6.0f} | {med(e[:,7]-e[:,2]):6.0f} || {med(a[:,st_+1,0]-e[:,0]):7.0f}')
