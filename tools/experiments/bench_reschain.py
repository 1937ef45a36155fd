#!/usr/bin/env python3
"""dev: rtg_reschain_forward / _backward of one ResBlock3 branch (32 channels, 32 clips x 8192) against the three resconv launches"""
import ctypes as C, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, 'transtacos-retunegan_amd'))
import torch
from rtg import ops
from models.generator import ResBlock3
from models.layers import BankedModel

class Net(BankedModel):
    def __init__(self, k):
        super().__init__()
        self.blk = ResBlock3(32, k, (9, 3, 1))
    def forward(self, x):
        return self.blk.run(self.token(), x)

def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

B, Lx = 32, 8192
for k in (3, 5, 7):
    net = Net(k).cuda(); net.bank()
    x = torch.randn(B, 32, Lx, device='cuda'); dy = torch.randn_like(x)
    res = {}
    for fused in (True, False):
        ops.RESCHAIN = fused
        with torch.no_grad():
            tf = timeit(lambda: net(x))
        xg = x.clone().requires_grad_(True)
        def fb():
            net.zero_grad(); y = net(xg); y.backward(dy)
        tfb = timeit(fb)
        res[fused] = (tf, tfb)
    flop = 3 * 2.0 * B * Lx * 32 * 32 * k
    print(f'k{k}: fused fwd {res[True][0]:6.1f} us ({flop / res[True][0] / 1e6:5.1f} TF/s)  unfused {res[False][0]:6.1f} us | fwd+bwd (with wgrads, token) fused {res[True][1]:7.1f} unfused {res[False][1]:7.1f}', flush=True)
