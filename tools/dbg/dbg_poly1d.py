"""dev: where does a dense block shape differ from the general kernel on the polyphase backward-data case of tests/test_dconv_gpu.py"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, 'transtacos-retunegan_amd')); sys.path.insert(0, os.path.join(REPO, 'tests'))
import numpy as np, torch, torch.nn.functional as F
import packref
import test_dconv_gpu as T
B, Cin, Cout, L = 9, 256, 512, 102
K, s, p = 5, 3, 2
gen = torch.Generator().manual_seed(19)
x = torch.randn(B, Cin, L, generator=gen, dtype=torch.float64)
w = torch.randn(Cout, Cin, K, generator=gen) / np.sqrt(Cin * K)
Lo = (L + 2 * p - K) // s + 1
dy = torch.randn(B, Cout, Lo, generator=gen)
W = packref.logical_dgrad_poly(w.numpy(), 1, s)
nt = W.shape[-1]
wp = torch.from_numpy(T._both_images(W)).cuda()
nq = (L - 1 + p) // s + 1
kw = T._desc(B, Cout, Lo, Cin * s, nt, 1, nt - 1, nq, Cin, L, shuf_S=s, shuf_P=p, mask_slope=0.15)
xm, dyd = x.float().cuda(), dy.cuda()
rc, base = T._run(kw, dyd, wp, mask=xm, out_shape=(B, Cin, L))
for c in T._codes(kw):
    rc, out = T._run(kw, dyd, wp, mask=xm, out_shape=(B, Cin, L), cfg=c)
    bad = (out != base)
    print(c, rc, int(bad.sum()), 'max abs diff', float((out - base).abs().max()))
    if bad.any():
        idx = bad.nonzero()
        print(' clips', sorted(set(idx[:, 0].tolist())), 'positions', sorted(set(idx[:, 2].tolist()))[:40], 'n channels', len(set(idx[:, 1].tolist())))
