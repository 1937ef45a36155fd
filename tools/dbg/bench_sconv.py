#!/usr/bin/env python3
"""dev: the stride-1 layers at the bottom of the UNet (1024 columns at batch 32): general block shapes against the split-K
codes of rtg_sconv.hip (9004 / 9008)."""
import ctypes as C
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, 'transtacos-retunegan_amd'))
sys.path.insert(0, os.path.join(REPO, 'tests'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import packref  # noqa: E402
from rtg import tune  # noqa: E402
from rtg.lib import lib, Conv1dDesc  # noqa: E402

P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
B = 32
for name, C1, C2, Cout, L, K, dil, split in (('conv_fuse fwd', 80, 128, 256, 32, 7, 1, 0), ('conv_fuse dgrad', 256, 0, 208, 32, 7, 1, 80),
                                             ('resstack k3 d1', 128, 0, 128, 32, 3, 1, 0), ('resstack k3 d9', 128, 0, 128, 32, 3, 9, 0),
                                             ('resblock3 128ch k7 L256', 128, 0, 128, 256, 7, 1, 0)):
    Cin = C1 + C2
    if L > 64:
        print(name, 'rows too long for rtg_sconv'); continue
    w = (np.random.RandomState(1).randn(Cout, Cin, K) / np.sqrt(Cin * K)).astype(np.float32)
    W = packref.logical_fwd(w, 1)
    tm = 16 if Cout % 32 else 32
    wp = torch.from_numpy(np.concatenate([packref.pack_logical(W, tm), packref.pack_frag16(W)])).cuda()
    x1 = torch.randn(B, C1, L, device='cuda'); x2 = torch.randn(B, C2, L, device='cuda') if C2 else None
    o1 = torch.empty(B, split or Cout, L, device='cuda'); o2 = torch.empty(B, Cout - split, L, device='cuda') if split else None
    pad = dil * (K - 1) // 2
    d = Conv1dDesc(B=B, C1=C1, C2=C2, L_in=L, groups=1, Cg=Cin, Mg=Cout, K=K, stride=1, dil=dil, pad=pad, Q=L, out_C=Cout, out_L=L,
                   shuf_S=1, shuf_P=0, pre_mode=1, pre_slope=0.15, mask_slope=1.0, out_scale=1.0, act=0, act_slope=1.0, accumulate=0,
                   tile_m=tm, out_split=split, wp16=2)
    cands = (C.c_int * 48)()
    n = lib.rtg_conv1d_tile_candidates(C.byref(d), cands, 48)
    fl = 2.0 * B * L * Cout * Cin * K
    line = f'{name:24s}'
    best = None
    tune.REPS = 20
    for c in list(cands[:n]):
        d.tile_cfg = c
        t = tune._time(lambda: lib.rtg_conv1d(C.byref(d), P(x1), P(x2), None, P(wp), None, None, None, P(o1), P(o2), None))
        if t is None:
            continue
        if c > 9000:
            line += f'  {c}: {t * 1e3:6.1f} us {fl / t / 1e9:5.1f} TF'
        elif best is None or t < best[1]:
            best = (c, t)
    print(line + f'  | general best {best[0]}: {best[1] * 1e3:6.1f} us {fl / best[1] / 1e9:5.1f} TF', flush=True)

# ---- the strided / transposed convs at the bottom (downs.2, ups.0): strided walk and polyphase operator
wd = (np.random.RandomState(2).randn(128, 64, 15) / 30).astype(np.float32)          # downs.2 weight [C_out, C_in, K]
wu = (np.random.RandomState(3).randn(256, 128, 15) / 30).astype(np.float32)         # ups.0 weight [C_in, C_out, K]
CASES = (
    # name, logical operator, C_in, rows, L_in, K, stride, pad, Q, out_C, out_L, shuf_S, shuf_P
    ('downs.2 fwd', packref.logical_fwd(wd, 1), 64, 128, 256, 15, 8, 7, 32, 128, 32, 1, 0),
    ('downs.2 dgrad', packref.logical_dgrad_poly(wd, 1, 8), 128, 512, 32, 2, 1, 1, 33, 64, 256, 8, 7),
    ('ups.0 fwd', packref.logical_convT_poly(wu, 8), 256, 1024, 32, 2, 1, 1, 33, 128, 256, 8, 7),
    ('ups.0 dgrad', packref.logical_convT_dgrad(wu), 128, 256, 256, 15, 8, 7, 32, 256, 32, 1, 0),
)
for name, W, Cin, Mg, L, K, S, pad, Q, out_C, out_L, sS, sP in CASES:
    wp = torch.from_numpy(np.concatenate([packref.pack_logical(W, 32), packref.pack_frag16(W)])).cuda()
    x1 = torch.randn(B, Cin, L, device='cuda')
    o1 = torch.empty(B, out_C, out_L, device='cuda')
    d = Conv1dDesc(B=B, C1=Cin, C2=0, L_in=L, groups=1, Cg=Cin, Mg=Mg, K=K, stride=S, dil=1, pad=pad, Q=Q, out_C=out_C, out_L=out_L,
                   shuf_S=sS, shuf_P=sP, pre_mode=1, pre_slope=0.15, mask_slope=1.0, out_scale=1.0, act=0, act_slope=1.0, accumulate=0,
                   tile_m=32, out_split=0, wp16=2)
    cands = (C.c_int * 48)()
    n = lib.rtg_conv1d_tile_candidates(C.byref(d), cands, 48)
    fl = 2.0 * B * Q * Mg * Cin * K
    line = f'{name:24s}'
    best = None
    for c in list(cands[:n]):
        d.tile_cfg = c
        t = tune._time(lambda: lib.rtg_conv1d(C.byref(d), P(x1), None, None, P(wp), None, None, None, P(o1), None, None))
        if t is None:
            continue
        if c > 9000:
            line += f'  {c}: {t * 1e3:6.1f} us {fl / t / 1e9:5.1f} TF'
        elif best is None or t < best[1]:
            best = (c, t)
    print(line + f'  | general best {best[0]}: {best[1] * 1e3:6.1f} us {fl / best[1] / 1e9:5.1f} TF', flush=True)
