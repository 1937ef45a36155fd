#!/usr/bin/env python3
"""dev: weight gradients of the MSD thin-group layers — every shape code the library lists (general matrix-core shapes,
9 = rtg_gconv_wgrad on the vector ALUs) at the train step's batch."""
import ctypes as C
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, 'transtacos-retunegan_amd'))
import torch  # noqa: E402
from rtg import tune  # noqa: E402
from rtg.lib import lib, WgradDesc  # noqa: E402

P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
B = int(os.environ.get('GW_B', '64'))
LAYERS = [(32, 64, 4, 2), (64, 128, 8, 2), (128, 512, 32, 4), (512, 512, 64, 4)]
for sub, L0 in ((0, 8192), (1, 4096), (2, 2048)):
    L = L0
    for cin, cout, g, s in LAYERS:
        Lo = (L + 40 - 40 - 1) // s + 1
        x = torch.randn(B, cin, L, device='cuda')
        dy = torch.randn(B, cout, Lo, device='cuda')
        need = cout * ((cin // g) * 41 + 1)
        fl = 2.0 * B * Lo * cout * (cin // g) * 41
        probe = WgradDesc(B=B, C1=cin, C2=0, L_in=L, groups=g, Cg=cin // g, Mg=cout // g, K=41, stride=s, dil=1, pad=20, Q=Lo,
                          dy_L=Lo, pre_mode=1, pre_slope=0.15, gy_mode=0, gy_slope=1.0, gy_scale=1.0, splits=1, part_stride=0)
        cands = (C.c_int * 12)()
        n = lib.rtg_wgrad_shape_candidates(C.byref(probe), cands, 12)
        line = f'd{sub} {cin}->{cout} g{g} s{s} L{L}:'
        for c in list(cands[:n]):
            wd = WgradDesc(B=B, C1=cin, C2=0, L_in=L, groups=g, Cg=cin // g, Mg=cout // g, K=41, stride=s, dil=1, pad=20, Q=Lo,
                           dy_L=Lo, pre_mode=1, pre_slope=0.15, gy_mode=0, gy_slope=1.0, gy_scale=1.0, splits=1, part_stride=0,
                           shape_cfg=c)
            sp = lib.rtg_wgrad_splits(C.byref(wd))
            if sp < 1:
                continue
            part = torch.empty(sp * need, device='cuda')
            wd.splits, wd.part_stride = sp, need
            tune.REPS = 20
            t = tune._time(lambda: lib.rtg_conv1d_wgrad(C.byref(wd), P(x), None, P(dy), None, P(part), None))
            if t:
                line += f'  s{c} x{sp}: {t * 1e3:6.1f} us {fl / t / 1e9:5.1f} TF'
        print(line, flush=True)
        L = Lo
