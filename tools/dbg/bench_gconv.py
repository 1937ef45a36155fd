#!/usr/bin/env python3
"""rtg_gconv_forward against the matrix-core path (rtg_conv1d, best block shape) on the MSD thin-group layers (dev tool)."""
import ctypes as C
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, 'transtacos-retunegan_amd'))
import torch  # noqa: E402
import hparam  # noqa: E402,F401
from models import MultiScaleDiscriminator  # noqa: E402
from rtg import ops, tune  # noqa: E402
from rtg.lib import lib, GconvDesc  # noqa: E402

def bank_v(ly):
    return C.c_void_p(bank.flat.data_ptr() + 4 * ly.v_off)


def bank_s(ly):
    return C.c_void_p(bank.scales.data_ptr() + 4 * ly.scale_off)


msd = MultiScaleDiscriminator().cuda()
tok = msd.token()
bank = msd.bank()
B = 64
for sub, L0 in ((0, 8192), (1, 4096), (2, 2048)):
    L = L0
    for li, c in enumerate(msd.discriminators[sub].convs):
        ly = c._layer
        Lo = ops._conv_out_len(ly, L)
        if ly.groups > 1:
            x = torch.randn(B, ly.cin, L, device='cuda')
            out = torch.empty(B, ly.cout, Lo, device='cuda')
            d, _ = ops._fwd_desc(ly, B, ly.cin, L, 0.15)
            args = (ops._p(x), None, None, bank.fwd_ptr(ly), bank.bias_ptr(ly), None, None, ops._p(out), None, None)
            tune.ACTIVE = True
            d.tile_cfg = tune.conv_cfg(d, lambda: lib.rtg_conv1d(C.byref(d), *args))
            tune.ACTIVE = False
            gd = GconvDesc(B, ly.groups, ly.cin // ly.groups, ly.cout // ly.groups, ly.k, ly.stride, ly.pad, L, Lo, 0.15)
            gargs = (ops._p(x), ops._p(bank.gconv_weights(ly, gd, tok._rtg_id)), bank.bias_ptr(ly), ops._p(out), None)
            tune.REPS = 20
            t0 = tune._time(lambda: lib.rtg_conv1d(C.byref(d), *args))
            t1 = tune._time(lambda: lib.rtg_gconv_forward(C.byref(gd), *gargs))
            t2 = None
            if lib.rtg_gmfma_ok(C.byref(gd)) == 1:
                ref = out.clone()
                margs = (ops._p(x), ops._p(bank.gmfma_weights(ly, gd)), bank.bias_ptr(ly), ops._p(out), None)
                assert lib.rtg_gmfma_forward(C.byref(gd), *margs) == 0
                torch.cuda.synchronize()
                err = ((out - ref).abs().max() / ref.abs().max()).item()
                t2 = tune._time(lambda: lib.rtg_gmfma_forward(C.byref(gd), *margs))
            tune.REPS = 3
            fl = 2.0 * B * Lo * ly.cout * (ly.cin // ly.groups) * ly.k
            print(f'd{sub}.convs.{li} Cg{ly.cin // ly.groups} Mg{ly.cout // ly.groups} s{ly.stride} L{L}: mfma {t0 * 1e3:7.1f} us '
                  f'{fl / t0 / 1e9:6.1f} TF/s (cfg {d.tile_cfg})   valu {t1 * 1e3:7.1f} us {fl / t1 / 1e9:6.1f} TF/s' +
                  (f'   gmfma {t2 * 1e3:7.1f} us {fl / t2 / 1e9:6.1f} TF/s (vs valu result: rel err {err:.1e})' if t2 else ''))
            # backward-data of the same layer
            dy = torch.randn(B, ly.cout, Lo, device='cuda')
            dx = torch.empty(B, ly.cin, L, device='cuda')
            dd = ops._dgrad_desc(ly, B, L, Lo, 0.15)
            dargs = (ops._p(dy), None, None, bank.bwd_ptr(ly), None, ops._p(x), None, ops._p(dx), None, None)
            tune.ACTIVE = True
            dd.tile_cfg = tune.conv_cfg(dd, lambda: lib.rtg_conv1d(C.byref(dd), *dargs))
            tune.ACTIVE = False
            wb = torch.empty(lib.rtg_gconv_workspace(C.byref(gd)), device='cuda')
            lib.rtg_gconv_prepare_bwd(C.byref(gd), bank_v(ly), bank_s(ly), ops._p(wb), None)
            bargs = (ops._p(dy), ops._p(wb), ops._p(x), None, ops._p(dx), None)
            tune.REPS = 20
            t0 = tune._time(lambda: lib.rtg_conv1d(C.byref(dd), *dargs))
            t1 = tune._time(lambda: lib.rtg_gconv_backward_data(C.byref(gd), *bargs))
            tune.REPS = 3
            print(f'   dgrad: mfma {t0 * 1e3:7.1f} us {fl / t0 / 1e9:6.1f} TF/s (cfg {dd.tile_cfg})   valu {t1 * 1e3:7.1f} us '
                  f'{fl / t1 / 1e9:6.1f} TF/s')
        L = Lo
