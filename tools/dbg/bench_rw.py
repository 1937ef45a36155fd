"""dev: reswgrad (shape code 8) against the general wgrad shapes on the ResBlock3 layers of config 2"""
import os, sys, ctypes as C
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, 'transtacos-retunegan_amd'))
import torch
from rtg.lib import lib, WgradDesc
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
for B, Cc, L in ((32, 32, 8192), (32, 64, 2048), (32, 32, 2048), (32, 64, 256)):
    for K, dil in ((3, 9), (5, 3), (7, 1)):
        pad = (K * dil - dil) // 2
        x = torch.randn(B, Cc, L, device='cuda'); dy = torch.randn(B, Cc, L, device='cuda')
        need = Cc * (Cc * K + 1)
        cands = (C.c_int * 12)()
        probe = WgradDesc(B=B, C1=Cc, C2=0, L_in=L, groups=1, Cg=Cc, Mg=Cc, K=K, stride=1, dil=dil, pad=pad, Q=L, dy_L=L,
                          pre_mode=1, pre_slope=0.15, gy_mode=0, gy_slope=1.0, gy_scale=1.0, splits=1, part_stride=0)
        n = lib.rtg_wgrad_shape_candidates(C.byref(probe), cands, 12)
        res = []
        for c in cands[:n]:
            wd = WgradDesc(B=B, C1=Cc, C2=0, L_in=L, groups=1, Cg=Cc, Mg=Cc, K=K, stride=1, dil=dil, pad=pad, Q=L, dy_L=L,
                           pre_mode=1, pre_slope=0.15, gy_mode=0, gy_slope=1.0, gy_scale=1.0, splits=1, part_stride=0, shape_cfg=c)
            sp = lib.rtg_wgrad_splits(C.byref(wd))
            if sp < 1: continue
            part = torch.empty(sp * need, device='cuda')
            wd.splits, wd.part_stride = sp, need
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            run = lambda: lib.rtg_conv1d_wgrad(C.byref(wd), p(x), None, p(dy), None, p(part), st)
            if run() != 0: continue
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(3): run()
            e0.record()
            for _ in range(20): run()
            e1.record(); e1.synchronize()
            res.append((e0.elapsed_time(e1) * 50, c, sp))
        flop = 2.0 * B * L * Cc * Cc * K
        gen = min(r for r in res if r[1] != 8); rw = [r for r in res if r[1] == 8]
        print(f'B{B} C{Cc} L{L} k{K} d{dil}: general best {gen[0]:6.1f} us (cfg{gen[1]}/s{gen[2]}, {flop/gen[0]/1e6:5.1f} TF/s) | reswgrad ' +
              (f'{rw[0][0]:6.1f} us s{rw[0][2]} {flop/rw[0][0]/1e6:5.1f} TF/s' if rw else 'n/a'))
