"""dev: thin wgrad (shape code 7) against the MFMA wgrad shapes on the 1-channel layers of config 2"""
import os, sys, ctypes as C
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, 'transtacos-retunegan_amd'))
import torch
from rtg.lib import lib, WgradDesc
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
CASES = [('MSD0 conv0', 64, 1, 32, 8192, 15, 1, 7), ('MSD2 conv0', 64, 1, 32, 2048, 15, 1, 7), ('MPD0 conv0', 192, 1, 32, 2731, 5, 3, 2),
         ('MPD3 conv0', 704, 1, 32, 745, 5, 3, 2), ('G conv_pre', 32, 1, 16, 8192, 7, 1, 3), ('G conv_post', 32, 32, 1, 8192, 7, 1, 3),
         ('MSD0 conv_post', 64, 512, 1, 128, 3, 1, 1), ('MPD0 conv_post', 192, 512, 1, 34, 3, 1, 1), ('MPD3 conv_post', 704, 512, 1, 10, 3, 1, 1)]
for name, B, Cin, Cout, L, K, s, pad in CASES:
    Lo = (L + 2 * pad - (K - 1) - 1) // s + 1
    x = torch.randn(B, Cin, L, device='cuda'); dy = torch.randn(B, Cout, Lo, device='cuda')
    need = Cout * (Cin * K + 1)
    cands = (C.c_int * 8)()
    probe = WgradDesc(B=B, C1=Cin, C2=0, L_in=L, groups=1, Cg=Cin, Mg=Cout, K=K, stride=s, dil=1, pad=pad, Q=Lo, dy_L=Lo,
                      pre_mode=1, pre_slope=0.15, gy_mode=0, gy_slope=1.0, gy_scale=1.0, splits=1, part_stride=0)
    n = lib.rtg_wgrad_shape_candidates(C.byref(probe), cands, 8)
    res = []
    for c in cands[:n]:
        wd = WgradDesc(B=B, C1=Cin, C2=0, L_in=L, groups=1, Cg=Cin, Mg=Cout, K=K, stride=s, dil=1, pad=pad, Q=Lo, dy_L=Lo,
                       pre_mode=1, pre_slope=0.15, gy_mode=0, gy_slope=1.0, gy_scale=1.0, splits=1, part_stride=0, shape_cfg=c)
        sp = lib.rtg_wgrad_splits(C.byref(wd))
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
        res.append((c, sp, e0.elapsed_time(e1) * 50))
    mb = 4 * (x.numel() + dy.numel()) / 1e6
    print(f'{name:16s} {mb:6.1f} MB  ' + '  '.join(f'cfg{c}/s{sp}: {t:6.1f}us' for c, sp, t in res))
