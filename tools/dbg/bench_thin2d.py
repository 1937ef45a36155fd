#!/usr/bin/env python3
"""Dev tool: the first-layer Conv2d kernels of rtg_thin2d.hip (forward, weight gradient) alone, at the step's sizes."""
import ctypes as C, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, 'transtacos-retunegan_amd')); sys.path.insert(0, os.path.join(REPO, 'tests'))
import numpy as np, torch
import packref
from rtg.lib import lib, Conv1dDesc, WgradDesc
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(f, iters=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for (B, H, W) in ((64, 1025, 35), (64, 257, 137), (64, 1025, 69)):
    Cin, Cout, kh, kw, sh = 2, 32, 3, 3, 2
    Ho = (H + 2 - kh) // sh + 1
    x = torch.randn(B, Cin, H, W, device='cuda'); dy = torch.randn(B, Cout, Ho, W, device='cuda')
    w = np.random.RandomState(0).randn(Cout, Cin, kh, kw).astype(np.float32)
    wp = torch.from_numpy(packref.pack_logical(w.reshape(1, Cout, Cin * kh, kw), 32)).cuda()
    bias = torch.randn(Cout, device='cuda'); out = torch.empty(B, Cout, Ho, W, device='cuda')
    d = Conv1dDesc(B=B * Ho, C1=Cin * kh, C2=0, L_in=W, groups=1, Cg=Cin * kh, Mg=Cout, K=kw, stride=1, dil=1, pad=1, Q=W, out_C=Cout,
                   out_L=W, shuf_S=1, shuf_P=0, pre_mode=0, pre_slope=1.0, mask_slope=1.0, out_scale=1.0, act=0, act_slope=1.0,
                   accumulate=0, tile_m=32, out_split=0, h_in=H, h_k=kh, h_stride=sh, h_pad=1, h_n=Ho, h_mode=0)
    us = timeit(lambda: lib.rtg_conv1d(C.byref(d), P(x), None, None, P(wp), P(bias), None, None, P(out), None, st))
    mb = (x.numel() + out.numel()) * 4 / 1e6
    line = f'B{B} {H}x{W}: fwd kind {lib.rtg_conv1d_variant(C.byref(d))} {us:7.1f} us {mb / us * 1e-3:6.2f} TB/s'
    wd = WgradDesc(B=B * Ho, C1=Cin * kh, C2=0, L_in=W, groups=1, Cg=Cin * kh, Mg=Cout, K=kw, stride=1, dil=1, pad=1, Q=W, dy_L=W,
                   pre_mode=0, pre_slope=1.0, gy_mode=0, gy_slope=1.0, gy_scale=1.0, splits=1, part_stride=0, h_in=H, h_k=kh,
                   h_stride=sh, h_pad=1, h_n=Ho, shape_cfg=7)
    sp = lib.rtg_wgrad_splits(C.byref(wd)); need = Cout * (Cin * kh * kw + 1)
    part = torch.empty(sp * need, device='cuda'); wd.splits, wd.part_stride = sp, need
    us = timeit(lambda: lib.rtg_conv1d_wgrad(C.byref(wd), P(x), None, P(dy), None, P(part), st))
    print(line + f' | wgrad x{sp} {us:7.1f} us {(x.numel() + dy.numel()) * 4 / 1e6 / us * 1e-3:6.2f} TB/s', flush=True)
