#!/usr/bin/env python3
"""Grouped vs separate launches of the MPD / MSD layer families through the C ABI (dev tool): bit equality + timing."""
import ctypes as C, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'transtacos-retunegan_amd'))
import torch
from rtg.lib import lib, Conv1dDesc, ConvPtrs, check
P = lambda t: t.data_ptr() if t is not None else None
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

def problems(shapes, cfg):
    out = []
    for (B, Cin, Cout, L, K, s, p) in shapes:
        Lo = (L + 2 * p - (K - 1) - 1) // s + 1
        x = torch.randn(B, Cin, L, device='cuda')
        n = lib.rtg_packed_size(1, Cout, Cin, K, 32)
        wp = torch.randn(n, device='cuda') * 0.05
        bias = torch.randn(Cout, device='cuda')
        o = torch.empty(B, Cout, Lo, device='cuda')
        d = Conv1dDesc(B=B, C1=Cin, C2=0, L_in=L, groups=1, Cg=Cin, Mg=Cout, K=K, stride=s, dil=1, pad=p, Q=Lo, out_C=Cout,
                       out_L=Lo, shuf_S=1, shuf_P=0, pre_mode=1, pre_slope=0.15, mask_slope=1.0, out_scale=1.0, act=0,
                       act_slope=1.0, accumulate=0, tile_m=32, out_split=0, tile_cfg=cfg)
        out.append((d, x, wp, bias, o, 2.0 * B * Lo * Cout * Cin * K))
    return out

def run(shapes, cfg, iters=30):
    pr = problems(shapes, cfg)
    def separate():
        for d, x, wp, b, o, _ in pr:
            check(lib.rtg_conv1d(C.byref(d), P(x), None, None, P(wp), P(b), None, None, P(o), None, st))
    descs = (Conv1dDesc * len(pr))(*[p[0] for p in pr])
    ptrs = (ConvPtrs * len(pr))(*[ConvPtrs(P(x), None, None, P(wp), P(b), None, None, P(o), None) for d, x, wp, b, o, _ in pr])
    def grouped():
        check(lib.rtg_conv1d_group(len(pr), descs, ptrs, st))
    separate(); torch.cuda.synchronize()
    ref = [p[4].clone() for p in pr]
    for p in pr: p[4].zero_()
    grouped(); torch.cuda.synchronize()
    same = all(torch.equal(a, p[4]) for a, p in zip(ref, pr))
    flop = sum(p[5] for p in pr)
    res = []
    for f in (separate, grouped):
        for _ in range(3): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): f()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / iters)
    print(f'cfg {cfg} bit-equal {same}  separate {res[0]*1e3:7.1f} us ({flop/res[0]/1e9:5.1f} TF)  grouped {res[1]*1e3:7.1f} us ({flop/res[1]/1e9:5.1f} TF)')

B = 64
mpd4 = [(B * p, 512, 512, -(-8192 // p) // 27 + (1 if p != 3 else 0), 5, 1, 2) for p in (3, 5, 7, 11)]
mpd4 = [(192, 512, 512, 34, 5, 1, 2), (320, 512, 512, 21, 5, 1, 2), (448, 512, 512, 15, 5, 1, 2), (704, 512, 512, 10, 5, 1, 2)]
mpd3 = [(192, 256, 512, 102, 5, 3, 2), (320, 256, 512, 61, 5, 3, 2), (448, 256, 512, 44, 5, 3, 2), (704, 256, 512, 28, 5, 3, 2)]
mpd2 = [(192, 128, 256, 304, 5, 3, 2), (320, 128, 256, 183, 5, 3, 2), (448, 128, 256, 131, 5, 3, 2), (704, 128, 256, 83, 5, 3, 2)]
msd5 = [(64, 512, 512, 128, 5, 1, 2), (64, 512, 512, 64, 5, 1, 2), (64, 512, 512, 32, 5, 1, 2)]
for name, sh in (('mpd convs.4', mpd4), ('mpd convs.3', mpd3), ('mpd convs.2', mpd2), ('msd convs.5', msd5)):
    print(name)
    for cfg in (122, 124, 222, 224, 121, 111):
        try:
            run(sh, cfg)
        except Exception as e:
            print('cfg', cfg, 'n/a', str(e)[:60])
