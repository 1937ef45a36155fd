#!/usr/bin/env python3
"""Micro-benchmark of single rtg_conv1d / rtg_conv1d_wgrad shapes through the C ABI (dev tool).
usage: bench_conv.py [fwd|wgrad] B Cin Cout L K stride dil pad groups [iters]"""
import ctypes as C
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'transtacos-retunegan_amd'))
import torch  # noqa: E402
from rtg.lib import lib, Conv1dDesc, WgradDesc, check  # noqa: E402


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def run(kind, B, Cin, Cout, L, K, s, d, p, g, iters=20):
    dev = 'cuda'
    Lo = (L + 2 * p - d * (K - 1) - 1) // s + 1
    x = torch.randn(B, Cin, L, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    flop = 2.0 * B * Lo * Cout * (Cin // g) * K
    if kind == 'fwd':
        tm = 32 if Cout // g >= 32 else 16
        bf = int(os.environ.get('BENCH_BF16', '0'))
        n = (lib.rtg_packed_size_bf16 if bf else lib.rtg_packed_size)(g, Cout // g, Cin // g, K, tm)
        wp = (torch.randn(2 * n, device=dev) * 0.05).bfloat16().view(torch.float32) if bf else torch.randn(n, device=dev) * 0.05
        out = torch.empty(B, Cout, Lo, device=dev)
        desc = Conv1dDesc(B=B, C1=Cin, C2=0, L_in=L, groups=g, Cg=Cin // g, Mg=Cout // g, K=K, stride=s, dil=d, pad=p,
                          Q=Lo, out_C=Cout, out_L=Lo, shuf_S=1, shuf_P=0, pre_mode=1, pre_slope=0.15, mask_slope=1.0,
                          out_scale=1.0, act=0, act_slope=1.0, accumulate=0, tile_m=tm, out_split=0, bf16=bf,
                          tile_cfg=int(os.environ.get('BENCH_TILE', '0')))
        var = lib.rtg_conv1d_variant(C.byref(desc))

        def call():
            check(lib.rtg_conv1d(C.byref(desc), P(x), None, None, P(wp), None, None, None, P(out), None, st))
    else:
        dy = torch.randn(B, Cout, Lo, device=dev)
        wd = WgradDesc(B=B, C1=Cin, C2=0, L_in=L, groups=g, Cg=Cin // g, Mg=Cout // g, K=K, stride=s, dil=d, pad=p, Q=Lo,
                       dy_L=Lo, pre_mode=1, pre_slope=0.15, gy_mode=0, gy_slope=1.0, gy_scale=1.0, splits=1, part_stride=0)
        splits = lib.rtg_wgrad_splits(C.byref(wd))
        stride = Cout * ((Cin // g) * K + 1)
        part = torch.empty(splits * stride, device=dev)
        wd.splits, wd.part_stride = splits, stride
        var = splits

        def call():
            check(lib.rtg_conv1d_wgrad(C.byref(wd), P(x), None, P(dy), None, P(part), st))
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        call()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f'{kind} B{B} {Cin}->{Cout} L{L} k{K} s{s} d{d} g{g}: {ms * 1e3:8.1f} us  {flop / ms / 1e9:6.1f} TF/s  variant/splits {var}')


SHAPES = [
    ('fwd', 32, 32, 32, 8192, 7, 1, 9, 27, 1), ('fwd', 32, 32, 32, 8192, 3, 1, 1, 1, 1),
    ('fwd', 32, 64, 64, 2048, 7, 1, 3, 9, 1), ('fwd', 32, 128, 128, 256, 7, 1, 1, 3, 1),
    ('fwd', 320, 512, 512, 21, 5, 1, 1, 2, 1), ('fwd', 64, 512, 512, 128, 5, 1, 1, 2, 1),
    ('fwd', 192, 256, 512, 102, 5, 3, 1, 2, 1), ('fwd', 64, 128, 512, 2048, 41, 4, 1, 20, 32),
    ('wgrad', 32, 32, 32, 8192, 7, 1, 9, 27, 1), ('wgrad', 32, 64, 64, 2048, 7, 1, 3, 9, 1),
    ('wgrad', 32, 128, 128, 256, 7, 1, 1, 3, 1), ('wgrad', 320, 512, 512, 21, 5, 1, 1, 2, 1),
    ('wgrad', 64, 512, 512, 128, 5, 1, 1, 2, 1), ('wgrad', 192, 256, 512, 102, 5, 3, 1, 2, 1),
]

if __name__ == '__main__':
    if len(sys.argv) > 2:
        a = sys.argv[1:]
        run(a[0], *[int(v) for v in a[1:10]], iters=int(a[10]) if len(a) > 10 else 20)
    else:
        for sh in SHAPES:
            run(*sh)
