#!/usr/bin/env python3
"""Dev tool: the dense-layer kernel (rtg_dconv.hip, codes 8xxx) against the general kernel's candidates on the layer
shapes it serves, at the train step's batch.  usage: bench_dconv.py [fwd|dgrad|poly ...]"""
import ctypes as C
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, 'transtacos-retunegan_amd'))
sys.path.insert(0, os.path.join(REPO, 'tests'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import packref  # noqa: E402
from rtg.lib import lib, Conv1dDesc, WgradDesc  # noqa: E402


BF = bool(os.environ.get('BD_BF'))       # bf16 operands (RtgConv1dDesc.bf16): the general bf16 kernel against the dense bf16 codes


def images(W):
    if not BF:
        return torch.from_numpy(np.concatenate([packref.pack_logical(W, 32), packref.pack_frag16(W)])).cuda()
    _, Mg, Cg, K = W.shape
    std = lib.rtg_packed_size_bf16(1, Mg, Cg, K, 32)
    junk = (np.random.RandomState(3).randn(std) * 0.05).astype(np.float32)      # (timing only: arbitrary bf16 pairs)
    return torch.from_numpy(np.concatenate([junk, packref.pack_frag16_bf16(W)])).cuda()


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


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
    return e0.elapsed_time(e1) / iters


def bench_wgrad(B, Cin, Cout, L, s):
    K, p = 5, 2
    Lo = (L + 2 * p - K) // s + 1
    x = torch.randn(B, Cin, L, device='cuda')
    dy = torch.randn(B, Cout, Lo, device='cuda')
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    need = Cout * (Cin * K + 1)
    flop = 2.0 * B * Lo * Cout * Cin * K
    probe = WgradDesc(B=B, C1=Cin, C2=0, L_in=L, groups=1, Cg=Cin, Mg=Cout, K=K, stride=s, dil=1, pad=p, Q=Lo, dy_L=Lo,
                      pre_mode=1, pre_slope=0.15, gy_mode=0, gy_slope=1.0, gy_scale=1.0, splits=1, part_stride=0, bf16=int(BF))
    cands = (C.c_int * 12)()
    n = lib.rtg_wgrad_shape_candidates(C.byref(probe), cands, 12)
    res, ref = [], None
    for c in list(cands[:n]):
        wd = WgradDesc(B=B, C1=Cin, C2=0, L_in=L, groups=1, Cg=Cin, Mg=Cout, K=K, stride=s, dil=1, pad=p, Q=Lo, dy_L=Lo,
                       pre_mode=1, pre_slope=0.15, gy_mode=0, gy_slope=1.0, gy_scale=1.0, splits=1, part_stride=0, shape_cfg=c, bf16=int(BF))
        splits = lib.rtg_wgrad_splits(C.byref(wd))
        if splits < 1:
            continue
        part = torch.empty(splits * need, device='cuda')
        wd.splits, wd.part_stride = splits, need
        if lib.rtg_conv1d_wgrad(C.byref(wd), P(x), None, P(dy), None, P(part), st):
            continue
        torch.cuda.synchronize()
        tot = part.view(splits, need).double().sum(0)
        if ref is None:
            ref = tot
        err = ((tot - ref).abs().max() / ref.abs().max()).item()
        ms = timeit(lambda: lib.rtg_conv1d_wgrad(C.byref(wd), P(x), None, P(dy), None, P(part), st))
        res.append((c, ms, splits, err))
    gen = min((r for r in res if r[0] < 10), key=lambda r: r[1])
    dn = [r for r in res if r[0] >= 10]
    line = f'wgrad B{B:4d} {Cin}->{Cout} L{L:4d} s{s}: general best s{gen[0]} x{gen[2]:3d} {gen[1] * 1e3:7.1f} us {flop / gen[1] / 1e9:6.1f} TF/s'
    for d in dn:
        line += f' | s{d[0]} x{d[2]:3d} {d[1] * 1e3:7.1f} us {flop / d[1] / 1e9:6.1f} TF/s  x{gen[1] / d[1]:.2f}  relerr {d[3]:.1e}'
    print(line, flush=True)


def bench_2d(B, Cin, Cout, H, W, kh, sh, sw, kw=3):
    """StftDiscriminator layer forward: (kh, kw) kernel, stride (sh, sw), 'same'-style padding; h_mode 0 on the (channel,
    kernel row) images against h_mode 2 on a kernel-row-major fragment image (timing only: same image bytes)"""
    ph, pw = kh // 2, kw // 2
    Ho, Wo = (H + 2 * ph - kh) // sh + 1, (W + 2 * pw - kw) // sw + 1
    w = (np.random.RandomState(1).randn(Cout, Cin, kh, kw) / np.sqrt(Cin * kh * kw)).astype(np.float32)
    Wl = w.reshape(1, Cout, Cin * kh, kw)
    wp = images(Wl)
    x = torch.randn(B, Cin, H, W, device='cuda')
    out = torch.empty(B, Cout, Ho, Wo, device='cuda')
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    d = Conv1dDesc(B=B * Ho, C1=Cin * kh, C2=0, L_in=W, groups=1, Cg=Cin * kh, Mg=Cout, K=kw, stride=sw, dil=1, pad=pw, Q=Wo,
                   out_C=Cout, out_L=Wo, shuf_S=1, shuf_P=0, pre_mode=1, pre_slope=0.15, mask_slope=1.0, out_scale=1.0,
                   act=0, act_slope=1.0, accumulate=0, tile_m=32, out_split=0, wp16=1, h_in=H, h_k=kh, h_stride=sh,
                   h_pad=ph, h_n=Ho, h_mode=0, bf16=int(BF))
    flop = 2.0 * B * Ho * Wo * Cout * Cin * kh * kw
    cands = (C.c_int * 48)()
    n = lib.rtg_conv1d_tile_candidates(C.byref(d), cands, 48)
    res, ref = [], None
    for c in list(cands[:n]):
        d.tile_cfg = c
        if lib.rtg_conv1d(C.byref(d), P(x), None, None, P(wp), None, None, None, P(out), None, st):
            continue
        torch.cuda.synchronize()
        if ref is None:
            ref = out.clone()
        same = BF or torch.equal(out, ref)
        ms = timeit(lambda: lib.rtg_conv1d(C.byref(d), P(x), None, None, P(wp), None, None, None, P(out), None, st))
        res.append((c, ms, same))
    gen = min((r for r in res if r[0] < 8000), key=lambda r: r[1])
    dc = [r for r in res if r[0] > 8000]
    line = f'fwd2d B{B} {Cin}->{Cout} {H}x{W} k({kh},{kw}) s({sh},{sw}): general best {gen[0]} {gen[1] * 1e3:7.1f} us {flop / gen[1] / 1e9:6.1f} TF/s'
    if dc:
        bd = min(dc, key=lambda r: r[1])
        line += f' | dconv best {bd[0]} {bd[1] * 1e3:7.1f} us {flop / bd[1] / 1e9:6.1f} TF/s  x{gen[1] / bd[1]:.2f}'
    d.h_mode, d.tile_cfg = 2, 0
    n2 = lib.rtg_conv1d_tile_candidates(C.byref(d), cands, 48)
    r2 = []
    for c in list(cands[:max(n2, 0)]):
        d.tile_cfg = c
        if lib.rtg_conv1d(C.byref(d), P(x), None, None, P(wp), None, None, None, P(out), None, st):
            continue
        r2.append((c, timeit(lambda: lib.rtg_conv1d(C.byref(d), P(x), None, None, P(wp), None, None, None, P(out), None, st))))
    if r2:
        b2 = min(r2, key=lambda r: r[1])
        line += f' | h_mode 2 best {b2[0]} {b2[1] * 1e3:7.1f} us {flop / b2[1] / 1e9:6.1f} TF/s'
    bad = [r[0] for r in res if not r[2]]
    print(line + (f'  MISMATCH {bad}' if bad else ''), flush=True)


def bench(kind, B, Cin, Cout, L, s):
    if kind == 'wgrad':
        return bench_wgrad(B, Cin, Cout, L, s)
    K, p = 5, 2
    w = (np.random.RandomState(1).randn(Cout, Cin, K) / np.sqrt(Cin * K)).astype(np.float32)
    Lo = (L + 2 * p - K) // s + 1
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    base = dict(C2=0, groups=1, dil=1, pre_slope=1.0, mask_slope=1.0, out_scale=1.0, act=0, act_slope=1.0, accumulate=0,
                tile_m=32, out_split=0, wp16=1, shuf_S=1, shuf_P=0, pre_mode=0)
    if kind == 'fwd':
        W = packref.logical_fwd(w, 1)
        x = torch.randn(B, Cin, L, device='cuda')
        out = torch.empty(B, Cout, Lo, device='cuda')
        d = Conv1dDesc(**dict(base, B=B, C1=Cin, L_in=L, Cg=Cin, Mg=Cout, K=K, stride=s, pad=p, Q=Lo, out_C=Cout, out_L=Lo,
                              pre_mode=1, pre_slope=0.15))
        mask = None
    elif kind == 'dgrad':
        W = packref.logical_dgrad_s1(w, 1)
        x = torch.randn(B, Cout, Lo, device='cuda')
        out = torch.empty(B, Cin, L, device='cuda')
        mask = torch.randn(B, Cin, L, device='cuda')
        d = Conv1dDesc(**dict(base, B=B, C1=Cout, L_in=Lo, Cg=Cout, Mg=Cin, K=K, stride=1, pad=K - 1 - p, Q=L, out_C=Cin,
                              out_L=L, mask_slope=0.15))
    else:
        W = packref.logical_dgrad_poly(w, 1, s)
        nt = W.shape[-1]
        x = torch.randn(B, Cout, Lo, device='cuda')
        out = torch.empty(B, Cin, L, device='cuda')
        mask = torch.randn(B, Cin, L, device='cuda')
        d = Conv1dDesc(**dict(base, B=B, C1=Cout, L_in=Lo, Cg=Cout, Mg=Cin * s, K=nt, stride=1, pad=nt - 1,
                              Q=(L - 1 + p) // s + 1, out_C=Cin, out_L=L, shuf_S=s, shuf_P=p, mask_slope=0.15))
    wp = images(W)
    d.bf16 = int(BF)
    flop = 2.0 * B * Lo * Cout * Cin * K
    cands = (C.c_int * 48)()
    n = lib.rtg_conv1d_tile_candidates(C.byref(d), cands, 48)
    res = []
    ref = None
    for c in list(cands[:n]):
        d.tile_cfg = c
        rc = lib.rtg_conv1d(C.byref(d), P(x), None, None, P(wp), None, P(mask), None, P(out), None, st)
        if rc:
            res.append((c, None, rc))
            continue
        torch.cuda.synchronize()
        if ref is None:
            ref = out.clone()
        same = BF or torch.equal(out, ref)
        ms = timeit(lambda: lib.rtg_conv1d(C.byref(d), P(x), None, None, P(wp), None, P(mask), None, P(out), None, st))
        res.append((c, ms, same))
    gen = [r for r in res if r[0] < 8000 and r[1]]
    dc = [r for r in res if r[0] > 8000 and r[1]]
    bg = min(gen, key=lambda r: r[1])
    bd = min(dc, key=lambda r: r[1]) if dc else None
    line = f'{kind:5s} B{B:4d} {Cin}->{Cout} L{L:4d} s{s}: general best {bg[0]} {bg[1] * 1e3:7.1f} us {flop / bg[1] / 1e9:6.1f} TF/s'
    if bd:
        line += f' | dconv best {bd[0]} {bd[1] * 1e3:7.1f} us {flop / bd[1] / 1e9:6.1f} TF/s  x{bg[1] / bd[1]:.2f}'
    bad = [r[0] for r in res if r[2] is not True]
    print(line + (f'  MISMATCH/ERR {bad}' if bad else ''), flush=True)
    if os.environ.get('BD_ALL'):
        for c, ms, same in res:
            if c > 8000 and ms:
                print(f'      {c}: {ms * 1e3:7.1f} us {flop / ms / 1e9:6.1f} TF/s')


SHAPES = [
    ('fwd', 704, 512, 512, 10, 1), ('fwd', 448, 512, 512, 15, 1), ('fwd', 320, 512, 512, 21, 1), ('fwd', 192, 512, 512, 34, 1),
    ('fwd', 64, 512, 512, 128, 1), ('fwd', 64, 512, 512, 64, 1), ('fwd', 64, 512, 512, 32, 1),
    ('dgrad', 704, 512, 512, 10, 1), ('dgrad', 192, 512, 512, 34, 1), ('dgrad', 352, 512, 512, 10, 1), ('dgrad', 64, 512, 512, 128, 1),
    ('fwd', 704, 256, 512, 28, 3), ('fwd', 192, 256, 512, 102, 3), ('fwd', 704, 128, 256, 83, 3), ('fwd', 192, 128, 256, 304, 3),
    ('poly', 704, 256, 512, 28, 3), ('poly', 192, 256, 512, 102, 3), ('poly', 704, 128, 256, 83, 3), ('poly', 192, 128, 256, 304, 3),
    ('fwd', 704, 32, 128, 249, 3), ('fwd', 192, 32, 128, 911, 3), ('poly', 704, 32, 128, 249, 3), ('poly', 192, 32, 128, 911, 3),
    ('wgrad', 704, 512, 512, 10, 1), ('wgrad', 192, 512, 512, 34, 1), ('wgrad', 64, 512, 512, 128, 1), ('wgrad', 64, 512, 512, 32, 1),
    ('wgrad', 704, 256, 512, 28, 3), ('wgrad', 192, 256, 512, 102, 3), ('wgrad', 704, 128, 256, 83, 3), ('wgrad', 192, 128, 256, 304, 3),
    ('wgrad', 704, 32, 128, 249, 3), ('wgrad', 192, 32, 128, 911, 3),
]

def bench_dgrad_2d(B, Cin, Cout, H, W, kh, sh, sw, kw=3):
    """StftDiscriminator layer backward-data (h_mode 1): the stride-1 3-tap operator, or — column stride > 1 — the 2-tap
    polyphase operator with rows (ci, phase) and the interleaving store; row stride > 1: class-ordered clips.  Timing only
    (arbitrary weights of the packed operator's shape)."""
    ph, pw = kh // 2, kw // 2
    Ho, Wo = (H + 2 * ph - kh) // sh + 1, (W + 2 * pw - kw) // sw + 1
    nt = -(-kw // sw)
    mg, cg, k = Cin * sw, Cout * kh, (kw if sw == 1 else nt)
    Wl = (np.random.RandomState(1).randn(1, mg, cg, k) / np.sqrt(cg * k)).astype(np.float32)
    wp = images(Wl)
    # BD_IO (with BD_BF=1): RTG_IO_* bits — 1 dy bf16, 2 dx bf16, 4 mask bf16 (bf16 tensors carry 16 readable bytes of slack)
    io = int(os.environ.get('BD_IO', '0')) if BF else 0

    def tensor(shape, b16):
        n = int(np.prod(shape))
        if not b16:
            return torch.randn(*shape, device='cuda')
        buf = torch.randn(n + 8, device='cuda').bfloat16()
        return buf[:n].view(*shape)
    dy = tensor((B, Cout, Ho, Wo), io & 1)
    xm = tensor((B, Cin, H, W), io & 4)
    dx = tensor((B, Cin, H, W), io & 2)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    common = dict(B=B * H, C1=Cout * kh, C2=0, L_in=Wo, groups=1, Cg=cg, Mg=mg, K=k, dil=1, out_C=Cin, out_L=W, pre_mode=0,
                  pre_slope=1.0, mask_slope=0.15, out_scale=1.0, act=0, act_slope=1.0, accumulate=0, tile_m=32, out_split=0,
                  wp16=1, h_in=Ho, h_k=kh, h_stride=sh, h_pad=ph, h_n=H, h_mode=1, bf16=int(BF), io_bf16=io, enc_slope=1.0)
    if sw == 1:
        d = Conv1dDesc(stride=1, pad=(kw - 1) - pw, Q=W, shuf_S=1, shuf_P=0, **common)
    else:
        d = Conv1dDesc(stride=1, pad=k - 1, Q=(W - 1 + pw) // sw + 1, shuf_S=sw, shuf_P=pw, **common)
    flop = 2.0 * B * Ho * Wo * Cout * Cin * kh * kw
    cands = (C.c_int * 48)()
    n = lib.rtg_conv1d_tile_candidates(C.byref(d), cands, 48)
    res = []
    for c in list(cands[:max(n, 0)]):
        d.tile_cfg = c
        if lib.rtg_conv1d(C.byref(d), P(dy), None, None, P(wp), None, P(xm), None, P(dx), None, st):
            continue
        torch.cuda.synchronize()
        res.append((c, timeit(lambda: lib.rtg_conv1d(C.byref(d), P(dy), None, None, P(wp), None, P(xm), None, P(dx), None, st))))
    gen = [r for r in res if r[0] < 8000]
    dc = sorted((r for r in res if r[0] > 8000), key=lambda r: r[1])
    line = f'dgrad2d io{io} B{B} {Cin}->{Cout} {H}x{W} k({kh},{kw}) s({sh},{sw}):'
    if gen:
        g = min(gen, key=lambda r: r[1])
        line += f' general best {g[0]} {g[1] * 1e3:7.1f} us {flop / g[1] / 1e9:6.1f} TF/s |'
    line += ' dconv ' + '  '.join(f'{r[0]} {r[1] * 1e3:6.1f} us {flop / r[1] / 1e9:5.1f} TF/s' for r in dc[:3])
    print(line, flush=True)


def bench_wgrad_2d(B, Cin, Cout, H, W, kh, sh, sw, kw=3):
    ph, pw = kh // 2, kw // 2
    Ho, Wo = (H + 2 * ph - kh) // sh + 1, (W + 2 * pw - kw) // sw + 1
    x = torch.randn(B, Cin, H, W, device='cuda')
    dy = torch.randn(B, Cout, Ho, Wo, device='cuda')
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    need = Cout * (Cin * kh * kw + 1)
    flop = 2.0 * B * Ho * Wo * Cout * Cin * kh * kw

    def desc(c):
        return WgradDesc(B=B * Ho, C1=Cin * kh, C2=0, L_in=W, groups=1, Cg=Cin * kh, Mg=Cout, K=kw, stride=sw, dil=1, pad=pw,
                         Q=Wo, dy_L=Wo, pre_mode=1, pre_slope=0.15, gy_mode=0, gy_slope=1.0, gy_scale=1.0, splits=1,
                         part_stride=0, h_in=H, h_k=kh, h_stride=sh, h_pad=ph, h_n=Ho, shape_cfg=c, bf16=int(BF))
    cands = (C.c_int * 16)()
    probe = desc(0)
    n = lib.rtg_wgrad_shape_candidates(C.byref(probe), cands, 16)
    res, ref = [], None
    for c in list(cands[:n]):
        wd = desc(c)
        splits = lib.rtg_wgrad_splits(C.byref(wd))
        if splits < 1:
            continue
        part = torch.empty(splits * need, device='cuda')
        wd.splits, wd.part_stride = splits, need
        if lib.rtg_conv1d_wgrad(C.byref(wd), P(x), None, P(dy), None, P(part), st):
            continue
        torch.cuda.synchronize()
        tot = part.view(splits, need).double().sum(0)
        if ref is None:
            ref = tot
        err = ((tot - ref).abs().max() / ref.abs().max()).item()
        ms = timeit(lambda: lib.rtg_conv1d_wgrad(C.byref(wd), P(x), None, P(dy), None, P(part), st))
        res.append((c, ms, splits, err))
    gen = min((r for r in res if r[0] < 10), key=lambda r: r[1])
    line = f'wgrad2d B{B} {Cin}->{Cout} {H}x{W} k({kh},{kw}) s({sh},{sw}): general best s{gen[0]} x{gen[2]:3d} {gen[1] * 1e3:7.1f} us {flop / gen[1] / 1e9:6.1f} TF/s'
    for d in (r for r in res if r[0] >= 10):
        line += f' | s{d[0]} x{d[2]:3d} {d[1] * 1e3:7.1f} us {flop / d[1] / 1e9:6.1f} TF/s  x{gen[1] / d[1]:.2f} err {d[3]:.0e}'
    print(line, flush=True)


MTD = [  # B = 64 (real + generated clips), resolution 0 (1025 x 35) and 2 (257 x 137)
    (64, 64, 256, 257, 18, 5, 3, 2), (64, 256, 512, 86, 9, 5, 3, 2), (64, 512, 512, 29, 5, 3, 1, 1), (64, 32, 64, 513, 35, 3, 2, 2),
    (64, 64, 256, 65, 69, 5, 3, 2), (64, 256, 512, 22, 35, 5, 3, 2), (64, 512, 512, 8, 18, 3, 1, 1),
]

# the same layers as the product runs them (spectrogram discriminators along the frequency axis: rows = frames, the operator
# walks the frequency bins): (B, Cin, Cout, frames, bins, kh, sh, sw, kw); BD_WT=1 selects this list
MTD_WT = [
    (64, 64, 256, 18, 257, 3, 2, 3, 5), (64, 256, 512, 9, 86, 3, 2, 3, 5), (64, 512, 512, 5, 29, 3, 1, 1, 3), (64, 32, 64, 35, 513, 3, 2, 2, 3),
    (64, 64, 256, 69, 65, 3, 2, 3, 5), (64, 256, 512, 35, 22, 3, 2, 3, 5), (64, 512, 512, 18, 8, 3, 1, 1, 3),
]
if os.environ.get('BD_WT'):
    MTD = MTD_WT

if __name__ == '__main__':
    if os.environ.get('BD_PICK'):           # e.g. BD_PICK=0,4: only those entries of the lists
        pick = [int(i) for i in os.environ['BD_PICK'].split(',')]
        MTD = [MTD[i] for i in pick if i < len(MTD)]
        SHAPES = [SHAPES[i] for i in pick if i < len(SHAPES)]
    if sys.argv[1:] == ['2d']:
        for sh in MTD:
            bench_2d(*sh)
        for sh in MTD:
            bench_wgrad_2d(*sh)
        sys.exit(0)
    if sys.argv[1:] == ['dgrad2d']:
        for sh in MTD:
            bench_dgrad_2d(*sh)
        sys.exit(0)
    kinds = sys.argv[1:]
    for sh in SHAPES:
        if not kinds or sh[0] in kinds:
            bench(*sh)
