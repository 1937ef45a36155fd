"""dev: bf16-tensor instances of the dense kernels against the fp32-tensor instances of the SAME block shape (same summation
order: the results must agree bit for bit up to the output rounding); prints where they differ"""
import ctypes as C
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..', 'transtacos-retunegan_amd'))
import hparam as hp
hp.compute_dtype = 'bf16'
from rtg import lib as L, ops
from rtg.lib import lib
from models.layers import WNConv, BankedModel

S = 0.15


class One(BankedModel):
    def __init__(self, cin, cout, k, stride, pad):
        super().__init__()
        self.c = WNConv('conv', cin, cout, k, stride=stride, pad=pad)


def wgrad_case(cin, cout, stride, B, Lx):
    torch.manual_seed(1)
    m = One(cin, cout, 5, stride, 2).cuda()
    bank = m.bank(); ly = bank.layers[0]; bank.prepare()
    Lo = (Lx + 4 - 5) // stride + 1
    x32 = torch.randn(B, cin, Lx, device='cuda'); xe = ops.bf16_encode(x32, S); xd = ops.bf16_decode(xe, S)
    dyb = ops.bf16_encode(torch.randn(B, cout, Lo, device='cuda'), 1.0); dy32 = dyb.float()
    res = {}
    for io, (xx, dd) in {0: (xd, dy32), 2: (xd, dyb), 3: (xe, dyb)}.items():
        wd = L.WgradDesc(B=B, C1=cin, C2=0, L_in=Lx, groups=1, Cg=cin, Mg=cout, K=5, stride=stride, dil=1, pad=2, Q=Lo, dy_L=Lo,
                         pre_mode=L.PRE_LRELU, pre_slope=S, gy_mode=0, gy_slope=1.0, gy_scale=1.0, splits=1, part_stride=0,
                         bf16=1, io_bf16=io)
        for code in (12, 11, 10):
            wd.shape_cfg = code
            cands = (C.c_int * 12)()
            n = lib.rtg_wgrad_shape_candidates(C.byref(wd), cands, 12)
            if code in list(cands[:n]):
                break
        wd.splits = 3
        need = cout * (cin * 5 + 1)
        wd.part_stride = need
        part = torch.zeros(3 * need, device='cuda')
        st = lib.rtg_conv1d_wgrad(C.byref(wd), xx.data_ptr(), None, dd.data_ptr(), None, part.data_ptr(), None)
        torch.cuda.synchronize()
        res[io] = (st, part.view(3, need).sum(0), code)
    ref = res[0][1]
    for io in (2, 3):
        st, p, code = res[io]
        d = (p - ref).abs()
        w = d[:cout * cin * 5].view(cout, cin, 5)
        bad = (w > 1e-4 * ref.abs().max()).nonzero()
        print(f'wgrad cin{cin} cout{cout} s{stride} B{B} L{Lx} io{io} code{code} st{st}: max diff {d.max().item():.3e} (ref max {ref.abs().max().item():.3e}) bad {len(bad)}',
              'first', bad[:6].tolist(), 'taps', sorted(set(bad[:, 2].tolist())) if len(bad) else '', 'chans', sorted(set(bad[:, 1].tolist()))[:12] if len(bad) else '')


def conv_case(cin, cout, stride, B, Lx, dgrad=False):
    torch.manual_seed(2)
    m = One(cin, cout, 5, stride, 2).cuda()
    bank = m.bank(); ly = bank.layers[0]; bank.prepare()
    Lo = (Lx + 4 - 5) // stride + 1
    if not dgrad:
        x32 = torch.randn(B, cin, Lx, device='cuda'); xe = ops.bf16_encode(x32, S); xd = ops.bf16_decode(xe, S)
        d, _ = ops._fwd_desc(ly, B, cin, Lx, S)
        wp, bias, oshape = bank.fwd_ptr(ly), bank.bias_ptr(ly), (B, cout, Lo)
    else:
        xe = ops.bf16_encode(torch.randn(B, cout, Lo, device='cuda'), 1.0); xd = xe.float()
        d = ops._dgrad_desc(ly, B, Lx, Lo, 1.0)
        wp, bias, oshape = bank.bwd_ptr(ly), None, (B, cin, Lx)
    outs = {}
    cands = (C.c_int * 16)()
    d.io_bf16 = 3; d.enc_slope = S if not dgrad else 1.0
    n = lib.rtg_conv1d_tile_candidates(C.byref(d), cands, 16)
    for code in list(cands[:n]):
        for io in (0, 1, 3):
            d.io_bf16, d.tile_cfg = io, code
            xx = xe if io & 1 else xd
            out = torch.zeros(oshape, device='cuda', dtype=torch.bfloat16 if io & 2 else torch.float32)
            st = lib.rtg_conv1d(C.byref(d), xx.data_ptr(), None, None, wp, bias, None, None, out.data_ptr(), None, None)
            torch.cuda.synchronize()
            outs[io] = (st, out)
        ref = outs[0][1]
        o1 = outs[1][1]
        dd = (o1 - ref).abs()
        bad = (dd > 1e-5 * ref.abs().max()).nonzero()
        enc = torch.nn.functional.leaky_relu(ref, d.enc_slope).bfloat16()
        neq = (outs[3][1] != enc).nonzero()
        print(f'{"dgrad" if dgrad else "fwd"} cin{cin} cout{cout} s{stride} B{B} L{Lx} code{code} st{[outs[i][0] for i in (0, 1, 3)]}: io1 bad {len(bad)} of {ref.numel()}',
              'first', bad[:5].tolist(), 'positions', sorted(set(bad[:, 2].tolist()))[:16] if len(bad) else '', '| io3 != enc(io0):', len(neq))


if __name__ == '__main__':
    for c in [(64, 128, 1, 4, 300), (128, 256, 3, 6, 304), (512, 512, 1, 12, 68), (256, 512, 3, 33, 30), (32, 128, 3, 3, 1821)]:
        wgrad_case(*c)
    for c in [(64, 128, 1, 4, 300), (128, 256, 3, 6, 304), (32, 128, 3, 3, 1821), (512, 512, 1, 12, 68)]:
        conv_case(*c)
        conv_case(*c, dgrad=True)
