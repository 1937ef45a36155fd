"""rtg_gconv.hip: the thin-group k41 layers of DiscriminatorS on the vector ALUs, through the C ABI, against torch's
grouped conv1d on the CPU (the same effective weights g * v / ||v||).  GPU only."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


CASES = [
    # B, C_in, C_out, groups, stride, L_in, pre_slope          (discrminator.py:39-43 at the three MSD scales, ragged)
    (3, 32, 64, 4, 2, 8192, 0.15),
    (2, 64, 128, 8, 2, 4096, 0.15),
    (2, 128, 512, 32, 4, 2048, 0.15),
    (3, 512, 512, 64, 4, 512, 0.15),
    (2, 512, 512, 64, 4, 128, 0.15),        # one partial tile per clip
    (2, 128, 512, 32, 4, 1001, 1.0),        # ragged length, no input activation
    (1, 64, 128, 8, 2, 37, 0.15),           # shorter than the kernel
    (5, 32, 64, 4, 2, 3000, 0.15),
]


@pytest.mark.parametrize('case', CASES)
def test_gconv_forward_matches_torch(case):
    from rtg.lib import lib, GconvDesc
    B, Cin, Cout, g, s, L, slope = case
    K, pad = 41, 20
    Lo = (L + 2 * pad - (K - 1) - 1) // s + 1
    gen = torch.Generator().manual_seed(Cin + L)
    x = torch.randn(B, Cin, L, generator=gen)
    v = torch.randn(Cout, Cin // g, K, generator=gen) * 0.2
    gg = torch.rand(Cout, generator=gen) + 0.5
    bias = torch.randn(Cout, generator=gen)
    scale = gg / v.flatten(1).norm(dim=1)
    w = v * scale[:, None, None]
    ref = F.conv1d(F.leaky_relu(x.double(), slope) if slope != 1.0 else x.double(), w.double(), bias.double(), stride=s,
                   padding=pad, groups=g)
    d = GconvDesc(B, g, Cin // g, Cout // g, K, s, pad, L, Lo, slope)
    assert lib.rtg_gconv_ok(C.byref(d)) == 1
    xd, vd, sd, bd = x.cuda(), v.cuda(), scale.cuda(), bias.cuda()
    out = torch.full((B, Cout, Lo), float('nan'), device='cuda')
    n_w = lib.rtg_gconv_workspace(C.byref(d))
    assert n_w >= v.numel()
    wbuf = torch.full((n_w,), float('nan'), device='cuda')
    assert lib.rtg_gconv_prepare(C.byref(d), _ptr(vd), _ptr(sd), _ptr(wbuf), None) == 0
    # [group][ci][tap][oc] = the effective weights, re-ordered
    wr = w.view(g, Cout // g, Cin // g, K).permute(0, 2, 3, 1).contiguous().flatten()
    assert torch.allclose(wbuf.cpu()[:wr.numel()], wr, rtol=1e-6, atol=0)
    assert lib.rtg_gconv_forward(C.byref(d), _ptr(xd), _ptr(wbuf), _ptr(bd), _ptr(out), None) == 0
    torch.cuda.synchronize()
    got = out.cpu().double()
    assert torch.isfinite(got).all()
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    assert err < 2e-6, err


@pytest.mark.parametrize('case', CASES)
def test_gconv_backward_data_matches_torch(case):
    """dx of out = conv(leaky_relu(x)) given dy: lrelu'(x) * conv_transpose(dy), against torch autograd in float64"""
    from rtg.lib import lib, GconvDesc
    B, Cin, Cout, g, s, L, slope = case
    K, pad = 41, 20
    Lo = (L + 2 * pad - (K - 1) - 1) // s + 1
    gen = torch.Generator().manual_seed(Cin + L + 1)
    x = torch.randn(B, Cin, L, generator=gen)
    dy = torch.randn(B, Cout, Lo, generator=gen)
    v = torch.randn(Cout, Cin // g, K, generator=gen) * 0.2
    gg = torch.rand(Cout, generator=gen) + 0.5
    scale = gg / v.flatten(1).norm(dim=1)
    w = v * scale[:, None, None]
    xr = x.double().requires_grad_(True)
    out = F.conv1d(F.leaky_relu(xr, slope) if slope != 1.0 else xr, w.double(), None, stride=s, padding=pad, groups=g)
    out.backward(dy.double())
    ref = xr.grad
    d = GconvDesc(B, g, Cin // g, Cout // g, K, s, pad, L, Lo, slope)
    xd, dyd, vd, sd = x.cuda(), dy.cuda(), v.cuda(), scale.cuda()
    wbuf = torch.full((lib.rtg_gconv_workspace(C.byref(d)),), float('nan'), device='cuda')
    assert lib.rtg_gconv_prepare_bwd(C.byref(d), _ptr(vd), _ptr(sd), _ptr(wbuf), None) == 0
    wr = w.view(g, Cout // g, Cin // g, K).permute(0, 1, 3, 2).contiguous().flatten()      # [group][oc][tap][ci]
    assert torch.allclose(wbuf.cpu()[:wr.numel()], wr, rtol=1e-6, atol=0)
    dx = torch.full((B, Cin, L), float('nan'), device='cuda')
    res = torch.randn(B, Cin, L, generator=gen)            # a residual gradient riding the epilogue (B even: tested with it)
    resd = res.cuda() if B % 2 == 0 else None
    assert lib.rtg_gconv_backward_data(C.byref(d), _ptr(dyd), _ptr(wbuf), _ptr(xd) if slope != 1.0 else None, _ptr(resd),
                                       _ptr(dx), None) == 0
    torch.cuda.synchronize()
    got = dx.cpu().double() - (res.double() if resd is not None else 0.0)
    assert torch.isfinite(got).all()
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    assert err < 2e-6, err


def test_gconv_refuses_other_shapes():
    from rtg.lib import lib, GconvDesc
    assert lib.rtg_gconv_ok(C.byref(GconvDesc(2, 4, 8, 16, 41, 2, 20, 100, 50, 0.15))) == 1
    assert lib.rtg_gconv_ok(C.byref(GconvDesc(2, 4, 8, 16, 5, 2, 2, 100, 50, 0.15))) == 0       # k5
    assert lib.rtg_gconv_ok(C.byref(GconvDesc(2, 4, 16, 16, 41, 2, 20, 100, 50, 0.15))) == 0    # 16 channels per group
    assert lib.rtg_gconv_ok(C.byref(GconvDesc(2, 4, 8, 16, 41, 2, 20, 100, 51, 0.15))) == 0     # wrong output length


@pytest.mark.parametrize('case', CASES + [(64, 32, 64, 4, 2, 8192, 0.15), (7, 512, 512, 64, 4, 64, 0.15), (2, 128, 512, 32, 4, 700, 0.15),
                                          # the layer with 8 output channels per group (position pairs): rows of 32 / 64 / 128
                                          # positions (MSD scales at 8192-sample clips), ragged clip counts, odd row lengths
                                          (9, 512, 512, 64, 4, 128, 0.15), (6, 512, 512, 64, 4, 256, 0.15),
                                          (3, 512, 512, 64, 4, 512, 1.0), (2, 512, 512, 64, 4, 1000, 0.15),
                                          (5, 64, 64, 8, 4, 198, 0.15), (64, 512, 512, 64, 4, 128, 0.15)])
def test_gmfma_forward_matches_torch(case):
    """rtg_gmfma.hip (round 4): the same forward on the matrix cores with exact-fit tiles, its weight image
    [group][oc][ci][44] written by rtg_weights_pack (RTG_PACK_GMFMA_FWD) from the raw weight-norm parameters."""
    from rtg import lib as L
    from rtg.lib import lib, GconvDesc
    B, Cin, Cout, g, s, Lin, slope = case
    K, pad = 41, 20
    Lo = (Lin + 2 * pad - (K - 1) - 1) // s + 1
    d = GconvDesc(B, g, Cin // g, Cout // g, K, s, pad, Lin, Lo, slope)
    pair = Cout // g == 8                                   # position-pair tiles (gmfma_pair_kernel)
    if Lo < 16 or not (Cout // g == 16 or (pair and Cin // g == 8 and s == 4)):
        assert lib.rtg_gmfma_ok(C.byref(d)) == 0            # (rows shorter than a column tile stay with rtg_gconv)
        return
    assert lib.rtg_gmfma_ok(C.byref(d)) == 1
    gen = torch.Generator().manual_seed(Cin + Lin)
    x = torch.randn(B, Cin, Lin, generator=gen)
    v = torch.randn(Cout, Cin // g, K, generator=gen) * 0.2
    gg = torch.rand(Cout, generator=gen) + 0.5
    bias = torch.randn(Cout, generator=gen)
    scale = gg / v.flatten(1).norm(dim=1)
    w = v * scale[:, None, None]
    ref = F.conv1d(F.leaky_relu(x.double(), slope) if slope != 1.0 else x.double(), w.double(), bias.double(), stride=s,
                   padding=pad, groups=g)
    # the image through the pack launch: params = [v], scales = [scale | 1 / norm]
    n_w = lib.rtg_gmfma_workspace(C.byref(d))
    kp = 48 if pair else 44
    assert n_w == (g * 16 if pair else Cout) * (Cin // g) * kp
    params = v.flatten().cuda()
    scales = torch.cat([scale, 1.0 / v.flatten(1).norm(dim=1)]).cuda()
    packed = torch.full((n_w + 64,), float('nan'), device='cuda')
    job = L.PackJob(0, 0, 0, n_w, L.PACK_GMFMA_FWD, g, Cout // g, Cin // g, K, K, Cin // g, kp, 16, s if pair else 0, 0, 0, 0)
    blocks, lds = L.assign_pack_blocks([job])
    tab = torch.frombuffer(bytearray(bytes(job)), dtype=torch.uint8).cuda()
    assert lib.rtg_weights_pack(_ptr(tab), 1, blocks, lds, _ptr(params), _ptr(scales), _ptr(packed), None) == 0
    if pair:
        # [group][parity r][oc][ci][48]: row (r, oc) = the taps shifted right by r * stride, zeros around them
        img = packed.cpu()[:n_w].view(g, 2, 8, Cin // g, 48)
        wg = w.view(g, 8, Cin // g, K)
        for r in (0, 1):
            assert torch.allclose(img[:, r, :, :, s * r:s * r + K], wg, rtol=1e-6, atol=0)
            assert (img[:, r, :, :, :s * r] == 0).all() and (img[:, r, :, :, s * r + K:] == 0).all()
    else:
        img = packed.cpu()[:n_w].view(Cout, Cin // g, 44)
        assert torch.allclose(img[:, :, :41], w, rtol=1e-6, atol=0) and (img[:, :, 41:] == 0).all()
    out = torch.full((B, Cout, Lo), float('nan'), device='cuda')
    assert lib.rtg_gmfma_forward(C.byref(d), _ptr(x.cuda()), _ptr(packed), _ptr(bias.cuda()), _ptr(out), None) == 0
    torch.cuda.synchronize()
    got = out.cpu().double()
    assert torch.isfinite(got).all()
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    assert err < 2e-6, err
