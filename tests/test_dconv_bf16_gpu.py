"""bf16 form of the dense-layer kernel (rtg_dconv.hip with RtgConv1dDesc.bf16: v_mfma_f32_16x16x32_bf16, 32-channel chunks).
On operands that are exactly representable in bf16 every product is exact in fp32, so the bf16 codes must reproduce the fp32
general kernel (itself checked against torch in tests/test_dconv_gpu.py) up to the order of the fp32 additions; one case
with a non-representable activation checks the round-to-nearest-even of the staged patch against torch."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import packref
from test_dconv_gpu import _codes, _desc, _ptr, _run

pytestmark = pytest.mark.gpu


def _r(t):
    return t.bfloat16().float()


def _images(W, tile_m=32):
    """(fp32 standard + fragment image, bf16 standard [unused by the dense codes: zeros] + bf16 fragment image)"""
    from rtg.lib import lib
    _, Mg, Cg, K = W.shape
    f32 = np.concatenate([packref.pack_logical(W, tile_m), packref.pack_frag16(W)])
    std = lib.rtg_packed_size_bf16(1, Mg, Cg, K, tile_m)
    frag = packref.pack_frag16_bf16(W)
    assert frag.size == lib.rtg_packed_size_frag16_bf16(Mg, Cg, K)
    return torch.from_numpy(f32).cuda(), torch.from_numpy(np.concatenate([np.zeros(std, np.float32), frag])).cuda()


def _compare(kw, W, x, out_shape, **ops):
    w32, wbf = _images(W)
    rc, base = _run(kw, x, w32, out_shape=out_shape, **ops)
    assert rc == 0
    kb = dict(kw, bf16=1)
    codes = _codes(kb)
    assert codes, 'no dense-layer code listed for the bf16 descriptor'
    scale = base.abs().max().item()
    for c in codes:
        rc, out = _run(kb, x, wbf, out_shape=out_shape, cfg=c, **ops)
        assert rc == 0, c
        err = (out - base).abs().max().item()
        assert err <= 2e-5 * scale, (c, err, scale)
    return base


@pytest.mark.parametrize('case', [(24, 512, 512, 10, 1), (5, 512, 512, 128, 1), (9, 256, 512, 102, 3), (6, 128, 256, 304, 3),
                                  (4, 128, 144, 83, 3), (3, 96, 128, 40, 1)])
def test_forward(case):
    B, Cin, Cout, L, s = case
    K, p = 5, 2
    gen = torch.Generator().manual_seed(5)
    x = _r(torch.randn(B, Cin, L, generator=gen))
    w = _r(torch.randn(Cout, Cin, K, generator=gen) / np.sqrt(Cin * K))
    bias = torch.randn(Cout, generator=gen)
    Lo = (L + 2 * p - K) // s + 1
    kw = _desc(B, Cin, L, Cout, K, s, p, Lo, Cout, Lo, pre_mode=1, pre_slope=0.5)        # (x / 2 is exact in bf16)
    base = _compare(kw, packref.logical_fwd(w.numpy(), 1), x.cuda(), (B, Cout, Lo), bias=bias.cuda())
    ref = F.conv1d(F.leaky_relu(x, 0.5).double(), w.double(), bias.double(), s, p).float()
    np.testing.assert_allclose(base.cpu().numpy(), ref.numpy(), rtol=1e-4, atol=2e-5)


def test_forward_rounds_the_activation_to_nearest_even():
    B, Cin, Cout, L, K, p = 6, 128, 128, 50, 5, 2
    gen = torch.Generator().manual_seed(7)
    x = torch.randn(B, Cin, L, generator=gen)
    w = _r(torch.randn(Cout, Cin, K, generator=gen) / np.sqrt(Cin * K))
    ref = F.conv1d(_r(F.leaky_relu(x, 0.15)).double(), w.double(), None, 1, p).float()
    _, wbf = _images(packref.logical_fwd(w.numpy(), 1))
    kw = _desc(B, Cin, L, Cout, K, 1, p, L, Cout, L, pre_mode=1, pre_slope=0.15, bf16=1)
    codes = _codes(kw)
    assert codes
    for c in codes:
        rc, out = _run(kw, x.cuda(), wbf, out_shape=(B, Cout, L), cfg=c)
        assert rc == 0
        np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize('case', [(10, 512, 512, 21), (3, 512, 512, 128), (5, 256, 384, 15)])
def test_dgrad_stride1(case):
    B, Cin, Cout, L = case
    K, p = 5, 2
    gen = torch.Generator().manual_seed(17)
    w = _r(torch.randn(Cout, Cin, K, generator=gen) / np.sqrt(Cin * K))
    dy = _r(torch.randn(B, Cout, L, generator=gen))
    xm = torch.randn(B, Cin, L, generator=gen)
    tap = torch.randn(B, Cin, L, generator=gen)
    kw = _desc(B, Cout, L, Cin, K, 1, (K - 1) - p, L, Cin, L, mask_slope=0.15, out_scale=0.5)
    _compare(kw, packref.logical_dgrad_s1(w.numpy(), 1), dy.cuda(), (B, Cin, L), mask=xm.cuda(), res=tap.cuda())


@pytest.mark.parametrize('case', [(9, 256, 512, 102), (13, 256, 512, 28), (6, 128, 256, 304)])
def test_dgrad_polyphase(case):
    B, Cin, Cout, L = case
    K, s, p = 5, 3, 2
    gen = torch.Generator().manual_seed(19)
    w = _r(torch.randn(Cout, Cin, K, generator=gen) / np.sqrt(Cin * K))
    Lo = (L + 2 * p - K) // s + 1
    dy = _r(torch.randn(B, Cout, Lo, generator=gen))
    xm = torch.randn(B, Cin, L, generator=gen)
    W = packref.logical_dgrad_poly(w.numpy(), 1, s)
    nt = W.shape[-1]
    nq = (L - 1 + p) // s + 1
    kw = _desc(B, Cout, Lo, Cin * s, nt, 1, nt - 1, nq, Cin, L, shuf_S=s, shuf_P=p, mask_slope=0.15)
    _compare(kw, W, dy.cuda(), (B, Cin, L), mask=xm.cuda())


CONV2D = [
    (2, 64, 256, 40, 18, (5, 3), (3, 2), (2, 1)),
    (2, 512, 512, 8, 5, (3, 3), (1, 1), (1, 1)),
    (2, 32, 96, 33, 35, (3, 3), (2, 2), (1, 1)),
    (1, 64, 128, 13, 69, (3, 3), (1, 1), (1, 1)),
]


@pytest.mark.parametrize('case', CONV2D)
def test_conv2d_forward(case):
    B, Cin, Cout, H, W, (kh, kw), (sh, sw), (ph, pw) = case
    gen = torch.Generator().manual_seed(29)
    x = _r(torch.randn(B, Cin, H, W, generator=gen))
    w = _r(torch.randn(Cout, Cin, kh, kw, generator=gen) / np.sqrt(Cin * kh * kw))
    bias = torch.randn(Cout, generator=gen)
    Ho, Wo = (H + 2 * ph - kh) // sh + 1, (W + 2 * pw - kw) // sw + 1
    Wl = w.numpy().reshape(1, Cout, Cin * kh, kw)
    if (Cin * kh) % 32:
        pytest.skip('bf16 chunks are 32 virtual channels')
    kw_ = _desc(B * Ho, Cin * kh, W, Cout, kw, sw, pw, Wo, Cout, Wo, pre_mode=1, pre_slope=0.5, h_in=H, h_k=kh, h_stride=sh,
                h_pad=ph, h_n=Ho, h_mode=0)
    _compare(kw_, Wl, x.cuda(), (B, Cout, Ho, Wo), bias=bias.cuda())


@pytest.mark.parametrize('case', [(2, 512, 512, 8, 5), (1, 64, 128, 13, 69)])
def test_conv2d_dgrad_stride1(case):
    B, Cin, Cout, H, W = case
    k, p = 3, 1
    gen = torch.Generator().manual_seed(31)
    w = _r(torch.randn(Cout, Cin, k, k, generator=gen) / np.sqrt(Cin * k * k))
    dy = _r(torch.randn(B, Cout, H, W, generator=gen))
    xm = torch.randn(B, Cin, H, W, generator=gen)
    Wl = np.ascontiguousarray(w.numpy().transpose(1, 2, 0, 3)[..., ::-1]).reshape(1, Cin, k * Cout, k)
    kw_ = _desc(B * H, Cout * k, W, Cin, k, 1, (k - 1) - p, W, Cin, W, mask_slope=0.15, h_in=H, h_k=k, h_stride=1, h_pad=p,
                h_n=H, h_mode=1)
    _compare(kw_, Wl, dy.cuda(), (B, Cin, H, W), mask=xm.cuda())


@pytest.mark.parametrize('case', [(2, 64, 256, 40, 18, (5, 3), (3, 2), (2, 1)), (3, 256, 512, 22, 9, (5, 3), (3, 2), (2, 1)),
                                  (2, 32, 64, 33, 35, (3, 3), (2, 2), (1, 1)), (1, 128, 96, 17, 20, (5, 3), (3, 2), (2, 1))])
def test_conv2d_dgrad_strided(case):
    B, Cin, Cout, H, W, (kh, kw), (sh, sw), (ph, pw) = case
    gen = torch.Generator().manual_seed(37)
    w = _r(torch.randn(Cout, Cin, kh, kw, generator=gen) / np.sqrt(Cin * kh * kw))
    Ho, Wo = (H + 2 * ph - kh) // sh + 1, (W + 2 * pw - kw) // sw + 1
    dy = _r(torch.randn(B, Cout, Ho, Wo, generator=gen))
    xm = torch.randn(B, Cin, H, W, generator=gen)
    nt = -(-kw // sw)
    wn = w.numpy()
    Wl = np.zeros((1, Cin * sw, kh * Cout, nt), dtype=np.float32)
    for r in range(sw):
        for tap in range(nt):
            jj = r + (nt - 1 - tap) * sw
            if jj < kw:
                Wl[0, r::sw, :, tap] = wn[:, :, :, jj].transpose(1, 2, 0).reshape(Cin, kh * Cout)
    nq = (W - 1 + pw) // sw + 1
    kw_ = _desc(B * H, Cout * kh, Wo, Cin * sw, nt, 1, nt - 1, nq, Cin, W, shuf_S=sw, shuf_P=pw, mask_slope=0.15, h_in=Ho,
                h_k=kh, h_stride=sh, h_pad=ph, h_n=H, h_mode=1)
    _compare(kw_, Wl, dy.cuda(), (B, Cin, H, W), mask=xm.cuda())


def test_pack_kernel_writes_the_bf16_fragment_image():
    from rtg import lib as L
    from test_dconv_gpu import _pack_on_gpu
    gen = torch.Generator().manual_seed(43)
    Cout, Cin, K = 144, 160, 5
    v = torch.randn(Cout, Cin, K, generator=gen)
    g = torch.rand(Cout, generator=gen) + 0.5
    w_eff = (v * (g / v.flatten(1).norm(dim=1)).view(-1, 1, 1)).numpy()
    for mode, W, S in ((L.PACK_FWD, packref.logical_fwd(w_eff, 1), 1), (L.PACK_DGRAD_S1, packref.logical_dgrad_s1(w_eff, 1), 1),
                       (L.PACK_DGRAD_POLY, packref.logical_dgrad_poly(w_eff, 1, 3), 3)):
        _, Mg, Cg, Kp = W.shape
        got = _pack_on_gpu(v, g, mode, 1, Mg, Cg, Kp, K, Cin, S, 16, frag16=1, bf16=1)
        ref = packref.pack_frag16_bf16(W)
        a, b = got.view(np.uint16).astype(np.int32), ref.view(np.uint16).astype(np.int32)
        assert a.shape == b.shape
        # (the device forms g * v / ||v|| in fp32 in its own order: an element on a rounding boundary may land one bf16 ulp away)
        assert np.abs(a - b).max() <= 1 and (a != b).mean() < 1e-3
