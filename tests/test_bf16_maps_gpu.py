"""bf16 feature maps in HBM (hparam.compute_dtype = 'bf16' with hparam.bf16_maps; BASELINE configs[2], round 5).

A dense discriminator layer stores bf16(leaky_relu(out, 0.15)) — its consumer's activation applied once — and reads such
tensors natively: one 16-byte load per 8 positions and channel, transposed through ds_read_b64_tr_b16 (rtg_dconv_kernel.h,
rtg_dwgrad.hip); gradients between those layers are plain bf16.  Checked here:
  * the conversions (rtg_bf16_encode / rtg_bf16_decode) against torch;
  * every native path — forward, backward-data (stride 1, polyphase, 2-D, class-ordered 2-D), weight gradient — against the
    SAME launch made through fp32 copies (rtg/ops.py falls back to them where no native kernel serves a shape): both round
    the same values at the same places, so they agree to summation-order noise plus rare one-ulp rounding flips;
  * a layer on stock torch with the roundings written out (what the oracle does for whole models).
GPU only."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda'
S = 0.15


@pytest.fixture()
def bf16_mode():
    import hparam as hp
    hp.compute_dtype = 'bf16'
    hp.bf16_maps = True
    yield
    hp.compute_dtype = 'fp32'


def enc_ref(x, slope=S):
    return F.leaky_relu(x, slope).bfloat16()


def dec_ref(a, slope=S):
    a = a.float()
    return torch.where(a > 0, a, a / slope)


def test_encode_decode_match_torch():
    from rtg import ops
    torch.manual_seed(0)
    for n in (1, 7, 8, 1000, 4099):
        x = torch.randn(n, device=DEV) * 3
        for slope in (S, 1.0):
            e = ops.bf16_encode(x, slope)
            assert e.dtype == torch.bfloat16
            assert torch.equal(e, enc_ref(x, slope))
            d = ops.bf16_decode(e, slope)
            np.testing.assert_allclose(d.cpu().numpy(), dec_ref(e, slope).cpu().numpy(), rtol=1e-6)
    # an odd offset into a buffer: the 2-byte aligned paths
    x = torch.randn(1000, device=DEV)
    e = ops.bf16_encode(x[3:], S)
    assert torch.equal(e, enc_ref(x[3:]))


def _close_bf16(a, b, what, flips=2e-3, beyond=1e-5):
    """two bf16-rounded evaluations of the same quantity: equal but for rare one-ulp flips (relative 2^-7) where fp32
    summation-order noise crosses a rounding boundary"""
    a, b = a.float().cpu(), b.float().cpu()
    scale = b.abs().max().item() + 1e-30
    bad = ((a - b).abs() > 2.0 ** -7 * b.abs() + 1e-6 * scale)
    assert bad.float().mean().item() <= beyond, (what, 'beyond one ulp', bad.float().mean().item())
    diff = (a != b).float().mean().item()
    assert diff <= flips, (what, 'fraction of differing elements', diff)


def _one_layer(kind, cin, cout, k, stride, pad):
    from models.layers import WNConv, BankedModel, conv

    class One(BankedModel):
        def __init__(self):
            super().__init__()
            self.c = WNConv(kind, cin, cout, k, stride=stride, pad=pad)

        def forward(self, x):
            return conv(self.token(), self.c, x, pre_slope=S)
    return One().to(DEV)


def _run(m, x_in, dy_in):
    """forward + backward of the one-layer model on a bf16 (encoded) or fp32 input -> (out, dx, grads)"""
    m.zero_grad()
    x = x_in.clone().requires_grad_(True)
    out = m(x)
    out.backward(dy_in)
    torch.cuda.synchronize()
    return out.detach(), x.grad.detach(), [p.grad.detach().clone() for p in (m.c.weight_v, m.c.weight_g, m.c.bias)]


CASES_1D = [(64, 128, 1, 4, 300), (128, 256, 3, 6, 304), (512, 512, 1, 12, 68), (256, 512, 3, 33, 30), (512, 512, 1, 40, 10),
            (32, 128, 3, 3, 1821)]


@pytest.mark.parametrize('cin,cout,stride,B,L', CASES_1D)
@pytest.mark.parametrize('x_bf', [True, False])
def test_dense_layer_native_bf16_io_equals_the_path_through_fp32_copies(bf16_mode, monkeypatch, cin, cout, stride, B, L, x_bf):
    """one k5 layer of the period / scale discriminators: bf16 (or, first dense layer of a stack, fp32) input, bf16 output,
    bf16 gradients; native kernels against the conversion fallback"""
    from rtg import ops
    torch.manual_seed(cin + L)
    m = _one_layer('conv', cin, cout, 5, stride, 2)
    ly = m.bank().layers[0]
    assert ly.maps_bf and ly.fwd_bf and ly.bwd_bf and ly.wgrad_bf
    x32 = torch.randn(B, cin, L, device=DEV)
    x = ops.bf16_encode(x32, S) if x_bf else x32
    Lo = (L + 4 - 5) // stride + 1
    dy = ops.bf16_encode(torch.randn(B, cout, Lo, device=DEV), 1.0)
    ops._NATIVE.clear()
    out, dx, gr = _run(m, x, dy)
    assert out.dtype == torch.bfloat16 and dx.dtype == x.dtype
    native = dict(ops._NATIVE)
    assert native and all(native.values()), 'this shape should be served natively: ' + str(list(native.values()))
    # the same through fp32 copies
    ops._NATIVE.clear()
    monkeypatch.setattr(ops, '_conv_native', lambda d: False)
    monkeypatch.setattr(ops, '_wgrad_native', lambda wd: False)
    out2, dx2, gr2 = _run(m, x, dy)
    _close_bf16(out, out2, 'out')
    if x_bf:
        _close_bf16(dx, dx2, 'dx')
    else:
        np.testing.assert_allclose(dx.cpu().numpy(), dx2.cpu().numpy(), rtol=1e-5, atol=1e-5 * dx2.abs().max().item())
    for a, b, nm in zip(gr, gr2, 'vgb'):
        err = ((a - b).norm() / (b.norm() + 1e-30)).item()
        assert err < 2e-4, (nm, err)
    # ... and against stock torch with the roundings written out
    rb = lambda t: t.bfloat16().float()                                       # noqa: E731
    w = rb(m.c.effective_weight().detach()).reshape(cout, cin, 5)
    xa = x.float() if x_bf else rb(F.leaky_relu(x32, S))
    ref = F.conv1d(xa, w, m.c.bias.detach(), stride, 2)
    # (the weights themselves flip a rounding here and there — g * v / ||v|| is formed in another order — which moves whole
    # output rows by more than an output ulp)
    _close_bf16(out, enc_ref(ref), 'out vs torch', flips=5e-2, beyond=2e-3)
    gin = torch.nn.grad.conv1d_input(xa.shape, w, dy.float(), stride, 2) * torch.where(xa > 0, 1.0, S)
    tol = 3e-4 * gin.abs().max().item()
    if x_bf:
        np.testing.assert_allclose(dx.float().cpu().numpy(), gin.cpu().numpy(), rtol=2.0 ** -7, atol=tol)
    else:
        np.testing.assert_allclose(dx.cpu().numpy(), gin.cpu().numpy(), rtol=1e-4, atol=tol)


CASES_2D = [((32, 64, (3, 3), (2, 2), (1, 1)), 2, 65, 33), ((64, 256, (5, 3), (3, 2), (2, 1)), 2, 129, 17),
            ((256, 512, (5, 3), (3, 2), (2, 1)), 3, 43, 9), ((512, 512, (3, 3), (1, 1), (1, 1)), 3, 15, 9)]


@pytest.mark.parametrize('spec,B,H,W', CASES_2D)
@pytest.mark.parametrize('x_bf', [True, False])
def test_conv2d_layer_native_bf16_io_equals_the_path_through_fp32_copies(bf16_mode, monkeypatch, spec, B, H, W, x_bf):
    """the Conv2d layers of StftDiscriminator (discrminator.py:255-262): forward, backward-data (row stride 1: the 3-tap
    instance; row-strided: class-ordered clips, 2 taps) and weight gradient on bf16 tensors"""
    from rtg import ops
    cin, cout, k, stride, pad = spec
    torch.manual_seed(cin + H)
    m = _one_layer('conv2d', cin, cout, k, stride, pad)
    ly = m.bank().layers[0]
    assert ly.maps_bf
    x32 = torch.randn(B, cin, H, W, device=DEV)
    x = ops.bf16_encode(x32, S) if x_bf else x32
    Ho, Wo = (H + 2 * pad[0] - k[0]) // stride[0] + 1, (W + 2 * pad[1] - k[1]) // stride[1] + 1
    dy = ops.bf16_encode(torch.randn(B, cout, Ho, Wo, device=DEV), 1.0)
    ops._NATIVE.clear()
    out, dx, gr = _run(m, x, dy)
    assert out.dtype == torch.bfloat16 and dx.dtype == x.dtype
    native = dict(ops._NATIVE)
    print('native paths:', sum(native.values()), 'of', len(native))
    ops._NATIVE.clear()
    monkeypatch.setattr(ops, '_conv_native', lambda d: False)
    monkeypatch.setattr(ops, '_wgrad_native', lambda wd: False)
    out2, dx2, gr2 = _run(m, x, dy)
    _close_bf16(out, out2, 'out')
    if x_bf:
        _close_bf16(dx, dx2, 'dx')
    else:
        np.testing.assert_allclose(dx.cpu().numpy(), dx2.cpu().numpy(), rtol=1e-5, atol=1e-5 * dx2.abs().max().item())
    for a, b, nm in zip(gr, gr2, 'vgb'):
        err = ((a - b).norm() / (b.norm() + 1e-30)).item()
        assert err < 2e-4, (nm, err)
    rb = lambda t: t.bfloat16().float()                                       # noqa: E731
    w = rb(m.c.effective_weight().detach())
    xa = x.float() if x_bf else rb(F.leaky_relu(x32, S))
    ref = F.conv2d(xa, w, m.c.bias.detach(), stride, pad)
    _close_bf16(out, enc_ref(ref), 'out vs torch', flips=5e-2, beyond=2e-3)


def test_feature_loss_over_bf16_maps(bf16_mode):
    """RTG_LOSS_L1_ENC: mean |dec(a) - dec(b)| over bf16 (leaky-relu encoded) feature maps, gradients as bf16"""
    from models import feature_loss
    from rtg import ops
    torch.manual_seed(3)
    shapes = [(4, 128, 203), (4, 512, 68), (4, 32, 607)]
    r32 = [torch.randn(*s, device=DEV) for s in shapes]
    g32 = [torch.randn(*s, device=DEV) for s in shapes]
    r = [ops.bf16_encode(t, S) for t in r32[:2]] + [r32[2]]                   # two bf16 maps and an fp32 one (a first layer's)
    g = [ops.bf16_encode(t, S).requires_grad_(True) for t in g32[:2]] + [g32[2].clone().requires_grad_(True)]
    loss = feature_loss([r], [g])
    loss.backward()
    rr = [dec_ref(t) for t in r[:2]] + [r32[2]]
    gg = [dec_ref(t.detach()).requires_grad_(True) for t in g[:2]] + [g32[2].clone().requires_grad_(True)]
    ref = sum(F.l1_loss(a, b) for a, b in zip(rr, gg))
    ref.backward()
    np.testing.assert_allclose(loss.item(), ref.item(), rtol=2e-6)
    # a lone pair of one encoded and one decoded map (what the scale discriminators hand over: the real half of convs.5's
    # output still bf16, the generated half decoded for conv_post)
    one = feature_loss([[r[0]]], [[dec_ref(g[0].detach())]])
    np.testing.assert_allclose(one.item(), F.l1_loss(rr[0], gg[0]).item(), rtol=2e-6)
    for a, b in zip(g, gg):
        if a.dtype == torch.bfloat16:
            assert a.grad.dtype == torch.bfloat16
            np.testing.assert_allclose(a.grad.float().cpu().numpy(), b.grad.cpu().numpy(), rtol=2.0 ** -8, atol=0)
        else:
            np.testing.assert_allclose(a.grad.cpu().numpy(), b.grad.cpu().numpy(), rtol=1e-6)
