"""rtg_dconv.hip — the dense-layer conv kernel (block-shape codes 8xxx of rtg_conv1d): every code the library lists for a
problem must reproduce the general kernel BIT FOR BIT (same accumulation order, same epilogue arithmetic) and agree with
an fp64 torch convolution; the 16-byte-fragment weight image of rtg_weights_pack must equal the host statement of it.
Shapes: the DiscriminatorP / DiscriminatorS layers the kernel was written for (discrminator.py:44,155-163) at reduced
batch, plus ragged rows / columns / clips."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import packref

pytestmark = pytest.mark.gpu


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _both_images(W, tile_m=32):
    """standard image followed by the 16-byte-fragment image (what a bank layer with wp16 holds)"""
    return np.concatenate([packref.pack_logical(W, tile_m), packref.pack_frag16(W)])


def _run(desc_kw, x, wp, bias=None, mask=None, res=None, out_shape=None, cfg=0, out_init=None):
    from rtg.lib import lib, Conv1dDesc
    d = Conv1dDesc(**desc_kw)
    d.tile_cfg = cfg
    out = torch.full(out_shape, float('nan'), device='cuda') if out_init is None else out_init.clone()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = lib.rtg_conv1d(C.byref(d), _ptr(x), None, None, _ptr(wp), _ptr(bias), _ptr(mask), _ptr(res), _ptr(out), None, st)
    torch.cuda.synchronize()
    return rc, out


def _codes(desc_kw):
    from rtg.lib import lib, Conv1dDesc
    d = Conv1dDesc(**desc_kw)
    cands = (C.c_int * 48)()
    n = lib.rtg_conv1d_tile_candidates(C.byref(d), cands, 48)
    return [c for c in cands[:n] if 8000 < c < 9000]      # (9xxx: rtg_sconv.hip, another summation order — its own tests)


def _desc(B, Cin, L_in, Mg, K, stride, pad, Q, out_C, out_L, **kw):
    d = dict(B=B, C1=Cin, C2=0, L_in=L_in, groups=1, Cg=Cin, Mg=Mg, K=K, stride=stride, dil=1, pad=pad, Q=Q, out_C=out_C,
             out_L=out_L, shuf_S=1, shuf_P=0, pre_mode=0, pre_slope=1.0, mask_slope=1.0, out_scale=1.0, act=0,
             act_slope=1.0, accumulate=0, tile_m=32, out_split=0, wp16=1)
    d.update(kw)
    return d


FWD = [
    # B, C_in, C_out, L, stride      (k = 5, pad = 2)
    (24, 512, 512, 10, 1),       # DiscriminatorP convs.4, period 11: rows of 10
    (7, 512, 512, 34, 1),        # period 3: rows of 34, a ragged clip count
    (5, 512, 512, 128, 1),       # DiscriminatorS convs.5: one clip per 128-column tile
    (3, 512, 512, 64, 1),
    (9, 256, 512, 102, 3),       # DiscriminatorP convs.3 (stride 3)
    (13, 256, 512, 28, 3),
    (6, 128, 256, 304, 3),       # convs.2
    (4, 128, 144, 83, 3),        # rows not a multiple of 128 / 64: clamped row tiles
]


@pytest.mark.parametrize('case', FWD)
def test_forward_codes_bit_identical_to_the_general_kernel(case):
    B, Cin, Cout, L, s = case
    K, p = 5, 2
    gen = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn(B, Cin, L, generator=gen)
    w = torch.randn(Cout, Cin, K, generator=gen) / np.sqrt(Cin * K)
    bias = torch.randn(Cout, generator=gen)
    ref = F.conv1d(F.leaky_relu(x, 0.15).double(), w.double(), bias.double(), s, p).float()
    Lo = ref.shape[-1]
    wp = torch.from_numpy(_both_images(packref.logical_fwd(w.numpy(), 1))).cuda()
    kw = _desc(B, Cin, L, Cout, K, s, p, Lo, Cout, Lo, pre_mode=1, pre_slope=0.15)
    xd, bd = x.cuda(), bias.cuda()
    rc, base = _run(kw, xd, wp, bias=bd, out_shape=(B, Cout, Lo))
    assert rc == 0
    np.testing.assert_allclose(base.cpu().numpy(), ref.numpy(), rtol=1e-4, atol=2e-5)
    codes = _codes(kw)
    assert codes, 'no dense-layer code listed'
    for c in codes:
        rc, out = _run(kw, xd, wp, bias=bd, out_shape=(B, Cout, Lo), cfg=c)
        assert rc == 0, c
        assert torch.equal(out, base), (c, (out - base).abs().max().item())


def test_without_the_fragment_image_no_dense_code_is_listed_or_accepted():
    kw = _desc(8, 512, 34, 512, 5, 1, 2, 34, 512, 34, wp16=0)
    assert _codes(kw) == []
    x = torch.zeros(8, 512, 34, device='cuda')
    wp = torch.zeros(512 * 512 * 5, device='cuda')
    rc, _ = _run(kw, x, wp, out_shape=(8, 512, 34), cfg=8107)
    assert rc == -1


@pytest.mark.parametrize('case', [(10, 512, 512, 21), (3, 512, 512, 128), (5, 256, 384, 15)])
def test_dgrad_stride1_codes(case):
    """backward-data of a stride-1 k5 layer: the same operator on flipped / transposed weights, leaky-relu-derivative mask
    and a residual gradient (PairConvFn's tap) in the epilogue, out_scale and accumulate too"""
    B, Cin, Cout, L = case
    K, p = 5, 2
    gen = torch.Generator().manual_seed(17)
    x = torch.randn(B, Cin, L, generator=gen, dtype=torch.float64, requires_grad=True)
    w = torch.randn(Cout, Cin, K, generator=gen) / np.sqrt(Cin * K)
    y = F.conv1d(F.leaky_relu(x, 0.15), w.double(), None, 1, p)
    dy = torch.randn(y.shape, generator=gen)
    tap = torch.randn(B, Cin, L, generator=gen)
    y.backward(dy.double())
    ref = (x.grad.float() + tap)
    wp = torch.from_numpy(_both_images(packref.logical_dgrad_s1(w.numpy(), 1))).cuda()
    kw = _desc(B, Cout, L, Cin, K, 1, (K - 1) - p, L, Cin, L, mask_slope=0.15)
    xm, dyd, tapd = x.detach().float().cuda(), dy.cuda(), tap.cuda()
    rc, base = _run(kw, dyd, wp, mask=xm, res=tapd, out_shape=(B, Cin, L))
    assert rc == 0
    np.testing.assert_allclose(base.cpu().numpy(), ref.numpy(), rtol=1e-4, atol=3e-5)
    init = torch.randn(B, Cin, L, device='cuda')
    kw2 = dict(kw, accumulate=1, out_scale=0.5)
    rc, base2 = _run(kw2, dyd, wp, mask=xm, res=tapd, out_shape=(B, Cin, L), out_init=init)
    codes = _codes(kw)
    assert codes
    for c in codes:
        rc, out = _run(kw, dyd, wp, mask=xm, res=tapd, out_shape=(B, Cin, L), cfg=c)
        assert rc == 0 and torch.equal(out, base), c
        rc, out = _run(kw2, dyd, wp, mask=xm, res=tapd, out_shape=(B, Cin, L), cfg=c, out_init=init)
        assert rc == 0 and torch.equal(out, base2), c


@pytest.mark.parametrize('case', [(9, 256, 512, 102), (13, 256, 512, 28), (6, 128, 256, 304), (5, 128, 256, 83)])
def test_dgrad_polyphase_codes(case):
    """backward-data of the stride-3 k5 layers: 2-tap stride-1 operator over dy with (channel, phase) rows and the
    interleaving store"""
    B, Cin, Cout, L = case
    K, s, p = 5, 3, 2
    gen = torch.Generator().manual_seed(19)
    x = torch.randn(B, Cin, L, generator=gen, dtype=torch.float64, requires_grad=True)
    w = torch.randn(Cout, Cin, K, generator=gen) / np.sqrt(Cin * K)
    y = F.conv1d(F.leaky_relu(x, 0.15), w.double(), None, s, p)
    Lo = y.shape[-1]
    dy = torch.randn(y.shape, generator=gen)
    y.backward(dy.double())
    ref = x.grad.float()
    W = packref.logical_dgrad_poly(w.numpy(), 1, s)
    nt = W.shape[-1]
    wp = torch.from_numpy(_both_images(W)).cuda()
    nq = (L - 1 + p) // s + 1
    kw = _desc(B, Cout, Lo, Cin * s, nt, 1, nt - 1, nq, Cin, L, shuf_S=s, shuf_P=p, mask_slope=0.15)
    xm, dyd = x.detach().float().cuda(), dy.cuda()
    rc, base = _run(kw, dyd, wp, mask=xm, out_shape=(B, Cin, L))
    assert rc == 0
    np.testing.assert_allclose(base.cpu().numpy(), ref.numpy(), rtol=1e-4, atol=3e-5)
    codes = _codes(kw)
    assert codes
    for c in codes:
        rc, out = _run(kw, dyd, wp, mask=xm, out_shape=(B, Cin, L), cfg=c)
        assert rc == 0 and torch.equal(out, base), c


def test_pack_kernel_writes_the_fragment_image():
    """rtg_weights_pack with RtgPackJob.frag16 against tests/packref.pack_frag16, forward and both backward-data operators"""
    from rtg import lib as L
    from rtg.lib import lib
    gen = torch.Generator().manual_seed(23)
    Cout, Cin, K = 144, 128, 5
    v = torch.randn(Cout, Cin, K, generator=gen)
    g = torch.rand(Cout, generator=gen) + 0.5
    w_eff = (v * (g / v.flatten(1).norm(dim=1)).view(-1, 1, 1)).numpy()
    params = torch.cat([g, v.flatten()]).cuda()
    scales = torch.empty(2 * Cout, device='cuda')
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    norm = (L.NormJob * 1)(L.NormJob(0, Cout, 0, Cout, Cin * K))
    norm_d = torch.frombuffer(bytearray(bytes(memoryview(norm).cast('B'))), dtype=torch.uint8).cuda()
    assert lib.rtg_weightnorm_scales(_ptr(norm_d), 1, Cout, _ptr(params), _ptr(scales), st) == 0
    for mode, W in ((L.PACK_FWD, packref.logical_fwd(w_eff, 1)), (L.PACK_DGRAD_S1, packref.logical_dgrad_s1(w_eff, 1)),
                    (L.PACK_DGRAD_POLY, packref.logical_dgrad_poly(w_eff, 1, 3))):
        _, Mg, Cg, Kp = W.shape
        size = lib.rtg_packed_size_frag16(Mg, Cg, Kp)
        ref = packref.pack_frag16(W)
        assert size == ref.size
        job = (L.PackJob * 1)(L.PackJob(Cout, 0, 0, size, mode, 1, Mg, Cg, Kp, K, Cin, 3 if mode == L.PACK_DGRAD_POLY else 1,
                                       16, 1, 0, 0, 1))
        blocks, lds = L.assign_pack_blocks(job)
        job_d = torch.frombuffer(bytearray(bytes(memoryview(job).cast('B'))), dtype=torch.uint8).cuda()
        packed = torch.full((size,), float('nan'), device='cuda')
        assert lib.rtg_weights_pack(_ptr(job_d), 1, blocks, lds, _ptr(params), _ptr(scales), _ptr(packed), st) == 0
        torch.cuda.synchronize()
        np.testing.assert_allclose(packed.cpu().numpy(), ref, rtol=1e-6, atol=1e-7)


CONV2D = [
    # B, Cin, Cout, H, W, (kh, kw), (sh, sw), (ph, pw)        StftDiscriminator layers (discrminator.py:255-262)
    (2, 64, 256, 40, 18, (5, 3), (3, 2), (2, 1)),
    (3, 256, 512, 22, 9, (5, 3), (3, 2), (2, 1)),
    (2, 512, 512, 8, 5, (3, 3), (1, 1), (1, 1)),
    (2, 32, 96, 33, 35, (3, 3), (2, 2), (1, 1)),
    (1, 64, 128, 13, 69, (3, 3), (1, 1), (1, 1)),
]


@pytest.mark.parametrize('case', CONV2D)
def test_conv2d_forward_codes(case):
    """the second dimension of the dense-layer kernel: clips = (item, output row), channels = (channel, kernel row), the
    patch row of a clip gathered per kernel row; bit-identical to the general kernel's 2-D mode, close to torch.conv2d"""
    B, Cin, Cout, H, W, (kh, kw), (sh, sw), (ph, pw) = case
    gen = torch.Generator().manual_seed(29)
    x = torch.randn(B, Cin, H, W, generator=gen)
    w = torch.randn(Cout, Cin, kh, kw, generator=gen) / np.sqrt(Cin * kh * kw)
    bias = torch.randn(Cout, generator=gen)
    ref = F.conv2d(F.leaky_relu(x, 0.15).double(), w.double(), bias.double(), (sh, sw), (ph, pw)).float()
    Ho, Wo = ref.shape[-2:]
    Wl = w.numpy().reshape(1, Cout, Cin * kh, kw)                  # virtual channel = ci * kh + row
    wp = torch.from_numpy(_both_images(Wl)).cuda()
    kw_ = _desc(B * Ho, Cin * kh, W, Cout, kw, sw, pw, Wo, Cout, Wo, pre_mode=1, pre_slope=0.15, h_in=H, h_k=kh,
                h_stride=sh, h_pad=ph, h_n=Ho, h_mode=0)
    xd, bd = x.cuda(), bias.cuda()
    rc, base = _run(kw_, xd, wp, bias=bd, out_shape=(B, Cout, Ho, Wo))
    assert rc == 0
    np.testing.assert_allclose(base.cpu().numpy(), ref.numpy(), rtol=1e-4, atol=3e-5)
    codes = _codes(kw_)
    assert codes, 'no dense-layer code listed for the 2-D problem'
    for c in codes:
        rc, out = _run(kw_, xd, wp, bias=bd, out_shape=(B, Cout, Ho, Wo), cfg=c)
        assert rc == 0, c
        assert torch.equal(out, base), (c, (out - base).abs().max().item())


@pytest.mark.parametrize('case', [(2, 512, 512, 8, 5), (1, 64, 128, 13, 69)])
def test_conv2d_dgrad_stride1_codes(case):
    """backward-data of the 3x3 stride-(1, 1) layer: channels ordered (kernel row, output channel), rows r + pad - kh"""
    B, Cin, Cout, H, W = case
    k, p = 3, 1
    gen = torch.Generator().manual_seed(31)
    x = torch.randn(B, Cin, H, W, generator=gen, dtype=torch.float64, requires_grad=True)
    w = torch.randn(Cout, Cin, k, k, generator=gen) / np.sqrt(Cin * k * k)
    y = F.conv2d(F.leaky_relu(x, 0.15), w.double(), None, 1, p)
    dy = torch.randn(y.shape, generator=gen)
    y.backward(dy.double())
    ref = x.grad.float()
    # rows = input channels, channels = (kh, co), taps along W flipped (RTG_PACK_DGRAD_2D with stride 1)
    Wl = np.ascontiguousarray(w.numpy().transpose(1, 2, 0, 3)[..., ::-1]).reshape(1, Cin, k * Cout, k)
    wp = torch.from_numpy(_both_images(Wl)).cuda()
    kw_ = _desc(B * H, Cout * k, W, Cin, k, 1, (k - 1) - p, W, Cin, W, mask_slope=0.15, h_in=H, h_k=k, h_stride=1,
                h_pad=p, h_n=H, h_mode=1)
    xm, dyd = x.detach().float().cuda(), dy.cuda()
    rc, base = _run(kw_, dyd, wp, mask=xm, out_shape=(B, Cin, H, W))
    assert rc == 0
    np.testing.assert_allclose(base.cpu().numpy(), ref.numpy(), rtol=1e-4, atol=3e-5)
    codes = _codes(kw_)
    assert codes
    for c in codes:
        rc, out = _run(kw_, dyd, wp, mask=xm, out_shape=(B, Cin, H, W), cfg=c)
        assert rc == 0 and torch.equal(out, base), c


@pytest.mark.parametrize('case', [(2, 64, 256, 40, 18, (5, 3), (3, 2), (2, 1)), (3, 256, 512, 22, 9, (5, 3), (3, 2), (2, 1)),
                                  (2, 32, 64, 33, 35, (3, 3), (2, 2), (1, 1)), (1, 128, 96, 17, 20, (5, 3), (3, 2), (2, 1)),
                                  # (round 5: the XCDs' item ranges are cut by work — residue classes with 2 / 1 / 1 / 1 kernel rows,
                                  # a class WITHOUT kernel rows (3 rows at stride 4), enough tiles for every XCD to get a range)
                                  (2, 64, 128, 35, 22, (5, 3), (4, 2), (2, 1)), (2, 64, 128, 33, 18, (3, 3), (4, 2), (1, 1)),
                                  (6, 64, 128, 48, 70, (3, 3), (2, 2), (1, 1))])
def test_conv2d_dgrad_strided_codes(case):
    """backward-data of the row-strided Conv2d layers: clips in residue-class order of their rows (a tile inside one class
    walks only that class's kernel rows), polyphase walk and interleaving store along the last axis"""
    B, Cin, Cout, H, W, (kh, kw), (sh, sw), (ph, pw) = case
    gen = torch.Generator().manual_seed(37)
    x = torch.randn(B, Cin, H, W, generator=gen, dtype=torch.float64, requires_grad=True)
    w = torch.randn(Cout, Cin, kh, kw, generator=gen) / np.sqrt(Cin * kh * kw)
    y = F.conv2d(F.leaky_relu(x, 0.15), w.double(), None, (sh, sw), (ph, pw))
    Ho, Wo = y.shape[-2:]
    dy = torch.randn(y.shape, generator=gen)
    y.backward(dy.double())
    ref = x.grad.float()
    nt = -(-kw // sw)
    wn = w.numpy()
    Wl = np.zeros((1, Cin * sw, kh * Cout, nt), dtype=np.float32)      # rows (ci, phase), channels (kh, co): RTG_PACK_DGRAD_2D
    for r in range(sw):
        for tap in range(nt):
            jj = r + (nt - 1 - tap) * sw
            if jj < kw:
                Wl[0, r::sw, :, tap] = wn[:, :, :, jj].transpose(1, 2, 0).reshape(Cin, kh * Cout)
    wp = torch.from_numpy(_both_images(Wl)).cuda()
    nq = (W - 1 + pw) // sw + 1
    kw_ = _desc(B * H, Cout * kh, Wo, Cin * sw, nt, 1, nt - 1, nq, Cin, W, shuf_S=sw, shuf_P=pw, mask_slope=0.15, h_in=Ho,
                h_k=kh, h_stride=sh, h_pad=ph, h_n=H, h_mode=1)
    xm, dyd = x.detach().float().cuda(), dy.cuda()
    rc, base = _run(kw_, dyd, wp, mask=xm, out_shape=(B, Cin, H, W))
    assert rc == 0
    np.testing.assert_allclose(base.cpu().numpy(), ref.numpy(), rtol=1e-4, atol=3e-5)
    codes = _codes(kw_)
    assert codes
    for c in codes:
        rc, out = _run(kw_, dyd, wp, mask=xm, out_shape=(B, Cin, H, W), cfg=c)
        assert rc == 0 and torch.equal(out, base), c


def _pack_on_gpu(v, g, mode, groups, Mg, Cg, Kp, src_K, src_inner_c, S, tile_m, tap_major=0, frag16=0, bf16=0):
    """rtg_weightnorm_scales + rtg_weights_pack of ONE tensor through the C ABI -> the packed image (numpy)"""
    from rtg import lib as L
    from rtg.lib import lib
    rows = v.shape[0]
    inner = int(np.prod(v.shape[1:]))
    params = torch.cat([g.flatten(), v.flatten()]).cuda()
    scales = torch.empty(2 * rows, device='cuda')
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def table(job):
        return torch.frombuffer(bytearray(bytes(memoryview((type(job) * 1)(job)).cast('B'))), dtype=torch.uint8).cuda()
    assert lib.rtg_weightnorm_scales(_ptr(table(L.NormJob(0, rows, 0, rows, inner))), 1, rows, _ptr(params), _ptr(scales), st) == 0
    if frag16:
        size = (lib.rtg_packed_size_frag16_bf16 if bf16 else lib.rtg_packed_size_frag16)(Mg, Cg, Kp)
    elif tap_major:
        size = lib.rtg_packed_size_tapmajor(groups, Mg, Cg, Kp, tile_m)
    else:
        size = lib.rtg_packed_size(groups, Mg, Cg, Kp, tile_m)
    job = L.PackJob(rows, 0, 0, size, mode, groups, Mg, Cg, Kp, src_K, src_inner_c, S, tile_m, 1, tap_major, bf16, frag16)
    blocks, lds = L.assign_pack_blocks([job])
    packed = torch.full((size,), float('nan'), device='cuda')
    assert lib.rtg_weights_pack(_ptr(table(job)), 1, blocks, lds, _ptr(params), _ptr(scales), _ptr(packed), st) == 0
    torch.cuda.synchronize()
    return packed.cpu().numpy()


PACK = [
    # kind, C_out (C_in for convT), C_in (C_out), K, stride, groups, tile_m
    ('fwd', 144, 40, 7, 1, 1, 32),          # ragged rows and channels (staged path)
    ('fwd', 64, 64, 15, 8, 1, 32),          # long taps (staged: 240-float runs)
    ('fwd', 128, 32, 41, 2, 4, 16),         # grouped k41: runs too long for the slab -> gather path
    ('dgrad_s1', 96, 48, 3, 1, 1, 32),
    ('dgrad_s1', 32, 48, 7, 1, 1, 16),      # 16-row tiles
    ('dgrad_poly', 512, 256, 5, 3, 1, 32),
    ('dgrad_poly', 64, 32, 15, 8, 1, 32),
    ('dgrad_poly', 128, 64, 41, 4, 8, 16),  # grouped, long taps -> gather
    ('convT', 128, 64, 15, 8, 1, 32),
]


@pytest.mark.parametrize('case', PACK)
def test_pack_kernel_standard_images(case):
    """every packed layout of rtg_weights_pack (coalesced slab staging and the gather fallback) against tests/packref.py"""
    from rtg import lib as L
    kind, c0, c1, K, s, groups, TM = case
    gen = torch.Generator().manual_seed(41)
    if kind == 'convT':
        v = torch.randn(c0, c1, K, generator=gen)                  # [C_in, C_out, K]
    else:
        v = torch.randn(c0, c1 // groups, K, generator=gen)        # [C_out, C_in / groups, K]
    g = torch.rand(v.shape[0], generator=gen) + 0.5
    w_eff = (v * (g / v.flatten(1).norm(dim=1)).view(-1, 1, 1)).numpy()
    if kind == 'fwd':
        W, mode, S = packref.logical_fwd(w_eff, groups), L.PACK_FWD, 1
    elif kind == 'dgrad_s1':
        W, mode, S = packref.logical_dgrad_s1(w_eff, groups), L.PACK_DGRAD_S1, 1
    elif kind == 'dgrad_poly':
        W, mode, S = packref.logical_dgrad_poly(w_eff, groups, s), L.PACK_DGRAD_POLY, s
    else:
        W, mode, S = packref.logical_convT_poly(w_eff, s), L.PACK_CONVT_POLY, s
    G, Mg, Cg, Kp = W.shape
    got = _pack_on_gpu(v, g, mode, G, Mg, Cg, Kp, K, v.shape[1], S, TM)
    np.testing.assert_allclose(got, packref.pack_logical(W, TM), rtol=1e-6, atol=1e-7)
    if G == 1 and kind != 'convT':
        got16 = _pack_on_gpu(v, g, mode, 1, Mg, Cg, Kp, K, v.shape[1], S, 16, frag16=1)
        np.testing.assert_allclose(got16, packref.pack_frag16(W), rtol=1e-6, atol=1e-7)
    if kind == 'fwd' and Cg <= 16:
        gott = _pack_on_gpu(v, g, mode, G, Mg, Cg, Kp, K, v.shape[1], S, TM, tap_major=1)
        np.testing.assert_allclose(gott, packref.pack_logical_tapmajor(W, TM), rtol=1e-6, atol=1e-7)
