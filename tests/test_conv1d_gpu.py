"""rtg_conv1d (HIP, fp32 MFMA) against torch CPU convolutions, through the C ABI.  GPU only."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import packref

pytestmark = pytest.mark.gpu


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def run_conv(desc_kw, x1, wp, x2=None, aux=None, bias=None, mask=None, res=None, out_shape=None, out_init=None):
    from rtg.lib import lib, Conv1dDesc, check
    d = Conv1dDesc(**desc_kw)
    dev = 'cuda'
    t = {k: (v.to(dev).contiguous() if v is not None else None)
         for k, v in dict(x1=x1, x2=x2, aux=aux, bias=bias, mask=mask, res=res).items()}
    wp_d = torch.from_numpy(wp).to(dev)
    out = torch.full(out_shape, float('nan'), device=dev) if out_init is None else out_init.to(dev).clone()
    st = torch.cuda.current_stream().cuda_stream
    check(lib.rtg_conv1d(C.byref(d), _ptr(t['x1']), _ptr(t['x2']), _ptr(t['aux']), _ptr(wp_d), _ptr(t['bias']),
                         _ptr(t['mask']), _ptr(t['res']), _ptr(out), None, C.c_void_p(st)), 'rtg_conv1d')
    torch.cuda.synchronize()
    return out.cpu()


def base_desc(B, C1, C2, L_in, groups, Cg, Mg, K, stride, dil, pad, Q, out_C, out_L, tile_m, **kw):
    d = dict(B=B, C1=C1, C2=C2, L_in=L_in, groups=groups, Cg=Cg, Mg=Mg, K=K, stride=stride, dil=dil, pad=pad, Q=Q,
             out_C=out_C, out_L=out_L, shuf_S=1, shuf_P=0, pre_mode=0, pre_slope=1.0, mask_slope=1.0, out_scale=1.0,
             act=0, act_slope=1.0, accumulate=0, tile_m=tile_m, out_split=0)
    d.update(kw)
    return d


FWD_CASES = [
    # B, C_in, C_out, L, K, stride, dil, pad, groups, tile_m
    (2, 32, 32, 1024, 3, 1, 1, 1, 1, 32),
    (2, 32, 32, 1000, 7, 1, 9, 27, 1, 32),       # ResBlock3 k7 d9, ragged length
    (2, 64, 64, 512, 5, 1, 3, 6, 1, 32),
    (3, 128, 128, 256, 7, 1, 1, 3, 1, 32),
    (2, 128, 128, 32, 3, 1, 9, 9, 1, 32),        # ResidualStack at T'=32 (dilation wider than the signal)
    (2, 16, 32, 2048, 7, 4, 1, 3, 1, 32),        # downs.0
    (2, 32, 64, 2048, 15, 8, 1, 7, 1, 32),       # downs.1
    (2, 1, 16, 1024, 7, 1, 1, 3, 1, 32),         # conv_pre  (C_in = 1)
    (2, 32, 1, 1024, 7, 1, 1, 3, 1, 16),         # conv_post (C_out = 1)
    (2, 1, 32, 1024, 15, 1, 1, 7, 1, 32),        # MSD conv0
    (2, 32, 64, 1024, 41, 2, 1, 20, 4, 16),      # MSD grouped k41 s2
    (2, 128, 512, 512, 41, 4, 1, 20, 32, 16),    # MSD grouped k41 s4, 4 in-ch per group
    (2, 512, 512, 128, 41, 4, 1, 20, 64, 16),    # MSD grouped, 8 rows per group
    (2, 512, 512, 64, 5, 1, 1, 2, 1, 32),        # MSD conv5
    (6, 32, 128, 911, 5, 3, 1, 2, 1, 32),        # MPD (5,1) stride 3 with the period folded into the batch
    (6, 512, 1, 34, 3, 1, 1, 1, 1, 16),          # D conv_post
    (1, 48, 32, 300, 7, 1, 1, 3, 1, 32),         # merge.2 shape, C_in not a multiple of 32
]


@pytest.mark.parametrize('case', FWD_CASES)
def test_conv_forward_matches_torch(case):
    B, Cin, Cout, L, K, s, d, p, g, TM = case
    gen = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn(B, Cin, L, generator=gen)
    w = torch.randn(Cout, Cin // g, K, generator=gen) / np.sqrt(Cin // g * K)
    bias = torch.randn(Cout, generator=gen)
    ref = F.conv1d(F.leaky_relu(x, 0.15).double(), w.double(), bias.double(), s, p, d, g)
    L_out = ref.shape[-1]
    res = torch.randn(B, Cout, L_out, generator=gen)
    ref = F.leaky_relu((ref + res.double()) * 0.5, 0.01).float()
    wp = packref.pack_logical(packref.logical_fwd(w.numpy(), g), TM)
    desc = base_desc(B, Cin, 0, L, g, Cin // g, Cout // g, K, s, d, p, L_out, Cout, L_out, TM,
                     pre_mode=1, pre_slope=0.15, out_scale=0.5, act=1, act_slope=0.01)
    out = run_conv(desc, x, wp, bias=bias, res=res, out_shape=(B, Cout, L_out))
    np.testing.assert_allclose(out.numpy(), ref.numpy(), rtol=1e-4, atol=2e-5)


def test_virtual_concat_and_accumulate():
    gen = torch.Generator().manual_seed(5)
    B, C1, C2, Cout, L, K = 2, 80, 128, 256, 32, 7
    x1, x2 = torch.randn(B, C1, L, generator=gen), torch.randn(B, C2, L, generator=gen)
    w = torch.randn(Cout, C1 + C2, K, generator=gen) / 30
    ref = F.conv1d(torch.cat([x1, x2], 1).double(), w.double(), None, 1, 3).float()
    wp = packref.pack_logical(packref.logical_fwd(w.numpy(), 1), 32)
    desc = base_desc(B, C1, C2, L, 1, C1 + C2, Cout, K, 1, 1, 3, L, Cout, L, 32, accumulate=1)
    init = torch.randn(B, Cout, L, generator=gen)
    out = run_conv(desc, x1, wp, x2=x2, out_shape=(B, Cout, L), out_init=init)
    np.testing.assert_allclose(out.numpy(), (ref + init).numpy(), rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize('case', [(2, 32, 32, 777, 7, 9, 27), (2, 64, 64, 256, 5, 3, 6), (2, 128, 128, 64, 3, 1, 1)])
def test_dgrad_stride1(case):
    """backward-data of a 'same' dilated conv = rtg_conv1d on RTG_PACK_DGRAD_S1 weights, with the leaky-relu
    derivative mask and the residual gradient fused in the epilogue (ResBlock3: y = conv(lrelu(x)) + x)."""
    B, Cin, Cout, L, K, d, p = case
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(B, Cin, L, generator=gen, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(Cout, Cin, K, generator=gen) / np.sqrt(Cin * K))
    y = F.conv1d(F.leaky_relu(x, 0.15), w.double(), None, 1, p, d) + x
    dy = torch.randn(y.shape, generator=gen)
    y.backward(dy.double())
    wp = packref.pack_logical(packref.logical_dgrad_s1(w.numpy(), 1), 32)
    desc = base_desc(B, Cout, 0, L, 1, Cout, Cin, K, 1, d, (K - 1) * d - p, L, Cin, L, 32, mask_slope=0.15)
    out = run_conv(desc, dy, wp, mask=x.detach().float(), res=dy, out_shape=(B, Cin, L))
    np.testing.assert_allclose(out.numpy(), x.grad.float().numpy(), rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize('case', [(2, 16, 32, 2048, 7, 4, 3, 1, 32), (2, 64, 128, 256, 15, 8, 7, 1, 32),
                                  (6, 32, 128, 911, 5, 3, 2, 1, 32), (2, 32, 64, 1024, 41, 2, 20, 4, 16),
                                  (2, 512, 512, 128, 41, 4, 20, 64, 16)])
def test_dgrad_strided_polyphase(case):
    B, Cin, Cout, L, K, s, p, g, TM = case
    gen = torch.Generator().manual_seed(13)
    x = torch.randn(B, Cin, L, generator=gen, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(Cout, Cin // g, K, generator=gen) / np.sqrt(Cin // g * K))
    y = F.conv1d(x, w.double(), None, s, p, 1, g)
    dy = torch.randn(y.shape, generator=gen)
    y.backward(dy.double())
    Lo = y.shape[-1]
    nt = -(-K // s)
    wp = packref.pack_logical(packref.logical_dgrad_poly(w.numpy(), g, s), TM)
    NQ = (L - 1 + p) // s + 1
    desc = base_desc(B, Cout, 0, Lo, g, Cout // g, (Cin // g) * s, nt, 1, 1, nt - 1, NQ, Cin, L, TM,
                     shuf_S=s, shuf_P=p)
    out = run_conv(desc, dy, wp, out_shape=(B, Cin, L))
    np.testing.assert_allclose(out.numpy(), x.grad.float().numpy(), rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize('case', [(2, 256, 128, 32, 15, 8, 7, 7), (2, 128, 64, 256, 15, 8, 7, 7),
                                  (2, 64, 32, 512, 7, 4, 3, 3), (2, 64, 32, 2048, 7, 4, 3, 3)])
def test_conv_transpose_forward_and_dgrad(case):
    B, Cin, Cout, L, K, s, p, op = case
    gen = torch.Generator().manual_seed(17)
    x = torch.randn(B, Cin, L, generator=gen, dtype=torch.float64, requires_grad=True)
    w = torch.randn(Cin, Cout, K, generator=gen) / np.sqrt(Cin * K / s)
    bias = torch.randn(Cout, generator=gen)
    y = F.conv_transpose1d(F.leaky_relu(x, 0.15), w.double(), bias.double(), s, p, op)
    Lo = y.shape[-1]
    assert Lo == L * s
    nt = -(-K // s)
    wp = packref.pack_logical(packref.logical_convT_poly(w.numpy(), s), 32)
    NQ = (Lo - 1 + p) // s + 1
    desc = base_desc(B, Cin, 0, L, 1, Cin, Cout * s, nt, 1, 1, nt - 1, NQ, Cout, Lo, 32, shuf_S=s, shuf_P=p,
                     pre_mode=1, pre_slope=0.15)
    out = run_conv(desc, x.detach().float(), wp, bias=bias, out_shape=(B, Cout, Lo))
    np.testing.assert_allclose(out.numpy(), y.detach().float().numpy(), rtol=1e-4, atol=2e-5)
    # backward-data of the transposed conv = strided conv of dy, times lrelu'(x)
    dy = torch.randn(y.shape, generator=gen)
    y.backward(dy.double())
    wp2 = packref.pack_logical(packref.logical_convT_dgrad(w.numpy()), 32)
    desc2 = base_desc(B, Cout, 0, Lo, 1, Cout, Cin, K, s, 1, p, L, Cin, L, 32, mask_slope=0.15)
    dx = run_conv(desc2, dy, wp2, mask=x.detach().float(), out_shape=(B, Cin, L))
    np.testing.assert_allclose(dx.numpy(), x.grad.float().numpy(), rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize('case', [(2, 32, 16, 32, 8192, 7), (2, 128, 64, 128, 256, 7), (2, 80, 128, 256, 32, 7)])
def test_dgrad_split_store_for_concat_inputs(case):
    """backward-data of conv(cat([x1, x2])) written straight into the two input gradients (out_split)."""
    B, C1, C2, Cout, L, K = case
    gen = torch.Generator().manual_seed(23)
    x1 = torch.randn(B, C1, L, generator=gen, dtype=torch.float64, requires_grad=True)
    x2 = torch.randn(B, C2, L, generator=gen, dtype=torch.float64, requires_grad=True)
    w = torch.randn(Cout, C1 + C2, K, generator=gen) / np.sqrt((C1 + C2) * K)
    y = F.conv1d(torch.cat([x1, x2], 1), w.double(), None, 1, K // 2)
    dy = torch.randn(y.shape, generator=gen)
    y.backward(dy.double())
    wp = packref.pack_logical(packref.logical_dgrad_s1(w.numpy(), 1), 32)
    from rtg.lib import lib, Conv1dDesc, check
    d = Conv1dDesc(**base_desc(B, Cout, 0, L, 1, Cout, C1 + C2, K, 1, 1, (K - 1) - K // 2, L, C1 + C2, L, 32,
                               out_split=C1))
    dyd, wpd = dy.cuda(), torch.from_numpy(wp).cuda()
    o1 = torch.full((B, C1, L), float('nan'), device='cuda')
    o2 = torch.full((B, C2, L), float('nan'), device='cuda')
    check(lib.rtg_conv1d(C.byref(d), _ptr(dyd), None, None, _ptr(wpd), None, None, None, _ptr(o1), _ptr(o2), None))
    torch.cuda.synchronize()
    np.testing.assert_allclose(o1.cpu().numpy(), x1.grad.float().numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(o2.cpu().numpy(), x2.grad.float().numpy(), rtol=1e-4, atol=2e-5)


def test_pre_modes_on_dy():
    """dy * lrelu'(out) and dy * (1 - out^2) applied while staging (post-activation layers' backward)."""
    gen = torch.Generator().manual_seed(19)
    B, Cc, L, K = 2, 32, 300, 3
    dy, outp = torch.randn(B, Cc, L, generator=gen), torch.randn(B, Cc, L, generator=gen).tanh()
    w = torch.randn(Cc, Cc, K, generator=gen) / 10
    wp = packref.pack_logical(packref.logical_fwd(w.numpy(), 1), 32)
    for mode, eff in ((2, dy * torch.where(outp > 0, 1.0, 0.15)), (3, dy * (1 - outp * outp))):
        ref = F.conv1d(eff.double(), w.double(), None, 1, 1).float()
        desc = base_desc(B, Cc, 0, L, 1, Cc, Cc, K, 1, 1, 1, L, Cc, L, 32, pre_mode=mode, pre_slope=0.15)
        out = run_conv(desc, dy, wp, aux=outp, out_shape=(B, Cc, L))
        np.testing.assert_allclose(out.numpy(), ref.numpy(), rtol=1e-4, atol=2e-5)


def test_bad_descriptor_is_refused():
    from rtg.lib import lib, Conv1dDesc
    d = Conv1dDesc(**base_desc(1, 8, 0, 16, 1, 7, 8, 3, 1, 1, 1, 16, 8, 16, 32))   # C1 != groups*Cg
    x = torch.zeros(1, 8, 16, device='cuda')
    assert lib.rtg_conv1d(C.byref(d), _ptr(x), None, None, _ptr(x), None, None, None, _ptr(x), None, None) == -1
    d = Conv1dDesc(**base_desc(1, 8, 0, 16, 1, 8, 8, 3, 1, 1, 1, 16, 8, 16, 32))
    assert lib.rtg_conv1d(C.byref(d), None, None, None, _ptr(x), None, None, None, _ptr(x), None, None) == -3


@pytest.mark.parametrize('case', [FWD_CASES[1], FWD_CASES[5], FWD_CASES[11], FWD_CASES[14], (40, 64, 96, 15, 5, 1, 1, 2, 1, 32)])
def test_every_block_shape_gives_identical_bits(case):
    """RtgConv1dDesc.tile_cfg: all candidates listed by rtg_conv1d_tile_candidates compute the same bits (the tuner in
    rtg/tune.py relies on it), and an unlisted code is refused."""
    from rtg.lib import lib, Conv1dDesc
    B, Cin, Cout, L, K, s, d, p, g, TM = case
    gen = torch.Generator().manual_seed(7)
    x = torch.randn(B, Cin, L, generator=gen)
    w = torch.randn(Cout, Cin // g, K, generator=gen) / np.sqrt(Cin // g * K)
    bias = torch.randn(Cout, generator=gen)
    L_out = (L + 2 * p - d * (K - 1) - 1) // s + 1
    wp = packref.pack_logical(packref.logical_fwd(w.numpy(), g), TM)
    desc = base_desc(B, Cin, 0, L, g, Cin // g, Cout // g, K, s, d, p, L_out, Cout, L_out, TM, pre_mode=1, pre_slope=0.15)
    cands = (C.c_int * 16)()
    n = lib.rtg_conv1d_tile_candidates(C.byref(Conv1dDesc(**desc)), cands, 16)
    assert n >= 2
    ref = run_conv(desc, x, wp, bias=bias, out_shape=(B, Cout, L_out))
    assert torch.isfinite(ref).all()
    for c in cands[:n]:
        out = run_conv(dict(desc, tile_cfg=c), x, wp, bias=bias, out_shape=(B, Cout, L_out))
        assert torch.equal(out, ref), f'tile_cfg {c} differs'
    bad = Conv1dDesc(**dict(desc, tile_cfg=999))
    assert lib.rtg_conv1d_variant(C.byref(bad)) < 0


# (round 4) conv_post on short rows: cout1_k3_rows_kernel — rows of 4 .. 64 positions (64 / L channel rows per wave-wide
# load, the last load of a wave partial), 8 .. 1024 channels, one clip and many
K3_ROWS_CASES = [(704, 512, 1, 10, 3, 1, 1, 1, 1, 16), (7, 512, 1, 15, 3, 1, 1, 1, 1, 32), (5, 1024, 1, 21, 3, 1, 1, 1, 1, 16),
                 (3, 128, 1, 64, 3, 1, 1, 1, 1, 16), (1, 512, 1, 32, 3, 1, 1, 1, 1, 16), (9, 256, 1, 4, 3, 1, 1, 1, 1, 16),
                 (2, 512, 1, 33, 3, 1, 1, 1, 1, 16), (4, 384, 1, 63, 3, 1, 1, 1, 1, 16),
                 # one INPUT channel with compile-time taps / stride (cin1_flat_kernel<KT, ST>): conv_post backward-data
                 # (1 -> 512, k3), the first layer of the period discriminators (1 -> 32, k5, stride 3), ragged rows
                 (37, 1, 512, 10, 3, 1, 1, 1, 1, 32), (5, 1, 512, 34, 3, 1, 1, 1, 1, 32), (3, 1, 512, 127, 3, 1, 1, 1, 1, 32),
                 (9, 1, 32, 2731, 5, 3, 1, 2, 1, 32), (4, 1, 32, 1171, 5, 3, 1, 2, 1, 32), (2, 1, 64, 50, 5, 1, 1, 2, 1, 32)]


@pytest.mark.parametrize('case', [FWD_CASES[7], FWD_CASES[8], FWD_CASES[9], FWD_CASES[15], (40, 1, 32, 911, 5, 3, 1, 2, 1, 32)] + K3_ROWS_CASES)
def test_thin_kernels_match_the_mfma_path(case):
    """One-input-channel / one-output-channel shapes run on the bandwidth kernels of rtg_thin.hip (variant 1 / 2) and
    agree with the MFMA kernel forced through tile_cfg, including mask and residual operands."""
    from rtg.lib import lib, Conv1dDesc
    B, Cin, Cout, L, K, s, d, p, g, TM = case
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(B, Cin, L, generator=gen)
    w = torch.randn(Cout, Cin // g, K, generator=gen) / np.sqrt(Cin // g * K)
    bias = torch.randn(Cout, generator=gen)
    L_out = (L + 2 * p - d * (K - 1) - 1) // s + 1
    mask, res = torch.randn(B, Cout, L_out, generator=gen), torch.randn(B, Cout, L_out, generator=gen)
    wp = packref.pack_logical(packref.logical_fwd(w.numpy(), g), TM)
    desc = base_desc(B, Cin, 0, L, g, Cin // g, Cout // g, K, s, d, p, L_out, Cout, L_out, TM, pre_mode=1, pre_slope=0.15,
                     mask_slope=0.2, out_scale=0.7, act=1, act_slope=0.1)
    assert lib.rtg_conv1d_variant(C.byref(Conv1dDesc(**desc))) in (1, 2)
    thin = run_conv(desc, x, wp, bias=bias, mask=mask, res=res, out_shape=(B, Cout, L_out))
    mfma = run_conv(dict(desc, tile_cfg=111), x, wp, bias=bias, mask=mask, res=res, out_shape=(B, Cout, L_out))
    assert lib.rtg_conv1d_variant(C.byref(Conv1dDesc(**dict(desc, tile_cfg=111)))) >= 100
    np.testing.assert_allclose(thin.numpy(), mfma.numpy(), rtol=2e-5, atol=2e-5)


PACKED_CASES = [
    # B, C_in, C_out, L, K, stride, dil, pad, groups, tile_m — rows much shorter than a block tile: clips are packed
    (37, 64, 64, 10, 5, 1, 1, 2, 1, 32),         # MPD tail, clip count not a multiple of the clips per block
    (5, 32, 64, 21, 5, 1, 1, 2, 1, 32),
    (9, 48, 32, 34, 5, 1, 1, 2, 1, 32),          # channels not a multiple of the chunk
    (7, 32, 32, 1, 3, 1, 1, 1, 1, 32),           # one position per clip
    (11, 32, 32, 2, 5, 1, 1, 2, 1, 32),          # rows shorter than the kernel
    (6, 32, 32, 16, 3, 1, 9, 9, 1, 32),          # dilation wider than the row
    (13, 32, 64, 61, 5, 3, 1, 2, 1, 32),         # strided, packed (phase-de-interleaved patch)
    (10, 32, 64, 28, 5, 3, 1, 2, 1, 32),
    (12, 64, 64, 40, 41, 4, 1, 20, 8, 16),       # grouped k41 s4 (tile_m 16), short rows
    (3, 32, 32, 63, 7, 1, 1, 3, 1, 32),          # just under half / a full tile
    (3, 32, 32, 65, 7, 1, 1, 3, 1, 32),
    (2, 32, 32, 127, 3, 1, 1, 1, 1, 32),
]


@pytest.mark.parametrize('case', PACKED_CASES)
def test_packed_short_clips_every_block_shape(case):
    """Packed clips (rtg_conv1d_kernel.h: columns enumerate (clip, q) densely, the staged patch keeps a halo gap per
    clip): every block shape against torch, with bias, mask, residual and activation in the epilogue."""
    from rtg.lib import lib, Conv1dDesc
    B, Cin, Cout, L, K, s, d, p, g, TM = case
    gen = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn(B, Cin, L, generator=gen)
    w = torch.randn(Cout, Cin // g, K, generator=gen) / np.sqrt(Cin // g * K)
    bias = torch.randn(Cout, generator=gen)
    ref = F.conv1d(F.leaky_relu(x, 0.15).double(), w.double(), bias.double(), s, p, d, g)
    L_out = ref.shape[-1]
    res = torch.randn(B, Cout, L_out, generator=gen)
    mask = torch.randn(B, Cout, L_out, generator=gen)
    ref = ref * torch.where(mask > 0, 1.0, 0.15).double()
    ref = F.leaky_relu((ref + res.double()) * 0.5, 0.01).float()
    wp = packref.pack_logical(packref.logical_fwd(w.numpy(), g), TM)
    desc = base_desc(B, Cin, 0, L, g, Cin // g, Cout // g, K, s, d, p, L_out, Cout, L_out, TM,
                     pre_mode=1, pre_slope=0.15, mask_slope=0.15, out_scale=0.5, act=1, act_slope=0.01)
    cands = (C.c_int * 16)()
    n = lib.rtg_conv1d_tile_candidates(C.byref(Conv1dDesc(**desc)), cands, 16)
    assert n >= 1
    first = None
    for c in cands[:n]:
        out = run_conv(dict(desc, tile_cfg=c), x, wp, bias=bias, mask=mask, res=res, out_shape=(B, Cout, L_out))
        np.testing.assert_allclose(out.numpy(), ref.numpy(), rtol=1e-4, atol=2e-5, err_msg=f'tile_cfg {c}')
        if first is None:
            first = out
        assert torch.equal(out, first), f'tile_cfg {c}: bits differ between block shapes'


@pytest.mark.parametrize('case', [(13, 32, 64, 61, 5, 3, 2, 1, 32), (37, 64, 64, 30, 5, 3, 2, 1, 32),
                                  (9, 64, 128, 40, 41, 4, 20, 8, 16), (5, 16, 32, 17, 7, 4, 3, 1, 32)])
def test_packed_polyphase_dgrad_every_block_shape(case):
    """Backward-data of strided convs on short rows: packed clips + polyphase shuffle store + lrelu' mask through the
    descriptor-based epilogue, every block shape."""
    from rtg.lib import lib, Conv1dDesc
    B, Cin, Cout, L, K, s, p, g, TM = case
    gen = torch.Generator().manual_seed(29)
    x = torch.randn(B, Cin, L, generator=gen, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(Cout, Cin // g, K, generator=gen) / np.sqrt(Cin // g * K))
    y = F.conv1d(F.leaky_relu(x, 0.15), w.double(), None, s, p, 1, g)
    dy = torch.randn(y.shape, generator=gen)
    y.backward(dy.double())
    Lo = y.shape[-1]
    nt = -(-K // s)
    wp = packref.pack_logical(packref.logical_dgrad_poly(w.numpy(), g, s), TM)
    NQ = (L - 1 + p) // s + 1
    desc = base_desc(B, Cout, 0, Lo, g, Cout // g, (Cin // g) * s, nt, 1, 1, nt - 1, NQ, Cin, L, TM,
                     shuf_S=s, shuf_P=p, mask_slope=0.15)
    mask = x.detach().float()                       # d lrelu(x) / dx = 1 for x > 0, 0.15 otherwise
    cands = (C.c_int * 16)()
    n = lib.rtg_conv1d_tile_candidates(C.byref(Conv1dDesc(**desc)), cands, 16)
    assert n >= 1
    for c in cands[:n]:
        out = run_conv(dict(desc, tile_cfg=c), dy, wp, mask=mask, out_shape=(B, Cin, L))
        np.testing.assert_allclose(out.numpy(), x.grad.float().numpy(), rtol=1e-4, atol=2e-5, err_msg=f'tile_cfg {c}')


RESCONV_CASES = [
    # B, C, L, K, dil — the stride-1 "same" convolutions of ResBlock3 / ResidualStack (generator.py:33-77,133-155)
    (3, 32, 1000, 7, 9),         # ragged length: the last tile of a clip is partial
    (2, 32, 8192, 3, 1),
    (2, 32, 2048, 5, 3),
    (2, 64, 2048, 5, 3),
    (2, 64, 260, 7, 9),
    (5, 64, 256, 3, 9),
    (7, 32, 36, 3, 3),           # rows shorter than a tile
]


@pytest.mark.parametrize('mode', ['resblock_fwd', 'resblock_dgrad', 'stack_fwd_act', 'mask_and_res'])
@pytest.mark.parametrize('case', RESCONV_CASES)
def test_resconv_kernel_matches_general_kernel_bitwise_and_torch(case, mode):
    """rtg_resconv.hip (block-shape codes 7001 / 7002: weights in registers, double-buffered raw window, activation on
    the read side, residual from LDS) against torch in float64 and bit for bit against the general MFMA kernel: same
    accumulation order, same epilogue arithmetic."""
    from rtg.lib import lib, Conv1dDesc, check
    B, Cc, L, K, d = case
    p = (K * d - d) // 2
    gen = torch.Generator().manual_seed(L + K)
    x = torch.randn(B, Cc, L, generator=gen)
    w = torch.randn(Cc, Cc, K, generator=gen) / np.sqrt(Cc * K)
    bias = torch.randn(Cc, generator=gen)
    other = torch.randn(B, Cc, L, generator=gen)
    msk = torch.randn(B, Cc, L, generator=gen)
    dev = 'cuda'
    xd, od, md, bd = x.to(dev), other.to(dev), msk.to(dev), bias.to(dev)
    xw, ww = x.double(), w.double()
    if mode == 'resblock_fwd':
        ref = F.conv1d(F.leaky_relu(xw, 0.15), ww, bias.double(), 1, p, d) + xw
        wp = packref.pack_logical(packref.logical_fwd(w.numpy(), 1), 32)
        kw = dict(pre_mode=1, pre_slope=0.15)
        ptrs = dict(bias=bd, mask=None, res=xd)
    elif mode == 'resblock_dgrad':
        # x plays dy; `msk` the forward input whose leaky-relu derivative masks the conv branch; residual gradient = dy
        gin = torch.nn.grad.conv1d_input(xw.shape, ww, xw, 1, p, d)
        ref = gin * torch.where(msk.double() > 0, 1.0, 0.15) + xw
        wp = packref.pack_logical(packref.logical_dgrad_s1(w.numpy(), 1), 32)
        kw = dict(mask_slope=0.15)
        ptrs = dict(bias=None, mask=md, res=xd)
    elif mode == 'stack_fwd_act':
        ref = F.leaky_relu(F.conv1d(F.leaky_relu(xw, 0.01), ww, bias.double(), 1, p, d) + other.double(), 0.15)
        wp = packref.pack_logical(packref.logical_fwd(w.numpy(), 1), 32)
        kw = dict(pre_mode=1, pre_slope=0.01, act=1, act_slope=0.15)
        ptrs = dict(bias=bd, mask=None, res=od)
    else:
        ref = (F.conv1d(xw, ww, bias.double(), 1, p, d) * torch.where(msk.double() > 0, 1.0, 0.3) + other.double()) * 0.5
        wp = packref.pack_logical(packref.logical_fwd(w.numpy(), 1), 32)
        kw = dict(mask_slope=0.3, out_scale=0.5)
        ptrs = dict(bias=bd, mask=md, res=od)
    pad = p if mode != 'resblock_dgrad' else (K - 1) * d - p
    desc = base_desc(B, Cc, 0, L, 1, Cc, Cc, K, 1, d, pad, L, Cc, L, 32, **kw)
    cands = (C.c_int * 32)()
    n = lib.rtg_conv1d_tile_candidates(C.byref(Conv1dDesc(**desc)), cands, 32)
    codes = [c for c in cands[:n] if c > 7000]
    assert codes == ([7002, 7001] if Cc == 32 else [7001]), list(cands[:n])
    wp_d = torch.from_numpy(wp).to(dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    outs = {}
    for code in [0] + codes:
        dd = Conv1dDesc(**dict(desc, tile_cfg=code))
        assert lib.rtg_conv1d_variant(C.byref(dd)) == (code if code else lib.rtg_conv1d_variant(C.byref(dd)))
        out = torch.full((B, Cc, L), float('nan'), device=dev)
        check(lib.rtg_conv1d(C.byref(dd), _ptr(xd), None, None, _ptr(wp_d), _ptr(ptrs['bias']), _ptr(ptrs['mask']),
                             _ptr(ptrs['res']), _ptr(out), None, st), f'rtg_conv1d cfg {code}')
        torch.cuda.synchronize()
        outs[code] = out.cpu()
    np.testing.assert_allclose(outs[0].numpy(), ref.float().numpy(), rtol=1e-4, atol=2e-5)
    for code in codes:
        assert torch.equal(outs[code], outs[0]), (code, (outs[code] - outs[0]).abs().max().item())


# ---------------------------------------------------------------------------------------------------------------
# rtg_sconv.hip (round 4): stride-1 "same" convs over few columns with split-K over the waves of a block (codes 9004 / 9008)
# ---------------------------------------------------------------------------------------------------------------
SCONV_CASES = [
    # B, C1, C2, C_out, L, K, dil, out_split, extras
    (32, 80, 128, 256, 32, 7, 1, 0, 'bias'),                  # conv_fuse forward: [mel | encoder] concatenated input
    (32, 256, 0, 208, 32, 7, 1, 80, ''),                      # its backward-data: 208 rows split into the two inputs
    (32, 128, 0, 128, 32, 3, 9, 0, 'pre bias res act'),       # ResidualStack conv, dilation 9 (wider than half the row)
    (32, 128, 0, 128, 32, 3, 3, 0, 'bias mask res'),          # backward-data shaped: mask and residual gradient
    (5, 128, 0, 144, 40, 5, 3, 0, 'bias res acc'),            # ragged: clips straddle the 64-column tiles, accumulate
    (3, 64, 64, 128, 64, 8, 1, 0, 'pre bias'),                # 8 taps, rows of 64, two 64-channel inputs
]


@pytest.mark.parametrize('case', SCONV_CASES)
def test_sconv_split_k_matches_the_general_kernel(case):
    from rtg.lib import lib, Conv1dDesc, check
    B, C1, C2, Cout, L, K, dil, split, extras = case
    Cin = C1 + C2
    pad = dil * (K - 1) // 2
    pad_r = dil * (K - 1) - pad                                # ("same": left pad, the right one follows from Q = L)
    gen = torch.Generator().manual_seed(C1 + K + L)
    x = torch.randn(B, Cin, L, generator=gen)
    w = torch.randn(Cout, Cin, K, generator=gen) / np.sqrt(Cin * K)
    bias = torch.randn(Cout, generator=gen) if 'bias' in extras else None
    mask = torch.randn(B, Cout, L, generator=gen) if 'mask' in extras else None
    res = torch.randn(B, Cout, L, generator=gen) if 'res' in extras else None
    W = packref.logical_fwd(w.numpy(), 1)
    wp = np.concatenate([packref.pack_logical(W, 32), packref.pack_frag16(W)])
    kw = dict(pre_mode=1, pre_slope=0.01) if 'pre' in extras else {}
    if 'act' in extras:
        kw.update(act=1, act_slope=0.2, out_scale=0.5)
    if mask is not None:
        kw.update(mask_slope=0.15)
    desc = base_desc(B, C1, C2, L, 1, Cin, Cout, K, 1, dil, pad, L, Cout, L, 32, out_split=split, wp16=2,
                     accumulate=int('acc' in extras), **kw)
    cands = (C.c_int * 48)()
    n = lib.rtg_conv1d_tile_candidates(C.byref(Conv1dDesc(**desc)), cands, 48)
    codes = [c for c in cands[:n] if c > 9000]
    assert codes and set(codes) <= {9004, 9008}, list(cands[:n])
    dev = 'cuda'
    xd = x.to(dev)
    x1, x2 = (xd[:, :C1].contiguous(), xd[:, C1:].contiguous()) if C2 else (xd, None)
    wpd = torch.from_numpy(wp).to(dev)
    t = {k_: (v.to(dev) if v is not None else None) for k_, v in dict(bias=bias, mask=mask, res=res).items()}
    init = torch.randn(B, Cout, L, generator=gen).to(dev) if 'acc' in extras else None
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run(code):
        d = Conv1dDesc(**dict(desc, tile_cfg=code))
        if split:
            o1 = torch.full((B, split, L), float('nan'), device=dev)
            o2 = torch.full((B, Cout - split, L), float('nan'), device=dev)
        else:
            o1 = init.clone() if init is not None else torch.full((B, Cout, L), float('nan'), device=dev)
            o2 = None
        check(lib.rtg_conv1d(C.byref(d), _ptr(x1), _ptr(x2), None, _ptr(wpd), _ptr(t['bias']), _ptr(t['mask']), _ptr(t['res']),
                             _ptr(o1), _ptr(o2), st), f'rtg_conv1d code {code}')
        torch.cuda.synchronize()
        return torch.cat([o1, o2], dim=1).cpu() if split else o1.cpu()

    ref = run(0)                                               # the library's heuristic: a general block shape
    assert torch.isfinite(ref).all()
    # (independent check of the reference itself against torch, fp64)
    xa = F.leaky_relu(x.double(), 0.01) if 'pre' in extras else x.double()
    tr = F.conv1d(F.pad(xa, (pad, pad_r)), w.double(), bias.double() if bias is not None else None, dilation=dil)
    if mask is not None:
        tr = tr * torch.where(mask.double() > 0, 1.0, 0.15)
    if res is not None:
        tr = tr + res.double()
    if 'act' in extras:
        tr = F.leaky_relu(tr * 0.5, 0.2)
    if init is not None:
        tr = tr + init.cpu().double()
    assert (ref.double() - tr).abs().max().item() < 2e-5 * max(1.0, tr.abs().max().item())
    for code in codes:
        got = run(code)
        assert torch.isfinite(got).all(), code
        err = (got.double() - tr).abs().max().item()
        assert err < 2e-5 * max(1.0, tr.abs().max().item()), (code, err)
        np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=2e-5, atol=2e-5)


def _sconv_codes_and_runs(desc, ins, wp_std, wp_frag, out_shape):
    """the general kernel (heuristic block shape) and every split-K code of rtg_sconv.hip on one problem: {code: output}"""
    from rtg.lib import lib, Conv1dDesc
    wp = np.concatenate([wp_std, wp_frag])
    desc = dict(desc, wp16=2)
    cands = (C.c_int * 48)()
    n = lib.rtg_conv1d_tile_candidates(C.byref(Conv1dDesc(**desc)), cands, 48)
    codes = [c for c in cands[:n] if c > 9000]
    assert codes and set(codes) <= {9004, 9008}, list(cands[:n])
    outs = {0: run_conv(dict(desc, tile_cfg=0), wp=wp, out_shape=out_shape, **ins)}
    for c in codes:
        outs[c] = run_conv(dict(desc, tile_cfg=c), wp=wp, out_shape=out_shape, **ins)
    return outs


SCONV_STRIDED = [
    # B, C_in, C_out, L, K, stride, pad
    (32, 64, 128, 256, 15, 8, 7),          # downs.2 (generator.py:700-703, hparam.py:61-62) at batch 32: 32 columns per clip
    (3, 64, 128, 256, 15, 8, 7),           # ragged: 96 columns, three 32-column tiles
    (5, 128, 256, 128, 8, 4, 4),           # another geometry: stride 4, 8 taps
    (4, 64, 144, 61 * 3, 5, 3, 2),         # stride 3, rows of 61 columns (tiles straddle clips), 144 rows
]


@pytest.mark.parametrize('case', SCONV_STRIDED)
def test_sconv_strided_forward_and_polyphase_backward_data(case):
    """rtg_sconv.hip on a strided conv of the UNet bottom: the strided walk forward (with bias, leaky-relu on the input) and
    its backward-data as the polyphase operator with the shuffle store, against fp64 torch and the general kernel"""
    B, Cin, Cout, L, K, s, p = case
    gen = torch.Generator().manual_seed(K + s + B)
    x = torch.randn(B, Cin, L, generator=gen, dtype=torch.float64, requires_grad=True)
    w = torch.randn(Cout, Cin, K, generator=gen) / np.sqrt(Cin * K)
    bias = torch.randn(Cout, generator=gen)
    y = F.conv1d(F.leaky_relu(x, 0.15), w.double(), bias.double(), s, p)
    Lo = y.shape[-1]
    W = packref.logical_fwd(w.numpy(), 1)
    desc = base_desc(B, Cin, 0, L, 1, Cin, Cout, K, s, 1, p, Lo, Cout, Lo, 32, pre_mode=1, pre_slope=0.15)
    outs = _sconv_codes_and_runs(desc, dict(x1=x.detach().float(), bias=bias), packref.pack_logical(W, 32),
                                 packref.pack_frag16(W), (B, Cout, Lo))
    ref = y.detach()
    for c, o in outs.items():
        assert torch.isfinite(o).all(), c
        err = (o.double() - ref).abs().max().item()
        assert err < 2e-5 * max(1.0, ref.abs().max().item()), (c, err)
    # backward-data: rows = (input channel, phase), 2 taps, shuffle store; times lrelu'(x) (the mask operand)
    dy = torch.randn(y.shape, generator=gen)
    y.backward(dy.double())
    nt = -(-K // s)
    Wd = packref.logical_dgrad_poly(w.numpy(), 1, s)
    NQ = (L - 1 + p) // s + 1
    desc2 = base_desc(B, Cout, 0, Lo, 1, Cout, Cin * s, nt, 1, 1, nt - 1, NQ, Cin, L, 32, shuf_S=s, shuf_P=p, mask_slope=0.15)
    outs = _sconv_codes_and_runs(desc2, dict(x1=dy, mask=x.detach().float()), packref.pack_logical(Wd, 32),
                                 packref.pack_frag16(Wd), (B, Cin, L))
    ref = x.grad
    for c, o in outs.items():
        assert torch.isfinite(o).all(), c
        err = (o.double() - ref).abs().max().item()
        assert err < 2e-5 * max(1.0, ref.abs().max().item()), (c, err)


SCONV_CONVT = [
    # B, C_in, C_out, L, K, stride, pad, out_pad
    (32, 256, 128, 32, 15, 8, 7, 7),       # ups.0 (generator.py:713-716, hparam.py:61-62) at batch 32
    (3, 256, 128, 32, 15, 8, 7, 7),
    (4, 128, 64, 40, 8, 4, 4, 3),
]


@pytest.mark.parametrize('case', SCONV_CONVT)
def test_sconv_transposed_conv_forward_and_backward_data(case):
    """rtg_sconv.hip on a transposed conv of the UNet bottom: the polyphase operator forward (bias indexed by the output
    channel, leaky-relu on the input), the strided conv of dy backward, against fp64 torch and the general kernel"""
    B, Cin, Cout, L, K, s, p, op = case
    gen = torch.Generator().manual_seed(K + s + B + 1)
    x = torch.randn(B, Cin, L, generator=gen, dtype=torch.float64, requires_grad=True)
    w = torch.randn(Cin, Cout, K, generator=gen) / np.sqrt(Cin * K / s)
    bias = torch.randn(Cout, generator=gen)
    y = F.conv_transpose1d(F.leaky_relu(x, 0.15), w.double(), bias.double(), s, p, op)
    Lo = y.shape[-1]
    nt = -(-K // s)
    W = packref.logical_convT_poly(w.numpy(), s)
    NQ = (Lo - 1 + p) // s + 1
    desc = base_desc(B, Cin, 0, L, 1, Cin, Cout * s, nt, 1, 1, nt - 1, NQ, Cout, Lo, 32, shuf_S=s, shuf_P=p,
                     pre_mode=1, pre_slope=0.15)
    outs = _sconv_codes_and_runs(desc, dict(x1=x.detach().float(), bias=bias), packref.pack_logical(W, 32),
                                 packref.pack_frag16(W), (B, Cout, Lo))
    ref = y.detach()
    for c, o in outs.items():
        assert torch.isfinite(o).all(), c
        err = (o.double() - ref).abs().max().item()
        assert err < 2e-5 * max(1.0, ref.abs().max().item()), (c, err)
    dy = torch.randn(y.shape, generator=gen)
    y.backward(dy.double())
    Wd = packref.logical_convT_dgrad(w.numpy())
    desc2 = base_desc(B, Cout, 0, Lo, 1, Cout, Cin, K, s, 1, p, L, Cin, L, 32, mask_slope=0.15)
    outs = _sconv_codes_and_runs(desc2, dict(x1=dy, mask=x.detach().float()), packref.pack_logical(Wd, 32),
                                 packref.pack_frag16(Wd), (B, Cin, L))
    ref = x.grad
    for c, o in outs.items():
        assert torch.isfinite(o).all(), c
        err = (o.double() - ref).abs().max().item()
        assert err < 2e-5 * max(1.0, ref.abs().max().item()), (c, err)


def test_sconv_is_offered_only_where_the_owner_allows_its_summation_order():
    """RtgConv1dDesc.wp16 == 1 promises that every listed block shape gives the general kernel's bits (the discriminators'
    grouped-launch / per-clip tests rely on it): the split-K codes 9004 / 9008 are listed and accepted with wp16 == 2 only."""
    from rtg.lib import lib, Conv1dDesc
    B, C_, L, K = 4, 128, 32, 3
    base = base_desc(B, C_, 0, L, 1, C_, C_, K, 1, 1, 1, L, C_, L, 32)
    cands = (C.c_int * 48)()
    for wp16, expect in ((0, False), (1, False), (2, True)):
        n = lib.rtg_conv1d_tile_candidates(C.byref(Conv1dDesc(**dict(base, wp16=wp16))), cands, 48)
        assert any(c > 9000 for c in cands[:n]) == expect, (wp16, list(cands[:n]))
    W = packref.logical_fwd(np.zeros((C_, C_, K), dtype=np.float32), 1)
    wp = torch.from_numpy(np.concatenate([packref.pack_logical(W, 32), packref.pack_frag16(W)])).cuda()
    x, out = torch.zeros(B, C_, L, device='cuda'), torch.zeros(B, C_, L, device='cuda')
    d = Conv1dDesc(**dict(base, wp16=1, tile_cfg=9008))
    rc = lib.rtg_conv1d(C.byref(d), _ptr(x), None, None, _ptr(wp), None, None, None, _ptr(out), None, None)
    assert rc != 0
