"""rtg_conv1d_wgrad + rtg_weightnorm_backward (HIP) against torch autograd on the CPU, through the C ABI.  GPU only."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


CASES = [
    # B, C_in, C_out, L, K, stride, dil, pad, groups
    (2, 32, 32, 1000, 7, 1, 9, 27, 1),
    (3, 64, 64, 300, 3, 1, 3, 3, 1),
    (2, 128, 128, 32, 3, 1, 9, 9, 1),        # dilation wider than the row
    (2, 16, 32, 2048, 7, 4, 1, 3, 1),
    (2, 64, 128, 256, 15, 8, 1, 7, 1),
    (2, 1, 16, 1024, 7, 1, 1, 3, 1),
    (2, 32, 1, 1024, 7, 1, 1, 3, 1),
    (2, 1, 32, 1024, 15, 1, 1, 7, 1),
    (2, 32, 64, 1024, 41, 2, 1, 20, 4),
    (2, 128, 512, 512, 41, 4, 1, 20, 32),
    (2, 512, 512, 128, 41, 4, 1, 20, 64),
    (22, 512, 512, 10, 5, 1, 1, 2, 1),       # MPD tail: 10 positions per clip -> clips packed per tile
    (14, 256, 512, 44, 5, 3, 1, 2, 1),       # MPD strided, short rows
    (10, 512, 1, 21, 3, 1, 1, 1, 1),
    (5, 48, 32, 300, 7, 1, 1, 3, 1),
]


@pytest.mark.parametrize('case', CASES)
def test_wgrad_and_weightnorm_backward(case):
    _check_case(case)


THIN_CASES = [
    (2, 1, 16, 1024, 7, 1, 1, 3, 1),         # conv_pre
    (3, 1, 32, 5000, 15, 1, 1, 7, 1),        # MSD conv0, several tiles per clip, ragged
    (6, 1, 32, 911, 5, 3, 1, 2, 1),          # MPD conv0: stride 3, period folded into the batch
    (2, 32, 1, 8192, 7, 1, 1, 3, 1),         # G conv_post
    (5, 32, 1, 1500, 7, 1, 1, 3, 1),
    (10, 512, 1, 21, 3, 1, 1, 1, 1),         # D conv_post, short rows
    (7, 512, 1, 128, 3, 1, 1, 1, 1),
    (3, 300, 1, 10, 3, 1, 1, 1, 1),          # channels not a multiple of the block
]


@pytest.mark.parametrize('act', ['lrelu', 'tanh', 'none'])
@pytest.mark.parametrize('case', THIN_CASES)
def test_thin_wgrad_kernels(case, act):
    """rtg_wgrad_thin.hip (shape code 7): weight / bias gradients of the one-input-channel and one-output-channel layers
    on the bandwidth kernels, through the weight-norm chain, against torch autograd; listed as a candidate for exactly
    these shapes."""
    from rtg.lib import lib, WgradDesc
    B, Cin, Cout, L, K, s, d, p, g = case
    Lo = (L + 2 * p - d * (K - 1) - 1) // s + 1
    probe = WgradDesc(B=B, C1=Cin, C2=0, L_in=L, groups=g, Cg=Cin // g, Mg=Cout // g, K=K, stride=s, dil=d, pad=p, Q=Lo,
                      dy_L=Lo, pre_mode=1, pre_slope=0.15, gy_mode=0, gy_slope=1.0, gy_scale=1.0, splits=1, part_stride=0)
    cands = (C.c_int * 8)()
    n = lib.rtg_wgrad_shape_candidates(C.byref(probe), cands, 8)
    assert 7 in list(cands[:n])
    _check_case(case, shape_cfg=7, act=act)


RES_CASES = [
    (3, 32, 32, 1000, 7, 1, 9, 27, 1),       # ragged: the last tile of a clip is partial
    (2, 32, 32, 8192, 3, 1, 1, 1, 1),
    (2, 32, 32, 2048, 5, 1, 3, 6, 1),
    (2, 64, 64, 2048, 5, 1, 3, 6, 1),
    (5, 64, 64, 260, 5, 1, 3, 6, 1),
    (9, 64, 64, 256, 3, 1, 9, 9, 1),
    (300, 32, 32, 128, 3, 1, 1, 1, 1),       # more tiles than blocks: several tiles per block
]


@pytest.mark.parametrize('case', RES_CASES)
def test_reswgrad_kernel(case):
    """rtg_reswgrad.hip (shape code 8): the weight / bias gradients of the stride-1 'same' convs of ResBlock3 /
    ResidualStack as a streaming reduction, through the weight-norm chain, against torch autograd."""
    from rtg.lib import lib, WgradDesc
    B, Cin, Cout, L, K, s, d, p, g = case
    probe = WgradDesc(B=B, C1=Cin, C2=0, L_in=L, groups=g, Cg=Cin // g, Mg=Cout // g, K=K, stride=s, dil=d, pad=p, Q=L,
                      dy_L=L, pre_mode=1, pre_slope=0.15, gy_mode=0, gy_slope=1.0, gy_scale=1.0, splits=1, part_stride=0)
    cands = (C.c_int * 8)()
    n = lib.rtg_wgrad_shape_candidates(C.byref(probe), cands, 8)
    assert 8 in list(cands[:n])
    _check_case(case, shape_cfg=8, act='none')


DENSE_CASES = [
    (22, 512, 512, 10, 5, 1, 1, 2, 1),       # DiscriminatorP convs.4: rows of 10, clips straddle the 64-wide reduction tiles
    (7, 512, 512, 34, 5, 1, 1, 2, 1),
    (3, 512, 512, 128, 5, 1, 1, 2, 1),       # DiscriminatorS convs.5
    (14, 256, 512, 44, 5, 3, 1, 2, 1),       # convs.3: stride 3
    (6, 128, 256, 304, 5, 3, 1, 2, 1),       # convs.2
    (5, 32, 128, 911, 5, 3, 1, 2, 1),        # convs.1: two channel chunks
    (1, 64, 128, 50, 5, 1, 1, 2, 1),         # a single partial tile
    (1, 32, 128, 10, 5, 1, 1, 2, 1),         # the 16-byte gy loads of the last row reach past the end of the tensor
    (3, 32, 128, 7, 5, 1, 1, 2, 1),          # rows shorter than two fragments: every other fragment straddles two clips
]


@pytest.mark.parametrize('case', DENSE_CASES)
def test_dense_wgrad_kernel(case):
    """rtg_dwgrad.hip (shape codes 10, 11): weight / bias gradients of the dense discriminator layers with 16-byte operand
    fragments (both tiles staged in matrix-core order, the column tile as the im2col of the reduction tile), through the
    weight-norm chain, against torch autograd; listed as a candidate for exactly these shapes."""
    from rtg.lib import lib, WgradDesc
    B, Cin, Cout, L, K, s, d, p, g = case
    Lo = (L + 2 * p - d * (K - 1) - 1) // s + 1
    probe = WgradDesc(B=B, C1=Cin, C2=0, L_in=L, groups=g, Cg=Cin // g, Mg=Cout // g, K=K, stride=s, dil=d, pad=p, Q=Lo,
                      dy_L=Lo, pre_mode=1, pre_slope=0.15, gy_mode=0, gy_slope=1.0, gy_scale=1.0, splits=1, part_stride=0)
    cands = (C.c_int * 12)()
    n = lib.rtg_wgrad_shape_candidates(C.byref(probe), cands, 12)
    codes = [c for c in cands[:n] if c >= 10]
    assert codes == [10, 11]                     # 128 rows x 1 / 2 channel chunks per block
    for c in codes:
        _check_case(case, shape_cfg=c, act='none')


def _check_case(case, shape_cfg=0, act='lrelu'):
    from rtg.lib import lib, WgradDesc, WnBwdJob, NormJob, check
    B, Cin, Cout, L, K, s, d, p, g = case
    gen = torch.Generator().manual_seed(abs(hash(case)) % (2 ** 31))
    x = torch.randn(B, Cin, L, generator=gen)
    v = torch.randn(Cout, Cin // g, K, generator=gen, dtype=torch.float64, requires_grad=True)
    gg = (1 + 0.1 * torch.randn(Cout, 1, 1, generator=gen, dtype=torch.float64)).requires_grad_(True)
    bias = torch.zeros(Cout, dtype=torch.float64, requires_grad=True)
    w = gg * v / v.flatten(1).norm(dim=1).reshape(-1, 1, 1)
    pre = F.conv1d(F.leaky_relu(x.double(), 0.15), w, bias, s, p, d, g)
    y = F.leaky_relu(pre, 0.2) if act == 'lrelu' else (torch.tanh(pre) if act == 'tanh' else pre)
    dy = torch.randn(y.shape, generator=gen)
    y.backward(dy.double())
    Lo = y.shape[-1]

    dev = 'cuda'
    rows, inner = Cout, (Cin // g) * K
    flat = torch.cat([gg.detach().flatten(), v.detach().flatten(), bias.detach()]).float().to(dev)
    gflat = torch.zeros_like(flat)
    scales = torch.empty(2 * rows, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def table(job):
        return torch.frombuffer(bytearray(bytes(memoryview((type(job) * 1)(job)).cast('B'))), dtype=torch.uint8).to(dev)

    check(lib.rtg_weightnorm_scales(_ptr(table(NormJob(0, rows, 0, rows, inner))), 1, rows, _ptr(flat), _ptr(scales), st))
    wd = WgradDesc(B=B, C1=Cin, C2=0, L_in=L, groups=g, Cg=Cin // g, Mg=Cout // g, K=K, stride=s, dil=d, pad=p, Q=Lo,
                   dy_L=Lo, pre_mode=1, pre_slope=0.15, gy_mode={'lrelu': 2, 'tanh': 3, 'none': 0}[act], gy_slope=0.2,
                   gy_scale=1.0, splits=1, part_stride=0, shape_cfg=shape_cfg)
    splits = lib.rtg_wgrad_splits(C.byref(wd))
    assert splits >= 1
    stride = rows * (inner + 1)
    part = torch.full((splits * stride,), float('nan'), device=dev)
    wd.splits, wd.part_stride = splits, stride
    xd, dyd, outd = x.to(dev), dy.to(dev), y.detach().float().to(dev)
    check(lib.rtg_conv1d_wgrad(C.byref(wd), _ptr(xd), None, _ptr(dyd), _ptr(outd), _ptr(part), st))
    job = WnBwdJob(0, rows, rows + rows * inner, 0, (part.data_ptr() - flat.data_ptr()) // 4, stride, splits, rows, inner)
    check(lib.rtg_weightnorm_backward(_ptr(table(job)), 1, rows, inner, _ptr(flat), _ptr(scales), _ptr(flat),
                                      _ptr(gflat), st))
    torch.cuda.synchronize()
    got = gflat.cpu().double()
    for name, ref, sl in (('g', gg.grad.flatten(), slice(0, rows)), ('v', v.grad.flatten(), slice(rows, rows + rows * inner)),
                          ('bias', bias.grad, slice(rows + rows * inner, None))):
        err = (got[sl] - ref).abs().max().item()
        assert err <= 2e-4 * ref.abs().max().item() + 1e-5, (name, err, ref.abs().max().item())


@pytest.mark.parametrize('case', [CASES[0], CASES[4], CASES[9], CASES[11]])
def test_every_wgrad_block_shape_agrees(case):
    """RtgWgradDesc.shape_cfg: every shape listed by rtg_wgrad_shape_candidates gives the same sum of partials as the
    heuristic one (fp32 summation order differs only through the number of splits)."""
    from rtg.lib import lib, WgradDesc, check
    B, Cin, Cout, L, K, s, d, p, g = case
    gen = torch.Generator().manual_seed(11)
    Lo = (L + 2 * p - d * (K - 1) - 1) // s + 1
    x, dy = torch.randn(B, Cin, L, generator=gen).cuda(), torch.randn(B, Cout, Lo, generator=gen).cuda()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    need = Cout * ((Cin // g) * K + 1)

    def run(cfg):
        wd = WgradDesc(B=B, C1=Cin, C2=0, L_in=L, groups=g, Cg=Cin // g, Mg=Cout // g, K=K, stride=s, dil=d, pad=p, Q=Lo,
                       dy_L=Lo, pre_mode=1, pre_slope=0.15, gy_mode=0, gy_slope=1.0, gy_scale=1.0, splits=1,
                       part_stride=0, shape_cfg=cfg)
        splits = lib.rtg_wgrad_splits(C.byref(wd))
        assert splits >= 1
        part = torch.full((splits * need,), float('nan'), device='cuda')
        wd.splits, wd.part_stride = splits, need
        check(lib.rtg_conv1d_wgrad(C.byref(wd), _ptr(x), None, _ptr(dy), None, _ptr(part), st))
        torch.cuda.synchronize()
        return part.view(splits, need).double().sum(0).cpu()

    cands = (C.c_int * 8)()
    probe = WgradDesc(B=B, C1=Cin, C2=0, L_in=L, groups=g, Cg=Cin // g, Mg=Cout // g, K=K, stride=s, dil=d, pad=p, Q=Lo,
                      dy_L=Lo, pre_mode=1, pre_slope=0.15, gy_mode=0, gy_slope=1.0, gy_scale=1.0, splits=1, part_stride=0)
    n = lib.rtg_wgrad_shape_candidates(C.byref(probe), cands, 8)
    assert n >= 2
    ref = run(0)
    assert torch.isfinite(ref).all()
    for c in cands[:n]:
        got = run(c)
        assert (got - ref).abs().max().item() <= 1e-4 * ref.abs().max().item(), f'shape_cfg {c}'
    probe.shape_cfg = 99
    assert lib.rtg_wgrad_splits(C.byref(probe)) < 0


GROUPS = [
    # members (C, L, K, dil) of one grouped launch; B = 3
    [(128, 256, 3, 9), (128, 256, 5, 3), (128, 256, 7, 1)],                       # ResBlock3 branches of a decoder stage
    [(128, 32, 3, 1), (128, 32, 3, 1), (128, 32, 3, 3), (128, 32, 3, 1), (128, 32, 3, 9), (128, 32, 3, 1)],   # ResidualStack
    [(64, 300, 7, 1), (64, 177, 3, 3)],                                            # ragged, different lengths
]


@pytest.mark.parametrize('div', [1, 3])
@pytest.mark.parametrize('members', GROUPS)
def test_grouped_wgrad_is_bit_identical_to_the_members_own_launches(members, div):
    """rtg_conv1d_wgrad_group: n weight-gradient problems of one kernel instance in one launch.  Every member's partials
    (all splits, weight and bias columns) equal the partials of its own rtg_conv1d_wgrad launch bit for bit, for every
    general block shape the members share; a group the kernel instance cannot serve is refused with RTG_EINVAL."""
    from rtg.lib import lib, WgradDesc, WgradPtrs
    B = 3
    g = torch.Generator().manual_seed(len(members) * 10 + div)
    descs, tens = [], []
    for (Cc, L, K, dil) in members:
        pad = dil * (K - 1) // 2
        x = torch.randn(B, Cc, L, generator=g).cuda()
        dy = torch.randn(B, Cc, L, generator=g).cuda()
        d = WgradDesc(B=B, C1=Cc, C2=0, L_in=L, groups=1, Cg=Cc, Mg=Cc, K=K, stride=1, dil=dil, pad=pad, Q=L, dy_L=L,
                      pre_mode=1, pre_slope=0.15, gy_mode=0, gy_slope=1.0, gy_scale=1.0, splits=1, part_stride=0)
        descs.append(d)
        tens.append((x, dy))
    lists = []
    for d in descs:
        cands = (C.c_int * 12)()
        k = lib.rtg_wgrad_shape_candidates(C.byref(d), cands, 12)
        lists.append([c for c in cands[:k] if 1 <= c <= 6])
    common = [c for c in lists[0] if all(c in l for l in lists[1:])]
    assert common
    n = len(members)
    for shape in common:
        singles, parts = [], []
        darr, parr = (WgradDesc * n)(), (WgradPtrs * n)()
        for i, (d, (x, dy)) in enumerate(zip(descs, tens)):
            d.shape_cfg, d.splits, d.part_stride = shape, 1, 0
            s = max(1, -(-lib.rtg_wgrad_splits(C.byref(d)) // div))
            need = d.Mg * (d.Cg * d.K + 1)
            d.splits, d.part_stride = s, need
            one = torch.full((s * need,), float('nan'), device='cuda')
            assert lib.rtg_conv1d_wgrad(C.byref(d), _ptr(x), None, _ptr(dy), None, _ptr(one), None) == 0
            singles.append(one)
            parts.append(torch.full((s * need,), float('nan'), device='cuda'))
            darr[i] = d
            parr[i] = WgradPtrs(x.data_ptr(), None, dy.data_ptr(), None, parts[-1].data_ptr())
        assert lib.rtg_conv1d_wgrad_group(n, darr, parr, None) == 0
        torch.cuda.synchronize()
        for one, grp in zip(singles, parts):
            assert torch.isfinite(one).all()
            assert torch.equal(one, grp)
    # refused: a member on a kernel of its own (shape 8), and a member whose rows use the 16-row tile
    darr[0].shape_cfg = 8
    assert lib.rtg_conv1d_wgrad_group(n, darr, parr, None) != 0


GCONV_CASES = [
    # the thin-group k41 layers of DiscriminatorS (discrminator.py:39-43), three scales, ragged batches / lengths
    (3, 32, 64, 8192, 41, 2, 1, 20, 4),
    (2, 64, 128, 4096, 41, 2, 1, 20, 8),
    (2, 128, 512, 2048, 41, 4, 1, 20, 32),
    (3, 512, 512, 512, 41, 4, 1, 20, 64),
    (5, 512, 512, 128, 41, 4, 1, 20, 64),
    (2, 128, 512, 1001, 41, 4, 1, 20, 32),    # ragged: the last position block of a clip is partial
    (1, 64, 128, 37, 41, 2, 1, 20, 8),        # shorter than the kernel
]


@pytest.mark.parametrize('case', GCONV_CASES)
def test_gconv_wgrad_on_the_vector_alus(case):
    """rtg_gconv.hip backward-weight (shape code 9): listed as a candidate for exactly these layers; weight, bias and
    weight-norm gradients through the usual split partials against torch autograd."""
    from rtg.lib import lib, WgradDesc
    B, Cin, Cout, L, K, s, d, p, g = case
    Lo = (L + 2 * p - d * (K - 1) - 1) // s + 1
    probe = WgradDesc(B=B, C1=Cin, C2=0, L_in=L, groups=g, Cg=Cin // g, Mg=Cout // g, K=K, stride=s, dil=d, pad=p, Q=Lo,
                      dy_L=Lo, pre_mode=1, pre_slope=0.15, gy_mode=0, gy_slope=1.0, gy_scale=1.0, splits=1, part_stride=0)
    cands = (C.c_int * 12)()
    n = lib.rtg_wgrad_shape_candidates(C.byref(probe), cands, 12)
    assert 9 in list(cands[:n])
    _check_case(case, shape_cfg=9, act='none')


@pytest.mark.parametrize('case', GCONV_CASES)
def test_gmfma_wgrad_on_the_matrix_cores(case):
    """rtg_gmfma.hip backward-weight (shape code 15, round 4): exact-fit matrix-core tiles for the groups of 16 output
    channels; listed for those layers (rows of at least 32 positions), weight, bias and weight-norm gradients through the usual
    split partials against torch autograd."""
    from rtg.lib import lib, WgradDesc
    B, Cin, Cout, L, K, s, d, p, g = case
    Lo = (L + 2 * p - d * (K - 1) - 1) // s + 1
    probe = WgradDesc(B=B, C1=Cin, C2=0, L_in=L, groups=g, Cg=Cin // g, Mg=Cout // g, K=K, stride=s, dil=d, pad=p, Q=Lo,
                      dy_L=Lo, pre_mode=1, pre_slope=0.15, gy_mode=0, gy_slope=1.0, gy_scale=1.0, splits=1, part_stride=0)
    cands = (C.c_int * 12)()
    n = lib.rtg_wgrad_shape_candidates(C.byref(probe), cands, 12)
    if Cout // g != 16 or Lo < 32:
        assert 15 not in list(cands[:n])                 # (8 output channels per group / short rows: rtg_gconv.hip's)
        return
    assert 15 in list(cands[:n])
    _check_case(case, shape_cfg=15, act='none')


def _partials(wd_kw, x, dy, cfg, bf16):
    """rtg_conv1d_wgrad -> the split partials summed in float64: [rows * (Cg * K + 1)]"""
    from rtg.lib import lib, WgradDesc
    wd = WgradDesc(**dict(wd_kw, shape_cfg=cfg, bf16=bf16, splits=1, part_stride=0))
    splits = lib.rtg_wgrad_splits(C.byref(wd))
    assert splits >= 1, (cfg, bf16, splits)
    need = wd.Mg * (wd.Cg * wd.K + 1)
    part = torch.full((splits * need,), float('nan'), device='cuda')
    wd.splits, wd.part_stride = splits, need
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.rtg_conv1d_wgrad(C.byref(wd), _ptr(x), None, _ptr(dy), None, _ptr(part), st) == 0
    torch.cuda.synchronize()
    return part.view(splits, need).double().sum(0)


DENSE_BF16 = [
    # B, C_in, C_out, L (or H, W), K (or kh), stride (or (sh, sw))
    ('1d', 22, 512, 512, 10, 5, 1), ('1d', 3, 512, 512, 128, 5, 1), ('1d', 14, 256, 512, 44, 5, 3), ('1d', 3, 32, 128, 7, 5, 1),
    ('2d', 2, 512, 512, (8, 5), 3, (1, 1)), ('2d', 2, 64, 256, (40, 18), 5, (3, 2)), ('2d', 1, 256, 512, (22, 9), 5, (3, 2)),
    ('2d', 2, 32, 64, (33, 35), 3, (2, 2)),          # 64 rows: the 4-wave block (code 14)
]


@pytest.mark.parametrize('case', DENSE_BF16)
def test_dense_wgrad_kernel_bf16(case):
    """bf16 form of rtg_dwgrad.hip (RtgWgradDesc.bf16: v_mfma_f32_16x16x32_bf16).  On operands that are exactly representable
    in bf16 every product is exact in fp32: the bf16 codes must reproduce the fp32 general kernel up to the order of the
    fp32 additions."""
    from rtg.lib import lib, WgradDesc
    kind, B, Cin, Cout, L, K, s = case
    gen = torch.Generator().manual_seed(11)

    def rb(t):
        return t.bfloat16().float()
    if kind == '1d':
        p = 2
        Lo = (L + 2 * p - K) // s + 1
        x, dy = rb(torch.randn(B, Cin, L, generator=gen)), rb(torch.randn(B, Cout, Lo, generator=gen))
        kw = dict(B=B, C1=Cin, C2=0, L_in=L, groups=1, Cg=Cin, Mg=Cout, K=K, stride=s, dil=1, pad=p, Q=Lo, dy_L=Lo, pre_mode=1,
                  pre_slope=0.5, gy_mode=0, gy_slope=1.0, gy_scale=0.25)
    else:
        (H, W), kh, (sh, sw) = L, K, s
        kw_, ph, pw = 3, kh // 2, 1
        Ho, Wo = (H + 2 * ph - kh) // sh + 1, (W + 2 * pw - kw_) // sw + 1
        x, dy = rb(torch.randn(B, Cin, H, W, generator=gen)), rb(torch.randn(B, Cout, Ho, Wo, generator=gen))
        kw = dict(B=B * Ho, C1=Cin * kh, C2=0, L_in=W, groups=1, Cg=Cin * kh, Mg=Cout, K=kw_, stride=sw, dil=1, pad=pw, Q=Wo,
                  dy_L=Wo, pre_mode=1, pre_slope=0.5, gy_mode=0, gy_slope=1.0, gy_scale=0.25, h_in=H, h_k=kh, h_stride=sh,
                  h_pad=ph, h_n=Ho)
    xd, dyd = x.cuda(), dy.cuda()
    ref = _partials(kw, xd, dyd, 0, 0)
    probe = WgradDesc(**dict(kw, bf16=1, splits=1, part_stride=0))
    cands = (C.c_int * 16)()
    n = lib.rtg_wgrad_shape_candidates(C.byref(probe), cands, 16)
    codes = [c for c in cands[:n] if c >= 10]
    assert codes, 'no dense code listed for the bf16 descriptor'
    scale = ref.abs().max().item()
    for c in codes:
        got = _partials(kw, xd, dyd, c, 1)
        err = (got - ref).abs().max().item()
        assert err <= 3e-5 * scale, (c, err, scale)
