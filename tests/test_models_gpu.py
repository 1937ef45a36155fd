"""Parity of the HIP hot path (through the reference-shaped Python API, i.e. through the C ABI) against the golden
fixtures produced by the reference and against the CPU oracle on the same seeded inputs.  GPU only."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def stats(t):
    t = t.detach().double().cpu()
    return np.array([t.sum().item(), t.abs().mean().item()])


def close_stats(a, b, numel, rtol=1e-3, atol=2e-5, sum_rel=2e-5):
    """column 1 = mean |x| (relative check); column 0 = plain sum, which cancels: its tolerance is relative to the
    sum of magnitudes numel * mean|x| (fp32 accumulation-order noise), not to the sum itself."""
    a, b = np.asarray(a), np.asarray(b)
    np.testing.assert_allclose(a[..., 1], b[..., 1], rtol=rtol, atol=atol)
    bad = np.abs(a[..., 0] - b[..., 0]) > sum_rel * np.asarray(numel) * b[..., 1] + 1e-4
    assert not bad.any(), (np.nonzero(bad), a[bad], b[bad])


def numels(m):
    return np.array([p.numel() for _, p in sorted(m.named_parameters())])


@pytest.fixture(scope='module')
def nets(oracle):
    import hparam  # noqa: F401
    from models import Generator_RefineGAN_small, MultiScaleDiscriminator, MultiPeriodDiscriminator
    torch.manual_seed(7)
    g, msd, mpd = Generator_RefineGAN_small(), MultiScaleDiscriminator(), MultiPeriodDiscriminator()
    for m in (g, msd, mpd):
        oracle.det_fill(m)
        m.to(DEV).train()
    return g, msd, mpd


@pytest.fixture(scope='module')
def onets(oracle):
    torch.manual_seed(7)
    g, msd, mpd = oracle.Generator(), oracle.MSD(), oracle.MPD()
    for m in (g, msd, mpd):
        oracle.det_fill(m)
        m.train()
    return g, msd, mpd


def test_state_dict_roundtrip_and_flat_views(nets, oracle):
    g = nets[0]
    bank = g.bank()
    assert bank.check_views()
    sd = {k: v.clone() for k, v in g.state_dict().items()}
    og = oracle.Generator()
    og.load_state_dict({k: v.cpu() for k, v in sd.items()})       # the oracle has the reference's key set
    g.load_state_dict(sd)
    assert bank.check_views() and g.bank() is bank
    assert bank.n_params == 2748371


@pytest.mark.parametrize('which', [0, 1, 2])
def test_pack_kernel_matches_host_packing(nets, which):
    import packref
    g = nets[which]
    bank = g.bank()
    g.token()
    torch.cuda.synchronize()
    packed = bank.packed.cpu().numpy()
    assert which == 0 or any(ly.fwd_tap for ly in bank.layers)       # the discriminators use the tap-major order
    for ly in bank.layers:
        w = ly.module.effective_weight().detach().cpu().numpy()
        w = w.reshape(w.shape[0], w.shape[1], -1)
        s = ly.stride
        if ly.kind == 'conv':
            fwd = packref.logical_fwd(w, ly.groups)
            bwd = packref.logical_dgrad_s1(w, ly.groups) if s == 1 else packref.logical_dgrad_poly(w, ly.groups, s)
        else:
            fwd, bwd = packref.logical_convT_poly(w, s), packref.logical_convT_dgrad(w)
        for L, off, size, tm, tap in ((fwd, ly.fwd_off, ly.fwd_size, ly.fwd_tm, ly.fwd_tap),
                                      (bwd, ly.bwd_off, ly.bwd_size, ly.bwd_tm, ly.bwd_tap)):
            ref = packref.pack_logical_tapmajor(L, tm) if tap else packref.pack_logical(L, tm)
            assert ref.size == size, ly.name
            np.testing.assert_allclose(packed[off:off + size], ref, rtol=2e-6, atol=1e-7, err_msg=ly.name)


def test_generator_forward_golden(nets, oracle, gold):
    x, y_tmpl, _ = oracle.golden_inputs()
    with torch.no_grad():
        y_hat = nets[0](x.to(DEV), y_tmpl.to(DEV))
    assert y_hat.shape == (2, 1, 8192)
    np.testing.assert_allclose(y_hat.cpu().numpy(), gold['y_hat'], atol=1e-4, rtol=0)


def test_generator_backward_golden(nets, oracle, gold):
    g = nets[0]
    x, y_tmpl, _ = oracle.golden_inputs()
    g.zero_grad()
    torch.manual_seed(4321)
    noise = []
    shapes = [(2, 128, 256), (2, 128, 256), (2, 64, 2048), (2, 64, 2048), (2, 32, 8192), (2, 32, 8192)]
    for s in shapes:                       # the six rand_like draws of the reference forward, same order
        noise.append(torch.rand(s).to(DEV))
    y_hat = g(x.to(DEV), y_tmpl.to(DEV), noise_list=noise)
    y_hat.backward(torch.from_numpy(gold['ggrad_yhat']).to(DEV))
    torch.cuda.synchronize()
    names = [n for n, _ in sorted(g.named_parameters())]
    assert names == list(gold['ggrad_g_names'])
    got = np.stack([stats(p.grad) for _, p in sorted(g.named_parameters())])
    close_stats(got, gold['ggrad_g_stats'], numels(g), rtol=2e-3, atol=1e-5, sum_rel=2e-3)   # see the flip note below


def _g_grad_errors(g, og, oracle):
    x, y_tmpl, _ = oracle.golden_inputs(seed=3)
    gen = torch.Generator().manual_seed(5)
    dy = torch.randn(2, 1, 8192, generator=gen)
    g.zero_grad(); og.zero_grad()
    g(x.to(DEV), y_tmpl.to(DEV)).backward(dy.to(DEV))
    og(x, y_tmpl).backward(dy)
    torch.cuda.synchronize()
    op = dict(og.named_parameters())
    return {n: (p.grad.cpu() - op[n].grad).norm().item() / (op[n].grad.norm().item() + 1e-20)
            for n, p in g.named_parameters() if n != 'noise.w'}


def test_generator_grads_elementwise_vs_oracle(nets, onets, oracle):
    """Every parameter gradient of G against the CPU oracle: relative L2 error per tensor.

    What limits the agreement is not the kernels but the kink of the leaky-relu: its derivative jumps at 0, and an
    activation that lands within fp32 rounding of 0 (|x| ~ 1e-7 of ~2e7 activations: zero to a few per pass) takes the
    other branch on the GPU than on the CPU.  One flip moves the gradients upstream of it by up to ~1/sqrt(positions)
    (measured 2e-4 .. 2e-3); without a flip the HIP path agrees with a float64 run of the oracle to 2e-6.  So: every
    run must stay below 5e-3 (a kernel bug is O(1) in its own layer), and over three initialisations (the
    reference's default init under three seeds, noise switched off because the device RNG differs by construction)
    the median of the worst-tensor error must show the flip-free accuracy."""
    from models import Generator_RefineGAN_small
    worst = []
    for seed in (114514, 11, 7):
        torch.manual_seed(seed)
        g = Generator_RefineGAN_small().to(DEV).train()
        og = oracle.Generator().train()
        og.load_state_dict({k: v.cpu() for k, v in g.state_dict().items()})
        with torch.no_grad():
            g.noise.w.zero_(); og.noise.w.zero_()
        err = _g_grad_errors(g, og, oracle)
        n_bad = max(err, key=err.get)
        assert err[n_bad] <= 5e-3, (seed, n_bad, err[n_bad])
        worst.append(err[n_bad])
    assert sorted(worst)[1] <= 1e-4, worst
    # det_fill weights (the golden fixtures' parameter set): unscaled +-1 weights put more activations next to 0, and in
    # the ResidualStacks that run on 2 x 32 .. 2 x 2048 positions one flip is worth up to 1/sqrt(positions)
    err = _g_grad_errors(nets[0], onets[0], oracle)
    n_bad = max(err, key=err.get)
    assert err[n_bad] <= 2e-2, (n_bad, err[n_bad])
    # (the median tensor: 1.5e-6 when no activation flips against the CPU run, 5e-4 when ONE flips downstream of the
    # bottleneck stack — which of the two a kernel lands on is rounding luck: the fused ResidualStack launch is closer to
    # float64 than the six-launch path, 1.07e-7 vs 1.43e-7 relative, and lands on the other side)
    assert sorted(err.values())[len(err) // 2] <= 2e-3, sorted(err.values())[len(err) // 2]


@pytest.mark.parametrize('which', ['msd', 'mpd'])
def test_discriminator_forward_golden(nets, oracle, gold, which):
    d = nets[1] if which == 'msd' else nets[2]
    _, _, y = oracle.golden_inputs()
    yd = torch.from_numpy(gold['y_hat'])
    with torch.no_grad():
        lr, lg, fr, fg = d(y.to(DEV), yd.to(DEV))
    for i, (r, gg) in enumerate(zip(lr, lg)):
        np.testing.assert_allclose(r.cpu().numpy(), gold[f'{which}_logit_r{i}'], rtol=1e-3, atol=2e-4)
        np.testing.assert_allclose(gg.cpu().numpy(), gold[f'{which}_logit_g{i}'], rtol=1e-3, atol=2e-4)
    shapes = np.array([list(f.shape) + [1] * (4 - f.dim()) for fl in fr for f in fl])
    assert (shapes == gold[f'{which}_fmap_shapes']).all()
    ne = np.array([f.numel() for fl in fr for f in fl])
    close_stats(np.stack([stats(f) for fl in fr for f in fl]), gold[f'{which}_fmap_r_stats'], ne)
    close_stats(np.stack([stats(f) for fl in fg for f in fl]), gold[f'{which}_fmap_g_stats'], ne)
    from models import discriminator_loss, generator_loss, feature_loss
    np.testing.assert_allclose(discriminator_loss(lr, lg).item(), gold[f'{which}_d_loss'], rtol=1e-4)
    np.testing.assert_allclose(generator_loss(lg, lr).item(), gold[f'{which}_g_loss'], rtol=1e-4)
    np.testing.assert_allclose(feature_loss(fr, fg).item(), gold[f'{which}_fm_loss'], rtol=1e-4)


@pytest.mark.parametrize('which', ['msd', 'mpd'])
def test_discriminator_backward_golden(nets, oracle, gold, which):
    from models import discriminator_loss
    d = nets[1] if which == 'msd' else nets[2]
    _, _, y = oracle.golden_inputs()
    yd = torch.from_numpy(gold['y_hat'])
    d.zero_grad()
    lr, lg, _, _ = d(y.to(DEV), yd.to(DEV))
    discriminator_loss(lr, lg).backward()
    torch.cuda.synchronize()
    got = np.stack([stats(p.grad) for _, p in sorted(d.named_parameters())])
    close_stats(got, gold[f'dgrad_{which}_stats'], numels(d), rtol=2e-3, atol=1e-6)


@pytest.mark.parametrize('which', ['msd', 'mpd'])
def test_generator_side_losses_through_frozen_discriminator(nets, onets, oracle, gold, which):
    """G-update path: D frozen, gradient w.r.t. the generated wave through feature + adversarial losses."""
    from models import generator_loss, feature_loss
    d, od = (nets[1], onets[1]) if which == 'msd' else (nets[2], onets[2])
    _, _, y = oracle.golden_inputs()
    yd = torch.from_numpy(gold['y_hat'])
    yh = yd.clone().to(DEV).requires_grad_(True)
    for p in d.parameters():
        p.requires_grad_(False)
    try:
        lr, lg, fr, fg = d(y.to(DEV), yh)
        (generator_loss(lg, lr) + 2 * feature_loss(fr, fg)).backward()
    finally:
        for p in d.parameters():
            p.requires_grad_(True)
    yo = yd.clone().requires_grad_(True)
    lr, lg, fr, fg = od(y, yo)
    (oracle.generator_loss(lg, lr) + 2 * oracle.feature_loss(fr, fg)).backward()
    ref = yo.grad
    err = (yh.grad.cpu() - ref).abs().max().item()
    assert err <= 2e-3 * ref.abs().max().item() + 1e-7, err


def test_stft_mel_and_losses_golden(oracle, gold):
    from audio import stft_mel_spec
    from models import multi_stft_loss, dynamic_loss
    _, _, y = oracle.golden_inputs()
    yd = torch.from_numpy(gold['y_hat'])
    for n_fft, win, hop in oracle.STFT_PARAMS:
        mel, spec = stft_mel_spec(y.squeeze(1).to(DEV), n_fft, win, hop, True)
        np.testing.assert_allclose(mel.cpu().numpy(), gold[f'stft{n_fft}_mel'], rtol=2e-4, atol=2e-6)
        idx = gold[f'stft{n_fft}_idx']
        logS = spec[:, 0].flatten().cpu().numpy()[idx]
        P = spec[:, 1].flatten().cpu().numpy()[idx] * oracle.PI
        Sg, Pg = gold[f'stft{n_fft}_S'], gold[f'stft{n_fft}_P']
        np.testing.assert_allclose(np.exp(logS), Sg, rtol=2e-4, atol=2e-5)
        dph = np.abs(P - Pg)
        dph = np.minimum(dph, 2 * np.pi - dph)
        assert np.all(dph[Sg > 1e-3] < 2e-3)
    loss = multi_stft_loss(y.to(DEV), yd.to(DEV), ret_loss=True)
    np.testing.assert_allclose(loss.item(), gold['loss_mstft'], rtol=1e-4)     # north-star tolerance: 1e-4 rel
    np.testing.assert_allclose(dynamic_loss(y.to(DEV), yd.to(DEV)).item(), gold['loss_dyn'], rtol=1e-5)


def test_stft_and_dyn_backward_golden(oracle, gold):
    from models import multi_stft_loss, dynamic_loss
    _, _, y = oracle.golden_inputs()
    yd = torch.from_numpy(gold['y_hat'])
    yh = yd.clone().to(DEV).requires_grad_(True)
    multi_stft_loss(y.to(DEV), yh, ret_loss=True).backward()
    ref = gold['grad_mstft_yhat']
    np.testing.assert_allclose(yh.grad.cpu().numpy(), ref, rtol=2e-3, atol=2e-3 * np.abs(ref).max())
    yh = yd.clone().to(DEV).requires_grad_(True)
    dynamic_loss(y.to(DEV), yh).backward()
    np.testing.assert_allclose(yh.grad.cpu().numpy(), gold['grad_dyn_yhat'], rtol=1e-5, atol=1e-9)


def test_cpu_input_is_refused(nets):
    from rtg.lib import RtgError
    with pytest.raises(RtgError):
        nets[0](torch.zeros(1, 80, 32), torch.zeros(1, 1, 8192))


def test_grouped_launch_path_matches_forked_streams(nets, oracle, gold, monkeypatch):
    """RTG_GROUP=1 (sibling sub-discriminators layer by layer through rtg_conv1d_group) gives the same logits, feature
    maps and parameter gradients as the default forked-stream execution: the grouped launch is bit-identical per
    member, the weight-gradient split-K order is the tuner's choice in both (rounding-level differences only)."""
    import models.discrminator as D
    from models import discriminator_loss
    from rtg import ops
    # (the statement is about the general kernel: the thin-group layers' other kernels — rtg_gconv.hip, rtg_gmfma.hip,
    # picked per problem by the tuner when an earlier test of this process timed these shapes — sum in another order)
    monkeypatch.setattr(ops, 'GCONV', False)
    _, _, y = oracle.golden_inputs()
    yd = torch.from_numpy(gold['y_hat'])
    for d in (nets[1], nets[2]):
        res = []
        for grouped in (False, True):
            D.GROUPED = grouped
            try:
                d.zero_grad()
                lr, lg, fr, fg = D.run_stacks([(d, y.to(DEV), yd.to(DEV))])[0]
                discriminator_loss(lr, lg).backward()
                torch.cuda.synchronize()
                res.append(([t.detach().clone() for t in lr + lg], [getattr(f, '_rtg_base', f).detach().clone() for fl in fr for f in fl],
                            d.bank().gflat.clone()))
            finally:
                D.GROUPED = False
        for a, b in zip(res[0][0] + res[0][1], res[1][0] + res[1][1]):
            assert torch.equal(a, b)
        ga, gb = res[0][2], res[1][2]
        assert ((ga - gb).norm() / ga.norm()).item() < 1e-5


@pytest.mark.parametrize('mode', [1, 2])
def test_grouped_resblock_branches_match_forked_streams(nets, oracle, mode):
    """RTG_MRF_GROUP: the three parallel ResBlock3 branches of a decoder stage conv by conv through rtg_conv1d_group
    (residual epilogue forward, dy + mask * convT(dy) backward) against each branch on its own stream.  mode 1 groups
    the stages of >= 64 channels (the default), 2 all of them.  The block shape, and with it the accumulation order
    inside a tile, is the tuner's choice in both: rounding-level differences only."""
    from rtg import ops
    x, y_tmpl, _ = oracle.golden_inputs()
    g = nets[0]
    res = []
    old = ops.MRF_GROUP
    try:
        for m in (0, mode):
            ops.MRF_GROUP = m
            g.zero_grad()
            out = g(x.to(DEV), y_tmpl.to(DEV))
            (out * torch.linspace(-1, 1, out.numel(), device=DEV).view_as(out)).sum().backward()
            torch.cuda.synchronize()
            res.append((out.detach().clone(), g.bank().gflat[:-2].clone()))    # without noise.w (fresh noise per call) and the flag slot
    finally:
        ops.MRF_GROUP = old
    assert ((res[0][0] - res[1][0]).abs().max() / res[0][0].abs().max()).item() < 1e-5
    assert ((res[0][1] - res[1][1]).norm() / res[0][1].norm()).item() < 1e-5


@pytest.mark.parametrize('B,T', [(1, 256 * 3), (3, 256 * 9), (2, 256 * 86)])
def test_generator_other_lengths_and_batches(nets, onets, oracle, B, T):
    """ragged / minimum / finetune-sized inputs: any frame count and batch size the reference accepts (T = 256 x frames;
    86 frames = the "1 s" clips of BASELINE configs[4])"""
    g = torch.Generator().manual_seed(B * 1000 + T)
    x = torch.randn(B, 80, T // 256, generator=g).abs()
    y_tmpl = torch.rand(B, 1, T, generator=g) * 2 - 1
    with torch.no_grad():
        got = nets[0](x.to(DEV), y_tmpl.to(DEV)).cpu()
        ref = onets[0](x, y_tmpl)
    assert got.shape == (B, 1, T)
    np.testing.assert_allclose(got.numpy(), ref.numpy(), atol=1e-4, rtol=0)


@pytest.mark.parametrize('which,T', [('msd', 8000), ('mpd', 8000), ('msd', 700), ('mpd', 700), ('mpd', 22016)])
def test_discriminators_on_lengths_that_are_not_multiples_of_anything(nets, onets, oracle, which, T):
    """clip lengths that are no multiple of the hop, of the periods or of the pooling stride (reflect padding of the
    period fold, odd AvgPool lengths), batch 1"""
    d, od = (nets[1], onets[1]) if which == 'msd' else (nets[2], onets[2])
    g = torch.Generator().manual_seed(T)
    y = torch.rand(1, 1, T, generator=g) * 2 - 1
    yh = torch.rand(1, 1, T, generator=g) * 2 - 1
    with torch.no_grad():
        lr, lg, fr, fg = d(y.to(DEV), yh.to(DEV))
        olr, olg, ofr, ofg = od(y, yh)
    for a, b in zip(lr + lg, olr + olg):
        assert a.shape == b.shape
        np.testing.assert_allclose(a.cpu().numpy(), b.numpy(), rtol=1e-3, atol=2e-4)
    for a, b in zip([f for fl in fr + fg for f in fl], [f for fl in ofr + ofg for f in fl]):
        assert a.shape == b.shape
        np.testing.assert_allclose(a.cpu().numpy(), b.numpy(), rtol=1e-3, atol=5e-4)


def test_full_size_batch_is_the_concatenation_of_its_clips(nets):
    """BASELINE configs[1] size (32 clips x 8192 samples), a size-independent property: every clip of the batch gets the
    result it gets in a batch of two (other block shapes, packing and grid: fp32 summation order only) — generator
    output, all MSD / MPD logits and feature maps."""
    gnet, msd, mpd = nets
    B, T = 32, 8192
    gen = torch.Generator().manual_seed(321)
    x = torch.randn(B, 80, T // 256, generator=gen).abs().to(DEV)
    y_tmpl = (torch.rand(B, 1, T, generator=gen) * 2 - 1).to(DEV)
    y = (torch.rand(B, 1, T, generator=gen) * 2 - 1).to(DEV)
    with torch.no_grad():
        full = gnet(x, y_tmpl)
        assert full.shape == (B, 1, T) and torch.isfinite(full).all()
        for b0 in (0, 14, 30):
            part = gnet(x[b0:b0 + 2].contiguous(), y_tmpl[b0:b0 + 2].contiguous())
            np.testing.assert_allclose(part.cpu().numpy(), full[b0:b0 + 2].cpu().numpy(), atol=2e-5, rtol=0)
        for d in (msd, mpd):
            lr, lg, fr, fg = d(y, full)
            b0 = 17
            plr, plg, pfr, pfg = d(y[b0:b0 + 2].contiguous(), full[b0:b0 + 2].contiguous())
            for a, p in zip(lr + lg, plr + plg):
                np.testing.assert_allclose(p.cpu().numpy(), a[b0:b0 + 2].cpu().numpy(), rtol=1e-3, atol=2e-4)
            for a, p in zip([f for fl in fr + fg for f in fl], [f for fl in pfr + pfg for f in fl]):
                np.testing.assert_allclose(p.cpu().numpy(), a[b0:b0 + 2].cpu().numpy(), rtol=1e-3, atol=5e-4)


def test_full_size_generator_side_gradient_is_per_clip(nets):
    """Same size, the backward side of the generator step: with the discriminators frozen (one 2B-clip launch per layer
    for real + generated clips, backward-data over the generated half only: ops.PairConvFn), d loss / d y_hat of a clip
    does not depend on the other clips of the batch."""
    from models.loss import generator_loss, feature_loss
    _, msd, mpd = nets
    B, T = 32, 8192
    gen = torch.Generator().manual_seed(654)
    y = (torch.rand(B, 1, T, generator=gen) * 2 - 1).to(DEV)
    yh = (torch.rand(B, 1, T, generator=gen) * 2 - 1).to(DEV)

    def grad_of(y_, yh_):
        yh_ = yh_.clone().requires_grad_(True)
        n = yh_.shape[0]
        total = 0.0
        params = [p for d in (msd, mpd) for p in d.parameters()]
        for p in params:
            p.requires_grad_(False)
        try:
            for d in (msd, mpd):
                lr, lg, fr, fg = d(y_, yh_)
                # batch-mean losses: scaled by the batch size, a clip's gradient is comparable across batch sizes
                total = total + (generator_loss(lg, lr) + feature_loss(fr, fg)) * n
            total.backward()
        finally:
            for p in params:
                p.requires_grad_(True)
        return yh_.grad.detach()

    full = grad_of(y, yh)
    assert torch.isfinite(full).all() and full.abs().max() > 0
    b0 = 9
    part = grad_of(y[b0:b0 + 2].contiguous(), yh[b0:b0 + 2].contiguous())
    scale = full.abs().max().item()
    # the batch of 32 and the batch of 2 are different problems for the tuner (matrix-core or vector-ALU kernel, block
    # shape: other summation orders), so a feature-map element within rounding of 0 can take the other leaky-relu branch
    # in one of them: relative L2 over the wave, and a bound on the few elements such a flip moves
    a, b = part.cpu().double(), full[b0:b0 + 2].cpu().double()
    assert ((a - b).norm() / b.norm()).item() < 2e-4
    assert (a - b).abs().max().item() < 3e-3 * scale
    assert ((a - b).abs() > 2e-4 * scale).double().mean().item() < 5e-3
