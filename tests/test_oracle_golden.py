"""Pins the CPU oracle (oracle/rtg_oracle.py) to fixtures produced by running the reference itself
(oracle/gen_golden.py).  CPU only."""
import numpy as np
import pytest
import torch


def stats(t):
    t = t.detach().double()
    return np.array([t.sum().item(), t.abs().mean().item()])


def close_stats(a, b, rtol=2e-4, atol=2e-5):
    a, b = np.asarray(a), np.asarray(b)
    # column 0 = plain sum (cancellation: absolute tolerance scaled by abs-mean is not available -> loose atol)
    np.testing.assert_allclose(a[..., 1], b[..., 1], rtol=rtol, atol=atol)
    scale = np.maximum(np.abs(b[..., 0]), 1.0)
    assert np.all(np.abs(a[..., 0] - b[..., 0]) <= 5e-3 * scale + 5e-3), np.abs(a[..., 0] - b[..., 0]).max()


@pytest.fixture(scope='module')
def nets(oracle):
    torch.manual_seed(114514)
    g, msd, mpd, mtd = oracle.Generator(), oracle.MSD(), oracle.MPD(), oracle.MTD()
    return g, msd, mpd, mtd


def test_construction_matches_reference(nets, gold):
    """Same keys, same parameter counts and (same seed => same RNG stream) the same initial values."""
    for tag, m in zip(('g', 'msd', 'mpd', 'mtd'), nets):
        assert sum(p.numel() for p in m.parameters()) == int(gold[f'init_{tag}_count'])
        assert sorted(m.state_dict().keys()) == list(gold[f'init_{tag}_keys'])
        st = np.stack([stats(p) for _, p in sorted(m.named_parameters())])
        np.testing.assert_allclose(st, gold[f'init_{tag}_stats'], rtol=1e-6, atol=1e-7)
    assert int(gold['init_g_count']) == 2748371          # retunegan/hparam.py:50


@pytest.fixture(scope='module')
def filled(nets, oracle):
    for m in nets:
        oracle.det_fill(m)
        m.train()
    return nets


def test_mel_basis(oracle, gold):
    for n_fft in (2048, 1024, 512):
        fb = oracle.mel_filterbank(n_fft)
        assert fb.dtype == np.float32 and fb.shape == (80, n_fft // 2 + 1)
        np.testing.assert_allclose(fb.astype(np.float64).sum(), gold[f'melbasis{n_fft}_sum'], rtol=1e-7)
        np.testing.assert_allclose(fb.astype(np.float64).sum(1), gold[f'melbasis{n_fft}_rowsum'], rtol=1e-6)


def test_stft(oracle, gold):
    _, _, y = oracle.golden_inputs()
    for n_fft, win, hop in oracle.STFT_PARAMS:
        S, M, P = oracle.stft_mag_mel_phase(y.squeeze(1), n_fft, win, hop)
        np.testing.assert_allclose(M.numpy(), gold[f'stft{n_fft}_mel'], rtol=2e-4, atol=1e-6)
        idx = gold[f'stft{n_fft}_idx']
        Sg, Pg = gold[f'stft{n_fft}_S'], gold[f'stft{n_fft}_P']
        np.testing.assert_allclose(S.flatten().numpy()[idx], Sg, rtol=1e-4, atol=2e-5)
        d = np.abs(P.flatten().numpy()[idx] - Pg)
        d = np.minimum(d, 2 * np.pi - d)                       # phase compared modulo 2*pi
        assert np.all(d[Sg > 1e-3] < 1e-3)
        close_stats(stats(torch.log(S)), gold[f'stft{n_fft}_logS_stats'])


def test_generator_forward(filled, oracle, gold):
    x, y_tmpl, _ = oracle.golden_inputs()
    with torch.no_grad():
        y_hat = filled[0](x, y_tmpl)
    np.testing.assert_allclose(y_hat.numpy(), gold['y_hat'], atol=2e-5, rtol=0)


def test_discriminators_and_losses(filled, oracle, gold):
    g, msd, mpd, mtd = filled
    _, _, y = oracle.golden_inputs()
    yd = torch.from_numpy(gold['y_hat'])
    with torch.no_grad():
        S, Sg = oracle.multi_stft_loss(y, yd, ret_specs=True)
        for tag, d, a, b in (('msd', msd, y, yd), ('mpd', mpd, y, yd), ('mtd', mtd, S, Sg)):
            lr, lg, fr, fg = d(a, b)
            for i, (r, gg) in enumerate(zip(lr, lg)):
                np.testing.assert_allclose(r.numpy(), gold[f'{tag}_logit_r{i}'], rtol=1e-3, atol=2e-4)
                np.testing.assert_allclose(gg.numpy(), gold[f'{tag}_logit_g{i}'], rtol=1e-3, atol=2e-4)
            shapes = np.array([list(f.shape) + [1] * (4 - f.dim()) for fl in fr for f in fl])
            assert (shapes == gold[f'{tag}_fmap_shapes']).all()
            close_stats(np.stack([stats(f) for fl in fg for f in fl]), gold[f'{tag}_fmap_g_stats'])
            np.testing.assert_allclose(oracle.discriminator_loss(lr, lg).item(), gold[f'{tag}_d_loss'], rtol=1e-4)
            np.testing.assert_allclose(oracle.generator_loss(lg, lr).item(), gold[f'{tag}_g_loss'], rtol=1e-4)
            np.testing.assert_allclose(oracle.feature_loss(fr, fg).item(), gold[f'{tag}_fm_loss'], rtol=1e-4)
        np.testing.assert_allclose(oracle.multi_stft_loss(y, yd, ret_loss=True).item(), gold['loss_mstft'], rtol=1e-5)
        np.testing.assert_allclose(oracle.dynamic_loss(y, yd).item(), gold['loss_dyn'], rtol=1e-5)
        np.testing.assert_allclose(oracle.envelope_loss(y, yd).item(), gold['loss_env'], rtol=1e-5)
        np.testing.assert_allclose(oracle.strip_mirror_loss(yd).item(), gold['loss_sm'], rtol=1e-5)


def test_backward(filled, oracle, gold):
    g, msd, mpd, mtd = filled
    x, y_tmpl, y = oracle.golden_inputs()
    yd = torch.from_numpy(gold['y_hat'])
    for m in filled:
        m.zero_grad()
    dl = oracle.d_losses(y, yd, msd, mpd, mtd)
    tot = sum(dl.values())
    tot.backward()
    np.testing.assert_allclose(tot.item(), gold['loss_disc_all'], rtol=1e-5)
    for tag, m in (('msd', msd), ('mpd', mpd), ('mtd', mtd)):
        close_stats(np.stack([stats(p.grad) for _, p in sorted(m.named_parameters())]), gold[f'dgrad_{tag}_stats'])
    # G-side losses: evaluated at the reference's own y_hat (the phase input of MTD wraps at +-pi, so the loss surface
    # is discontinuous in y_hat; chaining through a 1e-5-different y_hat would compare different branches)
    for m in filled:
        m.zero_grad()
    yh = yd.clone().requires_grad_(True)
    gl = oracle.g_losses(y, yh, msd, mpd, mtd)
    gl['total'].backward()
    np.testing.assert_allclose(gl['total'].item(), gold['loss_gen_all'], rtol=1e-5)
    np.testing.assert_allclose(yh.grad.numpy(), gold['ggrad_yhat'], rtol=2e-3, atol=2e-6)
    torch.manual_seed(4321)          # same six rand_like draws as gen_golden.py (d loss / d noise.w depends on them)
    y_hat = g(x, y_tmpl)
    y_hat.backward(torch.from_numpy(gold['ggrad_yhat']))
    close_stats(np.stack([stats(p.grad) for _, p in sorted(g.named_parameters())]), gold['ggrad_g_stats'],
                rtol=1e-3, atol=1e-5)
    yh = yd.clone().requires_grad_(True)
    oracle.multi_stft_loss(y, yh, ret_loss=True).backward()
    np.testing.assert_allclose(yh.grad.numpy(), gold['grad_mstft_yhat'], rtol=1e-3, atol=1e-7)
    yh = yd.clone().requires_grad_(True)
    oracle.dynamic_loss(y, yh).backward()
    np.testing.assert_allclose(yh.grad.numpy(), gold['grad_dyn_yhat'], rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize('name,cfg', [('cfg1', (True, False, False, 1)), ('cfg2', (True, True, False, 2)),
                                      ('cfg4', (True, True, True, 2))])
def test_train_steps(oracle, gold, name, cfg):
    """Two complete iterations of train.py:121-193 (D x n, then G, AdamW) for BASELINE configs 1, 2 and 4 at B=2."""
    use_msd, use_mpd, use_mtd, d_times = cfg
    torch.manual_seed(1234)
    g, msd, mpd, mtd = oracle.Generator(), oracle.MSD(), oracle.MPD(), oracle.MTD()
    for m in (g, msd, mpd, mtd):
        oracle.det_fill(m)
        m.train()
    msd, mpd, mtd = (msd if use_msd else None), (mpd if use_mpd else None), (mtd if use_mtd else None)
    discs = [d for d in (msd, mpd, mtd) if d is not None]
    og, od = oracle.make_optimizers(g, discs)
    x, y_tmpl, y = oracle.golden_inputs()
    rec, first = [], {}
    for i in range(2):
        dl, gl = oracle.train_step(g, og, od, x, y_tmpl, y, msd, mpd, mtd, d_times)
        rec.append([sum(dl.values()).item(), gl['total'].item()])
        if i == 0:                      # (round 4 fixture: the parameters after the first step as well)
            first = {tag: np.stack([stats(p) for _, p in sorted(m.named_parameters())])
                     for tag, m in (('g', g), ('msd', msd), ('mpd', mpd), ('mtd', mtd)) if m is not None}
    np.testing.assert_allclose(np.array(rec), gold[f'step_{name}_losses'], rtol=2e-4)
    for tag, got in first.items():
        want = gold[f'step_{name}_{tag}_stats1']
        if not use_mtd:
            close_stats(got, want)
        else:
            assert np.all(np.abs(got[:, 1] - want[:, 1]) <= 0.25 * 2e-4 * d_times)

    def check(mod, key):
        got = np.stack([stats(p) for _, p in sorted(mod.named_parameters())])
        if not use_mtd:
            return close_stats(got, gold[key])
        # with MTD in the loop the phase input wraps at +-pi: a few gradient elements change sign under 1e-7
        # perturbations and AdamW turns each into a +-lr move, so parameters are only pinned to a fraction of lr*steps
        numel = np.array([p.numel() for _, p in sorted(mod.named_parameters())])
        lr_steps = 2e-4 * 2 * d_times
        assert np.all(np.abs(got[:, 1] - gold[key][:, 1]) <= 0.25 * lr_steps)
        assert np.all(np.abs(got[:, 0] - gold[key][:, 0]) <= 0.25 * lr_steps * numel + 5e-3)

    check(g, f'step_{name}_g_stats')
    for tag, d in (('msd', msd), ('mpd', mpd), ('mtd', mtd)):
        if d is not None:
            check(d, f'step_{name}_{tag}_stats')
