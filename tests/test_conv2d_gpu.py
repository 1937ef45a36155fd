"""The 2-D mode of rtg_conv1d / rtg_conv1d_wgrad (StftDiscriminator layers) and the MTD stack against the CPU.  GPU only."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda'

CASES = [
    # B, Cin, Cout, H, W, (kh,kw), (sh,sw), (ph,pw)
    (2, 2, 32, 65, 35, (3, 3), (2, 1), (1, 1)),      # the first layer: rtg_thin2d.hip (forward; weight gradient with force 7)
    (3, 2, 32, 129, 137, (3, 3), (2, 1), (1, 1)),    # several blocks, a partial last one
    (1, 2, 32, 8, 5, (3, 3), (2, 1), (1, 1)),        # an even row count (the last kernel row of the last output row is padding)
    (2, 32, 64, 33, 35, (3, 3), (2, 2), (1, 1)),
    (2, 64, 256, 40, 18, (5, 3), (3, 2), (2, 1)),
    (3, 256, 512, 22, 9, (5, 3), (3, 2), (2, 1)),
    (2, 512, 512, 8, 5, (3, 3), (1, 1), (1, 1)),
    (2, 512, 1, 8, 18, (3, 3), (1, 1), (1, 1)),
    (3, 512, 1, 29, 5, (3, 3), (1, 1), (1, 1)),      # conv_post on the other map shapes (cout1_k3x3_kernel: 435 columns,
    (5, 512, 1, 15, 9, (3, 3), (1, 1), (1, 1)),      # 675 columns: partial last block)
]


class _Net(nn.Module):
    pass


def _serves(case, force):
    """whether the forced weight-gradient kernel exists for the layer (0: always)"""
    Cin, Cout = case[1], case[2]
    return not ((force == 7 and Cin != 2 and Cout != 1) or (force in (10, 11) and Cout % 128) or (force == 14 and Cout != 64))


@pytest.mark.parametrize('wt', [False, True])
@pytest.mark.parametrize('case,force', [(c, f) for c in CASES for f in (0, 7, 10, 11, 14) if _serves(c, f)])
def test_conv2d_layer_forward_backward(case, force, wt, monkeypatch):
    """force: the weight-gradient block shape the tuner would have to pick (0: the library's heuristic; 7: the bandwidth
    kernels of rtg_wgrad_thin.hip / rtg_thin2d.hip; 10, 11: the dense kernel of rtg_dwgrad.hip in its 2-D mode, where it
    serves the layer).  wt: the layer built with WNConv(wt=True) — run on the input with its last two axes swapped (how the
    spectrogram discriminators run along the frequency axis), same parameters: its output / input gradient are the
    transposes, its parameter gradients the very same tensors."""
    import ctypes as C
    from models.layers import WNConv, BankedModel, conv
    from rtg import tune
    from rtg.lib import lib
    B, Cin, Cout, H, W, k, s, p = case
    used = []
    if force:
        def forced(wd, run):
            wd.shape_cfg, wd.splits, wd.part_stride = 0, 1, 0
            cands = (C.c_int * 16)()
            n = lib.rtg_wgrad_shape_candidates(C.byref(wd), cands, 16)
            pick = force if force in list(cands[:n]) else 0
            used.append(pick)
            return pick
        monkeypatch.setattr(tune, 'wgrad_cfg', forced)

    class Net(BankedModel):
        def __init__(self):
            super().__init__()
            self.c = WNConv('conv2d', Cin, Cout, k, stride=s, pad=p, wt=wt)

        def forward(self, x):
            return conv(self.token(), self.c, x, pre_slope=0.15)

    tr = (lambda t: t.transpose(2, 3).contiguous()) if wt else (lambda t: t)      # noqa: E731
    torch.manual_seed(3)
    net = Net()
    x = torch.randn(B, Cin, H, W)
    dy_seed = torch.Generator().manual_seed(4)
    # CPU reference in float64
    v = net.c.weight_v.detach().double().requires_grad_(True)
    g = net.c.weight_g.detach().double().requires_grad_(True)
    bias = net.c.bias.detach().double().requires_grad_(True)
    xr = x.double().requires_grad_(True)
    w = g * v / v.flatten(1).norm(dim=1).reshape(-1, 1, 1, 1)
    y = F.conv2d(F.leaky_relu(xr, 0.15), w, bias, s, p)
    dy = torch.randn(y.shape, generator=dy_seed)
    y.backward(dy.double())
    net.to(DEV)
    xg = tr(x).to(DEV).requires_grad_(True)
    out = net(xg)
    out.backward(tr(dy).to(DEV))
    torch.cuda.synchronize()

    def close(a, b, name):
        err = (a.detach().cpu().double() - b).abs().max().item()
        assert err <= 2e-4 * b.abs().max().item() + 1e-5, (name, err, b.abs().max().item())

    close(tr(out), y.detach(), 'out')
    close(tr(xg.grad), xr.grad, 'dx')
    close(net.c.weight_v.grad, v.grad, 'dv')
    close(net.c.weight_g.grad, g.grad, 'dg')
    close(net.c.bias.grad, bias.grad, 'dbias')
    if force in (10, 11) and Cout % 128 == 0 and (Cin * (k[1] if wt else k[0])) % (32 if force == 11 else 16) == 0 and out.shape[-1] >= 4:
        assert used == [force], used              # the dense kernel really ran
    if force == 14 and Cout == 64:
        assert used == [14], used                 # the 64-row block really ran
    if force == 7 and Cin == 2:
        assert used == [7], used                  # the two-channel bandwidth kernel really ran


@pytest.mark.parametrize('wt', [False, True])
@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
def test_conv2d_forward_over_the_kernel_row_major_image(wt, dtype, monkeypatch):
    """Round 5: the dense Conv2d layers' forward reads a fragment image whose channels are ordered (kernel row, channel)
    (RtgConv1dDesc.h_mode 2, RtgPackJob.kh_major; rtg/bank.py: FWD_KH_MAJOR).  Checked: the h_mode-2 instances really run
    for every dense layer shape of StftDiscriminator; the same layer built with the (channel, kernel row) image
    (FWD_KH_MAJOR off: h_mode 0) gives the same output up to summation order (fp32: 1e-5 of the output's scale; bf16
    operands: identical products, fp32 sums in another order)."""
    import ctypes as C
    import hparam as hp
    from models.layers import WNConv, BankedModel, conv
    from rtg import bank, ops
    from rtg.lib import lib
    monkeypatch.setattr(hp, 'compute_dtype', dtype)
    monkeypatch.setattr(hp, 'bf16_maps', False)
    seen = []
    run0 = ops._run_conv

    def spy(d, args, flop, label, what):
        run0(d, args, flop, label, what)
        seen.append((d.h_mode, lib.rtg_conv1d_variant(C.byref(d))))
    monkeypatch.setattr(ops, '_run_conv', spy)
    for case in CASES[3:7]:
        B, Cin, Cout, H, W, k, s, p = case

        class Net(BankedModel):
            def __init__(self):
                super().__init__()
                self.c = WNConv('conv2d', Cin, Cout, k, stride=s, pad=p, wt=wt)

            def forward(self, x):
                return conv(self.token(), self.c, x, pre_slope=0.15)

        outs = []
        for khc in (True, False):
            monkeypatch.setattr(bank, 'FWD_KH_MAJOR', khc)
            torch.manual_seed(5)
            net = Net().to(DEV)
            x = torch.randn(B, Cin, H, W, device=DEV)
            if wt:
                x = x.transpose(2, 3).contiguous()
            del seen[:]
            with torch.no_grad():
                outs.append(net(x).float())
            torch.cuda.synchronize()
            assert net.bank().layers[0].fwd_khc == int(khc)
            assert seen and seen[-1][0] == (2 if khc else 0), seen
            if khc:
                assert seen[-1][1] > 8000, seen                  # a dense-layer block shape
        scale = outs[1].abs().max().item()
        err = (outs[0] - outs[1]).abs().max().item()
        assert err <= (1e-5 if dtype == 'fp32' else 2e-5) * scale, (case, err, scale)


def _stats(t):
    t = t.detach().double().cpu()
    return np.array([t.sum().item(), t.abs().mean().item()])


def _nets(oracle):
    from models import MultiStftDiscriminator
    torch.manual_seed(1)
    mtd, omtd = MultiStftDiscriminator(), oracle.MTD()
    oracle.det_fill(mtd)
    oracle.det_fill(omtd)
    mtd.to(DEV).train()
    omtd.train()
    return mtd, omtd


def test_spec_tensors_match_reference_modulo_phase_wrap(oracle, gold):
    """[log|D+1e-9|, angle(D)/PI] against the oracle.  Frame 0 of a centred, reflect-padded STFT is symmetric about its
    centre, so its spectrum is real up to rounding: the REFERENCE's own angle() there is +pi or -pi by rounding noise
    (float32 and float64 CPU runs disagree).  The phase is therefore compared modulo 2 (units of pi)."""
    from models import multi_stft_loss
    _, _, y = oracle.golden_inputs()
    yd = torch.from_numpy(gold['y_hat'])
    S, Sg = multi_stft_loss(y.to(DEV), yd.to(DEV), ret_specs=True)
    oS, oSg = oracle.multi_stft_loss(y, yd, ret_specs=True)
    for a, b in zip(S + Sg, oS + oSg):
        a = a.cpu()
        big = b[:, 0] > -5.0                                   # |D| > e^-5: phase well conditioned
        np.testing.assert_allclose(a[:, 0][big].numpy(), b[:, 0][big].numpy(), rtol=0, atol=2e-4)
        d = (a[:, 1] - b[:, 1]).abs()
        d = torch.minimum(d, 2 - d)
        assert d[big].max().item() < 2e-3
        assert (d > 1e-2).float().mean().item() < 1e-4


def test_mtd_stack_matches_oracle_on_the_same_spectra(oracle, gold):
    """MTD forward, D-side and G-side backward against the CPU oracle fed with the SAME (GPU-produced) spectra."""
    from models import multi_stft_loss, discriminator_loss, generator_loss, feature_loss
    mtd, omtd = _nets(oracle)
    _, _, y = oracle.golden_inputs()
    yd = torch.from_numpy(gold['y_hat'])
    S, Sg = multi_stft_loss(y.to(DEV), yd.to(DEV), ret_specs=True)
    cS, cSg = [s.cpu() for s in S], [s.cpu().requires_grad_(True) for s in Sg]
    gSg = [s.detach().clone().requires_grad_(True) for s in Sg]
    lr, lg, fr, fg = mtd(S, gSg)
    olr, olg, ofr, ofg = omtd(cS, cSg)
    for a, b in zip(lr + lg, olr + olg):
        np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().numpy(), rtol=1e-3, atol=2e-5)
    shapes = np.array([list(f.shape) for fl in fr for f in fl])
    assert (shapes == gold['mtd_fmap_shapes']).all()                     # discrminator.py:268-290
    loss = discriminator_loss(lr, lg) + generator_loss(lg, lr) + 2 * feature_loss(fr, fg)
    oloss = oracle.discriminator_loss(olr, olg) + oracle.generator_loss(olg, olr) + 2 * oracle.feature_loss(ofr, ofg)
    np.testing.assert_allclose(loss.item(), oloss.item(), rtol=1e-5)
    mtd.zero_grad(); omtd.zero_grad()
    loss.backward(); oloss.backward()
    torch.cuda.synchronize()
    op = dict(omtd.named_parameters())
    for n, p in mtd.named_parameters():
        err = (p.grad.cpu() - op[n].grad).norm().item() / (op[n].grad.norm().item() + 1e-20)
        assert err < 2e-3, (n, err)
    for a, b in zip(gSg, cSg):
        err = (a.grad.cpu() - b.grad).norm().item() / b.grad.norm().item()
        assert err < 2e-3, err
    # reference's own loss values: the phase of frame 0 is rounding noise in the reference (see the test above), which
    # moves these sums by a few 1e-3 relative
    np.testing.assert_allclose(discriminator_loss(lr, lg).item(), gold['mtd_d_loss'], rtol=2e-2)
    np.testing.assert_allclose(feature_loss(fr, fg).item(), gold['mtd_fm_loss'], rtol=2e-2)


def test_stft_backward_through_log_magnitude_and_phase(oracle, gold):
    """d/dy of <cotangent, [log|D+1e-9|, angle(D)/PI]> and of the mel output against CPU autograd (float64)."""
    from audio import stft_mel_spec
    yd = torch.from_numpy(gold['y_hat']).squeeze(1)
    gen = torch.Generator().manual_seed(9)
    for n_fft, win, hop in oracle.STFT_PARAMS:
        yh = yd.clone().to(DEV).requires_grad_(True)
        mel, spec = stft_mel_spec(yh, n_fft, win, hop, True)
        cs = torch.randn(spec.shape, generator=gen)
        cm = torch.randn(mel.shape, generator=gen)
        ((spec * cs.to(DEV)).sum() + (mel * cm.to(DEV)).sum()).backward()
        y64 = yd.double().requires_grad_(True)
        S, M, P = oracle.stft_mag_mel_phase(y64, n_fft, win, hop)
        o = torch.stack([torch.log(S), P / oracle.PI], dim=1)
        ((o * cs.double()).sum() + (M * cm.double()).sum()).backward()
        err = (yh.grad.cpu().double() - y64.grad).norm().item() / y64.grad.norm().item()
        assert err < 5e-3, (n_fft, err)      # fp32 vs float64: d angle / dD ~ 1/|D| amplifies rounding at weak bins


@pytest.mark.parametrize('n_fft,win,hop', [(128, 64, 16), (256, 128, 30), (4096, 2048, 480), (4096, 4096, 1024)])
def test_stft_sizes_outside_the_reference_defaults(oracle, gold, n_fft, win, hop):
    """multi_stft_params is user-editable (hparam.py:78-80) and torch.stft takes any size: the kernel serves every power of
    two in 128 .. 4096 (8 frames per block .. two butterflies per thread) — magnitudes, phases, mel and d/dy against float64."""
    from audio import stft_mel_spec
    yd = torch.from_numpy(gold['y_hat']).squeeze(1)
    gen = torch.Generator().manual_seed(n_fft)
    yh = yd.clone().to(DEV).requires_grad_(True)
    mel, spec = stft_mel_spec(yh, n_fft, win, hop, True)
    y64 = yd.double().requires_grad_(True)
    S, M, P = oracle.stft_mag_mel_phase(y64, n_fft, win, hop)
    assert tuple(spec.shape) == (yd.shape[0], 2, n_fft // 2 + 1, 1 + yd.shape[1] // hop)
    big = S > 1e-2
    np.testing.assert_allclose(spec[:, 0].detach().cpu().double()[big].numpy(), torch.log(S)[big].detach().numpy(), rtol=0, atol=2e-4)
    d = (spec[:, 1].detach().cpu().double() - P.detach() / oracle.PI).abs()
    d = torch.minimum(d, 2 - d)
    assert d[big].max().item() < 2e-3
    np.testing.assert_allclose(mel.detach().cpu().numpy(), M.detach().numpy(), rtol=2e-4, atol=2e-6)
    cs = torch.randn(spec.shape, generator=gen)
    cs[:, 1] *= big.float()       # (d angle / dD ~ 1 / |D|: a phase cotangent at a weak bin measures fp32 rounding, not the kernel)
    cm = torch.randn(mel.shape, generator=gen)
    ((spec * cs.to(DEV)).sum() + (mel * cm.to(DEV)).sum()).backward()
    o = torch.stack([torch.log(S), P / oracle.PI], dim=1)
    ((o * cs.double()).sum() + (M * cm.double()).sum()).backward()
    err = (yh.grad.cpu().double() - y64.grad).norm().item() / y64.grad.norm().item()
    assert err < 5e-3, (n_fft, err)


@pytest.mark.parametrize('want_spec', [False, True])
def test_all_resolutions_in_one_launch_equal_one_launch_each(gold, want_spec):
    """audio.multi_stft_mel_spec (rtg_stft_forward_multi / rtg_stft_backward_multi: the three resolutions of multi_stft_loss on the
    generated and the real wave in one launch, one frame launch + one overlap-add backward) against stft_mel_spec per
    resolution: the same bits forward and per resolution backward."""
    import hparam as hp
    from audio import stft_mel_spec, multi_stft_mel_spec
    yd = torch.from_numpy(gold['y_hat']).squeeze(1).to(DEV)
    yr = torch.flip(yd, dims=[1]).contiguous()
    gen = torch.Generator().manual_seed(5)
    y1 = yd.clone().requires_grad_(True)
    y2 = yd.clone().requires_grad_(True)
    (mels, specs), (mels_r, specs_r) = multi_stft_mel_spec(y1, hp.multi_stft_params, want_spec, y_real=yr)
    assert all(not t.requires_grad for t in mels_r) and all(t.requires_grad for t in mels)
    tot1 = tot2 = 0
    for i, (n_fft, win, hop) in enumerate(hp.multi_stft_params):
        m, sp = stft_mel_spec(y2, n_fft, win, hop, want_spec)
        with torch.no_grad():
            mr, spr = stft_mel_spec(yr, n_fft, win, hop, want_spec)
        assert torch.equal(m, mels[i]) and torch.equal(mr, mels_r[i])
        cm = torch.randn(m.shape, generator=gen).to(DEV)
        tot1 = tot1 + (mels[i] * cm).sum()
        tot2 = tot2 + (m * cm).sum()
        if want_spec:
            assert specs[i].shape == sp.shape and torch.equal(sp, specs[i]) and torch.equal(spr, specs_r[i])
            cs = torch.randn(sp.shape, generator=gen).to(DEV)
            tot1 = tot1 + (specs[i] * cs).sum()
            tot2 = tot2 + (sp * cs).sum()
        else:
            assert specs[i] is None and specs_r[i] is None
    tot1.backward()
    tot2.backward()
    # (one launch adds a sample's three contributions in resolution order, autograd adds the three launches' tensors in its own:
    # equal up to the rounding of two additions)
    np.testing.assert_allclose(y1.grad.cpu().numpy(), y2.grad.cpu().numpy(), rtol=1e-5, atol=1e-5 * y2.grad.abs().max().item())
    # a subset of the resolutions carrying a gradient, and the generated wave alone (the real one's spectra cached)
    y3 = yd.clone().requires_grad_(True)
    y4 = yd.clone().requires_grad_(True)
    mels3, _ = multi_stft_mel_spec(y3, hp.multi_stft_params, want_spec)
    n_fft, win, hop = hp.multi_stft_params[1]
    m4, _ = stft_mel_spec(y4, n_fft, win, hop, want_spec)
    mels3[1].sum().backward()
    m4.sum().backward()
    assert torch.equal(y3.grad, y4.grad)


def test_stft_size_that_is_not_served_fails_loudly():
    from audio import stft_mel_spec
    from rtg.lib import RtgError
    with pytest.raises(RtgError):
        stft_mel_spec(torch.zeros(1, 8192, device=DEV), 768, 384, 96, True)


@pytest.mark.parametrize('case', [CASES[1], CASES[3], CASES[4], (2, 32, 64, 9, 65, (3, 9), (1, 1), (1, 4))])
def test_conv2d_every_block_shape(case, monkeypatch):
    """2-D layers through every block shape the library lists for their forward and backward-data problems (whole-clip
    packing and the dense windows over the (item, row, q) column sequence): all agree with torch and with each other."""
    import ctypes as C
    from models.layers import WNConv, BankedModel, conv
    from rtg import tune
    from rtg.lib import lib
    B, Cin, Cout, H, W, k, s, p = case

    class Net(BankedModel):
        def __init__(self):
            super().__init__()
            self.c = WNConv('conv2d', Cin, Cout, k, stride=s, pad=p)

        def forward(self, x):
            return conv(self.token(), self.c, x, pre_slope=0.15)

    torch.manual_seed(11)
    net = Net()
    x = torch.randn(B, Cin, H, W)
    v = net.c.weight_v.detach().double()
    g = net.c.weight_g.detach().double()
    w = g * v / v.flatten(1).norm(dim=1).reshape(-1, 1, 1, 1)
    xr = x.double().requires_grad_(True)
    y = F.conv2d(F.leaky_relu(xr, 0.15), w, net.c.bias.detach().double(), s, p)
    dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(12))
    y.backward(dy.double())
    net.to(DEV)

    # candidate lists of the forward and the backward-data problem: record the descriptors of one ordinary pass
    seen = []
    monkeypatch.setattr(tune, 'conv_cfg', lambda d, launch: (seen.append(bytes(d)), 0)[1])
    xg = x.to(DEV).requires_grad_(True)
    net(xg).backward(dy.to(DEV))
    assert len(seen) == 2
    from rtg.lib import Conv1dDesc
    lists = []
    for raw in seen:
        d = Conv1dDesc.from_buffer_copy(raw)
        cands = (C.c_int * 32)()
        n = lib.rtg_conv1d_tile_candidates(C.byref(d), cands, 32)
        assert n >= 1
        lists.append(list(cands[:n]))
    outs, dxs = [], []
    for i in range(max(len(lists[0]), len(lists[1]))):
        pick = [lists[0][min(i, len(lists[0]) - 1)], lists[1][min(i, len(lists[1]) - 1)]]
        calls = []
        monkeypatch.setattr(tune, 'conv_cfg', lambda d, launch: (calls.append(1), pick[len(calls) - 1])[1])
        xg = x.to(DEV).requires_grad_(True)
        out = net(xg)
        out.backward(dy.to(DEV))
        torch.cuda.synchronize()
        for got, ref, name in ((out, y.detach(), 'out'), (xg.grad, xr.grad, 'dx')):
            err = (got.detach().cpu().double() - ref).abs().max().item()
            assert err <= 2e-4 * ref.abs().max().item() + 1e-5, (name, pick, err)
        outs.append(out.detach().cpu())
        dxs.append(xg.grad.detach().cpu())
    if W == 65:       # rows of 65 positions: half of a 128-column tile would be lost to whole-clip packing
        assert any(c >= 1000 for l in lists for c in l), 'no dense-window candidate exercised'
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
