"""Host-side logic that needs no GPU: reference-shaped construction API, state-dict keys, loud failure on CPU tensors,
mel filterbank / STFT plans, optimizer state-dict format."""
import numpy as np
import pytest
import torch


def test_models_star_import_surface():
    import models
    for name in ('Generator_RefineGAN_small', 'MultiScaleDiscriminator', 'MultiPeriodDiscriminator',
                 'MultiStftDiscriminator', 'multi_stft_loss', 'dynamic_loss', 'envelope_loss', 'strip_mirror_loss',
                 'discriminator_loss', 'generator_loss', 'feature_loss', 'get_param_cnt', 'scan_checkpoint',
                 'load_checkpoint', 'save_checkpoint', 'LRELU_SLOPE', 'PI', 'hp', 'torch', 'F', 'get_padding',
                 'get_same_padding', 'init_weights'):
        assert hasattr(models, name), name
    import hparam as hp
    assert hp.segment_size == 8192 and hp.generator_ver == 'RefineGAN_small' and hp.d_train_times == 2
    assert hp.multi_stft_params == [(2048, 1024, 240), (1024, 512, 120), (512, 256, 60)]
    assert (hp.learning_rate_g, hp.learning_rate_d, hp.adam_b1, hp.adam_b2) == (1.8e-4, 2e-4, 0.8, 0.99)
    assert models.get_padding(7, 9) == 27 and models.get_same_padding(3, 9) == 9


def test_construction_matches_reference_under_the_same_seed(gold):
    """Same key set, parameter counts and (same RNG stream) identical initial values as the reference's modules."""
    from models import (Generator_RefineGAN_small, MultiScaleDiscriminator, MultiPeriodDiscriminator,
                        MultiStftDiscriminator, get_param_cnt)
    torch.manual_seed(114514)
    nets = Generator_RefineGAN_small(), MultiScaleDiscriminator(), MultiPeriodDiscriminator(), MultiStftDiscriminator()
    for tag, m in zip(('g', 'msd', 'mpd', 'mtd'), nets):
        assert get_param_cnt(m) == int(gold[f'init_{tag}_count'])
        assert sorted(m.state_dict().keys()) == list(gold[f'init_{tag}_keys'])
        st = np.stack([[p.double().sum().item(), p.double().abs().mean().item()]
                       for _, p in sorted(m.named_parameters())])
        np.testing.assert_allclose(st, gold[f'init_{tag}_stats'], rtol=1e-6, atol=1e-7)
    sd = nets[0].state_dict()
    assert sd['ups.0.weight_g'].shape == (256, 1, 1) and sd['ups.0.weight_v'].shape == (256, 128, 15)
    assert sd['noise.w'].item() == pytest.approx(1e-6)
    sd = nets[2].state_dict()
    assert sd['discriminators.0.convs.1.weight_v'].shape == (128, 32, 5, 1)      # Conv2d-shaped, as in the reference


def test_hot_path_refuses_cpu_tensors_loudly():
    from models import Generator_RefineGAN_small, MultiScaleDiscriminator, multi_stft_loss
    from rtg.lib import RtgError
    g = Generator_RefineGAN_small()
    with pytest.raises(RtgError):
        g(torch.zeros(1, 80, 32), torch.zeros(1, 1, 8192))
    with pytest.raises(RtgError):
        MultiScaleDiscriminator()(torch.zeros(1, 1, 8192), torch.zeros(1, 1, 8192))
    with pytest.raises(RtgError):
        multi_stft_loss(torch.zeros(1, 1, 8192), torch.zeros(1, 1, 8192), ret_loss=True)


def test_mel_filterbank_and_stft_plan(oracle, gold):
    from audio import mel_filterbank, get_plan, mel_basis
    for n_fft in (2048, 1024, 512):
        fb = mel_filterbank(22050, n_fft, 80, 125, 7600)
        assert fb.dtype == np.float32
        np.testing.assert_allclose(fb.astype(np.float64).sum(), gold[f'melbasis{n_fft}_sum'], rtol=1e-7)
        np.testing.assert_array_equal(fb, oracle.mel_filterbank(n_fft))
    assert mel_basis.shape == (80, 1025)
    plan = get_plan(2048, 1024, 240)
    t = plan._host
    fb = plan.fb
    # band tables (forward) and per-bin pairs (backward) both reproduce the dense filterbank
    dense = np.zeros_like(fb)
    for m in range(80):
        lo, ln, off = int(t['mel_lo'][m]), int(t['mel_len'][m]), int(t['mel_woff'][m])
        dense[m, lo:lo + ln] = t['mel_w'][off:off + ln].numpy()
    np.testing.assert_array_equal(dense, fb)
    dense2 = np.zeros_like(fb)
    idx, w = t['binmel_idx'].numpy(), t['binmel_w'].numpy()
    for f in range(fb.shape[1]):
        for j in range(2):
            if idx[f, j] >= 0:
                dense2[idx[f, j], f] = w[f, j]
    np.testing.assert_array_equal(dense2, fb)
    np.testing.assert_array_equal(t["window"].numpy(), torch.hann_window(1024, periodic=True).numpy())


def test_exponential_lr_and_loss_switches():
    import train
    import hparam as hp

    class FakeOpt:
        initial_lr = 2e-4
        param_groups = [{'lr': 2e-4}]

    sch = train.ExponentialLR(FakeOpt(), gamma=hp.lr_decay)
    for _ in range(3):
        sch.step()
    assert sch.get_last_lr()[0] == pytest.approx(2e-4 * 0.999 ** 3)
    from models import envelope_loss, strip_mirror_loss
    from rtg.lib import RtgError
    with pytest.raises(RtgError):
        envelope_loss(torch.zeros(1, 1, 320), torch.zeros(1, 1, 320))
    with pytest.raises(RtgError):
        strip_mirror_loss(torch.zeros(1, 1, 320))
