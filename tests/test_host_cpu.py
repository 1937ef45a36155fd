"""Host-side logic that needs no GPU: reference-shaped construction API, state-dict keys, loud failure on CPU tensors,
mel filterbank / STFT plans, optimizer state-dict format."""
import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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


def test_parameter_registration_order_is_the_references(oracle, gold, gold4):
    """torch.optim state dicts (the `do_*` checkpoints, train.py:263-273) index their entries by parameter position:
    the modules must yield their parameters in the order the reference's modules do (bias, weight_g, weight_v per
    weight-normed conv; module registration order of the constructors).  The order lists come from the real reference
    modules (oracle/gen_golden.py, gen_golden_f4.py)."""
    import models as M
    torch.manual_seed(0)
    for tag, prod, orc in (('g', M.Generator_RefineGAN_small, oracle.Generator), ('msd', M.MultiScaleDiscriminator, oracle.MSD),
                           ('mpd', M.MultiPeriodDiscriminator, oracle.MPD), ('mtd', M.MultiStftDiscriminator, oracle.MTD)):
        ref = list(gold[f'init_{tag}_param_order'])
        assert [n for n, _ in prod().named_parameters()] == ref, tag
        assert [n for n, _ in orc().named_parameters()] == ref, tag
    ref = list(gold4['full_param_order'])
    assert [n for n, _ in M.Generator_RefineGAN().named_parameters()] == ref
    assert [n for n, _ in oracle.GeneratorFull().named_parameters()] == ref


def test_mel_filterbank_known_answer_from_librosa_docs():
    """A pin outside the shared restatement: the docstring example of librosa.filters.mel (librosa 0.8.1,
    `melfb = librosa.filters.mel(22050, 2048)`) prints `[[0., 0.016, ..., 0., 0.], [0., 0., ..., 0., 0.], ...]`:
    128 Slaney bands over 0..11025 Hz, element [0, 1] = 0.016 at 3 decimals, first and last columns zero."""
    from audio import mel_filterbank
    fb = mel_filterbank(22050, 2048, 128, 0.0, 11025.0)
    assert fb.shape == (128, 1025)
    assert np.round(fb[0, 1], 3) == np.float32(0.016)
    assert fb[0, 0] == 0 and fb[0, -1] == 0 and fb[0, -2] == 0 and fb[1, 0] == 0 and fb[1, 1] == 0
    assert fb[-1, 0] == 0 and fb[-1, -1] == 0
    # Slaney area normalisation: every band integrates to ~1 over frequency (bin width sr / n_fft), up to sampling
    area = fb.astype(np.float64).sum(1) * 22050 / 2048
    assert np.all(np.abs(area - 1.0) < 0.25) and abs(area.mean() - 1.0) < 0.01
    # the band edges follow the Slaney scale: linear below 1 kHz (200/3 Hz per mel), log above (step ln(6.4)/27)
    peak_hz = fb.argmax(1) * 22050 / 2048
    mel = np.where(peak_hz < 1000, peak_hz / (200 / 3), 15 + np.log(np.maximum(peak_hz, 1e-9) / 1000) / (np.log(6.4) / 27))
    step = np.diff(mel)
    assert np.all(np.abs(step - step.mean()) < 0.5 * step.mean() + 0.2)


def test_exponential_lr_matches_torch_fresh_and_resumed():
    """train.py:87-88: ExponentialLR(optim, gamma, last_epoch=-1) for a fresh run, last_epoch=<saved epoch> on resume.
    Compared against torch's own scheduler on a torch AdamW, including the extra step torch's constructor performs."""
    import train
    import hparam as hp
    from torch.optim.lr_scheduler import ExponentialLR as TorchLR

    w = torch.nn.Parameter(torch.zeros(3))
    topt = torch.optim.AdamW([w], hp.learning_rate_d, betas=[hp.adam_b1, hp.adam_b2])
    tsch = TorchLR(topt, gamma=hp.lr_decay, last_epoch=-1)
    opt = train.AdamW([], hp.learning_rate_d, betas=[hp.adam_b1, hp.adam_b2])
    sch = train.ExponentialLR(opt, gamma=hp.lr_decay)
    assert sch.get_last_lr() == tsch.get_last_lr() == [hp.learning_rate_d] and sch.last_epoch == tsch.last_epoch == 0
    for _ in range(3):
        topt.step(); tsch.step(); sch.step()
    assert sch.get_last_lr()[0] == tsch.get_last_lr()[0] == opt.param_groups[0]['lr']
    assert sch.get_last_lr()[0] == pytest.approx(2e-4 * 0.999 ** 3, rel=1e-12)
    # resume at epoch 3 from the saved optimizer state (param_groups carry lr and initial_lr)
    sd = topt.state_dict()
    topt2 = torch.optim.AdamW([w], hp.learning_rate_d, betas=[hp.adam_b1, hp.adam_b2])
    topt2.load_state_dict(sd)
    tsch2 = TorchLR(topt2, gamma=hp.lr_decay, last_epoch=3)
    opt2 = train.AdamW([], hp.learning_rate_d, betas=[hp.adam_b1, hp.adam_b2])
    opt2.load_state_dict({'state': {}, 'param_groups': sd['param_groups']})
    sch2 = train.ExponentialLR(opt2, gamma=hp.lr_decay, last_epoch=3)
    assert sch2.last_epoch == tsch2.last_epoch == 4
    assert sch2.get_last_lr()[0] == tsch2.get_last_lr()[0] == opt2.param_groups[0]['lr']
    topt2.step(); tsch2.step(); sch2.step()
    assert sch2.get_last_lr()[0] == tsch2.get_last_lr()[0]
    assert opt2.initial_lr == hp.learning_rate_d


def test_exponential_lr_legacy_resume(monkeypatch):
    """hparam.legacy_resume_lr: torch 1.8's constructor sent the loaded lr through get_lr() once more on resume (one extra
    factor of gamma per resume); a fresh run is unaffected."""
    import train
    import hparam as hp
    monkeypatch.setattr(hp, 'legacy_resume_lr', True, raising=False)
    opt = train.AdamW([], hp.learning_rate_d, betas=[hp.adam_b1, hp.adam_b2])
    sch = train.ExponentialLR(opt, gamma=hp.lr_decay)
    assert sch.get_last_lr() == [hp.learning_rate_d]
    lr3 = hp.learning_rate_d * hp.lr_decay ** 3
    opt2 = train.AdamW([], hp.learning_rate_d, betas=[hp.adam_b1, hp.adam_b2])
    opt2.load_state_dict({'state': {}, 'param_groups': [{'lr': lr3, 'initial_lr': hp.learning_rate_d}]})
    sch2 = train.ExponentialLR(opt2, gamma=hp.lr_decay, last_epoch=3)
    assert sch2.last_epoch == 4 and sch2.get_last_lr()[0] == pytest.approx(lr3 * hp.lr_decay, rel=1e-15)


def test_loss_switches_refuse_cpu():
    from models import envelope_loss, strip_mirror_loss
    from rtg.lib import RtgError
    with pytest.raises(RtgError):
        envelope_loss(torch.zeros(1, 1, 320), torch.zeros(1, 1, 320))
    with pytest.raises(RtgError):
        strip_mirror_loss(torch.zeros(1, 1, 320))


def _run_bench(args, env_extra=None, timeout=120):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'RTG_BENCH_REHEARSE')}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), *args], env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_bench_refuses_to_time_fewer_gpus_than_asked():
    """Round-4 verdict: `bench.py --gpus N` silently timed ONE GPU.  Outside a launcher it now starts N ranks itself and
    refuses when the node has fewer than N devices (this container has none); under a launcher --gpus must equal WORLD_SIZE.
    Both exits are non-zero with a message that names the numbers, and no JSON line is printed."""
    r = _run_bench(['--gpus', '2'])
    assert r.returncode != 0 and '--gpus 2' in r.stderr and 'GPU(s)' in r.stderr, (r.returncode, r.stderr[-400:])
    assert '"metric"' not in r.stdout
    r = _run_bench(['--gpus', '2'], {'WORLD_SIZE': '4', 'RANK': '0', 'LOCAL_RANK': '0'})
    assert r.returncode != 0 and 'WORLD_SIZE=4' in r.stderr, (r.returncode, r.stderr[-400:])
    assert '"metric"' not in r.stdout
