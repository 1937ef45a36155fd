"""Host-side data path (transtacos-retunegan_amd/audio.py host block, data.py).  librosa is absent, so
the restated algorithms are checked through their defining properties (parity unpinned, see the module headers) and
the Dataset contract (file format, cache, crop, shapes, sharding) directly.  CPU only."""
import os

import numpy as np
import torch


def test_stft_matches_torch_and_inverts():
    import audio as A
    rng = np.random.RandomState(0)
    y = ((rng.rand(22016) * 2 - 1) * 0.3).astype(np.float32)
    D = A.stft_np(y)
    assert D.shape == (1025, 87)
    Dt = torch.stft(torch.from_numpy(y), 2048, 256, 1024, torch.hann_window(1024), center=True, pad_mode='reflect',
                    return_complex=True).numpy()
    np.testing.assert_allclose(D, Dt, atol=2e-5)
    np.testing.assert_allclose(A.istft_np(D, length=len(y)), y, atol=1e-6)
    assert A.get_mag(y).shape == (1025, 87) and A.get_mag(y).min() >= np.log(1e-5) - 1e-6


def test_griffin_lim_properties():
    import audio as A
    rng = np.random.RandomState(1)
    t = np.arange(22016) / 22050.0
    y = (0.4 * np.sin(2 * np.pi * 220 * t) + 0.2 * np.sin(2 * np.pi * 1760 * t) + 0.01 * rng.randn(len(t))).astype(np.float32)
    S = np.abs(A.stft_np(y))
    w = A._griffinlim(S ** (1 / 1.2), wavlen=len(y))          # hp.gl_power = 1.2 is applied inside
    assert w.shape == y.shape and w.dtype == np.float32
    np.testing.assert_array_equal(w, A._griffinlim(S ** (1 / 1.2), wavlen=len(y)))     # random_state=hp.randseed
    # consistency: the magnitude of the rebuilt wave's STFT is much closer to S than that of a random-phase inverse
    sc = lambda v: np.linalg.norm(np.abs(A.stft_np(v)) - S) / np.linalg.norm(S)
    ang = np.exp(2j * np.pi * np.random.RandomState(2).rand(*S.shape))
    assert sc(w) < 0.5 * sc(A.istft_np(S * ang, length=len(y))) and sc(w) < 0.35
    # inv_mag: exp() of the log magnitude, a 1024-bin input gets a zero DC row
    mag = np.log(S.clip(min=1e-5))
    assert A.inv_mag(mag, wavlen=len(y)).shape == y.shape
    np.testing.assert_allclose(A.inv_mag(mag[1:], wavlen=len(y)),
                               A._griffinlim(np.concatenate([np.zeros((1, S.shape[1])), np.exp(mag[1:])]), len(y)), atol=1e-6)


def test_spec_to_natural_scale_known_values():
    import audio as A
    # [-4, 4] -> [-100, 0] dB -> + 20 dB -> amplitude (transtacos/audio.py:80-82,195-196)
    np.testing.assert_allclose(A.spec_to_natural_scale(np.array([4.0, -4.0, 0.0])), [10.0, 1e-4, 10 ** (-30 / 20)], rtol=1e-12)
    assert A.align_wav(np.zeros(1000)).shape == (1024,) and A.align_wav(np.zeros(1024)).shape == (1024,)
    m = A.augment_spec(np.random.RandomState(0).rand(80, 40).astype(np.float32), rounds=5, rng=np.random.RandomState(3))
    assert m.shape == (80, 40) and m.dtype == np.float32


def _make_corpus(tmp, n=3, seconds=(1.5, 0.3, 2.0)):
    from scipy.io import wavfile
    import audio as A
    wav_dir, data_dp = os.path.join(tmp, 'wavs'), os.path.join(tmp, 'prep')
    os.makedirs(wav_dir); os.makedirs(data_dp)
    names = [f'{i:06d}' for i in range(n)]
    rng = np.random.RandomState(0)
    for name, sec in zip(names, seconds):
        t = np.arange(int(22050 * sec)) / 22050.0
        y = (0.5 * np.sin(2 * np.pi * 200 * t) * (t > 0.05) + 0.001 * rng.randn(len(t))).astype(np.float32)
        wavfile.write(os.path.join(wav_dir, name + '.wav'), 22050, (y * 32767).astype(np.int16))
    open(os.path.join(data_dp, 'wav_path.txt'), 'w').write(wav_dir)
    for split in ('train', 'test'):
        open(os.path.join(data_dp, split + '.txt'), 'w', encoding='utf-8').write(
            ''.join(f'{nm}|#1 #2|ni3 hao3\n' for nm in names))
    return data_dp, names


def test_dataset_contract(tmp_path):
    import data as D
    import audio as A
    data_dp, names = _make_corpus(str(tmp_path))
    ds = D.Dataset('train', data_dp)
    assert len(ds) == 3
    mel, tmpl, wav = ds[0]                                   # long utterance: random 8192-sample crop
    assert mel.shape == (80, 32) and tmpl.shape == (8192,) and wav.shape == (8192,)
    assert all(a.dtype == np.float32 for a in (mel, tmpl, wav)) and ds.data[0] is not None
    mel, tmpl, wav = ds[1]                                   # short utterance: padded
    assert mel.shape == (80, 32) and wav.shape == (8192,) and np.all(wav[-100:] == 0)
    full = D.Dataset('test', data_dp)[2]                     # evaluation: full length, aligned to the hop
    assert full[2].shape[0] % 256 == 0 and full[0].shape[1] * 256 == full[2].shape[0] == full[1].shape[0]
    x, y_tmpl, y = D.collate([ds[0], ds[1]])
    assert x.shape == (2, 80, 32) and y_tmpl.shape == (2, 1, 8192) and y.shape == (2, 1, 8192)
    # finetune mode: the acoustic model's normalised spectrogram on disk drives mel and the Griffin-Lim reference
    for nm in names:
        wav = A.align_wav(D.trim_silence(D.load_wav(os.path.join(str(tmp_path), 'wavs', nm + '.wav'))))
        S = np.abs(A.stft_np(wav[:-1]))
        db = 20 * np.log10(np.maximum(1e-5, S)) - 20
        np.save(os.path.join(data_dp, f'mag-{nm}.npy'), np.clip(2 * 4 * ((db + 100) / 100) - 4, -4, 4).astype(np.float32))
    ft = D.Dataset('test', data_dp, finetune=True)
    mel, tmpl, wav = ft[0]
    assert mel.shape[0] == 80 and mel.shape[1] * 256 == len(wav) == len(tmpl) and np.isfinite(tmpl).all()
    # sharding: disjoint, equal-size, covering
    sh = [D.shard_indices(10, r, 4) for r in range(4)]
    assert all(len(s) == 3 for s in sh) and set(sum(sh, [])) == set(range(10))
    assert len(D.Dataset('train', data_dp, rank=1, world=2)) == 2


def test_trim_silence():
    import data as D
    y = np.concatenate([np.zeros(4000), 0.5 * np.sin(np.arange(8000) * 0.1), np.zeros(4000)]).astype(np.float32)
    t = D.trim_silence(y)
    assert 7900 <= len(t) <= 8700 and abs(t).max() > 0.49


# ---------------------------------------------------------------------------------------------------------------
# Round 4: what CAN be pinned of SURVEY.md 8 f1 without librosa — the transform pair against an independent implementation
# present in the image (torch.stft / torch.istft, fp64), and the fast Griffin-Lim recursion of retunegan/audio.py:131-136
# (n_iter 4, momentum 0.7, power 1.2, random phases from seed 114514) against a naive restatement that shares no code with
# audio.py (explicit DFT sums, explicit frames, explicit overlap-add).  librosa itself stays unpinned (absent).
# ---------------------------------------------------------------------------------------------------------------
import pytest


@pytest.mark.parametrize('n_fft,hop,win,n', [(2048, 256, 1024, 22016), (1024, 120, 600, 8192), (512, 50, 240, 8192),
                                              (2048, 256, 1024, 8191), (64, 16, 32, 500)])
def test_stft_pair_against_torch_fp64(n_fft, hop, win, n):
    import audio as A
    rng = np.random.RandomState(n_fft + n)
    y = ((rng.rand(n) * 2 - 1) * 0.5).astype(np.float32)
    window = torch.hann_window(win, periodic=True, dtype=torch.float64)
    Dt = torch.stft(torch.from_numpy(y).double(), n_fft, hop, win, window, center=True, pad_mode='reflect',
                    normalized=False, onesided=True, return_complex=True).numpy()
    D = A.stft_np(y, n_fft, hop, win)
    assert D.shape == Dt.shape == (n_fft // 2 + 1, 1 + n // hop) and D.dtype == np.complex64
    # complex64 output of an fp64 computation: error = one fp32 rounding of values up to ~|D|max
    assert np.abs(D - Dt).max() <= 2e-7 * np.abs(Dt).max() + 1e-7
    # the inverse on an ARBITRARY (inconsistent) spectrogram, with and without a target length — torch.istft states the
    # same operator (irfft, synthesis window, overlap-add, division by the window's squared-sum envelope, n_fft/2 trimmed)
    Z = (rng.randn(*Dt.shape) + 1j * rng.randn(*Dt.shape))
    Z[0].imag = 0; Z[-1].imag = 0
    for length in (None, n, n - 37, hop * (Dt.shape[1] - 1) + 11):
        it = torch.istft(torch.from_numpy(Z), n_fft, hop, win, window, center=True, length=length).numpy()
        ours = A.istft_np(Z.astype(np.complex64), hop, win, length=length)
        assert ours.shape == it.shape and ours.dtype == np.float32
        # positions past the last frame's support are zero-filled by both; the envelope division amplifies the complex64
        # rounding of the input near the edges where the envelope is small
        scale = np.abs(it).max()
        assert np.abs(ours - it).max() <= 3e-6 * scale, (length, np.abs(ours - it).max() / scale)
    # and the round trip on a consistent spectrogram
    np.testing.assert_allclose(A.istft_np(D, hop, win, length=n), y, atol=2e-6)


def _naive_fast_griffin_lim(S, hop, win, n_iter, momentum, seed, length):
    """librosa-0.8.1-style fast Griffin-Lim (Perraudin, Balazs, Sondergaard 2013: c_n = P_C1(P_C2(t_{n-1})), t_n = c_n +
    alpha (c_n - c_{n-1}); librosa projects `rebuilt - momentum / (1 + momentum) * previous rebuilt` onto unit phases) in
    plain fp64 loops: DFT by explicit sums, frames by explicit indexing, overlap-add by explicit accumulation."""
    F, T = S.shape
    N = 2 * (F - 1)
    w = np.zeros(N)
    lp = (N - win) // 2
    w[lp:lp + win] = [0.5 - 0.5 * np.cos(2 * np.pi * i / win) for i in range(win)]
    Wf = np.array([[np.exp(-2j * np.pi * k * n / N) for n in range(N)] for k in range(F)])        # rfft
    Wi = np.array([[np.exp(2j * np.pi * k * n / N) for k in range(N)] for n in range(N)]) / N      # ifft

    def istft(D):
        full = np.concatenate([D, np.conj(D[-2:0:-1])], axis=0)                    # hermitian extension
        y, env = np.zeros(N + hop * (T - 1)), np.zeros(N + hop * (T - 1))
        for t in range(T):
            fr = (Wi @ full[:, t]).real * w
            for n in range(N):
                y[t * hop + n] += fr[n]
                env[t * hop + n] += w[n] ** 2
        y = np.where(env > 1e-30, y / np.maximum(env, 1e-30), y)[N // 2:]
        return np.concatenate([y, np.zeros(max(0, length - len(y)))])[:length]

    def stft(y):
        yp = np.concatenate([y[N // 2:0:-1], y, y[-2:-N // 2 - 2:-1]])            # reflect padding
        return np.stack([Wf @ (w * yp[t * hop:t * hop + N]) for t in range(1 + (len(yp) - N) // hop)], axis=1)

    ang = np.exp(2j * np.pi * np.random.RandomState(seed).rand(F, T))
    prev = 0.0
    for _ in range(n_iter):
        reb = stft(istft(S * ang))
        ang = reb - momentum / (1 + momentum) * prev
        ang = ang / (np.abs(ang) + 1e-16)
        prev = reb
    return istft(S * ang)


def test_fast_griffin_lim_recursion_against_naive_restatement(monkeypatch):
    import audio as A
    import hparam as hp
    # a 3-frame case small enough for O(N^2) sums: n_fft 16, window 12, hop 6 -> 9 bins x 3 frames, 12 samples
    monkeypatch.setattr(hp, 'hop_length', 6)
    monkeypatch.setattr(hp, 'win_length', 12)
    assert (hp.gl_iters, hp.gl_momentum, hp.gl_power, hp.randseed) == (4, 0.7, 1.2, 114514)     # audio.py:131-136 / hparam
    rng = np.random.RandomState(7)
    S = np.abs(rng.randn(9, 3)) + 0.1
    for length in (12, 15):
        got = A._griffinlim(S, wavlen=length)
        want = _naive_fast_griffin_lim(S ** 1.2, 6, 12, 4, 0.7, 114514, length)
        assert got.shape == want.shape == (length,)
        np.testing.assert_allclose(got, want, atol=2e-5 * np.abs(want).max())
    # the momentum term matters at this tolerance (a plain Griffin-Lim differs visibly): the test can tell them apart
    plain = _naive_fast_griffin_lim(S ** 1.2, 6, 12, 4, 0.0, 114514, 12)
    assert np.abs(plain - A._griffinlim(S, wavlen=12)).max() > 1e-3 * np.abs(plain).max()
    # five-frame case, longer clip
    S5 = np.abs(rng.randn(9, 5)) + 0.05
    want = _naive_fast_griffin_lim(S5 ** 1.2, 6, 12, 4, 0.7, 114514, 24)
    np.testing.assert_allclose(A._griffinlim(S5, wavlen=24), want, atol=2e-5 * np.abs(want).max())
    # inv_mag on top: exp() of the log magnitude, a zero DC row for an (n_freq - 1)-bin input (audio.py:139-147)
    monkeypatch.setattr(hp, 'n_freq', 9)
    mag = np.log(S5[1:])
    want = _naive_fast_griffin_lim(np.concatenate([np.zeros((1, 5)), S5[1:]]) ** 1.2, 6, 12, 4, 0.7, 114514, 24)
    np.testing.assert_allclose(A.inv_mag(mag, wavlen=24), want, atol=2e-5 * np.abs(want).max())
