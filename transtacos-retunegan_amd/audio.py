"""On-device STFT path of retunegan/audio.py (`get_stft_torch`, :150-170) on the MI355X STFT kernel, plus the mel
filterbank it needs.  The host-side librosa DSP of the reference (load/save wav, augmentation, Griffin-Lim) is not on
the train-step hot path."""
import math

import numpy as np
import torch

import hparam as hp
from rtg import ops

eps = 1e-5
PI = 3.14159265358979


def mel_filterbank(sr, n_fft, n_mels=128, fmin=0.0, fmax=None):
    """The published Slaney-scale, area-normalised triangular filterbank that librosa 0.8.1's `filters.mel` builds
    (reference call sites: retunegan/audio.py:20,158, positional (sr, n_fft, n_mels, fmin, fmax)).  float32
    [n_mels, n_fft//2 + 1]."""
    if fmax is None:
        fmax = sr / 2.0
    lin_step, knee_hz = 200.0 / 3, 1000.0
    knee_mel, log_step = knee_hz / lin_step, math.log(6.4) / 27.0

    def to_mel(f):
        return f / lin_step if f < knee_hz else knee_mel + math.log(f / knee_hz) / log_step

    m = np.linspace(to_mel(float(fmin)), to_mel(float(fmax)), n_mels + 2)
    hz = np.where(m >= knee_mel, knee_hz * np.exp(log_step * (m - knee_mel)), lin_step * m)
    bins = np.linspace(0.0, sr / 2.0, 1 + n_fft // 2)
    width = np.diff(hz)
    rising = (bins[None, :] - hz[:-2, None]) / width[:-1, None]
    falling = (hz[2:, None] - bins[None, :]) / width[1:, None]
    fb = np.maximum(0.0, np.minimum(rising, falling)).astype(np.float32)
    fb *= (2.0 / (hz[2:] - hz[:-2]))[:, None]
    return fb


mel_basis = mel_filterbank(hp.sample_rate, hp.n_fft, n_mels=hp.n_mel, fmin=hp.fmin, fmax=hp.fmax)
mag_to_mel = lambda x: np.dot(mel_basis, x)  # noqa: E731  (audio.py:21)


class StftPlan:
    """Constant tables of one STFT resolution, built once on the host and cached per device:
    periodic hann window (torch.hann_window semantics), fp64-rounded twiddles, the mel filterbank as per-filter bands
    (forward) and per-bin (filter, weight) pairs (backward)."""

    def __init__(self, n_fft, win, hop, n_mel=None):
        self.n_fft, self.win, self.hop = n_fft, win, hop
        self.n_mel = hp.n_mel if n_mel is None else n_mel
        assert hp.window_fn == 'hann', 'only the hann window of hparam.py:36 is on the path'
        F = n_fft // 2 + 1
        k = np.arange(n_fft // 2, dtype=np.float64)
        tw = np.concatenate([np.cos(2 * np.pi * k / n_fft), np.sin(2 * np.pi * k / n_fft)]).astype(np.float32)
        window = getattr(torch, f'{hp.window_fn}_window')(win).numpy()      # exactly the tensor audio.py:155-156 builds
        fb = mel_filterbank(hp.sample_rate, n_fft, self.n_mel, hp.fmin, hp.fmax)
        self.fb = fb
        lo, ln, woff, wts = [], [], [], []
        for m in range(self.n_mel):
            nz = np.nonzero(fb[m])[0]
            a, b = (int(nz[0]), int(nz[-1]) + 1) if len(nz) else (0, 0)
            lo.append(a); ln.append(b - a); woff.append(len(wts)); wts.extend(fb[m, a:b].tolist())
        bidx = -np.ones((F, 2), dtype=np.int32)
        bw = np.zeros((F, 2), dtype=np.float32)
        for f in range(F):
            nz = np.nonzero(fb[:, f])[0]
            assert len(nz) <= 2, 'a bin is covered by at most two triangular filters'
            for j, m in enumerate(nz):
                bidx[f, j], bw[f, j] = m, fb[m, f]
        self._host = dict(window=torch.from_numpy(window), twiddle=torch.from_numpy(tw),
                          mel_lo=torch.tensor(lo, dtype=torch.int32), mel_len=torch.tensor(ln, dtype=torch.int32),
                          mel_woff=torch.tensor(woff, dtype=torch.int32),
                          mel_w=torch.tensor(wts if wts else [0.0], dtype=torch.float32),
                          binmel_idx=torch.from_numpy(bidx), binmel_w=torch.from_numpy(bw))
        self._dev = {}

    def tensors(self, device):
        key = str(device)
        if key not in self._dev:
            self._dev[key] = {k: v.to(device) for k, v in self._host.items()}
        return self._dev[key]


_plans = {}
mel_basis_torch = {}   # { n_fft: mel_basis } (name kept from audio.py:25)
window_fn_torch = {}   # { win_length: window }


def get_plan(n_fft, win_length, hop_length):
    key = (n_fft, win_length, hop_length)
    if key not in _plans:
        _plans[key] = StftPlan(n_fft, win_length, hop_length)
    return _plans[key]


def stft_mel_spec(y, n_fft, win_length, hop_length, want_spec=False):
    """y [B,T] -> (mel [B,80,frames], spec [B,2,F,frames] = stack(log|D+1e-9|, angle(D)/PI) or None)."""
    mel, spec = ops.StftFn.apply(y, get_plan(n_fft, win_length, hop_length), want_spec)
    if spec is not None and ops.SPEC_FREQ_MAJOR:
        spec = spec.transpose(2, 3)          # stored [B, 2, frames, F] (frequency contiguous); the map is its transposed view
    return mel, spec


def multi_stft_mel_spec(y, params, want_spec=False, y_real=None):
    """every (n_fft, win_length, hop_length) of `params` on y [B,T] — and on the constant y_real, if given — in one launch
    (ops.MultiStftFn).  -> (mels, specs) of y, or ((mels, specs), (mels_real, specs_real)); entries as stft_mel_spec's."""
    plans = [get_plan(*p) for p in params]
    n = len(plans)
    out = ops.MultiStftFn.apply(y, y_real, want_spec, *plans)

    def view(s):
        return s.transpose(2, 3) if (s is not None and ops.SPEC_FREQ_MAJOR) else s
    mine = (list(out[:n]), [view(s) for s in out[n:2 * n]])
    if y_real is None:
        return mine
    return mine, (list(out[2 * n:3 * n]), [view(s) for s in out[3 * n:4 * n]])


def get_stft_torch(y, n_fft, win_length, hop_length):
    """audio.py:150-170: returns (S, M, P) = (|D + 1e-9|, mel_basis @ S, angle(D)).  API-compatibility wrapper;
    the train step consumes `stft_mel_spec` directly (log-magnitude / phase-over-PI straight from the kernel)."""
    mel, spec = stft_mel_spec(y, n_fft, win_length, hop_length, want_spec=True)
    return torch.exp(spec[:, 0]), mel, spec[:, 1] * PI


# ---------------------------------------------------------------------------------------------------------------
# Host-side DSP of the finetune data path: numpy only.  librosa==0.8.1 (requirements.txt:1) is a
# third-party dependency absent from /root/reference and from this image, so its published algorithms are restated
# here; no reference test or fixture pins them: PARITY UNPINNED for this block (tests check the defining properties).
# ---------------------------------------------------------------------------------------------------------------
def align_wav(wav, r=hp.hop_length):
    """audio.py:37-41: zero-pad to a multiple of the hop."""
    d = len(wav) % r
    return np.pad(wav, (0, r - d)) if d else wav


def _padded_window(win_length, n_fft):
    """scipy/librosa 'hann' with fftbins=True (periodic), zero-padded to n_fft, centred (librosa.util.pad_center)"""
    w = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(win_length) / win_length)
    lp = (n_fft - win_length) // 2
    return np.pad(w, (lp, n_fft - win_length - lp))


def stft_np(y, n_fft=hp.n_fft, hop_length=hp.hop_length, win_length=hp.win_length):
    """librosa.stft(center=True, pad_mode='reflect'): [1 + n_fft/2, 1 + len(y) // hop] complex64."""
    w = _padded_window(win_length, n_fft)
    yp = np.pad(np.asarray(y, dtype=np.float32), n_fft // 2, mode='reflect')
    n_frames = 1 + (len(yp) - n_fft) // hop_length
    idx = np.arange(n_fft)[:, None] + hop_length * np.arange(n_frames)[None, :]
    return np.fft.rfft(w[:, None] * yp[idx], axis=0).astype(np.complex64)


def istft_np(D, hop_length=hp.hop_length, win_length=hp.win_length, length=None):
    """librosa.istft(center=True): windowed overlap-add divided by the window's squared-sum envelope, n_fft/2 trimmed
    from both ends, cut / zero-padded to `length`."""
    n_fft = 2 * (D.shape[0] - 1)
    n_frames = D.shape[1]
    w = _padded_window(win_length, n_fft)
    frames = np.fft.irfft(D, n=n_fft, axis=0) * w[:, None]
    n = n_fft + hop_length * (n_frames - 1)
    y = np.zeros(n, dtype=np.float64)
    env = np.zeros(n, dtype=np.float64)
    w2 = w * w
    for t in range(n_frames):
        s = t * hop_length
        y[s:s + n_fft] += frames[:, t]
        env[s:s + n_fft] += w2
    ok = env > np.finfo(np.float32).tiny
    y[ok] /= env[ok]
    y = y[n_fft // 2:]
    if length is None:
        y = y[:len(y) - n_fft // 2]
    else:
        y = y[:length] if len(y) >= length else np.pad(y, (0, length - len(y)))
    return y.astype(np.float32)


def get_mag(y, clamp_low=True):
    """audio.py:115-119: log magnitude spectrogram [F, T] (natural log, floor eps)."""
    S = np.abs(stft_np(y))
    return np.log(S.clip(min=eps) if clamp_low else S).astype(np.float32)


def _griffinlim(S, wavlen=None):
    """audio.py:131-136 -> librosa.griffinlim(S ** gl_power, n_iter=gl_iters, momentum=gl_momentum, init='random',
    random_state=randseed, length=wavlen): the fast Griffin-Lim of Perraudin et al. as librosa 0.8.1 states it."""
    if hp.gl_power:
        S = S ** hp.gl_power
    S = np.asarray(S, dtype=np.float32)
    rng = np.random.RandomState(seed=hp.randseed)
    angles = np.exp(2j * np.pi * rng.rand(*S.shape)).astype(np.complex64)
    rebuilt = 0.0
    n_fft = 2 * (S.shape[0] - 1)
    for _ in range(hp.gl_iters):
        tprev = rebuilt
        inverse = istft_np(S * angles, hp.hop_length, hp.win_length, length=wavlen)
        rebuilt = stft_np(inverse, n_fft, hp.hop_length, hp.win_length)
        angles = rebuilt - (hp.gl_momentum / (1 + hp.gl_momentum)) * tprev
        angles = (angles / (np.abs(angles) + 1e-16)).astype(np.complex64)
    return istft_np(S * angles, hp.hop_length, hp.win_length, length=wavlen).astype(np.float32)


def inv_mag(mag, wavlen=None):
    """audio.py:139-147: exp() of a log-magnitude [F or F-1, T] (a 1024-bin input gets a zero DC row) -> Griffin-Lim wave.
    NOTE (kept from the reference): the finetune path calls this with an already LINEAR magnitude (data.py:65,76), so
    exp() is applied to it again."""
    S = np.exp(mag)
    F, T = mag.shape
    if F == hp.n_freq - 1:
        S = np.concatenate([np.zeros([1, T]), S], axis=0)
    y = _griffinlim(S, wavlen)
    if wavlen:
        assert len(y) == wavlen
    return y


def _denormalize(S):
    """transtacos/audio.py:195-196: [-max_abs, max_abs] -> [min_level_db, 0]"""
    return ((S + hp.max_abs_value) * -hp.min_level_db) / (2 * hp.max_abs_value) + hp.min_level_db


def spec_to_natural_scale(spec):
    """transtacos/audio.py:80-82: normalised dB spectrogram -> linear magnitude, 10 ** ((denorm + ref_level_db) / 20)."""
    return np.power(10.0, (_denormalize(spec) + hp.ref_level_db) * 0.05)


def augment_spec(S, time_mask=True, freq_mask=True, prob=0.2, rounds=3, freq_width=9, time_width=3, rng=None):
    """audio.py:72-99: random frequency / time band masks then a 3x3 mean blur (zero padded, count includes padding).
    Kept from the reference: its masks are written through `torch.from_numpy(S)`, i.e. INTO THE CALLER'S ARRAY (a float32
    ndarray shares its memory with the tensor) — the caller's `mel / 2 + mel_aug / 2` (data.py:71-72) therefore blends
    the MASKED mel with its blur.  Pinned by tests/test_data_golden_cpu.py against the reference's own Dataset."""
    R = rng or np.random
    F, T = S.shape
    if not (isinstance(S, np.ndarray) and S.dtype == np.float32 and S.flags.writeable):
        S = np.array(S, dtype=np.float32, copy=True)
    for _ in range(rounds):
        if freq_mask and R.random() < prob:
            s, r = R.randint(0, F - freq_width), R.randint(1, freq_width)
            S[s:s + r, :] = R.uniform(low=S.min(), high=S.mean())
        if time_mask and R.random() < prob:
            s, r = R.randint(0, T - time_width), R.randint(1, time_width)
            S[:, s:s + r] = R.uniform(low=S.min(), high=S.mean())
    P = np.pad(S, 1)
    out = sum(P[i:i + F, j:j + T] for i in range(3) for j in range(3)) / 9.0
    return out.astype(np.float32)
