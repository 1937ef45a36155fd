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
    return ops.StftFn.apply(y, get_plan(n_fft, win_length, hop_length), want_spec)


def get_stft_torch(y, n_fft, win_length, hop_length):
    """audio.py:150-170: returns (S, M, P) = (|D + 1e-9|, mel_basis @ S, angle(D)).  API-compatibility wrapper;
    the train step consumes `stft_mel_spec` directly (log-magnitude / phase-over-PI straight from the kernel)."""
    mel, spec = stft_mel_spec(y, n_fft, win_length, hop_length, want_spec=True)
    return torch.exp(spec[:, 0]), mel, spec[:, 1] * PI
