"""Minimal stand-in for librosa==0.8.1 (requirements.txt:1), absent from this image and from /root/reference.
Used only by oracle/gen_golden*.py in the build container.  TEST INFRASTRUCTURE.

`filters.mel` — the one librosa call on the hot path (retunegan/audio.py:20,158) — restates the published Slaney
filterbank.  The rest exists so that the reference's HOST data path (retunegan/data.py, retunegan/audio.py:28-147,
transtacos/audio.py) can be imported and run for oracle/gen_golden_data.py: `load`, `stft`, `istft`, `griffinlim`,
`effects.trim`, `note_to_hz`, `hz_to_midi` restate the published librosa 0.8.1 algorithms from their documentation.  They
are STAND-INS: what the data fixtures pin is the reference's own code around these calls (crop / pad / align logic,
exp / power / DC-row handling of inv_mag, the TransTacoS de-normalisation, the mel projection, the augmentation blend);
Griffin-Lim, the STFT pair and the silence trimmer themselves stay "parity unpinned" (DESIGN.md section 5)."""
import numpy as np

from . import filters  # noqa: F401
from . import effects  # noqa: F401


def load(path, sr=22050, mono=True, res_type='kaiser_best'):
    """PCM / float wav -> float32 in [-1, 1) (soundfile's int16 / 32768 convention); the fixtures' files are written at
    the target rate, so no resampling happens (librosa's 'kaiser_best' resampler is not restated)."""
    from scipy.io import wavfile
    rate, y = wavfile.read(path)
    assert rate == sr, 'stand-in: write the fixture wavs at the target sample rate'
    if y.dtype.kind == 'i':
        y = y.astype(np.float32) / float(2 ** (8 * y.dtype.itemsize - 1))
    y = y.astype(np.float32)
    if y.ndim > 1 and mono:
        y = y.mean(axis=1)
    return y, rate


def _window(win_length, n_fft):
    w = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(win_length) / win_length)      # scipy 'hann', fftbins=True
    lpad = (n_fft - win_length) // 2
    return np.pad(w, (lpad, n_fft - win_length - lpad))                            # util.pad_center


def stft(y, n_fft=2048, hop_length=None, win_length=None, window='hann', center=True, pad_mode='reflect'):
    win_length = win_length or n_fft
    hop_length = hop_length or win_length // 4
    assert window == 'hann' and center and pad_mode == 'reflect'
    w = _window(win_length, n_fft)
    yp = np.pad(np.asarray(y, dtype=np.float32), n_fft // 2, mode='reflect')
    n_frames = 1 + (len(yp) - n_fft) // hop_length
    frames = np.stack([yp[t * hop_length:t * hop_length + n_fft] for t in range(n_frames)], axis=1)
    return np.fft.rfft(w[:, None] * frames, axis=0).astype(np.complex64)


def istft(D, hop_length=None, win_length=None, window='hann', center=True, length=None):
    n_fft = 2 * (D.shape[0] - 1)
    win_length = win_length or n_fft
    hop_length = hop_length or win_length // 4
    assert window == 'hann' and center
    w = _window(win_length, n_fft)
    n_frames = D.shape[1]
    total = n_fft + hop_length * (n_frames - 1)
    y = np.zeros(total)
    wss = np.zeros(total)
    seg = np.fft.irfft(D, n=n_fft, axis=0) * w[:, None]
    for t in range(n_frames):
        y[t * hop_length:t * hop_length + n_fft] += seg[:, t]
        wss[t * hop_length:t * hop_length + n_fft] += w * w                         # filters.window_sumsquare
    nz = wss > np.finfo(np.float32).tiny
    y[nz] /= wss[nz]
    y = y[n_fft // 2:]
    if length is None:
        y = y[:len(y) - n_fft // 2]
    elif len(y) >= length:
        y = y[:length]
    else:
        y = np.pad(y, (0, length - len(y)))                                          # util.fix_length
    return y.astype(np.float32)


def griffinlim(S, n_iter=32, hop_length=None, win_length=None, window='hann', center=True, length=None, momentum=0.99,
               init='random', random_state=None):
    """fast Griffin-Lim (Perraudin, Balazs, Soendergaard 2013) as librosa 0.8.1 documents it"""
    assert init == 'random'
    n_fft = 2 * (S.shape[0] - 1)
    rng = np.random.RandomState(seed=random_state)
    angles = np.exp(2j * np.pi * rng.rand(*S.shape)).astype(np.complex64)
    rebuilt = 0.0
    for _ in range(n_iter):
        previous = rebuilt
        wave = istft(S * angles, hop_length=hop_length, win_length=win_length, window=window, length=length)
        rebuilt = stft(wave, n_fft=n_fft, hop_length=hop_length, win_length=win_length, window=window)
        angles = rebuilt - (momentum / (1 + momentum)) * previous
        angles = (angles / (np.abs(angles) + 1e-16)).astype(np.complex64)
    return istft(S * angles, hop_length=hop_length, win_length=win_length, window=window, length=length)


_NOTE = {'C': 0, 'D': 2, 'E': 4, 'F': 5, 'G': 7, 'A': 9, 'B': 11}


def note_to_hz(note):
    """'D2' -> 440 * 2 ** ((midi - 69) / 12) with midi = 12 * (octave + 1) + pitch class"""
    pc = _NOTE[note[0].upper()]
    rest = note[1:]
    while rest and rest[0] in '#b':
        pc += 1 if rest[0] == '#' else -1
        rest = rest[1:]
    midi = 12 * (int(rest) + 1) + pc
    return 440.0 * 2.0 ** ((midi - 69) / 12.0)


def hz_to_midi(f):
    return 12.0 * (np.log2(np.asanyarray(f, dtype=float)) - np.log2(440.0)) + 69.0
