"""librosa.effects stand-ins (see the package docstring).  TEST INFRASTRUCTURE."""
import numpy as np


def trim(y, top_db=60, ref=np.max, frame_length=2048, hop_length=512):
    """keep the span between the first and last frame whose mean-square power is within top_db of the loudest frame
    (feature.rms with centred, reflect-padded frames; core.power_to_db with amin = 1e-10)"""
    yp = np.pad(np.asarray(y, dtype=np.float32), frame_length // 2, mode='reflect')
    n = 1 + (len(yp) - frame_length) // hop_length
    mse = np.array([np.mean(np.abs(yp[t * hop_length:t * hop_length + frame_length]) ** 2) for t in range(n)])
    db = 10.0 * np.log10(np.maximum(1e-10, mse)) - 10.0 * np.log10(np.maximum(1e-10, ref(mse)))
    keep = np.flatnonzero(db > -top_db)
    if keep.size == 0:
        return y[0:0], np.array([0, 0])
    start, end = int(keep[0]) * hop_length, min(len(y), (int(keep[-1]) + 1) * hop_length)
    return y[start:end], np.array([start, end])


def pitch_shift(*a, **k):
    raise NotImplementedError('stand-in: the fixture recipe never reaches the pitch-shift augmentation')


def time_stretch(*a, **k):
    raise NotImplementedError('stand-in: the fixture recipe never reaches the time-stretch augmentation')
