"""Synthetic corpus in the on-disk format of TransTacoS's preprocessor (transtacos/preprocess.py:16-41,
transtacos/datasets/databaker.py:113-116), by recipe: used by oracle/gen_golden_data.py (to run the REFERENCE's Dataset
on it) and by tests/test_data_golden_cpu.py (to run this repo's Dataset on the same files).  TEST INFRASTRUCTURE."""
import os

import numpy as np

SR = 22050
HOP = 256
NAMES = ('000001', '000002', '000003')


def utterance(i):
    """int16 PCM: 0.15 s of near-silence, a few partials with a slow envelope plus noise, near-silence again"""
    rng = np.random.RandomState(1000 + i)
    n = int(SR * (1.1 + 0.35 * i))
    t = np.arange(n) / SR
    f0 = 110.0 * (1 + 0.5 * i)
    y = sum(np.sin(2 * np.pi * f0 * k * t + rng.rand() * 6.28) / k for k in range(1, 9))
    y = y * (0.6 + 0.4 * np.sin(2 * np.pi * 2.5 * t)) * 0.12 + rng.randn(n) * 0.01
    lead = int(0.15 * SR)
    y[:lead] *= 1e-3
    y[-lead:] *= 1e-3
    return np.clip(np.round(y * 32767), -32768, 32767).astype(np.int16)


def normalised_mag(n_frames, i):
    """a smooth pseudo-spectrogram in the acoustic model's normalised scale [-4, 4], [1025, n_frames] float32"""
    rng = np.random.RandomState(2000 + i)
    f = np.linspace(0, 1, 1025)[:, None]
    t = np.linspace(0, 1, n_frames)[None, :]
    base = 2.5 * np.cos(6.0 * f + 3.0 * t + i) - 2.0 * f + 0.3 * rng.randn(1025, n_frames)
    return np.clip(base, -4, 4).astype(np.float32)


def write_corpus(root, frames_of):
    """frames_of(name, wav_path) -> number of spectrogram frames the utterance has after the finetune preprocessing
    (load, trim, align): the mag file must have as many (retunegan/data.py:123)"""
    from scipy.io import wavfile
    wav_dir = os.path.join(root, 'wavs')
    os.makedirs(wav_dir, exist_ok=True)
    with open(os.path.join(root, 'wav_path.txt'), 'w') as fh:
        fh.write(wav_dir + '\n')
    for i, name in enumerate(NAMES):
        wavfile.write(os.path.join(wav_dir, name + '.wav'), SR, utterance(i))
    for split in ('train', 'test'):
        with open(os.path.join(root, f'{split}.txt'), 'w', encoding='utf-8') as fh:
            for name in NAMES:
                fh.write(f'{name}|prosody|text\n')
    for i, name in enumerate(NAMES):
        n = frames_of(name, os.path.join(wav_dir, name + '.wav'))
        np.save(os.path.join(root, f'mag-{name}.npy'), normalised_mag(n, i))
