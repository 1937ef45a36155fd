#!/usr/bin/env python3
"""Generate tests/golden/retunegan_data.npz by importing and RUNNING the reference's host data path (retunegan/data.py,
retunegan/audio.py, transtacos/audio.py) in the build container on the synthetic corpus of oracle/data_recipe.py.

    cd /tmp && python /root/repo/oracle/gen_golden_data.py

*** TEST INFRASTRUCTURE. ***  Needs /root/reference.  librosa is absent: the stand-ins of oracle/stubs/librosa serve the
reference's librosa calls (see their docstring: Griffin-Lim, the STFT pair and the silence trimmer are restatements and
stay unpinned; everything the reference does AROUND them is what these fixtures pin: SURVEY.md 8 f1 / f2)."""
import os
import random
import sys
import tempfile

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, '/root/reference/retunegan')
sys.path.insert(0, os.path.join(HERE, 'stubs'))
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402

import hparam as hp  # noqa: E402  (reference)
import audio as A  # noqa: E402  (reference; seeds numpy's global RNG with hp.randseed at import)
import data as D  # noqa: E402  (reference; imports transtacos.audio through audio_proxy)
from audio_proxy import AP  # noqa: E402
import data_recipe as R  # noqa: E402


def main():
    gold = {}
    root = tempfile.mkdtemp(prefix='rtg_corpus_')

    def frames_of(name, wav_fp):
        wav = AP.align_wav(AP.trim_silence(AP.load_wav(wav_fp)))
        return len(wav) // hp.hop_length
    R.write_corpus(root, frames_of)

    # ---- pieces (reference functions on recipe inputs)
    mag_n = R.normalised_mag(40, 7)
    nat = AP.spec_to_natural_scale(mag_n)
    gold['natural_scale'] = nat.astype(np.float32)[::64, ::5]
    gold['mag_to_mel'] = A.mag_to_mel(nat).astype(np.float32)
    np.random.seed(4242)
    gold['augment_spec'] = A.augment_spec(A.mag_to_mel(nat).astype(np.float32), rounds=5)
    y = R.utterance(0).astype(np.float32) / 32768.0
    gold['align_len'] = np.array(len(A.align_wav(y)))
    gold['get_mag'] = A.get_mag(A.align_wav(y)[:-1])[::32, ::7]
    tr = AP.trim_silence(y)
    gold['trim_len'] = np.array(len(tr))

    # ---- Dataset, finetune feed, training crops: numpy's global RNG (augment_spec) and python's (crop) seeded here
    np.random.seed(hp.randseed)
    random.seed(hp.randseed)
    ds = D.Dataset('train', root, finetune=True)
    gold['train_len'] = np.array(len(ds))
    for rep in range(2):                              # second pass: from the per-utterance cache, new crops
        for i in range(len(ds)):
            mel, tmpl, wav = ds[i]
            gold[f'ft_train_{rep}_{i}_mel'] = mel
            gold[f'ft_train_{rep}_{i}_tmpl'] = tmpl[::4]
            gold[f'ft_train_{rep}_{i}_wav'] = wav[::4]
    # ---- Dataset, plain feed, evaluation items (full length, no augmentation)
    dt = D.Dataset('test', root, finetune=False, limit=2)
    for i in range(len(dt)):
        mel, tmpl, wav = dt[i]
        gold[f'test_{i}_shapes'] = np.array([mel.shape[0], mel.shape[1], len(tmpl), len(wav)])
        gold[f'test_{i}_mel'] = mel[:, ::3]
        gold[f'test_{i}_tmpl'] = tmpl[::16]
        gold[f'test_{i}_wav'] = wav[::16]
    out = os.path.join(REPO, 'tests', 'golden', 'retunegan_data.npz')
    np.savez_compressed(out, **gold)
    print('wrote', out, os.path.getsize(out), 'bytes,', len(gold), 'arrays')


if __name__ == '__main__':
    main()
