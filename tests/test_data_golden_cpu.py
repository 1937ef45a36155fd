"""Host data path against fixtures produced by the REFERENCE's own retunegan/data.py, retunegan/audio.py
and transtacos/audio.py (oracle/gen_golden_data.py ran them in the build container on the synthetic corpus of
oracle/data_recipe.py; tests/golden/retunegan_data.npz).  Pinned here: the TransTacoS de-normalisation, the mel projection,
the augmentation blend and its RNG consumption, align / trim lengths, the log-magnitude spectrogram, and the whole
Dataset.__getitem__ contract (finetune feed with its double exp() quirk, per-utterance cache, training crops drawn from
python's `random`, evaluation items at full length).  NOT pinned by the reference: Griffin-Lim, the STFT pair and the
silence trimmer came from stand-ins of the absent librosa (oracle/stubs/librosa) when the fixtures were made."""
import os
import random
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'oracle'))


@pytest.fixture(scope='module')
def gd():
    return dict(np.load(os.path.join(REPO, 'tests', 'golden', 'retunegan_data.npz'), allow_pickle=False))


@pytest.fixture(scope='module')
def corpus(tmp_path_factory):
    import audio as A
    import data as D
    import data_recipe as R
    import hparam as hp
    root = str(tmp_path_factory.mktemp('corpus'))

    def frames_of(name, wav_fp):
        return len(A.align_wav(D.trim_silence(D.load_wav(wav_fp)))) // hp.hop_length
    R.write_corpus(root, frames_of)
    return root


def test_pieces_match_the_reference(gd):
    import audio as A
    import data as D
    import data_recipe as R
    mag_n = R.normalised_mag(40, 7)
    nat = A.spec_to_natural_scale(mag_n)
    np.testing.assert_allclose(nat.astype(np.float32)[::64, ::5], gd['natural_scale'], rtol=1e-6)
    mel = A.mag_to_mel(nat).astype(np.float32)
    np.testing.assert_allclose(mel, gd['mag_to_mel'], rtol=2e-5, atol=1e-7)
    np.random.seed(4242)                                   # the reference draws from numpy's global generator
    np.testing.assert_allclose(A.augment_spec(mel, rounds=5), gd['augment_spec'], rtol=2e-5, atol=1e-6)
    y = R.utterance(0).astype(np.float32) / 32768.0
    assert len(A.align_wav(y)) == int(gd['align_len'])
    np.testing.assert_allclose(A.get_mag(A.align_wav(y)[:-1])[::32, ::7], gd['get_mag'], rtol=1e-4, atol=2e-4)
    assert len(D.trim_silence(y)) == int(gd['trim_len'])


def test_finetune_dataset_training_crops(gd, corpus):
    """Dataset('train', finetune=True): mag-<name>.npy -> natural scale -> mel (blended with its augmentation) and the
    Griffin-Lim reference wave of exp(linear magnitude) (the reference's quirk, data.py:65,76 + audio.py:140), cached per
    utterance; every __getitem__ draws a new 32-frame crop from python's `random`."""
    import data as D
    import hparam as hp
    np.random.seed(hp.randseed)
    random.seed(hp.randseed)
    ds = D.Dataset('train', corpus, finetune=True)
    assert len(ds) == int(gd['train_len'])
    for rep in range(2):
        for i in range(len(ds)):
            mel, tmpl, wav = ds[i]
            assert mel.dtype == tmpl.dtype == wav.dtype == np.float32
            assert mel.shape == (hp.n_mel, hp.segment_size // hp.hop_length) and tmpl.shape == wav.shape == (hp.segment_size,)
            # the crop position is pinned by the target wave (read straight from the file: bit-exact) ...
            np.testing.assert_array_equal(wav[::4], gd[f'ft_train_{rep}_{i}_wav'])
            # ... the mel by the same projection + augmentation stream, the reference wave by the shared Griffin-Lim recipe
            np.testing.assert_allclose(mel, gd[f'ft_train_{rep}_{i}_mel'], rtol=1e-4, atol=1e-5)
            np.testing.assert_allclose(tmpl[::4], gd[f'ft_train_{rep}_{i}_tmpl'], rtol=1e-3, atol=2e-4)


def test_plain_dataset_evaluation_items(gd, corpus):
    """Dataset('test', finetune=False): wav -> aligned -> log-magnitude STFT -> mel and Griffin-Lim reference wave, full
    length, no augmentation, `limit` honoured."""
    import data as D
    dt = D.Dataset('test', corpus, finetune=False, limit=2)
    assert len(dt) == 2
    for i in range(2):
        mel, tmpl, wav = dt[i]
        assert [mel.shape[0], mel.shape[1], len(tmpl), len(wav)] == gd[f'test_{i}_shapes'].tolist()
        np.testing.assert_array_equal(wav[::16], gd[f'test_{i}_wav'])
        np.testing.assert_allclose(mel[:, ::3], gd[f'test_{i}_mel'], rtol=2e-4, atol=1e-5)
        np.testing.assert_allclose(tmpl[::16], gd[f'test_{i}_tmpl'], rtol=1e-3, atol=2e-4)
