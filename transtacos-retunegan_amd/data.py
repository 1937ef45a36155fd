"""Host-side data path of the RetuneGAN trainer: the `Dataset` contract of retunegan/data.py:20-172 on
the on-disk format TransTacoS's preprocessor writes, per-rank sharding for the data-parallel trainer, and the pinned,
double-buffered host->HBM feed of BASELINE configs[4] ("Griffin-Lim ref wav on host, pinned async H2D").

On-disk format (transtacos/preprocess.py:16-41, transtacos/datasets/databaker.py:113-116): a folder holding
`wav_path.txt` (the directory of the recordings), `train.txt` / `test.txt` (one `name|prosody|text` line per
utterance) and, for finetuning, `mag-<name>.npy` = the acoustic model's linear spectrogram [1025, T] normalised to
[-4, 4] (`_normalize` of transtacos/audio.py:190-193).

Everything here is numpy / scipy on the host.  librosa (wav decoding with 'kaiser_best' resampling, silence trimming,
pitch-shift / time-stretch augmentation) is absent from this image and from /root/reference: `load_wav` reads PCM / float
wav files with scipy and resamples with a polyphase filter when the rate differs, `trim_silence` restates
librosa.effects.trim, and the two librosa augmentations of retunegan/audio.py:44-69 are left out (the dynamic-range one
is kept).  Parity: the Dataset contract (finetune and plain feed, per-utterance cache, training crops, evaluation items),
the TransTacoS de-normalisation, the mel projection and the augmentation blend are PINNED by fixtures the reference's own
data.py / audio.py produced on a synthetic corpus (oracle/gen_golden_data.py, tests/test_data_golden_cpu.py); librosa's
Griffin-Lim, STFT pair, silence trimmer and resampler were stand-ins there (oracle/stubs/librosa) and stay unpinned.
"""
import os
import threading
import queue
from random import randint

import numpy as np
import torch
from torch.utils.data import Dataset as _TorchDataset

import hparam as hp
import audio as A
from rtg.lib import new_stream

assert hp.segment_size % hp.hop_length == 0            # data.py:16
frames_per_seg = hp.segment_size // hp.hop_length


def load_wav(path):
    """float32 mono in (-1, 1) at hp.sample_rate (retunegan/audio.py:28-30, transtacos/audio.py:29-31)."""
    from scipy.io import wavfile
    sr, y = wavfile.read(path)
    if y.dtype.kind == 'i':
        y = y.astype(np.float32) / float(2 ** (8 * y.dtype.itemsize - 1))
    elif y.dtype.kind == 'u':
        y = (y.astype(np.float32) - 128.0) / 128.0
    y = y.astype(np.float32)
    if y.ndim > 1:
        y = y.mean(axis=1)
    if sr != hp.sample_rate:
        from math import gcd
        from scipy.signal import resample_poly
        g = gcd(int(sr), hp.sample_rate)
        y = resample_poly(y, hp.sample_rate // g, int(sr) // g).astype(np.float32)
    return y


def trim_silence(wav, frame_length=512, hop_length=128):
    """transtacos/audio.py:58-61 -> librosa.effects.trim(top_db=hp.trim_below_peak_db): keep the span between the first
    and the last frame whose RMS power is within top_db of the loudest frame."""
    yp = np.pad(wav, frame_length // 2, mode='reflect')
    n = 1 + (len(yp) - frame_length) // hop_length
    idx = np.arange(frame_length)[None, :] + hop_length * np.arange(n)[:, None]
    mse = np.mean(yp[idx].astype(np.float64) ** 2, axis=1)
    db = 10.0 * np.log10(np.maximum(1e-10, mse)) - 10.0 * np.log10(np.maximum(1e-10, mse.max()))
    nz = np.flatnonzero(db > -hp.trim_below_peak_db)
    if nz.size == 0:
        return wav[:0]
    return wav[nz[0] * hop_length:min(len(wav), (nz[-1] + 1) * hop_length)]


def augment_wav(y, rng=None):
    """retunegan/audio.py:44-69 without the librosa pitch-shift / time-stretch branches (see module docstring): 75 % of
    the clips get a global gain 2 ** N(0, 1/3), re-normalised if it clips."""
    R = rng or np.random
    R.random(); R.random()                 # the two draws that gate the omitted augmentations keep the RNG stream aligned
    if R.random() > 0.25:
        y = y * 2 ** R.normal(scale=1 / 3)
        absmax = max(y.max(), -y.min())
        if absmax > 1.0:
            y = y / absmax
    return y.astype(np.float32)


def shard_indices(n, rank=0, world=1):
    """Equal-size strided shards (DistributedSampler without shuffling: every rank gets ceil(n / world) indices, the
    tail wraps around) for the one-process-per-GPU trainer."""
    per = -(-n // world)
    return [(rank + i * world) % n for i in range(per)]


class Dataset(_TorchDataset):
    """data.py:20-172 (non-split generators): item = (mel [80, 32], wav_tmpl [8192], wav [8192]) float32 for training,
    full-length arrays otherwise.  Utterances are preprocessed on first use and cached; under data parallelism build one
    per rank with `rank` / `world` so that each process caches only its shard."""

    def __init__(self, name, data_dp, finetune=False, limit=None, rank=0, world=1):
        self.is_train = name == 'train'
        self.data_dp = data_dp
        self.finetune = finetune
        with open(os.path.join(data_dp, 'wav_path.txt')) as fh:
            wav_path = fh.read().strip()
        with open(os.path.join(data_dp, f'{name}.txt'), encoding='utf-8') as fh:
            fps = [os.path.join(wav_path, line.split('|')[0] + '.wav') for line in fh.readlines() if line.strip()]
        if limit:
            fps = fps[:limit]
        self.wav_fps = [fps[i] for i in shard_indices(len(fps), rank, world)] if world > 1 else fps
        self.data = [None] * len(self.wav_fps)

    def __len__(self):
        return len(self.wav_fps)

    def _prepare(self, index):
        wav_fp = self.wav_fps[index]
        if not self.finetune:
            wav = load_wav(wav_fp)
            if self.is_train:
                wav = augment_wav(wav)
            wav = A.align_wav(wav)
        else:                                       # identical to TransTacoS's make_metadata (data.py:49-52)
            wav = A.align_wav(trim_silence(load_wav(wav_fp)))
        wavlen = len(wav)
        if not self.finetune:
            mag = A.get_mag(wav[:-1])               # `[:-1]` avoids the extra trailing frame (data.py:58)
        else:
            name = os.path.splitext(os.path.basename(wav_fp))[0]
            mag = A.spec_to_natural_scale(np.load(os.path.join(self.data_dp, f'mag-{name}.npy')))
        mel = A.mag_to_mel(mag)
        if self.is_train:
            # (in this order: augment_spec writes its masks into `mel` itself, as the reference's does — audio.augment_spec)
            mel_aug = A.augment_spec(mel, rounds=5)
            mel = mel / 2 + mel_aug / 2
        wav_tmpl = np.pad(A.inv_mag(mag, wavlen=wavlen - 1), (0, 1))       # data.py:76-77
        if hp.ref_wav == 'dy':
            wav_tmpl = np.diff(np.pad(wav_tmpl, (0, 1)))
        assert len(wav) == len(wav_tmpl) == mel.shape[1] * hp.hop_length, (len(wav), len(wav_tmpl), mel.shape)
        return mel, wav, wav_tmpl

    def __getitem__(self, index):
        if self.data[index] is None:
            self.data[index] = self._prepare(index)
        mel, wav, wav_tmpl = self.data[index]
        if self.is_train:                           # data.py:135-160: wav[8192] <=> mel[32]
            wavlen, mellen = len(wav), mel.shape[1]
            if wavlen > hp.segment_size:
                cp = randint(0, mellen - frames_per_seg - 1)
                mel = mel[:, cp:cp + frames_per_seg]
                wav = wav[cp * hp.hop_length:(cp + frames_per_seg) * hp.hop_length]
                wav_tmpl = wav_tmpl[cp * hp.hop_length:(cp + frames_per_seg) * hp.hop_length]
            else:
                # the reference's np.pad call here (data.py:156) is malformed and would raise; the evident intent is
                # to pad the frame axis with the floor value
                mel = np.pad(mel, ((0, 0), (0, frames_per_seg - mellen)), constant_values=mel.min())
                wav = np.pad(wav, (0, hp.segment_size - wavlen))
                wav_tmpl = np.pad(wav_tmpl, (0, hp.segment_size - wavlen))
        return [x.astype(np.float32) for x in (mel, wav_tmpl, wav)]


def collate(items, device=None):
    """train.py:121-128: (x [B,80,T/256], y_tmpl [B,1,T], y [B,1,T])"""
    mel = torch.from_numpy(np.stack([it[0] for it in items]))
    tmpl = torch.from_numpy(np.stack([it[1] for it in items])).unsqueeze(1)
    wav = torch.from_numpy(np.stack([it[2] for it in items])).unsqueeze(1)
    return mel, tmpl, wav


def finetune_batch_from_mags(mags, wavs):
    """What the finetune path feeds per utterance (data.py:61-77): mel = mel_basis @ natural-scale magnitude, reference
    wave = Griffin-Lim of it on the host.  mags: list of normalised [1025, frames] spectrograms; wavs: list of [T]."""
    items = []
    for mag_n, wav in zip(mags, wavs):
        mag = A.spec_to_natural_scale(mag_n)
        mel = A.mag_to_mel(mag)
        tmpl = np.pad(A.inv_mag(mag, wavlen=len(wav) - 1), (0, 1))
        items.append((mel.astype(np.float32), tmpl.astype(np.float32), wav.astype(np.float32)))
    return items


class PinnedFeeder:
    """Double-buffered host -> HBM feed: a producer thread builds batches (e.g. Griffin-Lim reference waves) into
    page-locked buffers, the copy to the device runs on its own HIP stream, and `next()` hands the training stream
    tensors whose copy has been ordered before their first use (event wait, no host sync).  `depth` batches are in
    flight, so host DSP and PCIe traffic of batch i+1 overlap the train step of batch i."""

    def __init__(self, make_batch, device, depth=2):
        self.make_batch, self.device, self.depth = make_batch, torch.device(device), depth
        self.copy_stream = new_stream(device=self.device)
        self.free, self.ready = queue.Queue(), queue.Queue(maxsize=depth)
        self.slots = None
        self._stop = False
        self._thread = threading.Thread(target=self._produce_guarded, daemon=True)
        self._started = False
        self._held = None
        self._error = None

    def _produce_guarded(self):
        try:
            self._produce()
        except BaseException as e:  # noqa: BLE001  a failing producer must not leave next() blocked forever
            self._error = e
            self.ready.put(None)

    def _alloc(self, batch):
        self.slots = []
        for _ in range(self.depth + 1):
            host = [torch.empty(t.shape, dtype=t.dtype).pin_memory() for t in batch]
            dev = [torch.empty(t.shape, dtype=t.dtype, device=self.device) for t in batch]
            self.slots.append((host, dev, torch.cuda.Event(), torch.cuda.Event()))
        for i in range(len(self.slots)):
            self.free.put(i)

    def _produce(self):
        torch.cuda.set_device(self.device)
        step = 0
        while not self._stop:
            batch = self.make_batch(step)
            if self.slots is None:
                self._alloc(batch)
            i = self.free.get()
            if i is None:
                return
            host, dev, copied, consumed = self.slots[i]
            copied.synchronize()                               # the slot's previous H2D copy has left the host buffer
            for h, t in zip(host, batch):
                h.copy_(t)
            with torch.cuda.stream(self.copy_stream):
                self.copy_stream.wait_event(consumed)          # the previous user of this device slot is done
                for h, d in zip(host, dev):
                    d.copy_(h, non_blocking=True)
                copied.record(self.copy_stream)
            self.ready.put(i)
            step += 1

    def next(self):
        if self._stop:
            raise RuntimeError('PinnedFeeder.next() after close()')
        if not self._started:
            self._started = True
            self._thread.start()
        if self._held is not None:                             # release the slot handed out last time
            host, dev, copied, consumed = self.slots[self._held]
            consumed.record(torch.cuda.current_stream(self.device))
            self.free.put(self._held)
            self._held = None
        i = self.ready.get()
        if i is None:
            raise RuntimeError(f'PinnedFeeder: the producer thread failed: {self._error!r}') from self._error
        self._held = i
        host, dev, copied, consumed = self.slots[i]
        torch.cuda.current_stream(self.device).wait_event(copied)
        return dev

    def close(self):
        """stop the producer and WAIT for it: a daemon thread still inside torch when the interpreter finalises aborts
        the process ("terminate called without an active exception", exit code 134)"""
        if self._stop and not self._thread.is_alive():         # second close(): nothing left to stop
            return
        self._stop = True
        self.free.put(None)
        if self._started:
            while self._thread.is_alive():
                try:
                    self.ready.get(timeout=0.05)           # a producer blocked on a full queue can finish its put
                except queue.Empty:
                    pass
            self._thread.join()
        try:
            while True:
                self.ready.get_nowait()
        except queue.Empty:
            pass
