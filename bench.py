#!/usr/bin/env python3
"""bench.py — RetuneGAN train-step throughput on MI355X (metric of BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload config2|config1|config4]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

`--gpus N` with N > 1 outside a launcher (WORLD_SIZE unset) starts the N ranks itself — one child process per GPU, before
anything in this process touches a GPU — relays rank 0's JSON line and exits non-zero if a rank fails or the node has fewer
than N devices; under a launcher `--gpus` must equal WORLD_SIZE.  It never silently times fewer GPUs than it was asked for.

One "step" = one iteration of retunegan/train.py:121-193 (1 G forward, d_train_times D updates, 1 G update) on a batch
of synthetic clips already resident in HBM.  Default workload = BASELINE.json configs[1]: UNet-G + MSD + MPD,
per-GPU batch 32, 8192-sample clips, multi-STFT loss, fp32.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, 'transtacos-retunegan_amd'))
# HIP-runtime defaults of the package (rtg/config.py: HIP_FORCE_DEV_KERNARG) before anything here can initialise the runtime;
# ranks started by self_launch inherit the environment
from rtg import config as _rtg_config  # noqa: E402,F401

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_FP32_MATRIX_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / 16x16x4_f32, dense
PEAK_HBM_GBS = 8000.0                # MI355X_MICROARCH.md: HBM3E spec (6.3 TB/s measured achievable)
PEAK_BF16_MATRIX_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense bf16 MFMA peak (the headline figures include sparsity)
SAMPLE_RATE = 22050


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', default='config2', choices=['config1', 'config2', 'config3', 'config4', 'config5'])
    ap.add_argument('--dtype', default=None, choices=['fp32', 'bf16'], help='conv arithmetic (default: the workload\'s)')
    ap.add_argument('--batch', type=int, default=None, help='per-GPU batch (default: the workload\'s)')
    ap.add_argument('--bf16-maps', action='store_true', help='bf16 workloads: the dense discriminator layers keep their feature maps in HBM as bf16 (hparam.bf16_maps = True; measured slower than fp32 maps, DESIGN.md section 3)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    return ap.parse_args()


WORKLOADS = {
    # name: (description, use_mpd, use_mtd, d_train_times, per-GPU batch, T)
    'config1': ('UNet-G + MSD, 1 D-step (BASELINE configs[0] shape on the GPU)', False, False, 1, 2, 8192),
    'config2': ('UNet-G + MSD/MPD, multi-STFT loss, d_train_times=2 (BASELINE configs[1])', True, False, 2, 32, 8192),
    'config3': ('full stack at 16384-sample clips, bf16 operands on the bf16 matrix cores, fp32 accumulation / losses / '
                'optimizer (BASELINE configs[2]); the tap-major (grouped) and 1-channel layers stay fp32',
                True, True, 2, 32, 16384),
    'config4': ('UNet-G + MSD/MPD/MTD full stack (BASELINE configs[3])', True, True, 2, 32, 8192),
    # BASELINE configs[4]: "1 s" clips = 86 frames = 22016 samples (22050 is not a multiple of the hop)
    'config5': ('finetune-shaped feed (BASELINE configs[4]): full stack, linear-spec -> mel + host Griffin-Lim reference '
                'wave per utterance (cached like data.py:33-40), per-step crop, pinned double-buffered async H2D',
                True, True, 2, 16, 22016),
}


def finetune_feeder(batch, T, seed, device, pool=48):
    """Host side of BASELINE configs[4]: a pool of synthetic utterances (normalised linear spectrograms of ~2 s of noise
    -> natural scale -> mel and Griffin-Lim reference wave, data.py:61-77), then per step a random `T`-sample crop of
    `batch` of them, copied through pinned buffers on a side stream (data.PinnedFeeder)."""
    import numpy as np
    import hparam as hp
    import audio as A
    import data as D
    rng = np.random.RandomState(seed)
    frames = T // hp.hop_length
    utts = []
    for _ in range(pool):
        n = (frames + 40 + rng.randint(0, 40)) * hp.hop_length
        wav = (rng.rand(n).astype(np.float32) * 2 - 1) * 0.5
        S = np.abs(A.stft_np(wav[:-1]))
        db = 20 * np.log10(np.maximum(1e-5, S)) - hp.ref_level_db
        mag_n = np.clip(2 * hp.max_abs_value * ((db - hp.min_level_db) / -hp.min_level_db) - hp.max_abs_value,
                        -hp.max_abs_value, hp.max_abs_value).astype(np.float32)
        utts.extend(D.finetune_batch_from_mags([mag_n], [wav]))

    def make_batch(step):
        items = []
        for _ in range(batch):
            mel, tmpl, wav = utts[rng.randint(0, pool)]
            cp = rng.randint(0, mel.shape[1] - frames)
            items.append((mel[:, cp:cp + frames], tmpl[cp * hp.hop_length:(cp + frames) * hp.hop_length],
                          wav[cp * hp.hop_length:(cp + frames) * hp.hop_length]))
        return D.collate(items)

    return D.PinnedFeeder(make_batch, device)



def synthetic_batch(batch, T, seed, device):
    """x = mel_2048 @ |N(0,1)| linear spec (what the finetune path feeds), y_tmpl, y ~ U(-1,1)."""
    import hparam as hp
    from audio import mel_filterbank
    g = torch.Generator().manual_seed(seed)
    spec = torch.randn(batch, hp.n_freq, T // hp.hop_length, generator=g).abs()
    mel = torch.from_numpy(mel_filterbank(hp.sample_rate, hp.n_fft, hp.n_mel, hp.fmin, hp.fmax))
    x = torch.matmul(mel, spec)
    y_tmpl = torch.rand(batch, 1, T, generator=g) * 2 - 1
    y = torch.rand(batch, 1, T, generator=g) * 2 - 1
    return x.to(device), y_tmpl.to(device), y.to(device)


def cpu_baseline(budget_s=12.0):
    """The CPU oracle (oracle/rtg_oracle.py, stock PyTorch fp32 CPU ops; the reference's own files cannot travel to the
    GPU box) timed on this host: BASELINE configs[0] = UNet-G + MSD, batch 2, 8192-sample clips, 1 D-step + 1 G-step."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('rtg_oracle', os.path.join(REPO, 'oracle', 'rtg_oracle.py'))
    O = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(O)
    torch.manual_seed(0)
    g, msd = O.Generator(), O.MSD()
    og, od = O.make_optimizers(g, [msd])
    x, y_tmpl, y = O.synthetic_batch(2, 8192, 1)
    # the batch-2 step does not scale past ~16 threads (measured on the 256-CPU host: 8/16 threads 0.25 s/step, 32 threads
    # 0.62, 64 threads 2.0, all 128 default threads 7): use the best setting, not the default oversubscription
    cores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    for _ in range(2):
        O.train_step(g, og, od, x, y_tmpl, y, msd, None, None, 1)
    n, t0 = 0, time.perf_counter()
    while True:
        O.train_step(g, og, od, x, y_tmpl, y, msd, None, None, 1)
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 40:
            break
    return {'value': round(2 * 8192 / SAMPLE_RATE / (el / n), 4), 'unit': 'audio-s/s', 'cores': cores, 'kind': 'port',
            'sample': f'{n} steps of configs[0] (UNet-G + MSD, batch 2 x 8192 samples, 1 D-step + 1 G-step) '
                      f'in {el:.1f} s on {cores} torch CPU threads (host has {os.cpu_count()} logical CPUs; more threads are slower '
                      f'at batch 2); oracle/rtg_oracle.py'}


def _src_digest():
    """digest of the kernel sources of this build (csrc/*.hip, *.h + include/rtg.h): tools/pmc_summary.py stores it in the
    PMC summary, so a committed PMC pass that describes other code than the one benchmarked is flagged"""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(REPO, 'transtacos-retunegan_amd', 'csrc', '*.hip')) +
                   glob.glob(os.path.join(REPO, 'transtacos-retunegan_amd', 'csrc', '*.h')) +
                   [os.path.join(REPO, 'include', 'rtg.h')])
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]


def _pmc_for(prefix, workload='config2', bf16_maps=False):
    """launch-weighted HBM bytes and matrix-pipe busy fraction of the kernels whose name starts with `prefix`, from the
    newest profiles/*_bench_<workload>[_bf16maps]_pmc.json (None without a pass of this workload AND this kind of feature
    maps: a pass of the fp32-map build says nothing about the bytes the bf16-map kernels move)"""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, 'profiles', f'*_bench_{workload}{"_bf16maps" if bf16_maps else ""}_pmc.json')))
    if not files:
        return None
    doc = json.load(open(files[-1]))
    ks = doc['kernels']
    stale = doc.get('src_digest') != _src_digest()
    # HBM bytes: per launch, launch-weighted (like `achieved`); matrix-pipe occupancy: time-weighted over the instances the SQ
    # pass has the counter for (busy cycles of the family / its SIMD cycles)
    n = b = m = t = 0.0
    for name, e in ks.items():
        if name.startswith(prefix):
            n += e['launches']
            b += e['launches'] * (e['hbm_read_MB_per_launch'] + e['hbm_write_MB_per_launch']) * 1e6
            if 'mfma_busy_frac' in e:
                m += e['launches'] * e['avg_us'] * e['mfma_busy_frac']
                t += e['launches'] * e['avg_us']
    if n == 0:
        return None
    return {'traffic': round(b / n), 'mfma_busy': round(m / t, 3) if t else None,
            'source': 'profiles/' + os.path.basename(files[-1]), 'stale': bool(stale)}


def roofline(trainer, batch, bf16=False, workload='config2', bf16_maps=False):
    """Live per-kernel timing of extra (eager) steps: every conv launch bracketed by HIP events on its launch stream."""
    from rtg import ops
    # an eager step first (the timed steps were graph replays: allocator blocks and caches of the eager path are cold), then
    # three instrumented steps; a launch's duration is the median of its three (the launch order is the same every step)
    trainer.train_step(*batch)
    runs = []
    for _ in range(3):
        ops.PROFILE = []
        trainer.train_step(*batch)
        torch.cuda.synchronize()
        rec, ops.PROFILE = ops.PROFILE, None
        runs.append([(k, v, f, e0.elapsed_time(e1), lab, nb) for k, v, f, e0, e1, lab, nb in rec])
    if not all(len(r) == len(runs[0]) and all(a[:2] == b[:2] for a, b in zip(r, runs[0])) for r in runs):
        runs = runs[:1]                              # (a step that met a new problem shape: keep the first as it is)
    rec = [(r0[0], r0[1], r0[2], sorted(r[i][3] for r in runs)[len(runs) // 2], r0[4], r0[5])
           for i, r0 in enumerate(runs[0])]
    agg = {}
    bw = {}
    for kernel, variant, flop, ms, _label, nbytes in rec:
        # bandwidth kernels: the weight-norm / optimizer / STFT launches (ops.timed_bw) and the 1-channel conv shapes
        name = kernel[3:] if kernel.startswith('bw:') else ({1: 'thin_cin1', 2: 'thin_cout1', 3: 'thin_cout1'}.get(variant) if kernel == 'conv1d' else None)
        if name:
            e = bw.setdefault(name, [0, 0.0, 0.0])
            e[0] += 1; e[1] += ms * 1e-3; e[2] += nbytes
        if kernel.startswith('bw:'):
            continue
        # one entry per kernel TEMPLATE: wgrad:<rows per MFMA> and conv1d:<instance> as before; the block shapes of the
        # dense-layer kernel (codes 8xxx: rows per wave x waves x column tiles of one template) are one kernel, dconv
        k = ('dconv', 16) if kernel == 'conv1d' and 8000 < variant < 9000 else (kernel, variant)       # (9xxx: rtg_sconv.hip)
        a = agg.setdefault(k, [0, 0.0, 0.0, 0.0])
        a[0] += 1
        a[1] += ms * 1e-3
        a[2] += flop
        a[3] += nbytes
    # the dominant MATRIX kernel (the bandwidth kernels of the 1-channel layers, variants < 100, are reported in by_kernel)
    (kernel, variant), (n, secs, flop, nbytes) = max(((k, v) for k, v in agg.items() if not (k[0] == 'conv1d' and k[1] < 100)),
                                             key=lambda kv: kv[1][1])
    if kernel == 'dconv':
        name = 'dconv_kernel<rows per wave, waves, 16-column tiles, stride, taps> (rtg_dconv.hip, all block shapes)'
    elif kernel == 'conv1d':
        name = f'conv1d_mfma_group_kernel<{variant // 100},{variant // 10 % 10},{variant % 10}>'
    else:
        name = f'wgrad_kernel<{variant}>'
    achieved = flop / secs / 1e12
    out = {'bound': 'mfma', 'achieved': round(achieved, 3), 'peak': PEAK_FP32_MATRIX_TFLOPS, 'unit': 'TFLOP/s',
           'frac': round(achieved / PEAK_FP32_MATRIX_TFLOPS, 4), 'traffic': None, 'kernel': name,
           'launches_per_step': n, 'avg_launch_us': round(secs / n * 1e6, 2),
           'algorithmic_gflop_per_launch': round(flop / n / 1e9, 4),
           'algorithmic_bytes_per_launch': round(nbytes / n) if nbytes else None}
    # HBM traffic and matrix-pipe occupancy of that kernel from the committed PMC passes (rocprofv3 --pmc cannot run inside
    # this process): profiles/*_pmc.json, written by tools/pmc_pass.sh + tools/pmc_summary.py from this same command
    if kernel == 'dconv':
        pmc = _pmc_for('dconv_kernel<', workload, bf16_maps)
    else:
        pmc = _pmc_for(f'conv1d_mfma_group_kernel<{variant // 100}, {variant // 10 % 10}, {variant % 10},' if kernel == 'conv1d'
                       else f'wgrad_kernel<{variant},', workload, bf16_maps)
    if pmc:                                  # (a pass of this workload: its instances and arithmetic type)
        out['traffic'] = pmc['traffic']
        out['traffic_unit'] = 'bytes/launch (HBM read + write, PMC FETCH_SIZE x2 + WRITE_SIZE)'
        out['mfma_busy_frac_pmc'] = pmc['mfma_busy']
        out['pmc_source'] = pmc['source']
        out['pmc_stale'] = pmc['stale']      # True: the committed PMC pass was taken on other kernel sources than this build
    # the UNet-G conv stack alone (north-star target: >= 30 % of the fp32 matrix peak)
    conv_rec = [r for r in rec if not r[0].startswith('bw:')]
    g_flop = sum(f for k_, v_, f, ms_, lb, _nb in conv_rec if not lb.split()[1].startswith('discriminators'))
    g_s = sum(ms_ * 1e-3 for k_, v_, f, ms_, lb, _nb in conv_rec if not lb.split()[1].startswith('discriminators'))
    if g_s > 0:
        out['unet_g_conv_stack'] = {'tflops': round(g_flop / g_s / 1e12, 3),
                                    'frac': None if bf16 else round(g_flop / g_s / 1e12 / PEAK_FP32_MATRIX_TFLOPS, 4),
                                    'ms_per_step': round(g_s * 1e3, 3), 'gflop_per_step': round(g_flop / 1e9, 2)}
    if bf16 and kernel != 'conv1d':
        # weight-gradient kernel with bf16 operand fragments: priced against the dense bf16 matrix peak
        out.update({'peak': PEAK_BF16_MATRIX_TFLOPS, 'frac': round(achieved / PEAK_BF16_MATRIX_TFLOPS, 4)})
    if bf16 and kernel == 'conv1d' and nbytes > 0:
        # at bf16 matrix rates (2.5 PFLOP/s dense: ridge ~312 flop/B) the conv layers (60-180 flop/B, fp32 tensors) are
        # bound by HBM: price the launches by their algorithmic bytes (every operand tensor once) against 8 TB/s
        gbs = nbytes / secs / 1e9
        out.update({'bound': 'hbm', 'achieved': round(gbs, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                    'frac': round(gbs / PEAK_HBM_GBS, 4), 'algorithmic_MB_per_launch': round(nbytes / n / 1e6, 2),
                    'matrix_tflops': round(achieved, 2), 'matrix_peak_bf16_tflops': PEAK_BF16_MATRIX_TFLOPS})
    total_flop = sum(a[2] for a in agg.values())
    total_s = sum(a[1] for a in agg.values())
    out['all_conv_kernels'] = {'tflops': round(total_flop / total_s / 1e12, 3), 'ms_per_step': round(total_s * 1e3, 3),
                               'gflop_per_step': round(total_flop / 1e9, 2)}
    out['by_kernel'] = {(f'{k[0]}:{k[1]}'): {'n': v[0], 'ms': round(v[1] * 1e3, 3),
                                             'tflops': round(v[2] / v[1] / 1e12, 2)} for k, v in sorted(agg.items())}
    # HBM-bound kernels: algorithmic bytes (every operand once) / event-timed duration against the 8 TB/s spec
    out['bandwidth_kernels'] = {
        name: {'n': n_, 'ms': round(s_ * 1e3, 3), 'algorithmic_MB_per_launch': round(b_ / n_ / 1e6, 2),
               'GBps': round(b_ / s_ / 1e9, 1), 'frac_of_hbm_peak': round(b_ / s_ / 1e9 / PEAK_HBM_GBS, 4)}
        for name, (n_, s_, b_) in sorted(bw.items()) if s_ > 0}
    return out


def self_launch(a):
    """`--gpus N` (N > 1) outside a launcher: start the N ranks as child processes of this script — nothing in this process
    has touched a GPU yet (torch.cuda.device_count() does not initialise the runtime on this image), no exec — relay rank
    0's stdout (the JSON line), and fail loudly: fewer than N devices, or any rank exiting non-zero (the others are killed:
    they would wait in a collective forever), ends this process with a non-zero code and a message on stderr."""
    import signal
    import socket
    import subprocess
    rehearse = os.environ.get('RTG_BENCH_REHEARSE') == '1'
    n_dev = torch.cuda.device_count()
    if n_dev < 1 or (n_dev < a.gpus and not rehearse):
        raise SystemExit(f'bench.py --gpus {a.gpus}: this node exposes {n_dev} GPU(s); refusing to time fewer ranks than asked '
                         f'(RTG_BENCH_REHEARSE=1 rehearses {a.gpus} gloo ranks on one GPU)')
    s_ = socket.socket(); s_.bind(('127.0.0.1', 0)); port = s_.getsockname()[1]; s_.close()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        # rank 0's stdout is this process's (the JSON line); the other ranks' goes to stderr.  Own sessions: a failed job is
        # killed group by group, never by name
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env,
                                      stdout=None if r == 0 else sys.stderr, start_new_session=True))

    def kill_all():
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
        for p in procs:
            p.wait()

    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                kill_all()
                raise SystemExit(f'bench.py --gpus {a.gpus}: rank {bad[0][0]} exited with code {bad[0][1]}; the other ranks were stopped')
            if all(c == 0 for c in codes):
                return
            time.sleep(0.2)
    except BaseException:
        kill_all()
        raise


def main():
    a = parse()
    if a.gpus < 1:
        raise SystemExit('bench.py: --gpus must be >= 1')
    if a.steps < 1 or a.warmup < 0:
        raise SystemExit('bench.py: --steps must be >= 1 and --warmup >= 0')
    if 'WORLD_SIZE' not in os.environ and a.gpus > 1:
        return self_launch(a)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != a.gpus:
        raise SystemExit(f'bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks')
    if os.environ.get('RTG_BENCH_REHEARSE') == '1':      # dev aid: all ranks on cuda:0, gradients over gloo (a one-GPU box)
        local = 0
    # RTG_DP_FORCE=1 with one rank: the whole data-parallel machinery (RCCL all-reduces on the communication stream, graph
    # segments cut at the exchanges, tuner broadcast) over a 1-rank 'nccl' group — what a one-GPU box can rehearse of --gpus N
    forced = world == 1 and os.environ.get('RTG_DP_FORCE') == '1'
    if forced:
        import socket
        s_ = socket.socket(); s_.bind(('127.0.0.1', 0)); port = s_.getsockname()[1]; s_.close()
        os.environ.setdefault('MASTER_PORT', str(port))
        os.environ.update(RANK='0', WORLD_SIZE='1')
    multi = world > 1 or forced
    backend_desc = None
    if multi:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(local)
        if os.environ.get('RTG_BENCH_REHEARSE') == '1':
            dist.init_process_group('gloo')
            backend_desc = 'gloo (rehearsal: all ranks on one GPU)'
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
            try:
                backend_desc = 'nccl = RCCL ' + '.'.join(str(v) for v in torch.cuda.nccl.version())
            except Exception:  # noqa: BLE001
                backend_desc = 'nccl = RCCL (version unavailable)'
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the hot path has no CPU fallback')
    device = torch.device('cuda', local)
    torch.cuda.set_device(device)

    import hparam as hp
    import train as train_mod
    from train import Trainer
    from rtg import config as rtg_config
    t_start = time.perf_counter()

    def note(msg):
        """progress on stderr (a multi-rank set-up takes minutes: tuner, two captures, the policy trial)"""
        if multi or os.environ.get('RTG_BENCH_VERBOSE') == '1':
            print(f'[bench rank {rank} +{time.perf_counter() - t_start:6.1f}s] {msg}', file=sys.stderr, flush=True)
    desc, use_mpd, use_mtd, d_times, batch, T = WORKLOADS[a.workload]
    if a.batch:
        batch = a.batch
    dtype = a.dtype or ('bf16' if a.workload == 'config3' else 'fp32')
    hp.compute_dtype = dtype                # read when the weight banks are built
    hp.bf16_maps = bool(a.bf16_maps) and dtype == 'bf16'
    torch.manual_seed(hp.randseed)          # identical initial weights on every rank (and broadcast from rank 0)
    tr = Trainer(use_mpd=use_mpd, use_mtd=use_mtd, d_train_times=d_times, dev=device)
    feeder = None
    if a.workload == 'config5':
        feeder = finetune_feeder(batch, T, hp.randseed + rank, device)
        next_batch = feeder.next
    else:
        data = synthetic_batch(batch, T, hp.randseed + rank, device)
        next_batch = lambda: data  # noqa: E731

    # set-up, outside warm-up and timing: the first train step of a Trainer also times the candidate block shapes of
    # every conv / weight-gradient launch and keeps the fastest per problem (rtg/tune.py) — part of building the step,
    # like compiling a kernel; with it here --warmup 0 still times tuned steps only
    # The timed step is replayed from HIP graphs (Trainer.train_step_graphed): the same kernels in the same order, issued by
    # the graph executor instead of ~770 Python-side launches per step (1.4 % faster on one GPU; with 8 ranks per host the
    # launches of all ranks compete for the host's cores).  The graphs are cut where data parallelism exchanges
    # gradients; the all-reduces run between the segments (tests/test_zz_dp_gpu.py::test_two_ranks_graphed_step).
    # RTG_GRAPH=0: the eager step.  Capture happens here, outside warm-up and timing; if it fails the eager step is timed
    # and the record says so.
    note('trainer built; first step (block shapes are timed here)')
    tr.train_step(*next_batch())
    torch.cuda.synchronize()
    note('first step done')

    def setup_step(policy):
        """-> (step function, launch mode) under the exchange policy `policy`: graphs captured and replayed once, or the
        eager step if that failed on any rank (or RTG_GRAPH=0)"""
        mode = 'hip-graph replay'
        if multi:
            tr.set_exchange(policy)                       # (drops the graphs of another policy: they are cut elsewhere)
        if os.environ.get('RTG_GRAPH', '1') == '0':
            return tr.train_step, 'eager'
        # capture WITHOUT replaying, then agree on the outcome, then replay: a rank whose capture failed must not head for
        # the eager step's collectives while the others are inside a replay's (different buffers and sizes: RCCL would hang)

        def agree(ok_, why):
            """MIN over the ranks of `ok_`: either every rank goes on with the graphs or every rank drops them"""
            nonlocal mode
            if multi:
                flag = torch.tensor([ok_], device=device)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if flag.item() == 0.0 and ok_:
                    ok_, mode = 0.0, f'eager ({why} failed on another rank)'
            if not ok_:
                tr._graphs = None
            return ok_

        ok = 1.0
        note(f'exchange {policy!r}: capturing')
        try:
            tr.prepare_graphs(*next_batch())
        except Exception as e:  # noqa: BLE001
            ok = 0.0
            mode = f'eager (graph capture failed: {type(e).__name__}: {e})'[:200]
            import traceback
            print(f'[rank {rank}] graph capture failed:\n' + traceback.format_exc(), file=sys.stderr, flush=True)
        ok = agree(ok, 'graph capture')
        note(f'exchange {policy!r}: captured ok={ok}; first replay')
        if ok:
            # the first replay, still outside warm-up and timing; a rank on which it raises takes every rank back to the
            # eager step (the collectives between the segments are the eager step's: same buffers, same order)
            try:
                tr.train_step_graphed(*next_batch())
                torch.cuda.synchronize()
            except Exception as e:  # noqa: BLE001
                ok = 0.0
                mode = f'eager (first graph replay failed: {type(e).__name__}: {e})'[:200]
                import traceback
                print(f'[rank {rank}] first graph replay failed:\n' + traceback.format_exc(), file=sys.stderr, flush=True)
            ok = agree(ok, 'the first graph replay')
        if ok:
            return tr.train_step_graphed, mode
        tr.dp.pending = []
        tr.train_step(*next_batch())             # (the eager step once more after a failed capture / replay)
        torch.cuda.synchronize()
        return tr.train_step, mode

    def timed(step_fn, k):
        """seconds for k steps, barrier + synchronize on both sides, MAX over the ranks"""
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()
        t0_ = time.perf_counter()
        out_ = None
        for _ in range(k):
            out_ = step_fn(*next_batch())
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()
        el_ = time.perf_counter() - t0_
        mine_ = el_
        if multi:
            t_ = torch.tensor([el_], device=device, dtype=torch.float64)
            dist.all_reduce(t_, op=dist.ReduceOp.MAX)
            el_ = t_.item()
        return el_, mine_, out_

    # Exchange policy (data parallel only): both are set up and timed HERE, on the job's own ranks and links — round 4 picked
    # 'update' from a 1-rank RCCL group, where an all-reduce moves no bytes — each after `TRIAL_WARM` untimed steps, in two
    # blocks of `TRIAL_STEPS` steps whose two readings are the trial's own spread (MAX over ranks, so every rank sees the same
    # numbers).  The faster one by its better block runs the warm-up and the timed steps — unless the two are closer than the
    # larger spread (box noise is 1-2 %): then the first policy of train.EXCHANGE_POLICIES, the compute-stream all-reduce with
    # one graph segment per optimizer update.  RTG_DP_CUT set: that policy, no trial.
    TRIAL_WARM, TRIAL_STEPS = 3, 8
    exchange_trials = exchange_blocks = None
    if multi and 'RTG_DP_CUT' not in os.environ:
        exchange_trials, exchange_blocks = {}, {}
        for pol in train_mod.EXCHANGE_POLICIES:
            step, mode = setup_step(pol)
            for _ in range(TRIAL_WARM):
                step(*next_batch())
            blocks = [round(timed(step, TRIAL_STEPS)[0] / TRIAL_STEPS * 1e3, 3) for _ in range(2)]
            exchange_blocks[pol] = blocks
            exchange_trials[pol] = min(blocks)
            note(f'exchange {pol!r}: {blocks} ms/step over 2 x {TRIAL_STEPS} steps ({mode})')
        first = train_mod.EXCHANGE_POLICIES[0]
        best = min(train_mod.EXCHANGE_POLICIES, key=lambda p_: exchange_trials[p_])
        spread = max(abs(b[0] - b[1]) for b in exchange_blocks.values())
        if best != first and exchange_trials[first] - exchange_trials[best] <= spread:
            best = first                                    # (within the trial's own noise: the simpler policy)
        if best != tr.dp.exchange:
            step, mode = setup_step(best)
    else:
        step, mode = setup_step(tr.dp.exchange)
    note(f'set-up done: exchange {tr.dp.exchange!r}, {mode}; warm-up and timing')
    for _ in range(a.warmup):
        step(*next_batch())
    # EXACTLY a.steps steps between barrier + synchronize pairs; `elapsed` is the MAX over the ranks
    elapsed, my_elapsed, (dl, gl) = timed(step, a.steps)
    per_rank = None
    if multi:
        # every rank's own time and a digest of its tuner's picks (rank 0 tunes and broadcasts its tables,
        # train.DataParallel.sync_tuner: the digests must be equal)
        from rtg import tune
        mine = torch.tensor([my_elapsed / a.steps * 1e3, float(int(tune.digest(), 16))],
                            device=device, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = {'ms_per_step': [round(t[0].item(), 3) for t in allr],
                    'tuner_picks_digest': [f'{int(t[1].item()):06x}' for t in allr]}
    loss_g = gl['gen_all'].item()
    loss_d = dl['disc_all'].item()

    roof = None
    data = next_batch()
    if feeder is not None:
        feeder.close()
    if not a.no_roofline:
        # every rank runs the instrumented steps (they are train steps: under data parallelism their gradient all-reduces
        # need all ranks), rank 0 reports its own launches
        roof = roofline(tr, data, bf16=dtype == 'bf16', workload=a.workload, bf16_maps=bool(hp.bf16_maps))
    if multi:
        dist.barrier()
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline()

    if rank == 0:
        ms = elapsed / a.steps * 1e3
        value = world * batch * T / SAMPLE_RATE / (elapsed / a.steps)
        out = {
            'metric': 'G+D train-step audio-seconds/sec', 'value': round(value, 2), 'unit': 'audio-s/s',
            'n_gpus': dist.get_world_size() if multi else 1, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(ms, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32' if dtype == 'fp32' else ('bf16 operands, f32 accumulate' + (' (bf16 feature maps in HBM)' if hp.bf16_maps else '')), 'data': 'synthetic' if feeder is None else 'synthetic, fed from pinned host memory every step',
            'config': {'workload': f'{a.workload}: {desc}', 'per_gpu_batch': batch, 'clip_samples': T,
                       'global_batch': world * batch, 'd_train_times': d_times,
                       'parallelism': f'dp{world}' if world > 1 else ('single (1-rank RCCL group, data-parallel path forced)' if forced else 'single'), 'launch': mode,
                       **({'per_rank': per_rank, 'world_size': dist.get_world_size(), 'backend': backend_desc,
                           'exchange': {'policy': tr.dp.exchange,
                                        'what': ('RCCL all-reduce of each model\'s flat gradient buffer '
                                                 + ('on the compute stream, graphs cut per optimizer update' if tr.dp.exchange == 'update'
                                                    else 'on a high-priority communication stream from the banks\' flush hooks, graphs cut per discriminator')
                                                 + ('' if mode.startswith('hip-graph') else ' (eager step)')),
                                        'chosen_by': ('RTG_DP_CUT' if exchange_trials is None else
                                                      f'{TRIAL_WARM} untimed + 2 x {TRIAL_STEPS} timed steps per policy in set-up (MAX over ranks); the '
                                                      f'better block decides, a difference inside the blocks\' spread keeps {train_mod.EXCHANGE_POLICIES[0]!r}'),
                                        'trial_ms_per_step': exchange_trials, 'trial_blocks_ms_per_step': exchange_blocks}} if per_rank else {})},
            'knobs': rtg_config.non_default(),       # environment switches set to a non-default value (rtg/config.py)
            'roofline': roof, 'cpu_baseline': cpu,
            **({'cpu_baseline_ref': 'timed on rank 0 at N = 1 only (the --gpus 1 line of the same sources carries it): the host cores are '
                                    'busy with the other ranks here'} if world > 1 else {}),
            'final_losses': {'gen_all': round(loss_g, 4), 'disc_all': round(loss_d, 4)},
        }
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
