"""Train-step orchestration of retunegan/train.py:47-88,121-193 on the MI355X hot path.

What is kept from the reference: model construction through `models` (zero-argument constructors looked up by
`hp.generator_ver`, train.py:48-52), AdamW hyper-parameters (train.py:80-81), ExponentialLR per epoch (train.py:87-88),
the D x d_train_times -> G update order with its detach semantics and loss weights (train.py:131-193), the NaN guard
(train.py:158,191) and the checkpoint dict layout (train.py:263-273).

What is new (absent from the reference): one flat fused AdamW launch per optimizer, device-side NaN guard (no host
sync inside a step), the real wave's spectra computed once per step, and plain data parallelism over clips: one process
per GPU, gradients of each model's flat buffer all-reduced with RCCL (torch.distributed backend "nccl").  Two exchange
policies (DataParallel.exchange, Trainer.set_exchange): 'update' — the all-reduces run on the compute stream between the
backward and the optimizer launch, one HIP-graph segment per optimizer update; 'disc' — each discriminator's all-reduce
is issued from its bank's flush on a high-priority communication stream and overlaps the next discriminator's backward
(graphs cut per discriminator).  bench.py times both on the job's actual links when it runs on more than one rank.
"""
import ctypes as C
import itertools  # noqa: F401
import os
import weakref

import torch
import torch.distributed as dist

import hparam as hp
import hparam as h  # noqa: F401  (train.py:17 imports it under both names)
from models import *  # noqa: F401,F403
from models import (multi_stft_loss, dynamic_loss, discriminator_loss, generator_loss, feature_loss,
                    MultiScaleDiscriminator, MultiPeriodDiscriminator, MultiStftDiscriminator)
from models.layers import BankedModel, fork_join  # noqa: F401
from models.discrminator import run_stacks
from models.loss import stft_cache
from rtg import ops, tune, config
from rtg.lib import lib, check, RtgError, new_stream, current_stream_ptr as _lib_stream_ptr

device = 'cuda' if torch.cuda.is_available() else 'cpu'
CAPTURE_ERROR_MODE = config.get('RTG_CAPTURE_MODE')
LEAN_PACK = True              # (Trainer.lean_pack_enabled: tests build trainers without it)


EXCHANGE_POLICIES = ('update', 'disc')
# the generator step's wave-only loss terms (multi-resolution STFT, dynamic / envelope / strip-mirror) as a branch of the
# discriminator stacks' fork (Trainer.g_step; round 6, same-box A/B of the config-2 step: 26.23 / 26.32 ms against 26.43 / 26.49)
WAVE_LOSSES_FORKED = True


def default_exchange():
    """The exchange policy a Trainer starts with (RTG_DP_CUT: 'update' | 'disc'; Trainer.set_exchange changes it).
    'update': all-reduces on the compute stream itself (async_op=False: stream-ordered after the kernels that wrote the
    gradients and before the optimizer launch, no communication stream, no event hops).  Measured on one MI355X over a
    1-RANK RCCL group (round 4): 28.1 ms/step against 29.9-30.8 for 'disc' (27.7 with the collectives left out) — the
    cross-stream hops and the serialised discriminator backward passes cost more than they hide when the exchange moves no
    bytes over xGMI.  A 1-rank measurement cannot decide the N > 1 case: bench.py times both policies on the job's ranks
    and keeps the faster (config.exchange in its record)."""
    v = config.get('RTG_DP_CUT')
    if v not in EXCHANGE_POLICIES:
        raise RtgError(f'RTG_DP_CUT={v!r}: expected one of {EXCHANGE_POLICIES}')
    return v



def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return _lib_stream_ptr()


# ---------------------------------------------------------------------------------------------------------------
# optimizer
# ---------------------------------------------------------------------------------------------------------------
class AdamW:
    """torch.optim.AdamW semantics (decoupled weight decay 0.01, eps 1e-8, bias correction) over the flat parameter
    buffers of one or more BankedModels: ONE kernel launch per model instead of 550 per-tensor updates per step
    (SURVEY.md 2.1).  Constructible like the reference does it: AdamW(module.parameters(), lr, betas=[b1, b2])."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        params = list(params)
        if params and isinstance(params[0], BankedModel):
            models = params
        else:
            pos = {id(p): i for i, p in enumerate(params)}
            models = [m for m in BankedModel._registry if any(id(p) in pos for p in m.parameters())]
            # in the order the caller listed the parameters (the registry is an unordered weak set): the per-parameter
            # indices of state_dict() / load_state_dict() must follow torch.optim's numbering
            models.sort(key=lambda m: min(pos.get(id(p), len(pos)) for p in m.parameters()))
            covered = sum(sum(1 for _ in m.parameters()) for m in models)
            if covered != len(params):
                raise RtgError('AdamW: parameters must cover whole RetuneGAN models (generator / msd / mpd / mtd)')
        self.models = models
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), (float(betas[0]), float(betas[1])), eps, weight_decay
        self.initial_lr = self.lr
        self.state = {}
        self.param_groups = [{'lr': self.lr, 'betas': self.betas, 'eps': eps, 'weight_decay': weight_decay,
                              'initial_lr': self.lr}]
        self.grad_scale = 1.0

    def _st(self, m):
        bank = m.bank()
        s = self.state.get(id(m))
        if s is None or s['bank'] is not bank:
            dev = bank.flat.device
            old = s
            s = {'bank': bank, 'exp_avg': torch.zeros_like(bank.flat), 'exp_avg_sq': torch.zeros_like(bank.flat),
                 'step': torch.zeros(1, device=dev)}
            if old is not None:      # the model was moved: carry the moments over
                s['exp_avg'].copy_(old['exp_avg']); s['exp_avg_sq'].copy_(old['exp_avg_sq']); s['step'].copy_(old['step'])
            self.state[id(m)] = s
        return s

    def zero_grad(self, set_to_none=False):
        for m in self.models:
            m.bank().zero_grad()

    def step(self, loss_flag=None, use_bank_flag=False):
        """loss_flag: optional device scalar; if it is NaN the update (and the step counter) is skipped on the device
        — the reference's `if not torch.isnan(loss): loss.backward()` guard without a host round trip.
        use_bank_flag: test the flag slot of each model's gradient buffer instead (WeightBank.set_flag: the loss value
        rides the data-parallel all-reduce of the gradients, so the guard is collective without a collective of its own)."""
        lr = self.param_groups[0]['lr']
        for m in self.models:
            s = self._st(m)
            bank = s['bank']
            bank.sync_grads()
            from rtg import ops
            check(ops.timed_bw('adamw', 28 * bank.n_params, lambda: lib.rtg_adamw(
                _p(bank.flat), _p(bank.gflat), _p(s['exp_avg']), _p(s['exp_avg_sq']), bank.n_params, _p(s['step']),
                _p(bank.flag() if use_bank_flag else loss_flag), lr, self.betas[0], self.betas[1], self.eps, self.weight_decay, self.grad_scale,
                _stream()), type(m).__name__), 'adamw')

    def step_tensor(self, m=None):
        return self._st(m or self.models[0])['step']

    # -- checkpoint format: per-parameter entries in parameter order, like torch.optim.AdamW.state_dict()
    def state_dict(self):
        state, idx = {}, 0
        for m in self.models:
            s = self._st(m)
            bank = s['bank']
            base = bank.flat.data_ptr()
            for p in m.parameters():
                off = (p.data_ptr() - base) // 4
                state[idx] = {'step': s['step'].clone().reshape(()), 'exp_avg': s['exp_avg'][off:off + p.numel()].view(p.shape).clone(),
                              'exp_avg_sq': s['exp_avg_sq'][off:off + p.numel()].view(p.shape).clone()}
                idx += 1
        g = dict(self.param_groups[0])
        g['params'] = list(range(idx))
        return {'state': state, 'param_groups': [g]}

    def load_state_dict(self, sd):
        """Accepts what torch.optim.AdamW.state_dict() of the reference wrote (train.py:84-85): entries indexed by
        parameter position, which is why the models register their parameters in the reference's order."""
        idx = 0
        for m in self.models:
            s = self._st(m)
            base = s['bank'].flat.data_ptr()
            for name, p in m.named_parameters():
                e = sd['state'].get(idx)
                if e is not None:
                    for key in ('exp_avg', 'exp_avg_sq'):
                        if tuple(e[key].shape) != tuple(p.shape):
                            raise RtgError(f'AdamW.load_state_dict: state {idx} ({name}) has {key} of shape '
                                           f'{tuple(e[key].shape)}, the parameter is {tuple(p.shape)}: the checkpoint '
                                           'was written for another parameter order or architecture')
                    off = (p.data_ptr() - base) // 4
                    s['exp_avg'][off:off + p.numel()].copy_(e['exp_avg'].reshape(-1))
                    s['exp_avg_sq'][off:off + p.numel()].copy_(e['exp_avg_sq'].reshape(-1))
                    s['step'].fill_(float(e['step']))
                idx += 1
        g = sd['param_groups'][0]
        self.param_groups[0]['lr'] = float(g['lr'])
        if 'initial_lr' in g:
            self.initial_lr = self.param_groups[0]['initial_lr'] = float(g['initial_lr'])


class ExponentialLR:
    """torch.optim.lr_scheduler.ExponentialLR(optim, gamma, last_epoch) as used at train.py:87-88,326-327, with the
    semantics of the torch this package runs on (2.x; tests/test_host_cpu.py compares with torch's own class): the
    constructor counts one step without touching the learning rate — a fresh run (last_epoch=-1) ends at last_epoch 0
    with lr = initial_lr, a resumed run (last_epoch=e, lr loaded from the checkpoint's param_groups) at e+1 with the
    loaded lr — and every step() multiplies the CURRENT lr by gamma (the chainable form).
    Deviation: torch 1.8.0, the version the reference's README.md:15 names, multiplied the loaded lr by
    gamma once more inside the constructor on resume; `hparam.legacy_resume_lr = True` reproduces that, so a run resumed
    here follows the schedule of a reference run resumed under its era's torch."""

    def __init__(self, optimizer, gamma, last_epoch=-1):
        self.optimizer, self.gamma = optimizer, gamma
        group = optimizer.param_groups[0]
        if last_epoch == -1:
            group.setdefault('initial_lr', group['lr'])
        elif 'initial_lr' not in group:
            raise KeyError("param 'initial_lr' is not specified in param_groups[0] when resuming an optimizer")
        self.base_lrs = [group['initial_lr']]
        self.last_epoch = last_epoch + 1          # the constructor's initial step
        if last_epoch != -1 and getattr(hp, 'legacy_resume_lr', False):
            group['lr'] = group['lr'] * gamma     # torch 1.x: that initial step went through get_lr()

    def step(self):
        self.last_epoch += 1
        self.optimizer.param_groups[0]['lr'] = self.optimizer.param_groups[0]['lr'] * self.gamma

    def get_last_lr(self):
        return [self.optimizer.param_groups[0]['lr']]


# ---------------------------------------------------------------------------------------------------------------
# data parallel (one process per GPU, RCCL over xGMI)
# ---------------------------------------------------------------------------------------------------------------
class DataParallel:
    """Plain data parallelism over utterance clips: identical replicas, per-rank batches, gradients
    summed with all_reduce on each model's flat gradient buffer and averaged inside the AdamW kernel (grad_scale)."""

    def __init__(self, models, process_group=None):
        # RTG_DP_FORCE=1: keep the whole data-parallel machinery (flush hooks, communication stream, collectives between
        # graph segments) on in a group of ONE rank — how tests/test_zz_dp_gpu.py runs RCCL itself on a one-GPU box
        self.enabled = dist.is_available() and dist.is_initialized() and (
            dist.get_world_size(process_group) > 1 or config.get('RTG_DP_FORCE') == '1')
        self.group = process_group
        self.world = dist.get_world_size(process_group) if self.enabled else 1
        self.models = [m for m in models if m is not None]
        self.pending = []
        self.comm_stream = None
        self.exchange = default_exchange()

    def broadcast_parameters(self):
        if not self.enabled:
            return
        for m in self.models:
            dist.broadcast(m.bank().flat, src=0, group=self.group)

    def reduce_async(self, flat_grad):
        """all-reduce `flat_grad` on the communication stream once the kernels already queued on the current stream
        (which produce it) have finished; returns immediately."""
        if not self.enabled:
            return
        if flat_grad.is_cuda and self.exchange == 'update':
            # stream-ordered on the compute stream: after the kernels that produced the gradients, before the optimizer
            # launch that reads them; no communication stream, no event hops (nothing to overlap with: see _capture)
            dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=self.group, async_op=False)
            return
        if flat_grad.is_cuda:
            if self.comm_stream is None:
                # high priority: its own hardware queue, so that the RCCL kernels neither wait behind nor hold up the
                # compute kernels of a forked stream that would otherwise share a queue with it (4 queues per process)
                self.comm_stream = new_stream(priority=-1)
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                w = dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self.pending.append(w)
        else:
            self.pending.append(dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def wait(self):
        for w in self.pending:
            w.wait()
        self.pending = []
        if self.comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self.comm_stream)

    def drain(self):
        """Quiesce the process group before a HIP-graph capture: every all-reduce this trainer issued has been waited for and
        has FINISHED on the device, and no rank starts capturing while another still runs eager collectives.  (Round 3:
        ProcessGroupNCCL's watchdog thread polls the events of collectives it still tracks; a capture in the default
        'global' error mode makes that hipEventQuery fail with hipErrorStreamCaptureUnsupported and the watchdog ends the
        process.  Trainer._capture therefore captures 'thread_local' — the fix — and drains first so that the graphs also
        never depend on a collective in flight.)"""
        if not self.enabled:
            return
        self.wait()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        dist.barrier(group=self.group)
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    def sync_tuner(self, src=0):
        """-> whether rank `src` met a problem its tables did not hold yet (every rank then tunes once more).
        Rank `src` timed the block shapes (rtg/tune.py); every other rank takes its pick tables, so that all ranks run
        the same kernels (the same step time: the slowest rank paces a data-parallel step) and only one tuner runs per
        host.  The table keys are descriptor bytes (shapes only, no pointers): identical on every rank for equal
        per-rank batches.  Collective: every rank calls it at the same point."""
        if not self.enabled:
            return tune.MISSED
        mine = dist.get_rank(self.group) == src
        box = [(tune.export_tables(), tune.MISSED) if mine else None]
        dist.broadcast_object_list(box, src=src, group=self.group)
        tables, missed = box[0]
        if not mine:
            tune.import_tables(tables)
        return bool(missed)



# ---------------------------------------------------------------------------------------------------------------
# the step
# ---------------------------------------------------------------------------------------------------------------
def _detached(losses):
    return {k: (v.detach() if torch.is_tensor(v) else v) for k, v in losses.items()}


class Trainer:
    """Owns the models and optimizers exactly as train.py:47-88 builds them and runs train.py:121-193 per batch."""

    def __init__(self, generator=None, msd=None, mpd=None, mtd=None, use_mpd=True, use_mtd=True, d_train_times=None,
                 dev=None, process_group=None):
        dev = torch.device(dev or device)
        if dev.type == 'cuda':
            # librtg launches on the current stream of the CURRENT device (rtg/lib.py:current_stream_ptr): make the
            # trainer's device that device, as one process per GPU does anyway
            torch.cuda.set_device(dev if dev.index is not None else torch.cuda.current_device())
        Generator = globals().get(f'Generator_{hp.generator_ver}')
        self.generator = (generator or Generator()).to(dev)
        self.msd = (msd or MultiScaleDiscriminator()).to(dev)
        self.mpd = (mpd or MultiPeriodDiscriminator()).to(dev) if use_mpd else None
        self.mtd = (mtd or MultiStftDiscriminator()).to(dev) if use_mtd else None
        self.discs = [d for d in (self.msd, self.mpd, self.mtd) if d is not None]
        self.d_train_times = hp.d_train_times if d_train_times is None else d_train_times
        self.optim_g = AdamW(self.generator.parameters(), hp.learning_rate_g, betas=[hp.adam_b1, hp.adam_b2])
        self.optim_d = AdamW(itertools.chain(*[d.parameters() for d in self.discs]), hp.learning_rate_d,
                             betas=[hp.adam_b1, hp.adam_b2])
        self.scheduler_g = ExponentialLR(self.optim_g, gamma=hp.lr_decay)
        self.scheduler_d = ExponentialLR(self.optim_d, gamma=hp.lr_decay)
        self.dp = DataParallel([self.generator, *self.discs], process_group)
        self.optim_g.grad_scale = self.optim_d.grad_scale = 1.0 / self.dp.world
        self.dp.broadcast_parameters()
        self.generator.noise.salt = None
        self.steps = 0
        self._tuned = False
        self._graphs = None
        self._graph_gen = None
        self._static_in = self._static_out = None
        self._capture_hook = None
        self._cap_stream = None
        self._settle = False
        self._lean_pending = True
        self.lean_pack_enabled = LEAN_PACK
        self.lean_dropped = 0
        for m in (self.generator, *self.discs):
            m.train()
        self.set_exchange(self.dp.exchange)

    def set_exchange(self, policy):
        """'update' | 'disc' (module docstring).  Drops captured graphs: the policy decides where they are cut."""
        if policy not in EXCHANGE_POLICIES:
            raise RtgError(f'exchange policy {policy!r}: expected one of {EXCHANGE_POLICIES}')
        self.dp.wait()
        self.dp.exchange = policy
        self._graphs = None
        # 'disc': each discriminator's all-reduce is issued from its bank's flush, as soon as ITS backward is complete, and
        # overlaps the rest of the backward.  'update': no hooks — the flushes run on the forked streams of the stacks, and
        # collectives of one communicator must not be in flight on two streams at once; _d_reduce issues them one after the
        # other on the main stream instead.
        for d in self.discs:
            d.bank().on_flush = self.dp.reduce_async if (self.dp.enabled and policy == 'disc') else None

    def _table_gen(self):
        return sum(m.bank().table_gen for m in (self.generator, *self.discs))

    def _freeze(self, flag):
        for d in self.discs:
            for p in d.parameters():
                p.requires_grad_(not flag)

    def _d_forward(self, y, y_g_hat_detach):
        """train.py:133-157 up to the backward: -> (losses, [(discriminator, its loss term)])"""
        self.optim_d.zero_grad()
        S = S_g = None
        if self.mtd is not None:
            S, S_g = multi_stft_loss(y, y_g_hat_detach, ret_specs=True)
        losses = {}
        # the discriminator stacks are independent: run them on separate streams (models.layers.fork_join)
        jobs = [('disc_s', (self.msd, y, y_g_hat_detach))]
        if self.mpd is not None:
            jobs.append(('disc_p', (self.mpd, y, y_g_hat_detach)))
        if self.mtd is not None:
            jobs.append(('disc_t', (self.mtd, S, S_g)))
        for (tag, _), (r, g, _, _) in zip(jobs, run_stacks([j for _, j in jobs])):
            losses[tag] = discriminator_loss(r, g)
        total = ops.weighted_sum(list(losses.values()))        # (one launch; the reference adds the terms one by one)
        losses['disc_all'] = total
        for d in self.discs:                 # before the backward: the flush hooks all-reduce the buffers, flag included
            d.bank().set_flag(total)
        return losses, [(job[0], losses[tag]) for tag, job in jobs]

    def d_step(self, y, y_g_hat_detach, apply=True):
        """train.py:133-160.  apply=False stops after the backward (graph segments: reduce + update come later)."""
        losses, _parts = self._d_forward(y, y_g_hat_detach)
        losses['disc_all'].backward()
        losses = _detached(losses)           # nobody differentiates them again: let the autograd graph go now
        if not apply:
            for d in self.discs:             # join the streams the gradient flushes ran on (a graph segment ends here)
                d.bank().sync_grads()
            return losses
        self._d_reduce()
        self.optim_d.step(use_bank_flag=True)
        return losses

    def _d_reduce(self, replay=False):
        """replay: called between replayed graph segments — the flush hooks that issue the all-reduces in the eager step
        do not run there (no Python runs inside a replay), so every bank is reduced here"""
        if self.dp.enabled:
            for d in self.discs:
                if replay or getattr(d.bank(), 'on_flush', None) is None:
                    d.bank().sync_grads()
                    self.dp.reduce_async(d.bank().gflat)
            self.dp.wait()

    def g_step(self, y, y_g_hat, apply=True):
        """train.py:163-193."""
        self.optim_g.zero_grad()
        losses = {}

        def wave_losses():
            """the loss terms that depend on the two waves only (train.py:165-168)"""
            out = {'env': envelope_loss(y, y_g_hat) if hp.envelope_loss else None,
                   'dyn': dynamic_loss(y, y_g_hat) if hp.dynamic_loss else None,
                   'sm': strip_mirror_loss(y_g_hat) if hp.strip_mirror_loss else None}
            if self.mtd is None:
                out['mstft'] = multi_stft_loss(y, y_g_hat, ret_loss=True)
            return out
        # They share nothing with the discriminators: with WAVE_LOSSES_FORKED they are one more branch of the stacks' fork — their
        # launches (and, autograd replaying every backward on its forward's stream, their backward) run beside the stacks'
        # instead of in front of them.  The spectrogram stack takes the spectra as its input: with it the multi-resolution STFT
        # stays in front.
        if self.mtd is not None:
            losses['mstft'], (S, S_g_hat) = multi_stft_loss(y, y_g_hat, ret_loss=True, ret_specs=True)
        forked = WAVE_LOSSES_FORKED
        if not forked:
            losses.update(wave_losses())
        self._freeze(True)       # the reference lets D weight gradients accumulate and discards them at the next
        try:                     # optim_d.zero_grad() (train.py:133): skipping them changes no result
            jobs = [('s', (self.msd, y, y_g_hat))]
            if self.mpd is not None:
                jobs.append(('p', (self.mpd, y, y_g_hat)))
            if self.mtd is not None:
                jobs.append(('t', (self.mtd, S, S_g_hat)))
            stack_outs = run_stacks([j for _, j in jobs], extra=[wave_losses] if forked else ())
            if forked:
                losses.update(stack_outs.pop())
            terms, weights = [losses['mstft']], [hp.w_loss_mstft]  # the total is ONE weighted sum over the terms, in this order
            for key, w in (('env', hp.w_loss_env), ('dyn', hp.w_loss_dyn), ('sm', hp.w_loss_sm)):
                if losses[key] is not None:
                    terms.append(losses[key]); weights.append(w)
            for (tag, _), (r, g, fr, fg) in zip(jobs, stack_outs):
                losses['gen_' + tag] = generator_loss(g, r)
                losses['fm_' + tag] = feature_loss(fr, fg)
                terms += [losses['gen_' + tag], losses['fm_' + tag]]; weights += [1.0, hp.w_loss_fm]
            total = ops.weighted_sum(terms, weights)
            losses['gen_all'] = total
            self.generator.bank().set_flag(total)
            with ops.noise_grad_accumulate():
                total.backward()
        finally:
            self._freeze(False)
        losses = _detached(losses)
        if not apply:
            self.generator.bank().sync_grads()
            return losses
        self._g_reduce()
        self.optim_g.step(use_bank_flag=True)
        return losses

    def _g_reduce(self):
        if self.dp.enabled:
            self.generator.bank().sync_grads()
            self.dp.reduce_async(self.generator.bank().gflat)
            self.dp.wait()

    def train_step(self, x, y_tmpl, y, noise_list=None):
        """One iteration of the batch loop (train.py:121-193).  x [B,80,T/256], y_tmpl / y [B,1,T] on the GPU.
        Returns (d_losses, g_losses) as device scalars — nothing here synchronises with the host."""
        # the first step (and one after any step that met a new problem shape) also times the block shapes of every
        # conv / wgrad launch and keeps the fastest (rtg/tune.py); it is an ordinary train step otherwise
        tuning = tune.ENABLED and (not self._tuned or tune.MISSED)
        # data parallel: rank 0 times, the others run this step on the library's heuristic shapes and take rank 0's picks
        # after it (DataParallel.sync_tuner); `tuning` is the same on every rank (same shapes, same tables)
        tune.ACTIVE = tuning and (not self.dp.enabled or dist.get_rank(self.dp.group) == 0)
        tune.MISSED = False
        # the first step that runs entirely on settled block shapes is watched: standard weight images no launch of it read
        # leave the pack launches (rtg/bank.py: lean_pack; Trainer.lean_pack_enabled = False keeps everything)
        observe = self._lean_pending and not tuning and self.lean_pack_enabled
        if observe:
            for m in (self.generator, *self.discs):
                m.bank().observe_std()
        try:
            with stft_cache():
                y_g_hat = self.generator(x, y_tmpl, noise_list) if noise_list is not None else self.generator(x, y_tmpl)
                assert y.shape[-1] == y_g_hat.shape[-1]
                y_det = y_g_hat.detach()
                dl = {}
                for _ in range(self.d_train_times):
                    dl = self.d_step(y, y_det)
                gl = self.g_step(y, y_g_hat)
        finally:
            self._tuned = self._tuned or tuning
            # a step that timed block shapes (or, data parallel, ran on the heuristic shapes while rank 0 timed) is not yet
            # the steady state: the next step may take other split counts, i.e. other partial buffers and a rebuilt
            # weight-norm job table (a host-to-device copy: illegal under capture) — prepare_graphs runs one more first
            self._settle = tuning
            tune.ACTIVE = False
        if tuning:
            self._lean_pending = True
        elif observe:
            self._lean_pending = False
            self.lean_dropped = sum(m.bank().lean_pack() for m in (self.generator, *self.discs))
        if tuning and self.dp.enabled:
            tune.MISSED = self.dp.sync_tuner()
        self.steps += 1
        return dl, gl

    # -- the same step replayed from HIP graphs
    def prepare_graphs(self, x, y_tmpl, y):
        """tune (eagerly, two steps) and capture the step's HIP graphs WITHOUT replaying them: under data parallelism every
        rank must know that every rank captured before the first replay — a rank whose capture failed would otherwise
        enter other collectives than the ranks already replaying (bench.py agrees on the outcome in between)."""
        if self._graphs is None:
            # block shapes are timed eagerly, never under capture, and the step captured is the steady state: eager steps
            # until one ran entirely on tuned shapes (data parallel: the same count on every rank — `tuning` is agreed)
            for _ in range(4):
                if self._tuned and not tune.MISSED and not self._settle:
                    break
                self.train_step(x, y_tmpl, y)
            self.dp.drain()                                     # no collective in flight on any rank while one captures
            self._capture(x, y_tmpl, y)

    def train_step_graphed(self, x, y_tmpl, y):
        """train_step() captured once into HIP graphs and replayed: the ~900 kernel launches of a step (and the forks /
        joins of the sub-network streams) are issued by the graph executor instead of the Python autograd machinery.
        Measured on MI355X at batch 32: 31.9 vs 32.4 ms eager (round 1, with slower kernels: 41.6 vs 41.7); bench.py
        times this path, the default trainer API stays eager.  ROCm 7.2 notes: a fork inside a forked stream crashes
        hipStreamEndCapture (hence the flat fork of run_stacks), and a capturing stream must not wait on a stream of
        an earlier capture (WeightBank.sync_grads re-homes the flush stream).  The step is cut where data parallelism exchanges
        gradients — [G forward, D backward] | [D update, D backward] ... | [D update, G backward] | [G update] — and the
        all-reduces run eagerly between the segments, so one code path serves 1 and N GPUs.  Inputs are copied into
        static buffers; the returned loss dicts hold static device scalars overwritten by every replay.  The learning
        rate is a launch argument: end_epoch() drops the graphs and the next step captures them again."""
        if self._graphs is not None and self._graph_gen != self._table_gen():
            # a bank rebuilt its device job tables since the capture (an eager launch at another shape restored a weight
            # image the lean pack had dropped, rtg/bank.py:restore_std): the captured pack launches still name the old
            # tables (kept alive, so the replay is memory-safe) — capture the step again on the new ones
            self._graphs = None
        self.prepare_graphs(x, y_tmpl, y)
        sx, sy_tmpl, sy = self._static_in
        for dst, src in ((sx, x), (sy_tmpl, y_tmpl), (sy, y)):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        for graph, after in self._graphs:
            graph.replay()
            if after is not None:
                after()
        self.steps += 1
        return self._static_out

    def _capture(self, x, y_tmpl, y):
        assert x.is_cuda and not tune.ACTIVE
        self._static_in = (x.clone(), y_tmpl.clone(), y.clone())
        sx, sy_tmpl, sy = self._static_in
        if self.generator.noise.salt is None:   # the noise seeds are launch arguments: mix in a device
            self.generator.noise.salt = self.optim_g.step_tensor(self.generator)   # word that changes every step
        hooks = [(d, d.bank().on_flush) for d in self.discs]
        for d, _ in hooks:
            d.bank().on_flush = None                            # collectives stay outside the graphs
        pool = torch.cuda.graph_pool_handle()
        if self._cap_stream is None:               # one capture stream per trainer, outside torch's stream pool (the RCCL
            self._cap_stream = new_stream()        # stream is a pool stream: rtg/lib.py:new_stream)
        cap = self._cap_stream
        cap.wait_stream(torch.cuda.current_stream())
        state = {}
        n_d = self.d_train_times

        # Where the graphs are cut under data parallelism (RTG_DP_CUT):
        #   'update' (default, round 4): one segment per optimizer update, as on one GPU — the discriminators' backward passes
        #       stay in ONE graph, forked over their streams, and every bank's all-reduce is issued after the segment.  The
        #       exchange is not overlapped with compute: 54 MB (G + MSD + MPD) to 109 MB (full stack) per D update, ~0.3-1 ms
        #       over xGMI, 2-4 % of the step.
        #   'disc' (round 3): the D backward cut per discriminator, largest gradient buffer first (MTD, MPD, MSD: the stacks
        #       share no parameter and their loss terms are summed, so three backward calls give the bits of one); a stack's
        #       all-reduce runs on the communication stream while the next stack's backward segment replays.
        #       Measured on one MI355X over a 1-rank RCCL group (profiles/r04_dp_capture.txt): 30.8 ms/step against 27.8
        #       without the cuts — serialising the stacks' backward passes and five more graph launches per step cost more
        #       than the exchange they hide, so this is the opt-in for slow links.
        dp = self.dp.enabled and self.dp.exchange == 'disc'
        order = sorted(self.discs, key=lambda d: -d.bank().n_params)
        def d_forward_and_first():
            losses, parts = self._d_forward(sy, state['y_hat'].detach())
            state['dl'] = _detached(losses)
            state['parts'] = {id(d): t for d, t in parts}
            d_backward(order[0])()

        def d_backward(d):
            def run():
                state['parts'].pop(id(d)).backward()
                d.bank().sync_grads()
            return run

        def reduce_of(d, last):
            def run():
                self.dp.reduce_async(d.bank().gflat)
                if last:
                    self.dp.wait()
            return run

        def seg_first():
            state['cache'] = stft_cache()
            state['cache'].__enter__()
            state['y_hat'] = self.generator(sx, sy_tmpl)
            if dp:
                d_forward_and_first()
            else:
                state['dl'] = self.d_step(sy, state['y_hat'].detach(), apply=False)

        def seg_d(i):
            def run():
                self.optim_d.step(use_bank_flag=True)
                if i < n_d:
                    if dp:
                        d_forward_and_first()
                    else:
                        state['dl'] = self.d_step(sy, state['y_hat'].detach(), apply=False)
                else:
                    state['gl'] = self.g_step(sy, state['y_hat'], apply=False)
                    state['cache'].__exit__(None, None, None)
            return run

        def seg_last():
            self.optim_g.step(use_bank_flag=True)

        def after_g():
            self._g_reduce()

        def d_tail():
            """the segments that finish a D update whose forward + first backward sit in the previous segment"""
            if not dp:
                return []
            return [(d_backward(d), reduce_of(d, j == len(order) - 1)) for j, d in enumerate(order) if j > 0]

        def after_d():
            self._d_reduce(replay=True)

        first_after = reduce_of(order[0], len(order) == 1) if dp else after_d
        segs = [(seg_first, first_after)] + d_tail()
        for i in range(1, n_d + 1):
            last_d = i == n_d
            segs.append((seg_d(i), after_g if last_d else first_after))
            if not last_d:
                segs += d_tail()
        segs.append((seg_last, None))
        graphs = []
        try:
            with torch.cuda.stream(cap):
                for body, after in segs:
                    g = torch.cuda.CUDAGraph()
                    # 'thread_local': other threads of this process keep their right to call the HIP runtime while this
                    # one captures — ProcessGroupNCCL's watchdog polls events of the collectives it tracks (in the
                    # default 'global' mode its hipEventQuery fails, invalidates the capture and ends the process:
                    # GPUTEST_r03); the autograd thread that launches the backward is not policed either way
                    with torch.cuda.graph(g, pool=pool, stream=cap, capture_error_mode=CAPTURE_ERROR_MODE):
                        body()
                        if config.get('RTG_TEST_FAIL_CAPTURE') == '1':          # (bench.py's fallback, exercised on the GPU box)
                            torch.zeros(4).to(sx.device)                        # a synchronous copy: illegal under capture
                        if self._capture_hook is not None:      # (tests hold a capture open: tests/test_zz_dp_gpu.py)
                            self._capture_hook()
                    graphs.append((g, after if self.dp.enabled else None))
        except BaseException:
            # leave no stream in a (possibly invalidated) capture: every later allocation would fail, the eager step too
            lib.rtg_stream_end_capture(C.c_void_p(cap.cuda_stream))
            try:
                torch.cuda.synchronize()
            except Exception:       # noqa: BLE001  (the error of the failed capture, reported once more)
                pass
            raise
        finally:
            for d, h in hooks:
                d.bank().on_flush = h
        torch.cuda.current_stream().wait_stream(cap)
        self._graphs = graphs
        self._graph_gen = self._table_gen()
        self._static_out = (state['dl'], state['gl'])

    def end_epoch(self):
        self.scheduler_g.step()
        self.scheduler_d.step()
        self._graphs = None            # the learning rate is baked into the captured AdamW launches

    # -- checkpoints in the reference's layout (train.py:263-273)
    def checkpoint_dicts(self, epoch):
        do = {'msd': self.msd.state_dict(), 'optim_g': self.optim_g.state_dict(), 'optim_d': self.optim_d.state_dict(),
              'steps': self.steps, 'epoch': epoch}
        if self.mpd is not None:
            do['mpd'] = self.mpd.state_dict()
        if self.mtd is not None:
            do['mtd'] = self.mtd.state_dict()
        return {'generator': self.generator.state_dict()}, do

    def save(self, log_path, epoch):
        g, do = self.checkpoint_dicts(epoch)
        save_checkpoint(os.path.join(log_path, f'g_{self.steps:08d}'), g)        # noqa: F405
        save_checkpoint(os.path.join(log_path, f'do_{self.steps:08d}'), do)      # noqa: F405

    def resume(self, log_path):
        cp_g, cp_do = scan_checkpoint(log_path, 'g_'), scan_checkpoint(log_path, 'do_')   # noqa: F405
        if cp_g is None or cp_do is None:
            return -1
        self.generator.load_state_dict(load_checkpoint(cp_g, device)['generator'])         # noqa: F405
        sd = load_checkpoint(cp_do, device)                                                # noqa: F405
        for tag, m in (('msd', self.msd), ('mpd', self.mpd), ('mtd', self.mtd)):
            if m is not None and tag in sd:
                m.load_state_dict(sd[tag])
        self.optim_g.load_state_dict(sd['optim_g'])
        self.optim_d.load_state_dict(sd['optim_d'])
        self.steps = sd['steps']
        # train.py:87-88: the schedulers are rebuilt on the loaded optimizers with last_epoch = the saved epoch
        self.scheduler_g = ExponentialLR(self.optim_g, gamma=hp.lr_decay, last_epoch=sd['epoch'])
        self.scheduler_d = ExponentialLR(self.optim_d, gamma=hp.lr_decay, last_epoch=sd['epoch'])
        self._graphs = None            # the learning rate is baked into captured AdamW launches
        return sd['epoch']
