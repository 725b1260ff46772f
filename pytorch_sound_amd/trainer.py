"""Training loop with the public surface of pytorch_sound/trainer.py (drop-in surface #2).

Same constructor signature, same overridable ``forward(*inputs, is_logging=False) -> (loss, meta)``,
same checkpoint layout ({save_dir}/models/{save_prefix}/{ModelClass}/step_{step:06d}.chkpt with keys
step / model / optim / pretrained_step / seed [/ scheduler], plus {ModelClass}.best.chkpt), same log
lines.  What is new sits behind that surface:

* data parallelism: when a torch.distributed process group exists (one process per GPU, RCCL over
  xGMI), gradients are all-reduced through ``FlatGradReducer`` overlapped with backward, rank 0 alone
  logs and saves, the validation loss is averaged over ranks and the NaN-skip decision is made
  collectively (every rank skips or none does - otherwise the ranks would deadlock in the all-reduce);
* the H2D copy of the batch stays ``to_device`` (lazy ``.cuda(non_blocking=True)``); on a machine
  without a GPU the batch is passed through unchanged so the loop is testable on CPU.
"""
import abc
import enum
import glob
import os
from pytorch_sound_amd import _switches as _sw
from collections import defaultdict
from typing import Any, Dict, Tuple

import numpy as np
import torch
import torch.nn as nn

from pytorch_sound_amd import distributed as pdist
from pytorch_sound_amd.settings import SAMPLE_RATE
from pytorch_sound_amd.utils.commons import get_loadable_checkpoint, log
from pytorch_sound_amd.utils.tensor import to_device, to_numpy


class LogType(enum.Enum):
    SCALAR: int = 1
    IMAGE: int = 2
    ENG: int = 3
    AUDIO: int = 4
    PLOT: int = 5
    TEXT: int = 6


class _MemoryWriter:
    """Stand-in used when neither tensorboardX nor torch.utils.tensorboard is importable (and on
    non-zero ranks): keeps scalars in memory so behaviour stays observable."""

    def __init__(self, log_dir=None, flush_secs=10):
        self.log_dir = log_dir
        self.scalars = []

    def add_scalar(self, tag, value, global_step=None):
        self.scalars.append((tag, float(value), global_step))

    def add_image(self, *args, **kwargs):
        pass

    add_audio = add_text = add_image


def _make_writer(log_dir: str, active: bool):
    if active:
        try:
            from tensorboardX import SummaryWriter
            return SummaryWriter(log_dir=log_dir, flush_secs=10)
        except ImportError:
            pass
    return _MemoryWriter(log_dir=log_dir)


def _to_buf(kind: str, value):
    """matplotlib rendering for IMAGE / PLOT metas (utils/plots.py of the reference) - cosmetic,
    only on logging steps; skipped silently when matplotlib is unavailable."""
    try:
        import matplotlib
        matplotlib.use('Agg')
        import matplotlib.pyplot as plt
    except ImportError:
        return None
    fig = plt.figure(figsize=(8, 4))
    if kind == 'image':
        plt.imshow(value, aspect='auto', origin='lower')
        plt.colorbar()
    else:
        plt.plot(value)
    fig.canvas.draw()
    buf = np.asarray(fig.canvas.buffer_rgba())[..., :3].transpose(2, 0, 1).copy()
    plt.close(fig)
    return buf



def _independent_stream(tries: int = 6):
    """a side stream whose host->device copies really run next to the current stream's kernels.  HIP multiplexes streams onto a
    few hardware queues; a prefetch stream that lands on the compute stream's queue makes every step wait for the NEXT batch's copy
    (measured on MI355X: 0.96 ms per step when the queues differ, 2.0-2.9 ms when they coincide - which of the two a fresh stream gets
    varies from run to run).  So candidates are probed: a copy enqueued behind a busy compute stream must finish before the compute does."""
    cur = torch.cuda.current_stream()
    dev = cur.device
    busy = torch.empty(1 << 24, device=dev)                   # 64 MB: a fill takes ~15-20 us, 120 of them ~2 ms
    host = torch.empty(1 << 18, dtype=torch.float32).pin_memory()
    dst = torch.empty(1 << 18, dtype=torch.float32, device=dev)
    best = None
    for attempt in range(tries):
        cand = torch.cuda.Stream(device=dev)
        torch.cuda.synchronize(dev)
        done_busy, done_copy = torch.cuda.Event(), torch.cuda.Event()
        for _ in range(120):
            busy.fill_(1.0)
        done_busy.record(cur)
        with torch.cuda.stream(cand):
            dst.copy_(host, non_blocking=True)
            done_copy.record(cand)
        done_copy.synchronize()
        overlapped = not done_busy.query()                    # the copy is through while the fills are still running
        torch.cuda.synchronize(dev)
        if best is None:
            best = cand
        if overlapped:
            log('prefetch stream: candidate %d of %d runs next to the compute stream' % (attempt + 1, tries))
            return cand
    log('prefetch stream: no candidate of %d overlapped with the compute stream - copies may serialise with the step' % tries)
    return best


def dist_backend_is_nccl() -> bool:
    return torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_backend() == 'nccl'


class Trainer:

    def __init__(self, model: nn.Module, optimizer: torch.optim.Optimizer,
                 train_dataset, valid_dataset,
                 max_step: int, valid_max_step: int, save_interval: int, log_interval: int,
                 save_dir: str, save_prefix: str = 'save',
                 grad_clip: float = 0.0, grad_norm: float = 0.0,
                 pretrained_path: str = None, sr: int = None, scheduler=None,
                 seed: int = None):
        self.pretrained_path = pretrained_path
        self.model = model
        self.optimizer = optimizer
        self.scheduler = scheduler

        n_params = sum(p.numel() for p in self.model.parameters() if p.requires_grad)
        self._log('Model {} was loaded. Total {} params.'.format(self.model.__class__.__name__, n_params))

        self.train_dataset = self.repeat(train_dataset)
        self.valid_dataset = self.repeat(valid_dataset)

        self.step = 0
        self.sr = sr if sr else SAMPLE_RATE
        self.max_step = max_step
        self.save_interval = save_interval
        self.log_interval = log_interval
        self.save_dir = save_dir
        self.save_prefix = save_prefix
        self.grad_clip = grad_clip
        self.grad_norm = grad_norm
        self.valid_max_step = valid_max_step

        self.log_dir = os.path.join(save_dir, 'logs', self.save_prefix)
        self.model_dir = os.path.join(save_dir, 'models')
        os.makedirs(self.model_dir, exist_ok=True)
        os.makedirs(self.log_dir, exist_ok=True)
        self.writer = _make_writer(self.log_dir, pdist.is_main())

        self.seed = None
        self.load()                                   # restores step / weights (and a seed, see below)
        # as in the reference (trainer.py:124-130) the constructor argument wins over a restored seed
        self.seed = seed
        if not self.seed:
            self.seed = np.random.randint(np.iinfo(np.int32).max)
            if pdist.is_dist():                        # every rank must initialise identically
                self.seed = pdist.broadcast_int(self.seed, 0)
        np.random.seed(self.seed)
        torch.manual_seed(self.seed)
        if torch.cuda.is_available():
            torch.cuda.manual_seed(self.seed)

        if self.step == 0 and pretrained_path:
            self.load_pretrained_model()

        self.best_valid_loss = np.finfo(np.float32).max
        self.cur_best_valid_loss = self.best_valid_loss
        self.save_valid_loss = np.finfo(np.float32).max

        # data parallel: identical start on every rank, overlapped gradient all-reduce
        self._reducer = None
        force = os.environ.get('PSND_DDP_FORCE') == '1' and torch.distributed.is_available() and torch.distributed.is_initialized()
        if pdist.is_dist() or force:                   # force: a one-rank process group still runs the whole reducer path
            pdist.broadcast_module(self._bare_model)
            self._reducer = pdist.FlatGradReducer(self._bare_model, force=force, comm_dtype=self.ddp_comm_dtype)
            if pdist.is_dist() and self.ddp_block_granularity:
                from pytorch_sound_amd import cl
                cl.NODE_GRANULARITY = 'block'

    # `step` is a plain attribute in the reference; here reads are noticed while forward() runs under graph_steps = 'auto'
    @property
    def step(self):
        if self.__dict__.get('_watch_forward'):
            self.__dict__['_step_read_in_forward'] = True
        return self.__dict__.get('_step', 0)

    @step.setter
    def step(self, v):
        self.__dict__['_step'] = v

    def _maybe_adopt_optimizer(self):
        if self._opt_adopted is not None:
            return
        self._opt_adopted = False
        opt = self.optimizer
        if not self.adopt_optimizer or type(opt) not in (torch.optim.Adam, torch.optim.AdamW) or 'step' in opt.__dict__:
            return                                    # (an LR scheduler patched the instance's step(): leave it alone)
        from pytorch_sound_amd import optim as poptim
        for g in opt.param_groups:
            if (g.get('amsgrad') or g.get('maximize') or g.get('differentiable') or g.get('capturable')
                    or any(isinstance(g[k], torch.Tensor) for k in ('lr', 'eps', 'weight_decay')) or any(isinstance(b, torch.Tensor) for b in g['betas'])
                    or not all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in g['params'])):
                return
        opt.__class__ = poptim.AdamW if type(opt) is torch.optim.AdamW else poptim.Adam
        opt._plans = {}
        self._opt_adopted = True
        self._log('optimizer %s adopted: its steps run as one HIP launch (pytorch_sound_amd.optim)' % type(opt).__name__)

    # ------------------------------------------------------------------------------------------
    @property
    def _bare_model(self) -> nn.Module:
        m = self.model
        return m.module if isinstance(m, (nn.DataParallel, nn.parallel.DistributedDataParallel)) else m

    @staticmethod
    def _log(msg: str):
        if pdist.is_main():
            log(msg)

    @abc.abstractmethod
    def forward(self, *inputs, is_logging: bool = False) -> Tuple[torch.Tensor, Dict]:
        """override: returns (loss tensor, {name: (value, LogType)})"""
        raise NotImplementedError

    def _forward_resolved(self, *inputs, is_logging: bool = False):
        """self.forward(...) with deferred tensors resolved (deferred.py: a loss written with torch ops on a model's deferred estimate becomes
        the fused loss node here; anything else in `meta` the plain tensor it stands for)"""
        from pytorch_sound_amd.deferred import Deferred, resolve
        watch = (self.graph_steps == 'auto' and self._graph_auto_ok is None and torch.cuda.is_available()
                 and not torch.cuda.is_current_stream_capturing())
        if watch:
            import random
            before = (hash(random.getstate()), hash(np.random.get_state()[1].tobytes()), hash(torch.get_rng_state().numpy().tobytes()))
            self.__dict__['_watch_forward'], self.__dict__['_step_read_in_forward'] = True, False
        try:
            loss, meta = self.forward(*inputs, is_logging=is_logging)
        finally:
            if watch:
                self.__dict__['_watch_forward'] = False
        if watch:
            after = (hash(random.getstate()), hash(np.random.get_state()[1].tobytes()), hash(torch.get_rng_state().numpy().tobytes()))
            why = ('forward() reads self.step' if self.__dict__.get('_step_read_in_forward') else
                   'forward() draws host-side random numbers' if after != before else None)
            if why is not None:
                self._graph_auto_ok = False
                self._log("graph_steps = 'auto': %s - the steps stay eager (set graph_steps = True to capture anyway)" % why)
        if isinstance(loss, Deferred):
            real = resolve(loss)
            if meta:
                meta = {k: ((real if v[0] is loss else resolve(v[0])), *v[1:]) if isinstance(v, tuple) and v else v for k, v in meta.items()}
            loss = real
        elif meta and any(isinstance(v, tuple) and v and isinstance(v[0], Deferred) for v in meta.values()):
            meta = {k: (resolve(v[0]), *v[1:]) if isinstance(v, tuple) and v else v for k, v in meta.items()}
        return loss, meta

    def prepare(self, *inputs):
        """Optional override (not in the reference): parameter-free preprocessing of a batch (feature extraction)
        that runs eagerly BEFORE forward() - in graph mode it stays outside the captured graph, so its kernels can
        be timed individually and it may change shapes from step to step.  Returns the inputs of forward()."""
        return inputs

    # (not in the reference) data-parallel runs: one weight-norm backward per residual block instead of one per conv chain
    # (cl.NODE_GRANULARITY = 'block'), so that the parameter gradients ARRIVE over the course of the backward and the all-reduce of
    # the early buckets overlaps it.  Off by default: on one GPU it costs 24 % of the config-2 step (0.81 -> 1.00 ms: the chained pair
    # backward launches are per-block then), about what hiding ~5/6 of a 22 MB all-reduce over xGMI returns; set it
    # before constructing the Trainer when the interconnect is slower than that.
    ddp_block_granularity = False
    # (not in the reference) data-parallel runs: what crosses the wire per gradient bucket - None: the fp32 bucket; torch.bfloat16: a bf16
    # image of it (FlatGradReducer(comm_dtype=...): half the bytes per xGMI link, bf16-rounded contributions).  PSND_DDP_COMM=bf16 does the same.
    ddp_comm_dtype = None

    # (not in the reference) stage the NEXT training batch - host->device copy and prepare() - on a side stream while the
    # current step computes; the step's stream waits on an event, never the host.  prepare() must then be parameter-free
    # (it runs one step early).  Off by default: the data iterator is advanced one batch ahead of the step that uses it.
    prefetch_prepare = False
    # (not in the reference) only the host->device COPY of the next batch runs one step ahead on a side stream; prepare() stays
    # on the compute stream at the step that uses the batch (no second stream competing for the CUs)
    # None (default) = decided at the first training batch: on when the data set hands over PINNED host tensors (the case the copy can
    # overlap at all), off otherwise (device-resident or pageable batches: the reference's in-line `.cuda()`, trainer.py:202)
    prefetch_copy = None

    def _decide_prefetch_copy(self):
        """first training batch with prefetch_copy = None: look at what the data set hands over, use this batch in line"""
        raw = next(self.train_dataset)
        items = raw if isinstance(raw, (tuple, list)) else (raw,)
        gpu = torch.cuda.is_available() and self.move_batches_to_gpu
        pinned = gpu and any(isinstance(t, torch.Tensor) and not t.is_cuda and t.is_pinned() for t in items)
        self.prefetch_copy = bool(pinned) and not self.prefetch_prepare
        if self.prefetch_copy:
            self._log('batches arrive in pinned host memory: the next batch is copied on a side stream while a step computes '
                      '(Trainer.prefetch_copy; set it to False for the in-line copy)')
        batch = to_device(raw) if gpu else items
        return self.prepare(*batch)

    # (not in the reference) prepare() of batch k+1 on a side stream BEHIND step k's forward / backward: the side stream waits for
    # the step's last backward kernel (so persistent feature buffers - static_prepare - are free again) and its feature extraction runs
    # next to step k's optimizer launch and step k+1's weight prep, which leave most of the chip idle; step k+1 waits on an event.
    # prepare() must be parameter-free.  The data iterator is advanced one batch ahead of the step that uses it.
    overlap_prepare = False

    def _stage_overlapped(self):
        """called right behind the step's backward (graph replay) on the compute stream"""
        if not (self.overlap_prepare and torch.cuda.is_available()) or self.prefetch_prepare or self.prefetch_copy:
            return
        if getattr(self, '_ovl_stream', None) is None:
            self._ovl_stream = _independent_stream()
        side, cur = self._ovl_stream, torch.cuda.current_stream()
        side.wait_stream(cur)
        try:
            with torch.cuda.stream(side):
                raw = tuple(self._next_batch(self.train_dataset))
                for t in raw:
                    if isinstance(t, torch.Tensor) and t.is_cuda:
                        t.record_stream(side)
                batch = self.prepare(*raw)
                event = torch.cuda.Event()
                event.record(side)
            self._ovl = (batch, event)
        except StopIteration as e:                       # surfaces at the step that would have used the batch
            self._ovl = (e, None)

    def _stage_train_batch(self):
        side = self._pre_stream
        try:
            with torch.cuda.stream(side):
                batch = tuple(self._next_batch(self.train_dataset))
                if self.prefetch_prepare:
                    batch = self.prepare(*batch)
                event = torch.cuda.Event()
                event.record(side)
            return batch, event
        except StopIteration as e:                       # surfaces at the step that would have used the batch
            return e, None

    def setup_prefetch(self):
        """probe for the side stream now (a 64 MB scratch buffer, ~20 ms) instead of inside the first training step - call it
        before the model's memory peaks when the GPU is nearly full"""
        if (self.prefetch_prepare or self.prefetch_copy) and torch.cuda.is_available() and getattr(self, '_pre_stream', None) is None:
            self._pre_stream = _independent_stream()

    def _take_train_batch(self):
        if self.prefetch_copy is None and getattr(self, '_ovl', None) is None:
            return self._decide_prefetch_copy()
        if not ((self.prefetch_prepare or self.prefetch_copy) and torch.cuda.is_available()):
            ovl = getattr(self, '_ovl', None)
            if ovl is not None:                          # staged behind the previous step (overlap_prepare)
                self._ovl = None
                batch, event = ovl
                if event is None:
                    raise batch
                cur = torch.cuda.current_stream()
                cur.wait_event(event)
                for t in batch:
                    if isinstance(t, torch.Tensor) and t.is_cuda:
                        t.record_stream(cur)
                return batch
            return self.prepare(*self._next_batch(self.train_dataset))
        if self.prefetch_prepare and self.static_prepare:
            # prepare() of batch k+1 runs on the side stream BEFORE step k's graph replay is enqueued: with persistent feature
            # buffers (static_prepare) it would overwrite the very tensors that replay is about to read
            raise ValueError('Trainer.prefetch_prepare and Trainer.static_prepare are mutually exclusive: a prepare() that runs '
                             'one step ahead must return fresh tensors (set static_prepare = False), or only the copy may run '
                             'ahead (prefetch_copy)')
        if getattr(self, '_pre_stream', None) is None:
            self._pre_stream = _independent_stream()
        if getattr(self, '_pre', None) is None:
            self._pre = self._stage_train_batch()
        batch, event = self._pre
        if event is None:
            raise batch
        self._pre = self._stage_train_batch()            # enqueued ahead of this step's own kernels: overlaps with them
        cur = torch.cuda.current_stream()
        cur.wait_event(event)
        for t in batch:
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(cur)
        if not self.prefetch_prepare:
            batch = self.prepare(*batch)
        return batch

    # (not in the reference) False keeps the batches where the data set put them - a CPU training run on a machine that has a GPU
    # (the reference moves every batch with .cuda() unconditionally, trainer.py:202)
    move_batches_to_gpu = True

    def _next_batch(self, iterator):
        batch = next(iterator)
        if torch.cuda.is_available() and self.move_batches_to_gpu:
            return to_device(batch)
        return batch if isinstance(batch, (tuple, list)) else (batch,)

    # ------------------------------------------------------------------------------------------
    def run(self) -> float:
        try:
            for i in range(self.step + 1, self.max_step + 1):
                self.step = i
                if i % self.save_interval == 1:
                    self._log('------------- TRAIN step : %d -------------' % i)
                self.model.train()
                self.train(i)
                if i % self.save_interval == 0:
                    self._log('------------- VALID step : %d -------------' % i)
                    self.model.eval()
                    self.validate(i)
                    self.save(i)
        except KeyboardInterrupt:
            self._log('Train is canceled !!')
        return self.best_valid_loss

    def clip_grad(self):
        """trainer.py:184-191: clamp every gradient to +-grad_clip, then clip the global norm to grad_norm.  Called as it is by the
        synchronous path and by user code; the device-skip path folds both into the HIP optimizer's step (`_fused_clip`)."""
        if self.grad_clip:
            for p in self.model.parameters():
                if p.grad is not None:
                    p.grad = p.grad.clamp(-self.grad_clip, self.grad_clip)
        if self.grad_norm:
            torch.nn.utils.clip_grad_norm_([p for p in self.model.parameters() if p.requires_grad], self.grad_norm)

    # ---- NaN skip without a host sync --------------------------------------------------------------
    # The reference tests `loss != loss` on the host every step (trainer.py:205): a device->host sync that
    # idles the GPU between forward and backward.  When the optimizer implements the AMP `found_inf`
    # protocol (fused Adam/AdamW/SGD) and no scheduler is attached, the same decision is taken ON THE DEVICE:
    # the flag isnan(loss) (MAX-all-reduced under DDP, asynchronously) is handed to the optimizer kernel, which
    # skips the update (and its step count) itself; the log line is emitted as soon as the flag has reached the
    # host on its own.  Anything else falls back to the reference's synchronous check.
    async_nan_check = True

    def _can_skip_on_device(self, loss: torch.Tensor) -> bool:
        return (self.async_nan_check and loss.is_cuda and self.scheduler is None
                and getattr(self.optimizer, '_step_supports_amp_scaling', False))

    # The skip flag reaches the host without a device-to-host copy when the optimizer can leave it in host-visible memory itself
    # (pytorch_sound_amd.optim: psnd_adam_step_logged writes {flag, seq} into a pinned ring): a blit kernel + an event behind every
    # optimizer launch were ~14 us of a 0.67 ms config-2 step.  Other optimizers: a pinned scalar, an asynchronous copy and an event.
    _NAN_RING = 256

    def _nan_ring(self):
        ring = getattr(self, '_nan_ring_buf', None)
        if ring is None:
            ring = self._nan_ring_buf = torch.zeros((self._NAN_RING, 2), dtype=torch.int32).pin_memory()
            self._nan_seq = 0
        return ring

    def _poll_nan_log(self, block: bool = False):
        pending = getattr(self, '_nan_pending', None)
        if not pending:
            return
        if block and torch.cuda.is_available():
            torch.cuda.synchronize()
        keep = []
        for entry in pending:
            step, host_flag, event = entry
            if isinstance(event, int):                 # ring slot: host_flag = the slot, event = the sequence number to wait for
                ring = self._nan_ring_buf
                if int(ring[host_flag, 1]) == event:
                    if int(ring[host_flag, 0]) != 0:
                        self._log('{} cur step NAN is occured'.format(step))
                else:
                    keep.append(entry)
                continue
            if block:
                event.synchronize()
            if event.query():
                if float(host_flag.item()) > 0:
                    self._log('{} cur step NAN is occured'.format(step))
            else:
                keep.append(entry)
        self._nan_pending = keep

    @staticmethod
    def _nan_flag(loss: torch.Tensor) -> torch.Tensor:
        """fp32 scalar on the loss's device: 1 if the loss is NaN (`loss != loss`, trainer.py:205) - one launch (psnd_nan_flag) for an
        fp32 loss on the GPU, torch.isnan + cast otherwise"""
        ready = getattr(loss, 'psnd_nan_flag', None)       # a fused loss node formed the flag in the launch that formed the loss
        if ready is not None and ready.device == loss.device and ready.dtype == torch.float32 and ready.dim() == 0:
            return ready
        x = loss.detach()
        if x.is_cuda and x.dtype == torch.float32 and x.is_contiguous():
            from ._lib import lib, ptr, stream_ptr, check
            flag = torch.empty((), dtype=torch.float32, device=x.device)
            with torch.cuda.device(x.device):
                check(lib().psnd_nan_flag(ptr(x), x.numel(), ptr(flag), stream_ptr(x.device)), 'psnd_nan_flag')
            return flag
        return torch.isnan(x).any().to(torch.float32).reshape(())

    def _backward(self, loss: torch.Tensor):
        """loss.backward() with a persistent root gradient (autograd's own ones_like is a fill launch per step)"""
        if loss.dim() == 0 and loss.is_cuda and loss.dtype == torch.float32:
            one = getattr(self, '_root_grad', None)
            if one is None or one.device != loss.device:
                one = self._root_grad = torch.ones((), dtype=torch.float32, device=loss.device)
            loss.backward(gradient=one)
        else:
            loss.backward()

    def _train_device_skip(self, step: int, loss: torch.Tensor):
        flag = self._nan_flag(loss)
        if self._reducer is not None and (pdist.is_dist() or self._reducer.active):   # (the same condition as _finish_device_skip: a forced one-rank group)
            self._reducer.set_flag(flag)               # rides along with the last gradient bucket
        self._backward(loss)
        self._finish_device_skip(step, flag)

    def _loss_is_nan(self, loss: torch.Tensor) -> bool:
        bad = bool(loss != loss)                       # host sync, as in the reference (trainer.py:205)
        if self._reducer is not None:
            bad = pdist.all_reduce_scalar(1.0 if bad else 0.0, 'max', loss.device if loss.is_cuda else None) > 0
        return bad

    # ---- hipGraph mode ---------------------------------------------------------------------------------------
    # A config-2 step is ~300 kernel launches of 5-20 us each: enqueueing them one by one costs the host more
    # (3.7 ms) than the GPU needs to run them (~2 ms).  With `graph_steps = True` the launch-bound part of a
    # non-logging step - gradient zeroing, forward(), the NaN flag and backward - is captured ONCE per input
    # signature into a hipGraph (torch.cuda.CUDAGraph) and replayed; prepare() before it and the tail after it
    # (gradient all-reduce over RCCL, clipping, the fused optimizer step with its on-device NaN skip) stay
    # eager, so nothing of RCCL is ever captured.  The replayed backward writes its gradients into static tensors of
    # the graph's memory pool; under DDP they are copied into the flat buckets of FlatGradReducer (one multi-tensor
    # copy, the per-bucket backward hooks are deferred to one all-reduce after the replay).  Requirements:
    # CUDA inputs, no scheduler, an optimizer with the found_inf protocol (fused Adam/AdamW/SGD), a forward()
    # free of host synchronisation.  Logging steps and the first `graph_warmup` steps of a signature run eagerly.
    #
    # graph_steps = 'auto' (default, round 6): capture when it is SAFE to - no gradient reducer (a data-parallel run opts in with True, all
    # ranks alike), and during the eager warm-up steps forward() neither consumed host-side randomness (python `random`, numpy, torch's
    # CPU generator) nor read `self.step`: a replayed step repeats the HOST-side decisions of the captured call, so a forward that draws
    # a crop on the host or anneals a weight by the step count must keep running.  A capture that fails (a host synchronisation inside
    # forward: `.item()`, a pageable copy) is logged once and the Trainer stays eager.  What cannot be detected: side effects of forward()
    # on Python state (appending to a list every step) - such a Trainer sets graph_steps = False.  True: always capture (errors raise);
    # False: never.
    graph_steps = 'auto'
    graph_warmup = 3
    _graph_auto_ok = None          # 'auto': None undecided / True / False (stay eager)
    # Trainer adopts a stock `torch.optim.Adam` / `AdamW` handed to it (exact types, amsgrad / maximize / capturable off, fp32 HIP
    # parameters): the instance keeps its identity, param_groups and state, its class becomes pytorch_sound_amd.optim.Adam / AdamW - one
    # launch per step, on-device NaN skip, capturable steps (trainer.py:215-216 calls whatever optimizer the recipe built).
    adopt_optimizer = True
    _opt_adopted = None
    # (not in the reference) a prepare() that writes its outputs into persistent buffers of its own and returns those same
    # tensors every step may say so: the captured graph then reads them where they are (no copy into graph-owned inputs per step)
    static_prepare = False

    def _train_graph(self, step: int, batch) -> bool:
        if not (self.async_nan_check and self.scheduler is None
                and getattr(self.optimizer, '_step_supports_amp_scaling', False)
                and all(isinstance(t, torch.Tensor) and t.is_cuda for t in batch)):
            return False
        if not hasattr(self, '_graphs'):
            self._graphs = {}
        sig = tuple((tuple(t.shape), t.dtype) for t in batch)
        st = self._graphs.pop(sig, None) or {'seen': 0}
        self._graphs[sig] = st                         # most recently used last
        if 'graph' not in st:
            if st['seen'] < self.graph_warmup:
                st['seen'] += 1
                self._trim_graph_cache()
                return False
            if self.graph_steps == 'auto':
                # variable-length batches (a bucketed loader) bring a new input signature every few steps: a capture costs ~0.1 s and pays only
                # when its signature comes back - once 8 signatures are captured and each was replayed fewer than 8 times on average, new
                # signatures run eagerly (the captured ones keep replaying)
                caps, reps = getattr(self, '_auto_caps', 0), getattr(self, '_auto_replays', 0)
                if caps >= 8 and reps < 8 * caps:
                    return False
                try:
                    self._capture(st, batch)
                    self._graph_auto_ok = True
                    self._auto_caps = caps + 1
                except Exception as e:                 # noqa: BLE001 - a forward() that cannot be captured: stay eager, say why once
                    self._graph_auto_ok = False
                    self._graphs.pop(sig, None)
                    try:
                        torch.cuda.synchronize()
                    except Exception:                  # noqa: BLE001
                        pass
                    for p in self._bare_model.parameters():
                        p.grad = None
                    self._log("graph_steps = 'auto': the step could not be captured (%s) - the steps stay eager" % repr(e)[:200])
                    return False
            else:
                self._capture(st, batch)
            self._trim_graph_cache()
        for dst, src in zip(st['inputs'], batch):
            if src.data_ptr() != dst.data_ptr():           # static_prepare: already there
                dst.copy_(src, non_blocking=True)
        st['graph'].replay()
        self._auto_replays = getattr(self, '_auto_replays', 0) + 1
        self._stage_overlapped()                       # the next batch's prepare(): side stream, next to this step's optimizer launch
        if self._reducer is not None and st.get('ddp') in ('events', 'capture'):
            # the captured backward filled the flat buckets itself and marked where each is complete: bucket i is all-reduced
            # while the replay is still producing bucket i + 1 (FlatGradReducer, graph mode)
            self._reducer.after_replay(st['ddp'], time_marks=getattr(self, '_ddp_time_marks', False))
        elif self._reducer is not None:                # deferred: into the flat buckets, reduced in _finish_device_skip
            self._reducer.load_grads(st['grads'])
            if self._reducer.active:
                self._reducer.set_flag(st['flag'])
        else:
            for p, g in st['grads'].items():
                p.grad = g
        self._finish_device_skip(step, st['flag'])
        return True

    # every captured signature owns a memory pool with that shape's activations: variable-length batches (bucketed loaders)
    # must not pile them up.  Least recently used graphs go first; bare counters of shapes still warming up are capped too.
    graph_cache_size = 8

    def _trim_graph_cache(self):
        captured = [k for k, v in self._graphs.items() if 'graph' in v]
        for k in captured[:max(0, len(captured) - self.graph_cache_size)]:
            del self._graphs[k]
        if len(self._graphs) > 64 * max(1, self.graph_cache_size):
            for k in [k for k, v in self._graphs.items() if 'graph' not in v][:len(self._graphs) // 2]:
                del self._graphs[k]

    def _capture(self, st, batch):
        red = self._reducer
        mode = red.graph_mode() if red is not None and red.active else None
        # static_prepare: prepare() returns the SAME buffers every step (features written in place) - they are the graph's inputs
        st['inputs'] = tuple(batch) if self.static_prepare else tuple(t.clone() for t in batch)
        from . import cl
        # see cl.py: no extra graph branches (batch sections, resblock / parameter-side branches) next to other live streams; decided per
        # capture - a Trainer without such streams captured later in the same process gets its branches back
        # Round 5: graph branches and a gradient reducer in one captured step.  The reducer releases every bucket from a stream of its own
        # behind the events of all streams that produced its gradients (distributed.FlatGradReducer._release_on: a captured RCCL all-reduce
        # issued from a graph branch ended the process in hipStreamEndCapture), and a parameter handed over from inside a node arrives once
        # (the engine calls the post-accumulate hooks of its AccumulateGrad node although the node returned None for it; counted twice,
        # buckets left before their last gradients - zeros next to branches, an all-reduce ahead of its data on several ranks).
        # PSND_DDP_BRANCHES=0: round 4's behaviour (no branches next to a reducer) for A/B runs.
        red_blocks = red is not None and red.active and _sw.lab('PSND_DDP_BRANCHES', '1') != '1'
        # Round 5: next to `prefetch_copy` (a COPY-only side stream, picked by _independent_stream so that it runs next to the compute
        # stream) the branches stay on: config 3 with pinned host batches 2.86-2.93 ms with them, 3.55-3.65 without, 2.91 with the pool
        # on the device - six fresh processes each for configs 3 and 4, no slow step among them (tools/r05/prefetch_branches.py;
        # PSND_PREFETCH_BRANCHES=0 switches them off).  `prefetch_prepare` runs KERNELS on its side stream (prepare() of the next batch):
        # that stream competes with the branches for the hardware queues, as measured in round 3 with batch sections - branches off.
        pre_copy_only = (bool(self.prefetch_copy) or getattr(self, '_pre_stream', None) is not None) and not bool(self.prefetch_prepare)
        pre_blocks = bool(self.prefetch_prepare) or (pre_copy_only and _sw.lab('PSND_PREFETCH_BRANCHES', '1') != '1')
        cl.AUTO_SECTIONS = not (red_blocks or pre_blocks)
        params = [p for p in self._bare_model.parameters() if p.requires_grad]
        if getattr(self, '_root_grad', None) is None or self._root_grad.device != st['inputs'][0].device:
            self._root_grad = torch.ones((), dtype=torch.float32, device=st['inputs'][0].device)   # not inside the capture
        while True:
            for p in params:
                p.grad = None                          # backward then WRITES its gradients (no zero fill, no += kernels)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            ok = True
            try:
                # thread_local: the process group's watchdog thread keeps querying events of earlier collectives while this
                # thread captures ("operation not permitted when stream is capturing" under the default global mode)
                with torch.cuda.graph(graph, capture_error_mode='thread_local' if red is not None else 'global'):
                    loss, _ = self._forward_resolved(*st['inputs'], is_logging=False)
                    st['flag'] = self._nan_flag(loss)
                    if mode in ('events', 'capture'):
                        red.capture_begin(st['flag'], mode)   # the captured backward fills and releases the buckets itself
                    elif red is not None:
                        red.deferred = True            # no collective from inside the captured backward
                    self._backward(loss)
                    if mode in ('events', 'capture'):
                        red.capture_end()
            except Exception as e:                     # noqa: BLE001 - e.g. a runtime that cannot capture the release nodes
                if mode in (None, 'deferred'):
                    raise
                ok = False
                self._log('graph capture with DDP mode %s failed (%s)' % (mode, repr(e)[:200]))
                red._capturing = None
                torch.cuda.synchronize()
            finally:
                if red is not None:
                    red.deferred = False               # eager (logging) steps keep the overlapped per-bucket all-reduce
            if mode in ('events', 'capture') and pdist.is_dist():
                # every rank must replay the SAME collective sequence: a rank that fell back on its own would issue its all-reduces
                # after the replay while the others have them inside the graph - agree (MIN over ranks) and fall back together
                dev = st['flag'].device if dist_backend_is_nccl() else None
                ok = pdist.all_reduce_scalar(1.0 if ok else 0.0, 'min', dev) > 0
            if ok:
                break
            self._log('falling back to the deferred all-reduce on every rank')
            del graph
            mode = 'deferred'
        st['graph'] = graph
        st['ddp'] = mode
        st['grads'] = {p: p.grad for p in params}      # static tensors of the graph's memory pool
        self._log('captured the training step as a hipGraph for inputs %s' % (
            ', '.join('x'.join(map(str, t.shape)) for t in batch)))

    def _optimizer_covers_model(self) -> bool:
        """the fused clip takes its global norm over the OPTIMIZER's parameters; Trainer.clip_grad (trainer.py:184-191) over every model
        parameter that requires a gradient.  The two agree only when the sets agree - otherwise the stock clip_grad() runs.
        (With the fused clip `p.grad` itself stays unclipped - and un-averaged under DDP; read `optimizer.last_grad_norm`.)"""
        key = (id(self.optimizer), sum(len(g['params']) for g in self.optimizer.param_groups))
        cached = getattr(self, '_opt_cover', None)
        if cached is None or cached[0] != key:
            model = {id(p) for p in self._bare_model.parameters() if p.requires_grad}
            opt = {id(p) for g in self.optimizer.param_groups for p in g['params']}
            cached = self._opt_cover = (key, model == opt)
        return cached[1]

    def _finish_device_skip(self, step: int, flag: torch.Tensor):
        """eager tail of a step whose backward has run: NaN flag to the host (asynchronously), gradient all-reduce,
        clipping, optimizer step with on-device skip"""
        grad_scale = None
        # K18: clamp + global-norm clipping inside the optimizer launch (psnd_grad_sumsq + psnd_adam_step) when the optimizer can and
        # clip_grad() is the stock one; the averaging over ranks is then the kernel's grad_scale as well
        fused_clip = ((self.grad_clip or self.grad_norm) and getattr(self.optimizer, '_supports_fused_clip', False)
                      and type(self).clip_grad is Trainer.clip_grad and self._optimizer_covers_model())
        if self._reducer is not None and (pdist.is_dist() or self._reducer.active):      # (active without is_dist: a forced one-rank group)
            # ONE collective per bucket: the flag sits behind the last bucket (set_flag), and when nothing clips the gradients the
            # division by the world size is left to the optimizer kernel (grad_scale) instead of a pass over the buckets
            in_opt = fused_clip or not (self.grad_clip or self.grad_norm)
            self._reducer.finish(average=not in_opt)
            # the sum of the ranks' 0 / 1 flags (divided by the world size when the buckets are averaged): psnd_adam_step skips on any
            # non-zero value and takes it as it is; torch's fused optimizers test `found_inf == 1`, so they get it normalised (two launches)
            rf = self._reducer.flag
            raw_ok = rf.dtype == torch.float32 and getattr(self.optimizer, '_supports_flag_log', False)
            flag = rf.reshape(()) if raw_ok else (rf > 0).to(torch.float32).reshape(())
            if in_opt:
                if getattr(self, '_world_scale', None) is None or self._world_scale.device != flag.device:
                    self._world_scale = torch.full((), float(pdist.world_size()), dtype=torch.float32, device=flag.device)
                grad_scale = self._world_scale
        elif self._reducer is not None:
            self._reducer.finish()
        if fused_clip:
            self.optimizer.fused_clip = (self.grad_clip, self.grad_norm)
        else:
            self.clip_grad()
        if not hasattr(self, '_nan_pending'):
            self._nan_pending = []
        use_ring = bool(getattr(self.optimizer, '_supports_flag_log', False)) and flag.is_cuda
        if use_ring:
            ring = self._nan_ring()
            if len(self._nan_pending) >= self._NAN_RING - 2:   # the host is a whole ring ahead of the device: let it catch up
                self._poll_nan_log(block=True)
            self._nan_seq += 1
            slot, seq = self._nan_seq % self._NAN_RING, self._nan_seq
            self.optimizer.flag_log = (ring, slot, seq)
        self.optimizer.found_inf = flag
        self.optimizer.grad_scale = grad_scale
        try:
            self.optimizer.step()
        finally:
            del self.optimizer.found_inf
            del self.optimizer.grad_scale
            if fused_clip:
                del self.optimizer.fused_clip
            if use_ring:
                del self.optimizer.flag_log
        if use_ring and getattr(self.optimizer, 'flag_logged', False):
            self._nan_pending.append((step, slot, seq))    # written by the optimizer launch itself
        else:
            # the flag goes to the host BEHIND the optimizer launch (which reads it on the device): the copy is off the step's critical path
            host_flag = torch.empty((), dtype=torch.float32, pin_memory=True)
            host_flag.copy_(flag, non_blocking=True)
            event = torch.cuda.Event()
            event.record()
            self._nan_pending.append((step, host_flag, event))
        self._poll_nan_log()

    # Python's cyclic collector walks every tracked object of the process in a full (generation-2) pass: with torch, numpy and a model
    # imported that is ~10^6 objects and tens of milliseconds during which no kernel is launched - the 50-120 ms steps of an eager loop
    # (tools/r06/stall.py: every slow step coincides with a gen-2 pass).  After the first steps - modules, plans, graph and autograd closures built -
    # everything alive is moved to the permanent generation once (gc.freeze): later passes only look at what the loop itself allocates.
    gc_freeze = True
    _gc_frozen = False
    _gc_freeze_after = 4

    def _maybe_freeze_gc(self):
        if self.gc_freeze and not self._gc_frozen:             # (per Trainer: what a later Trainer of the process builds is frozen after ITS first steps)
            self._steps_seen = getattr(self, '_steps_seen', 0) + 1
            if self._steps_seen > self._gc_freeze_after:
                import gc
                gc.collect()
                gc.freeze()
                self._gc_frozen = True

    def train(self, step: int):
        self._maybe_freeze_gc()
        log_flag = step % self.log_interval == 0
        batch = self._take_train_batch()
        self._maybe_adopt_optimizer()
        use_graph = self.graph_steps is True or (self.graph_steps == 'auto' and self._graph_auto_ok is not False and self._reducer is None)
        if use_graph and not log_flag and self._train_graph(step, batch):
            return
        if self._reducer is not None:
            self._reducer.zero_grad()
        else:
            self.optimizer.zero_grad()
        loss, meta = self._forward_resolved(*batch, is_logging=log_flag)

        if self._can_skip_on_device(loss):
            self._train_device_skip(step, loss)
            if log_flag and pdist.is_main():
                self.console_log('train', meta, step)
                try:
                    self.tensorboard_log('train', meta, step)
                except OverflowError:
                    pass
            return

        if self._loss_is_nan(loss):
            self._log('{} cur step NAN is occured'.format(step))
            return

        loss.backward()
        if self._reducer is not None:
            self._reducer.finish()
        self.clip_grad()
        self.optimizer.step()
        if self.scheduler is not None:
            self.scheduler.step()

        if log_flag and pdist.is_main():
            self.console_log('train', meta, step)
            try:
                self.tensorboard_log('train', meta, step)
            except OverflowError:
                pass

    def validate(self, step: int):
        self._poll_nan_log(block=True)
        loss = 0.
        stat = defaultdict(float)
        for i in range(self.valid_max_step):
            with torch.no_grad():
                batch_loss, meta = self._forward_resolved(*self.prepare(*self._next_batch(self.valid_dataset)), is_logging=True)
                loss += batch_loss
            for key, (value, log_type) in meta.items():
                if log_type == LogType.SCALAR:
                    stat[key] += value
            if (i % self.log_interval == 0 or i == self.valid_max_step - 1) and pdist.is_main():
                self.console_log('valid', meta, i + 1)

        loss /= self.valid_max_step
        for key in stat.keys():
            if key == 'loss':
                continue
            stat[key] = stat[key] / self.valid_max_step
        if pdist.is_dist():                            # mean over ranks
            w = pdist.world_size()
            dev = loss.device if isinstance(loss, torch.Tensor) and loss.is_cuda else None
            loss = pdist.all_reduce_scalar(float(loss), 'sum', dev) / w
            for key in sorted(stat.keys()):
                if key != 'loss':
                    stat[key] = pdist.all_reduce_scalar(float(stat[key]), 'sum', dev) / w
        stat['loss'] = loss

        if loss < self.best_valid_loss:
            self.best_valid_loss = loss

        msg = 'step {} / total stat'.format(step)
        for key, value in sorted(stat.items()):
            msg += '\t{}: {:.6f}'.format(key, value)
        self._log(msg)
        for key, value in stat.items():
            self.writer.add_scalar('valid/{}'.format(key), value, global_step=step)

    # ------------------------------------------------------------------------------------------
    @property
    def save_name(self) -> str:
        return self.save_prefix + '/' + self._bare_model.__class__.__name__

    def load(self, load_optim: bool = True):
        save_path = os.path.join(self.model_dir, self.save_name)
        check_files = glob.glob(os.path.join(save_path, '*'))
        if not check_files:
            self._log('No any checkpoint in {}. Loading network skipped.'.format(save_path))
            return
        latest_file = max(check_files, key=os.path.getctime)
        state_dict = torch.load(latest_file, map_location='cpu', weights_only=False)
        if 'seed' in state_dict:
            self.seed = state_dict['seed']
        self._bare_model.load_state_dict(get_loadable_checkpoint(state_dict['model']))
        if load_optim:
            self.optimizer.load_state_dict(state_dict['optim'])
        if self.scheduler is not None:
            self.scheduler.load_state_dict(state_dict['scheduler'])
        self.step = state_dict['step']
        self._log('checkpoint \'{}\' is loaded. previous step={}'.format(latest_file, self.step))

    def save(self, step: int):
        if not pdist.is_main():
            self.cur_best_valid_loss = self.best_valid_loss
            return
        state_dict = {
            'step': step,
            'model': get_loadable_checkpoint(self.model.state_dict()),
            'optim': self.optimizer.state_dict(),
            'pretrained_step': step,
            'seed': self.seed,
        }
        if self.scheduler is not None:
            state_dict['scheduler'] = self.scheduler.state_dict()

        save_path = os.path.join(self.model_dir, self.save_name)
        os.makedirs(save_path, exist_ok=True)
        torch.save(state_dict, os.path.join(save_path, 'step_{:06d}.chkpt'.format(step)))

        if self.best_valid_loss != self.cur_best_valid_loss:
            torch.save(state_dict, os.path.join(self.model_dir, self.save_name + '.best.chkpt'))
            self.cur_best_valid_loss = self.best_valid_loss
        self._log('step %d / saved model.' % step)

    def load_pretrained_model(self):
        assert os.path.exists(self.pretrained_path), 'You must define pretrained path!'
        ckpt = torch.load(self.pretrained_path, map_location='cpu', weights_only=False)
        self._bare_model.load_state_dict(get_loadable_checkpoint(ckpt['model']))

    # ------------------------------------------------------------------------------------------
    def console_log(self, tag: str, meta: Dict[str, Any], step: int):
        msg = '{}\t{:06d} it'.format(tag, step)
        for key, (value, log_type) in sorted(meta.items()):
            if log_type == LogType.SCALAR:
                msg += '\t{}: {:.6f}'.format(key, value)
        log(msg)

    def tensorboard_log(self, tag: str, meta: Dict[str, Any], step: int):
        for key, (value, log_type) in meta.items():
            if log_type != LogType.SCALAR and type(value) == torch.Tensor:
                value = to_numpy(value)
            name = '{}/{}'.format(tag, key)
            if log_type == LogType.IMAGE:
                buf = _to_buf('image', value)
                if buf is not None:
                    self.writer.add_image(name, buf, global_step=step)
            elif log_type == LogType.AUDIO:
                self.writer.add_audio(name, value, global_step=step, sample_rate=self.sr)
            elif log_type == LogType.SCALAR:
                self.writer.add_scalar(name, value, global_step=step)
            elif log_type == LogType.PLOT:
                buf = _to_buf('plot', value)
                if buf is not None:
                    self.writer.add_image(name, buf, global_step=step)
            elif log_type == LogType.TEXT:
                self.writer.add_text(name, value, global_step=step)

    @staticmethod
    def repeat(iterable):
        while True:
            for x in iterable:
                yield x
