"""Data-parallel plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL on ROCm, over
xGMI inside a node; "gloo" for the CPU tests).  The reference has no distributed code at all
(SURVEY 2.2); the only collective this adds to the hot path is the gradient all-reduce.

FlatGradReducer
  * parameters are packed, in REVERSE registration order (the order backward produces gradients),
    into a few large flat fp32 buckets; every ``p.grad`` is a view into its bucket, so there is no
    gather/scatter copy around the collective;
  * a post-accumulate-grad hook per parameter counts arrivals; the moment a bucket is complete its
    all-reduce (SUM) is launched asynchronously - RCCL runs it on its own stream while backward
    keeps producing the next bucket;
  * ``finish()`` (called before clip_grad / optimizer.step) waits for the outstanding collectives
    and scales by 1/world.
Bucket size: gradients here are small (hifi_gan_v1 55.7 MB fp32, the config-2 separator 22 MB, v2 3.7 MB):
xGMI rings are per-link bound (~153 GB/s/link), so latency, not bandwidth, dominates -> few large
buckets, but more than one: by default about six per model (`auto_bucket_bytes`: whole MiB, 4..32 MiB),
so that all but the last sixth of the reduction can run under the backward.
"""
import collections
import os
from pytorch_sound_amd import _switches as _sw
from typing import List

import torch
import torch.distributed as dist


def is_dist() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def is_main() -> bool:
    return rank() == 0


def init_from_env(backend: str = None) -> bool:
    """Initialise the default process group from torchrun's environment (RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT / LOCAL_RANK).  Returns True when running distributed."""
    if 'RANK' not in os.environ or int(os.environ.get('WORLD_SIZE', '1')) <= 1:
        return False
    if dist.is_initialized():
        return True
    use_gpu = torch.cuda.is_available()
    # PSND_DIST_SHARE_GPU=1 (testing only): every rank on device 0 over gloo - exercises the multi-process step loop
    # (graph replay + flat-bucket all-reduce) on a single-GPU box, where RCCL refuses two ranks on one device
    share = os.environ.get('PSND_DIST_SHARE_GPU') == '1'
    if use_gpu:
        torch.cuda.set_device(0 if share else int(os.environ.get('LOCAL_RANK', '0')))
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if share:
        backend = 'gloo'
    dist.init_process_group(backend or ('nccl' if use_gpu else 'gloo'))
    return True


def broadcast_module(module: torch.nn.Module, src: int = 0):
    """rank-`src` parameters and buffers to every rank (start of training / after load)."""
    if not is_dist():
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src)


def _collective_device(device=None):
    """device a small collective tensor must live on: the process group's backend decides (RCCL has no CPU path,
    gloo handles both); `device` is honoured when the backend can take it"""
    backend = dist.get_backend()
    if backend == 'nccl':
        return torch.device('cuda', torch.cuda.current_device())
    return device or torch.device('cpu')


def broadcast_int(value: int, src: int = 0) -> int:
    """an exact integer (e.g. the seed) from rank `src` to every rank, as int64 on the backend's device"""
    if not is_dist():
        return int(value)
    t = torch.tensor([int(value)], dtype=torch.int64, device=_collective_device())
    dist.broadcast(t, src)
    return int(t.item())


def all_reduce_scalar(value, op: str = 'sum', device=None) -> float:
    if not is_dist():
        return float(value)
    device = _collective_device(device)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op={'sum': dist.ReduceOp.SUM, 'max': dist.ReduceOp.MAX, 'min': dist.ReduceOp.MIN}[op])
    return float(t.item())


class FlatGradReducer:
    """see the module docstring.  Graph mode (Trainer.graph_steps): the captured backward fills the buckets ITSELF and
    marks, per bucket, the point where it is complete -

      'events'   an external event-record node per bucket (psnd_event_record_external).  After the replay has been enqueued the
                 host enqueues, per bucket and in fixed order, `side.wait(event_i); all_reduce(bucket_i)` on a side stream:
                 RCCL runs bucket i while the replayed backward is still producing bucket i+1 (any backend; opt-in with
                 PSND_DDP_GRAPH=events where the HIP runtime captures external event records - never run multi-rank so far)
      'capture'  the all-reduce itself is captured into the graph on RCCL's stream (fork after the bucket, join at the end):
                 no host involvement at all (backend nccl only, where it is the default)
      'deferred' round 1's behaviour: every collective after the replay (the default under gloo; the fallback ALL ranks take
                 together when a capture with one of the other modes fails on any rank - Trainer._capture)
    """

    def __init__(self, module: torch.nn.Module, bucket_bytes: int = None, force: bool = False, comm_dtype=None):
        """comm_dtype: None = the fp32 buckets themselves cross the wire; torch.bfloat16 (or PSND_DDP_COMM=bf16) = a bf16 image of each
        bucket does (pack -> all-reduce -> unpack on the release stream): half the bytes per xGMI link, the buckets, the kernels that
        write them and the optimizer that reads them stay fp32.  What the ranks add up is then the bf16 ROUNDING of each rank's
        gradient (8 mantissa bits; RCCL / gloo accumulate the ring steps in bf16 as well) - the usual gradient-compression trade,
        identical on every rank (same collective, same result), not bit-compatible with the fp32 wire."""
        if comm_dtype is None and os.environ.get('PSND_DDP_COMM', '').lower() in ('bf16', 'bfloat16'):
            comm_dtype = torch.bfloat16
        if comm_dtype not in (None, torch.float32, torch.bfloat16):
            raise TypeError('FlatGradReducer: comm_dtype %s (None / torch.float32 / torch.bfloat16)' % (comm_dtype,))
        self.comm_dtype = torch.bfloat16 if comm_dtype is torch.bfloat16 else None
        self.world = world_size()
        self.active = self.world > 1 or force   # force: single-rank process group (tests the collective plumbing on one GPU)
        self.deferred = False        # True: no per-bucket all-reduce from the backward hooks, finish() reduces everything
        self._capturing = None       # graph capture in progress: its mode
        self._stall_logged = False
        self._graph_mode = None      # mode of the graph that was replayed last (None: eager step)
        self._events = None
        self._side = None
        self.release_marks = []      # (tests) per bucket: timing event recorded on the side stream when its wait was released
        self.params: List[torch.nn.Parameter] = [p for p in module.parameters() if p.requires_grad]
        self.buckets = []            # dicts: flat, params, pending, work
        self._next = 0               # first bucket whose all-reduce has not been issued in this step
        self.launch_log = collections.deque(maxlen=4096)   # bucket indices in the order their collectives were issued (tests)
        self._bucket_of = {}
        self._slot = {}              # parameter -> (bucket, offset in floats); a slot is 64-float aligned
        self._deliver_seq = 0        # deliver() calls so far; _in_deliver: the running one's number (0: none) - handover_log (tests, profiles)
        self._in_deliver = 0
        self.handover_log = collections.deque(maxlen=4096)   # (bucket, number of the deliver() call that released it, or 0 = a hook / finish())
        self._sunk = set()           # parameters whose gradient was written straight into its slot this step (sink protocol)
        self._delivered = set()      # ... and that a node has handed over (deliver): a later hook call for them is not an arrival
        # Streams.  A backward pass may produce gradients on several streams (inside a captured step: graph branches - the resblocks of a
        # HiFi-GAN stage, the parameter-side branch of a transposed conv / a 1x1 projection): every arrival leaves an event on the stream
        # that PRODUCES the gradient, and the stream a bucket is released from waits for the events of the bucket's other streams first
        # (round 5: graph branches and the reducer in one step).
        self._release = None         # the stream buckets are released from (_release_on)
        self._release_used = False
        self._bucket_events = {}     # bucket index -> {stream id: (stream, event)}
        self._producer = {}          # parameter -> stream its gradient is being written on, when that is not the node's stream (note_producer)
        self._event_pool = []
        self.sink_enabled = True     # cl.GRAD_SINK protocol: conv-chain nodes write into the buckets and hand parameters over mid-backward
        self._handles = []
        order = list(reversed(self.params))
        if bucket_bytes is None:
            lab_mib = _sw.lab('PSND_DDP_BUCKET_MIB')       # A/B: bucket size in MiB
            bucket_bytes = (int(float(lab_mib) * (1 << 20)) if lab_mib else self.auto_bucket_bytes(sum(p.numel() * 4 for p in order)))
        self.bucket_bytes = bucket_bytes
        # the parameters of one leaf module (a conv's weight_v / weight_g / bias) stay in ONE bucket: their gradients appear together, and
        # a bucket that holds the bias of the NEXT conv as well would wait for that conv's block (measured at config 2: bucket 3 of 6
        # waited for the first block because two 1 KB tensors of it had slipped in behind five convs of the second)
        owner = {}
        for m in module.modules():
            for q in m.parameters(recurse=False):
                owner.setdefault(q, m)
        units = []
        for p in order:
            if units and owner.get(p) is not None and owner.get(units[-1][-1]) is owner.get(p) and p.device == units[-1][-1].device:
                units[-1].append(p)
            else:
                units.append([p])
        groups, cur, cur_bytes = [], [], 0
        for u in units:
            nbytes = sum(q.numel() for q in u) * 4
            if cur and (cur_bytes + nbytes > bucket_bytes or u[0].device != cur[0].device):
                groups.append(cur)
                cur, cur_bytes = [], 0
            cur.extend(u)
            cur_bytes += nbytes
        if cur:
            groups.append(cur)
        for gi, g in enumerate(groups):
            # every view starts on a 256-byte boundary: the fused optimizer kernels take their vectorised path only
            # for 16-byte aligned gradients (measured on MI355X: 92 us vs 46 us per multi_tensor_apply launch)
            offs, off = [], 0
            for p in g:
                if p.dtype != torch.float32:
                    raise TypeError('FlatGradReducer keeps fp32 master gradients; got %s' % p.dtype)
                offs.append(off)
                off += (p.numel() + 63) // 64 * 64
            flag_off = None
            if gi == len(groups) - 1:        # one spare slot behind the last bucket: the step's NaN flag rides along with the
                flag_off, off = off, off + 64    # gradients (SUM over ranks > 0 <=> some rank saw a NaN) - no collective of its own
            flat = torch.zeros(off, dtype=torch.float32, device=g[0].device)
            if flag_off is not None:
                self._flag = flat[flag_off:flag_off + 1]
            for p, o in zip(g, offs):
                p.grad = flat[o:o + p.numel()].view_as(p)
            b = {'flat': flat, 'params': g, 'offs': offs, 'pending': len(g), 'work': None}
            self.buckets.append(b)
            for p, o in zip(g, offs):
                self._bucket_of[p] = b
                self._slot[p] = (b, o)
        if self.active:
            for p in self.params:
                self._handles.append(p.register_post_accumulate_grad_hook(self._on_grad))
            from . import cl
            cl.GRAD_SINK = self      # conv-chain nodes hand their weight gradients over from inside their backward (cl.py)
            cl.NAN_FLAG_DEST[0] = self._flag     # a fused loss node writes the step's NaN flag straight into the spare slot (set_flag: no copy)

    class _Done:
        """the collective has been enqueued in stream order already (bf16 wire): nothing to wait for"""
        def wait(self):
            return True

    def _reduce(self, b):
        """SUM over the ranks of bucket b, enqueued on the CURRENT stream (the release stream); returns a work object"""
        flat = b['flat']
        if self.comm_dtype is None:
            return dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)
        c = b.get('comm')
        if c is None:
            c = b['comm'] = torch.empty(flat.numel(), dtype=torch.bfloat16, device=flat.device)
        if flat.is_cuda:
            from ._lib import lib, check, ptr, stream_ptr
            with torch.cuda.device(flat.device):
                st = stream_ptr(flat.device)
                check(lib().psnd_grad_pack_bf16(ptr(flat), ptr(c), flat.numel(), 1.0, st), 'psnd_grad_pack_bf16')
                dist.all_reduce(c, op=dist.ReduceOp.SUM)         # stream-ordered: this stream waits for RCCL's, the host does not
                check(lib().psnd_grad_unpack_bf16(ptr(c), ptr(flat), flat.numel(), 1.0, st), 'psnd_grad_unpack_bf16')
        else:                                                    # the gloo tests on CPU tensors
            c.copy_(flat)
            dist.all_reduce(c, op=dist.ReduceOp.SUM)
            flat.copy_(c)
        return FlatGradReducer._Done()

    @staticmethod
    def auto_bucket_bytes(total_bytes: int, target_buckets: int = 6) -> int:
        """bucket size when none is given: about `target_buckets` buckets per model, whole MiB, between 4 and 32 MiB.  One 32 MiB bucket
        holds ALL 22 MB of the config-2 separator's gradients - its all-reduce could not start before the last gradient of the
        backward exists, i.e. no overlap at all; six buckets leave one sixth of the reduction exposed.  Below 4 MiB a bucket costs more
        than its overlap returns: besides RCCL's ~20-30 us per collective over xGMI, every bucket is a release point of the step graph
        (copy into the flat buffer, fork to the release stream, captured collective, join) - measured with a one-rank RCCL group on the
        config-4 block (4.2 MB of gradients, round 6): 5 buckets of 1 MiB 2.35 ms, 3 of 2 MiB 2.39, ONE bucket 2.18 (2.15 without a
        reducer); the 22 MB separator: 0.717 ms at 1 MiB, 0.705 at 4 MiB."""
        mib = 1 << 20
        per = -(-total_bytes // target_buckets)
        return int(min(32 * mib, max(4 * mib, -(-per // mib) * mib)))

    # ---- cl.GRAD_SINK protocol: gradients produced INSIDE a node's backward ------------------------------------------------------
    def sink_active(self) -> bool:
        """true while a backward runs whose gradients belong in the buckets: an eager data-parallel step, or a capture in mode
        'capture' / 'events' (the deferred mode keeps the graph's own gradient tensors: load_grads copies them)"""
        if not (self.active and self.sink_enabled):
            return False
        if self._capturing is not None:
            return self._capturing in ('capture', 'events')
        return not self.deferred

    def dest(self, p, numel: int):
        """the slot of parameter `p` as a flat fp32 tensor of `numel` elements (>= p.numel(): a padded bias), to be OVERWRITTEN with its
        gradient - or None: not a parameter of this reducer, no backward of ours running, or already written this step (a shared
        parameter: the node then returns the gradient and autograd accumulates it as usual)"""
        if p is None or not self.sink_active():
            return None
        slot = self._slot.get(p)
        if slot is None or p in self._sunk:
            return None
        from . import cl
        if not cl.single_use(p):         # a second node of this backward produces a gradient for p too: autograd must accumulate both BEFORE the
            return None                  # parameter counts as arrived (the post-accumulate hook does that) - no early hand-over
        b, off = slot
        if numel > (p.numel() + 63) // 64 * 64:
            return None
        self._sunk.add(p)
        return b['flat'][off:off + numel]

    def note_producer(self, params, stream):
        """the gradients of these parameters are being written on `stream` (a side stream / graph branch), not on the stream the node
        that returns them runs on: the bucket's release waits for THAT stream"""
        for p in params:
            if p is not None and p in self._slot:
                self._producer[p] = stream

    def _arrival(self, p):
        """an event behind the gradient of p on the stream that produces it"""
        b = self._bucket_of[p]
        if not b['flat'].is_cuda:                     # host tensors (gloo): no streams
            return
        st = self._producer.pop(p, None)
        if st is None:
            st = torch.cuda.current_stream(b['flat'].device)
        evs = self._bucket_events.setdefault(id(b), {})
        ent = evs.get(st.cuda_stream)
        if ent is None:
            ev = self._event_pool.pop() if (self._event_pool and self._capturing is None) else torch.cuda.Event()
            ent = evs[st.cuda_stream] = (st, ev)
        ent[1].record(st)

    def _join_streams(self, b):
        """the current stream waits for the gradients of bucket b that other streams produced"""
        cur = torch.cuda.current_stream(b['flat'].device)
        for key, (st, ev) in self._bucket_events.get(id(b), {}).items():
            if key != cur.cuda_stream:
                cur.wait_event(ev)

    def _reset_streams(self):
        if self._capturing is None:
            for evs in self._bucket_events.values():
                self._event_pool.extend(ev for _, ev in evs.values())
        self._bucket_events = {}
        self._producer.clear()

    def deliver(self, params):
        """the gradients of these parameters have been written into their slots (enqueued on the current stream): count them as
        arrived - a bucket that is complete leaves now, while the backward goes on"""
        self._deliver_seq += 1
        self._in_deliver = self._deliver_seq
        try:
            for p in params:
                b, off = self._slot[p]
                if p.grad is None or p.grad.data_ptr() != b['flat'].data_ptr() + off * 4:
                    p.grad = b['flat'][off:off + p.numel()].view_as(p)
                self._on_grad(p)
        finally:
            self._in_deliver = 0

    # gradients must stay views of the flat buffers: zero in place instead of dropping them
    def zero_grad(self):
        self._next = 0
        self._sunk.clear()
        self._delivered.clear()
        self._reset_streams()
        for b in self.buckets:
            b['flat'].zero_()
            b['pending'] = len(b['params'])
            b['work'] = None
            self._repoint(b)

    def _repoint(self, b):
        for p, off in zip(b['params'], b['offs']):
            if p.grad is None or p.grad.data_ptr() != b['flat'].data_ptr() + off * 4:
                p.grad = b['flat'][off:off + p.numel()].view_as(p)

    def load_grads(self, grads):
        """graph mode: the replayed backward left its gradients in the graph's own static tensors (`grads[p]`, None for
        an unused parameter); copy them into the buckets (one multi-tensor copy per bucket) and make the buckets'
        views the parameters' .grad again, ready for finish()."""
        self._next = 0
        self._reset_streams()                         # (events of an earlier eager step say nothing about the copies below)
        for b in self.buckets:
            b['pending'] = len(b['params'])
            b['work'] = None
            views = [b['flat'][off:off + p.numel()].view_as(p) for p, off in zip(b['params'], b['offs'])]
            src = [grads.get(p) for p in b['params']]
            have = [(v, g) for v, g in zip(views, src) if g is not None]
            # (only the views of parameters WITHOUT a gradient are cleared - as _emit_bucket does: the last bucket's spare slot carries the
            #  step's NaN flag, written by the replayed loss launch before this runs (cl.NAN_FLAG_DEST); zeroing the whole buffer wiped it)
            missing = [v for v, g in zip(views, src) if g is None]
            if missing:
                torch._foreach_zero_(missing)
            if have:
                torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
            for p, v in zip(b['params'], views):
                p.grad = v

    def set_flag(self, flag):
        """place this rank's NaN flag (0 / 1) in the spare slot BEFORE the last bucket is reduced"""
        if flag.data_ptr() == self._flag.data_ptr() and flag.dtype == self._flag.dtype:
            return                                   # written in place by the launch that formed the loss (cl.NAN_FLAG_DEST)
        self._flag.copy_(flag.detach().reshape(1).to(self._flag.dtype))

    @property
    def flag(self):
        """after finish(): > 0 iff any rank raised its flag"""
        return self._flag

    # ---- graph mode --------------------------------------------------------------------------------------------------
    def graph_mode(self) -> str:
        """the mode a captured step uses - by default the ones that have run end to end: the captured RCCL all-reduce under backend
        nccl, `deferred` under any other backend.  `events` (external event-record nodes; any backend) is opt-in through
        PSND_DDP_GRAPH=events and only where the HIP runtime captures such records: the runtime bundled with this torch build refuses
        them, so that path has never been exercised by a multi-rank run.  PSND_DDP_GRAPH=capture | deferred force those; a mode the
        runtime / backend cannot do is replaced by the default."""
        from ._lib import lib
        nccl = dist.get_backend() == 'nccl'
        default = 'capture' if nccl else 'deferred'
        mode = _sw.lab('PSND_DDP_GRAPH', default)
        if mode not in ('events', 'capture', 'deferred'):
            raise ValueError('PSND_DDP_GRAPH=%s (events | capture | deferred)' % mode)
        if mode == 'events' and not lib().psnd_event_external_supported():
            mode = default
        if mode == 'capture' and not nccl:           # only RCCL can be captured
            mode = 'deferred'
        return mode

    def capture_begin(self, flag, mode: str):
        """inside the stream capture, after forward: arm the hooks for the captured backward.  The NaN flag goes into its slot
        behind the last bucket now (a captured copy), ahead of every bucket's release point."""
        self._capturing = mode
        self._next = 0
        self._sunk.clear()
        self._delivered.clear()
        self._reset_streams()
        self._cap_works = []
        self._arrived = 0
        self.emit_log = []           # (bucket, gradients that had arrived when its release point was captured) - tests
        self.zeroed_log = []         # (bucket, its pending count, shape) of parameters zero-filled at a release point: no gradient had arrived
        for b in self.buckets:
            b['pending'] = len(b['params'])
            b['work'] = None
        self.set_flag(flag)
        if mode == 'events' and self._events is None:
            from ._lib import lib
            self._events = []
            for _ in self.buckets:
                ev = lib().psnd_event_create()
                if not ev:
                    raise RuntimeError('psnd_event_create failed')
                self._events.append(ev)

    def _release_on(self, b):
        """the stream a bucket is released from: the reducer's RELEASE stream, behind the events of every stream that produced one of the
        bucket's gradients.  Neither the backward's main stream nor a branch waits for the others at a release, and the collective is
        never issued from a graph branch (capturing RCCL's all-reduce from a resblock branch of the HiFi-GAN step ended the process in
        hipStreamEndCapture).  PSND_DDP_RELEASE=current: round 4's behaviour, released from whatever stream the last hook ran on."""
        dev = b['flat'].device
        if not b['flat'].is_cuda:
            import contextlib
            return contextlib.nullcontext()
        if _sw.lab('PSND_DDP_RELEASE', 'stream') != 'stream':
            self._join_streams(b)
            return torch.cuda.current_stream(dev)
        if self._release is None:
            self._release = torch.cuda.Stream(device=dev)
        rel = self._release
        # the releasing stream itself (where the last gradient arrived, or - nothing having arrived through a hook: load_grads / capture_end /
        # finish on an untouched bucket - where the gradients were copied in), then every other stream that produced one
        rel.wait_stream(torch.cuda.current_stream(dev))
        for st, ev in self._bucket_events.get(id(b), {}).values():
            rel.wait_event(ev)
        self._release_used = True
        return rel

    def _join_release(self, dev):
        """the current stream waits for what was enqueued on the release stream (the end of a backward pass / a capture)"""
        if self._release is not None and self._release_used and dev.type == 'cuda':
            torch.cuda.current_stream(dev).wait_stream(self._release)
            self._release_used = False

    def _emit_bucket(self, i):
        """captured: gradients of bucket i (static tensors of the graph's pool) -> its flat buffer, then the release point"""
        b = self.buckets[i]
        rel = self._release_on(b)
        with (torch.cuda.stream(rel) if b['flat'].is_cuda else rel):
            views = [b['flat'][off:off + p.numel()].view_as(p) for p, off in zip(b['params'], b['offs'])]
            # (a gradient that a node wrote straight into its slot - cl.GRAD_SINK - is there already)
            have = [(v, p.grad) for v, p in zip(views, b['params']) if p.grad is not None and p.grad.data_ptr() != v.data_ptr()]
            for v, p in zip(views, b['params']):
                if p.grad is None:
                    v.zero_()            # a parameter the captured backward never reached
                    self.zeroed_log.append((i, b['pending'], tuple(p.shape)))
            if have:
                torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
                if b['flat'].is_cuda:
                    for _, g in have:
                        g.record_stream(rel)
            mode = self._capturing
            if mode == 'capture':
                self._cap_works.append(self._reduce(b))
            elif mode == 'events':
                from ._lib import lib, check
                check(lib().psnd_event_record_external(self._events[i], rel.cuda_stream), 'psnd_event_record_external')   # (events mode: HIP tensors only)
        self.launch_log.append(i)
        self.handover_log.append((i, self._in_deliver))
        self.emit_log.append((i, self._arrived))

    def capture_end(self):
        """still inside the capture, after backward: buckets whose hooks did not all fire, then (mode capture) the join"""
        while self._next < len(self.buckets):
            self._emit_bucket(self._next)
            self._next += 1
        for w in self._cap_works:
            w.wait()
        self._cap_works = []
        self._join_release(self.buckets[0]['flat'].device)
        mode, self._capturing = self._capturing, None
        self._sunk.clear()
        self._delivered.clear()
        self._next = 0
        for b in self.buckets:
            b['pending'] = len(b['params'])
        return mode

    def after_replay(self, mode: str, time_marks: bool = False):
        """the replay of a captured step has just been enqueued on the current stream: make the buckets' views the
        parameters' .grad again and (mode events) enqueue every bucket's all-reduce behind its release event"""
        self._graph_mode = mode
        self._next = 0
        for b in self.buckets:
            b['pending'] = len(b['params'])
            b['work'] = None
            self._repoint(b)
        if mode != 'events' or not self.active:
            return
        from ._lib import lib, check
        dev = self.buckets[0]['flat'].device
        if self._side is None:
            self._side = torch.cuda.Stream(device=dev)
        self.release_marks = []
        with torch.cuda.stream(self._side):
            for i, b in enumerate(self.buckets):
                check(lib().psnd_stream_wait_event(self._side.cuda_stream, self._events[i]), 'psnd_stream_wait_event')
                if time_marks:
                    m = torch.cuda.Event(enable_timing=True)
                    m.record(self._side)
                    self.release_marks.append(m)
                b['work'] = self._reduce(b)
                self.launch_log.append(i)
        self._next = len(self.buckets)

    def _on_grad(self, p):
        # A parameter that a node handed over itself (deliver) arrives ONCE: autograd still runs the parameter's AccumulateGrad node with an
        # undefined gradient when the node returns None for it, and the engine calls the post-accumulate hooks of that node all the same -
        # counted a second time, a bucket's `pending` reached zero while other parameters' gradients were still to come: the bucket was
        # released (copied / zero-filled / all-reduced) early, and what arrived afterwards landed in the flat buffer behind the
        # collective - invisible on one rank and one stream, wrong gradients on several ranks, zeros next to graph branches (round 5,
        # tools/r05/dbg_branch_reducer.py: every resblock parameter of the HiFi-GAN generator arrived twice).
        if self._in_deliver == 0 and p in self._delivered:
            return
        if self._in_deliver != 0:
            self._delivered.add(p)
        if self._capturing is not None:
            b = self._bucket_of[p]
            self._arrival(p)
            b['pending'] -= 1
            self._arrived += 1
            if self._capturing != 'deferred':
                while self._next < len(self.buckets) and self.buckets[self._next]['pending'] == 0:
                    self._emit_bucket(self._next)
                    self._next += 1
            return
        if self.deferred:            # backward is being captured / replayed as a hipGraph: no collective from inside it
            return
        b = self._bucket_of[p]
        self._arrival(p)
        b['pending'] -= 1
        # collectives are issued in FIXED bucket order on every rank: a completed bucket launches from the hook only when all
        # earlier buckets have launched (a parameter without a gradient on one rank only - a data-dependent branch - would
        # otherwise reorder that rank's all-reduces against the others: deadlock or mixed-up gradients); what is left goes
        # out, in order, in finish()
        while self._next < len(self.buckets) and self.buckets[self._next]['pending'] == 0:
            nb = self.buckets[self._next]
            with (torch.cuda.stream(self._release_on(nb)) if nb['flat'].is_cuda else self._release_on(nb)):
                nb['work'] = self._reduce(nb)
            self.launch_log.append(self._next)
            self.handover_log.append((self._next, self._in_deliver))
            self._next += 1

    def finish(self, average: bool = True):
        """wait for every bucket, average (average=False: the buckets keep the SUM - the caller divides, e.g. through the
        optimizer kernel's grad_scale).  Buckets whose hooks did not all fire (unused parameters) are reduced here so that
        ranks never diverge."""
        if not self.active:
            return
        if self._graph_mode == 'capture':            # reduced inside the replayed graph
            self._graph_mode = None
            if average:
                for b in self.buckets:
                    b['flat'].mul_(1.0 / self.world)
            return
        self._graph_mode = None
        if self._next < len(self.buckets) and not self.deferred and not self._stall_logged:
            # eager hooks launch in bucket order: a parameter of bucket `_next` that got no gradient holds back every later bucket,
            # whose all-reduces are then issued here, serially, after the backward - say so once instead of losing the overlap silently
            b = self.buckets[self._next]
            if b['pending'] > 0 and any(q['pending'] == 0 for q in self.buckets[self._next + 1:]):
                self._stall_logged = True
                import logging
                logging.getLogger('pytorch_sound_amd').warning(
                    'FlatGradReducer: %d parameter(s) of bucket %d of %d received no gradient this step; the all-reduces of the '
                    'buckets behind it were issued after the backward (no overlap). Unused parameters are best excluded '
                    '(requires_grad = False).', b['pending'], self._next, len(self.buckets))
        for i in range(self._next, len(self.buckets)):
            b = self.buckets[i]
            with (torch.cuda.stream(self._release_on(b)) if b['flat'].is_cuda else self._release_on(b)):
                b['work'] = self._reduce(b)
            self.launch_log.append(i)
        self._join_release(self.buckets[0]['flat'].device)
        for b in self.buckets:
            b['work'].wait()
            if average:
                b['flat'].mul_(1.0 / self.world)
            b['work'] = None
            b['pending'] = len(b['params'])
        self._next = 0
        self._sunk.clear()
        self._delivered.clear()

    def remove(self):
        from . import cl
        if cl.GRAD_SINK is self:
            cl.GRAD_SINK = None
        if cl.NAN_FLAG_DEST[0] is not None and cl.NAN_FLAG_DEST[0].data_ptr() == self._flag.data_ptr():
            cl.NAN_FLAG_DEST[0] = None
        for h in self._handles:
            h.remove()
        self._handles = []
        if self._events:
            from ._lib import lib
            for ev in self._events:
                lib().psnd_event_destroy(ev)
            self._events = None
