"""Adam / AdamW for ``Trainer`` (pytorch_sound/trainer.py:215-216 calls ``optimizer.step()`` on whatever optimizer the
recipe built - ``torch.optim.Adam`` in the reference's recipes) as ONE HIP launch over every parameter tensor
(``psnd_adam_step``), with torch's numerics, state layout (``state[p] = {'step', 'exp_avg', 'exp_avg_sq'}`` - state
dicts load into ``torch.optim.Adam`` and back) and the AMP ``found_inf`` / ``grad_scale`` protocol, which is what lets the
Trainer skip a NaN step on the device.  fp32 CUDA parameters only; no CPU fallback.
"""
import struct

import numpy as np
import torch

from . import _lib
from ._lib import lib, check, ptr, stream_ptr


class Adam(torch.optim.Optimizer):
    _step_supports_amp_scaling = True           # honours optimizer.found_inf / optimizer.grad_scale on the device
    # Trainer.clip_grad (trainer.py:184-191) inside the step: `optimizer.fused_clip = (grad_clip, grad_norm)` for one step() makes
    # the kernel clamp every (scaled) gradient element and apply clip_grad_norm_'s factor - ONE extra pass over the gradients
    # (psnd_grad_sumsq) instead of a clamp per parameter + torch's multi-launch norm; `last_grad_norm` (device scalar) = the norm
    _supports_fused_clip = True
    # `optimizer.flag_log = (pinned int32 ring tensor (R, 2), slot, seq)` for one step(): the launch leaves found_inf and the sequence number
    # in host-visible memory (psnd_adam_step_logged) - the trainer's NaN log needs no device-to-host copy
    _supports_flag_log = True
    _decoupled = False                          # True: AdamW (weight decay applied to the parameter, not the gradient)

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False):
        if amsgrad:
            raise NotImplementedError('amsgrad is not implemented by the HIP optimizer kernel')
        if not 0.0 <= lr:
            raise ValueError('Invalid learning rate: {}'.format(lr))
        if not 0.0 <= eps:
            raise ValueError('Invalid epsilon value: {}'.format(eps))
        if not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError('Invalid beta parameters: {}'.format(betas))
        if not 0.0 <= weight_decay:
            raise ValueError('Invalid weight_decay value: {}'.format(weight_decay))
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=False,
                                      maximize=False, foreach=None, capturable=False, differentiable=False, fused=True))
        self._plans = {}

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._plans = {}                        # state tensors were replaced: cached device pointers are stale

    # ---- launch plan: parameter table + work list; rebuilt when the set of tensors changes, the table re-uploaded when a
    # gradient lives at a new address (eager steps usually get the same blocks back from the caching allocator, a replayed
    # hipGraph always does) ----
    def _build_plan(self, params):
        device = params[0].device
        chunk = int(lib().psnd_adam_chunk())
        which, off, static = [], [], []
        for i, p in enumerate(params):
            if not p.is_cuda or p.device != device or p.dtype != torch.float32 or not p.is_contiguous():
                raise _lib.PsndError('pytorch_sound_amd.optim: the parameters of a group must be contiguous fp32 tensors on one HIP '
                                     'device (got %s %s on %s)' % (p.dtype, tuple(p.shape), p.device))
            st = self.state[p]
            if len(st) == 0:
                st['step'] = torch.zeros((), dtype=torch.float32, device=device)
                st['exp_avg'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            else:
                if not (torch.is_tensor(st['step']) and st['step'].device == device and st['step'].dtype == torch.float32):
                    # a state dict saved by a non-fused torch optimizer keeps the step count on the host
                    st['step'] = torch.as_tensor(float(st['step']), dtype=torch.float32, device=device)
                for k in ('exp_avg', 'exp_avg_sq'):
                    st[k] = st[k].to(device=device, dtype=torch.float32).contiguous()
            static.append((p.data_ptr(), st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr(), st['step'].data_ptr(), p.numel()))
            starts = np.arange(0, p.numel(), chunk, dtype=np.int64)
            which.append(np.full(len(starts), i, dtype=np.int32))
            off.append(starts)
        return {'device': device, 'static': static, 'gptrs': None, 'table': None, 'grads': None,
                'chunk_tensor': torch.from_numpy(np.concatenate(which)).to(device),
                'chunk_off': torch.from_numpy(np.concatenate(off)).to(device),
                'corr': torch.empty(2 * len(params), dtype=torch.float32, device=device)}

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        found_inf = getattr(self, 'found_inf', None)
        grad_scale = getattr(self, 'grad_scale', None)
        clip_value, max_norm = getattr(self, 'fused_clip', None) or (0.0, 0.0)
        clip_value, max_norm = float(clip_value or 0.0), float(max_norm or 0.0)
        work = []
        for gi, group in enumerate(self.param_groups):
            params = [p for p in group['params'] if p.grad is not None]
            if not params:
                continue
            key = (gi, tuple(map(id, params)))
            plan = self._plans.get(key)
            if plan is None:
                if len(self._plans) > 17:
                    self._plans.clear()
                plan = self._plans[key] = self._build_plan(params)
            device = plan['device']
            gptrs = tuple(p.grad.data_ptr() for p in params)
            if gptrs != plan['gptrs']:
                grads = []
                for p in params:
                    g = p.grad
                    if g.is_sparse:
                        raise RuntimeError('sparse gradients are not supported')
                    if g.device != device:
                        raise _lib.PsndError('gradient on %s, parameter on %s' % (g.device, device))
                    grads.append(g if g.is_contiguous() and g.dtype == torch.float32 else g.contiguous().float())
                raw = b''.join(struct.pack('<5Qq', s[0], g.data_ptr(), s[1], s[2], s[3], s[4]) for s, g in zip(plan['static'], grads))
                assert len(raw) == len(params) * int(lib().psnd_adam_table_bytes())
                # stream-ordered upload from page-locked memory: a plain .to(device) of a pageable tensor is a SYNCHRONOUS copy - the host waits
                # for everything enqueued so far (the whole backward), the GPU then idles while the host catches up: every eager step whose
                # gradients came back at new addresses took ~4 ms instead of ~1.7 (round 6, tools/r06/stall.py)
                stage = plan.get('stage')
                if stage is None or stage[0][0].numel() != len(raw):
                    stage = plan['stage'] = [[torch.empty(len(raw), dtype=torch.uint8).pin_memory(), None] for _ in range(2)]
                    plan['stage_i'] = 0
                    plan['uploads'] = 0
                slot = stage[plan['stage_i']]
                plan['stage_i'] ^= 1
                plan['uploads'] += 1
                if slot[1] is not None:
                    slot[1].synchronize()                   # the copy that last read this staging buffer (two uploads ago) has run
                slot[0].numpy()[:] = np.frombuffer(raw, dtype=np.uint8)
                if plan['table'] is not None and plan['table'].numel() == len(raw) and not torch.cuda.is_current_stream_capturing():
                    plan['table'].copy_(slot[0], non_blocking=True)
                else:
                    plan['table'] = slot[0].to(device, non_blocking=True)
                slot[1] = torch.cuda.Event()
                slot[1].record()
                # converted copies (non-contiguous / non-fp32 gradients) are not stable addresses: look again next step
                stable = all(g is p.grad for g, p in zip(grads, params))
                plan['gptrs'] = gptrs if stable else None
                plan['grads'] = None if stable else grads
            work.append((group, plan, len(params)))
        coef = None
        if max_norm > 0.0 and work:
            device = work[0][1]['device']
            if any(w[1]['device'] != device for w in work):
                raise _lib.PsndError('fused gradient-norm clipping needs every parameter group on one HIP device')
            scratch = self._plans.get('clip')
            if scratch is None or scratch['sumsq'].device != device:
                scratch = self._plans['clip'] = {'sumsq': torch.zeros((), dtype=torch.float64, device=device),
                                                 'coef': torch.ones(2, dtype=torch.float32, device=device)}
            gs = grad_scale.to(device=device, dtype=torch.float32) if grad_scale is not None else None
            with torch.cuda.device(device):
                for wi, (group, plan, n) in enumerate(work):
                    if plan.get('partial') is None:
                        plan['partial'] = torch.empty(plan['chunk_off'].numel(), dtype=torch.float64, device=device)
                    check(lib().psnd_grad_sumsq(ptr(plan['table']), n, ptr(plan['chunk_tensor']), ptr(plan['chunk_off']),
                                                plan['chunk_off'].numel(), clip_value, ptr(gs), int(wi > 0), max_norm,
                                                ptr(plan['partial']), ptr(scratch['sumsq']), ptr(scratch['coef']), stream_ptr(device)),
                          'psnd_grad_sumsq')
            coef = scratch['coef']
            self.last_grad_norm = coef[1]
        flag_log = getattr(self, 'flag_log', None)
        self.flag_logged = flag_log is not None and len(work) > 0           # (no launch, no record: the caller falls back to a copy)
        for wi, (group, plan, n) in enumerate(work):
            device = plan['device']
            b1, b2 = group['betas']
            log_ptr, log_slot, log_seq = (ptr(flag_log[0]), int(flag_log[1]), int(flag_log[2])) if (flag_log is not None and wi == 0) else (None, 0, 0)
            fi = found_inf.to(device=device, dtype=torch.float32) if found_inf is not None else None
            gs = grad_scale.to(device=device, dtype=torch.float32) if grad_scale is not None else None
            with torch.cuda.device(device):
                check(lib().psnd_adam_step_logged(ptr(plan['table']), n, ptr(plan['chunk_tensor']), ptr(plan['chunk_off']),
                                                  plan['chunk_off'].numel(), float(group['lr']), float(b1), float(b2), float(group['eps']),
                                                  float(group['weight_decay']), int(self._decoupled), ptr(fi), ptr(gs), ptr(plan['corr']),
                                                  clip_value, ptr(coef), log_ptr, log_slot, log_seq, stream_ptr(device)), 'psnd_adam_step')
        return loss


class AdamW(Adam):
    _decoupled = True

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad)
