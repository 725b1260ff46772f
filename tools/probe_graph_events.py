#!/usr/bin/env python
"""Probes for the graph-mode DDP design (run on the 1-GPU box):
 1. external events: an event recorded INSIDE a captured graph (hipEventRecordExternal) and waited on by a side stream after
    the replay was enqueued - does the side stream start when the graph reaches the record node (overlap), when the whole
    graph is done (correct but serial), or at once (broken)?
 2. an RCCL all-reduce (backend nccl, world size 1) captured inside a torch.cuda.graph and replayed.
"""
import os
import sys
import time

import torch

dev = torch.device('cuda:0')
torch.cuda.set_device(0)


def probe_external_event():
    a = torch.zeros(1 << 26, device=dev)
    b = torch.ones(1 << 26, device=dev)
    m1 = torch.zeros(1, device=dev)
    m2 = torch.zeros(1, device=dev)
    side = torch.cuda.Stream()
    try:
        ev = torch.cuda.Event(external=True)
    except TypeError as e:
        print('external events: not constructible:', e)
        return
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            a.add_(1.0)
            b.mul_(1.0001)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(10):
            a.add_(1.0)            # segment A
        m1.fill_(1.0)
        ev.record()
        for _ in range(40):
            b.mul_(1.0001)         # segment B (long)
        m2.fill_(2.0)
    torch.cuda.synchronize()
    for trial in range(3):
        m1.zero_()
        m2.zero_()
        torch.cuda.synchronize()
        t0 = torch.cuda.Event(enable_timing=True)
        t_side = torch.cuda.Event(enable_timing=True)
        t_end = torch.cuda.Event(enable_timing=True)
        t0.record()
        g.replay()
        t_end.record()
        with torch.cuda.stream(side):
            side.wait_event(ev)
            snap = torch.cat([m1, m2]).clone()
            t_side.record(side)
        torch.cuda.synchronize()
        print('external event trial %d: markers seen by the side stream (A done, B done) = %s ; side stream ran at %.3f ms, graph '
              'ended at %.3f ms' % (trial, snap.tolist(), t0.elapsed_time(t_side), t0.elapsed_time(t_end)))
    # verdict
    v = snap.tolist()
    if v == [1.0, 0.0]:
        print('EXTERNAL_EVENT: OVERLAP (side stream released at the record node)')
    elif v == [1.0, 2.0]:
        print('EXTERNAL_EVENT: SERIAL (side stream released after the whole graph)')
    else:
        print('EXTERNAL_EVENT: BROKEN (side stream did not wait)')


def probe_rccl_capture():
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    try:
        dist.init_process_group('nccl', rank=0, world_size=1)
    except Exception as e:           # noqa: BLE001
        print('RCCL_CAPTURE: init failed:', repr(e))
        return
    t = torch.full((1 << 20,), 3.0, device=dev)
    dist.all_reduce(t)               # communicator set up eagerly first
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g):
            t.mul_(2.0)
            w = dist.all_reduce(t, async_op=True)
            w.wait()
            t.add_(1.0)
        torch.cuda.synchronize()
        t.fill_(3.0)
        g.replay()
        g.replay()
        torch.cuda.synchronize()
        print('RCCL_CAPTURE: OK value', float(t[0]), '(expected 15.0)')
    except Exception as e:           # noqa: BLE001
        print('RCCL_CAPTURE: capture failed:', repr(e)[:500])
    dist.destroy_process_group()


if __name__ == '__main__':
    t = time.time()
    try:
        probe_external_event()
    except Exception as e:           # noqa: BLE001
        print('EXTERNAL_EVENT: torch refuses:', repr(e))
    probe_rccl_capture()
    print('probe done in %.1f s' % (time.time() - t))
