"""One-GPU sanity check of the RCCL path: a world-size-1 NCCL group, halo tensors that alias the
engine's device memory, send-to-self through batch_isend_irecv. Verifies RCCL accepts the aliased
memory and measures the host + device cost of one exchange."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
from babelbrain_amd import harness as H, slab, _engine

os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
dt_fn = lambda ml, f, h, a: _engine.stable_dt(ml, f, True, h, a)
a, k, info = H.make_problem('C3', N=(512, 512, 64), steps=20, stable_dt_fn=dt_fn)
s, sinfo = slab.create_hip_slab(a, k, 0, 1, 0)
s.eng.run(10)
torch.cuda.synchronize()
for g in (0, 1):
    for f in range(3):
        src = s.halo(g, f, 1, True); dst = s.halo(g, f, 0, False)
        ref = src.clone()
        ops = [dist.P2POp(dist.isend, src, 0), dist.P2POp(dist.irecv, dst, 0)]
        for r in dist.batch_isend_irecv(ops):
            r.wait()
        torch.cuda.synchronize()
        assert torch.equal(dst, ref), (g, f)
print('self send/recv through RCCL on aliased engine memory: OK, |Vz halo| max', float(s.halo(0, 2, 1, True).abs().max()))
# cost of one full exchange (3 fields both directions, as an interior rank does per half-step)
def exchange(fields):
    ops = []
    for f in fields:
        ops.append(dist.P2POp(dist.isend, s.halo(0, f, 1, True), 0)); ops.append(dist.P2POp(dist.irecv, s.halo(0, f, 0, False), 0))
        ops.append(dist.P2POp(dist.isend, s.halo(0, f, 0, True), 0)); ops.append(dist.P2POp(dist.irecv, s.halo(0, f, 1, False), 0))
    for r in dist.batch_isend_irecv(ops):
        r.wait()
for fields in ([0, 1, 2], [2]):
    for _ in range(5): exchange(fields)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): exchange(fields)
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    tt = time.perf_counter() - t0
    print('fields %s: host %.1f us per exchange, host+device %.1f us per exchange (%.1f MB each way per neighbour)' % (
        fields, th / 50 * 1e6, tt / 50 * 1e6, len(fields) * 2 * 512 * 512 * 4 / 1e6))
dist.destroy_process_group()
