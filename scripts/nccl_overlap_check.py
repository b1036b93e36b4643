"""One-GPU timing of the halo exchange machinery: world-size-1 NCCL group, send-to-self of the real halo tensors.
(The ghost contents are meaningless here; this only measures how much of the exchange the split half-steps hide.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
from babelbrain_amd import harness as H, slab, _engine, RayleighAndBHTE
from babelbrain_amd._engine import HALO_STRESS, HALO_VELOCITY

os.environ.setdefault('TORCH_NCCL_HIGH_PRIORITY', '1')
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29534')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
dt_fn = lambda ml, f, h, a: _engine.stable_dt(ml, f, True, h, a)
a, k, info = H.make_problem('C3', steps=400, stable_dt_fn=dt_fn, forward=RayleighAndBHTE.ForwardSimple, full_sensors=False)
s, sinfo = slab.create_hip_slab(a, k, 0, 1, 0)
fields = s.halo_fields()

def start(group):
    ops = []
    for f in fields[group]:
        for side in (0, 1):
            ops.append(dist.P2POp(dist.isend, s.halo(group, f, side, True), 0))
            ops.append(dist.P2POp(dist.irecv, s.halo(group, f, 1 - side, False), 0))
    return dist.batch_isend_irecv(ops)

def finish(reqs):
    for r in reqs: r.wait()

def step(mode):
    if mode == 'none':
        s.half_step_stress(); s.half_step_velocity()
    elif mode == 'blocking':
        finish(start(HALO_VELOCITY)); s.half_step_stress(); finish(start(HALO_STRESS)); s.half_step_velocity()
    elif mode == 'overlap':
        s.half_step_stress(1); w = start(HALO_STRESS); s.half_step_stress(2); finish(w)
        s.half_step_velocity(1); w = start(HALO_VELOCITY); s.half_step_velocity(2); finish(w)
    else:               # 'streams': SlabRunner's two-stream choreography
        for half in (HALO_STRESS, HALO_VELOCITY):
            runner.launch_parts(half, M, B)
            with torch.cuda.stream(B):
                finish(start(half))
            M.wait_stream(B)

runner = slab.SlabRunner(s, 0, 1)
M, B = s.streams()
modes = sys.argv[1:] or ['none', 'blocking', 'overlap', 'streams', 'none', 'blocking', 'overlap', 'streams']
for mode in modes:
    for _ in range(10): step(mode)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): step(mode)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 50
    print('%-9s %.3f ms/step' % (mode, t * 1e3), flush=True)
dist.destroy_process_group()
