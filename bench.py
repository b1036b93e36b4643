#!/usr/bin/env python3
"""Benchmark of the Step-2 FDTD hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one full time step (stress half-step + velocity half-step + RMS accumulation
+ sensor capture when due) over the whole domain. Workload at N=1: BASELINE.json configs[2]
("512^3 CT-derived heterogeneous skull, CTX500 transducer, PML on"), the configuration the metric
is quoted on. At N>1 every rank owns a 512x512x512 Z-slab of a 512x512x(512 N) domain (weak
scaling), with neighbour halo exchange over RCCL.

Prints ONE JSON line on rank 0. `value` = voxel-steps of all ranks / max-over-ranks wall time of
the K timed steps (inputs resident in HBM), in Mvoxel-steps/s.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
BYTES_STRESS = 112.0           # algorithmic bytes per voxel, stress half-step   
BYTES_VELOCITY = 52.0          # velocity half-step
BYTES_RMS = 8.0                # Pressure RMS accumulate
BYTES_STEP = BYTES_STRESS + BYTES_VELOCITY + BYTES_RMS   # 172


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=300)
    ap.add_argument('--warmup', type=int, default=50)
    ap.add_argument('--config', default='C3')
    ap.add_argument('--size', type=int, nargs=3, default=None, help='override per-GPU grid N1 N2 N3')
    ap.add_argument('--variant', type=int, default=0, help='kernel variant (0 default, 1 simple, 2 LDS-tiled)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-dense-reference', action='store_true', help='skip the extra timing of the dense kernels (variant 2) at N=1')
    ap.add_argument('--debug-gloo-shared-gpu', action='store_true', help='debug only: N ranks on GPU 0, gloo backend, halos staged through the host (validates the multi-rank code path on a 1-GPU box)')
    ap.add_argument('--lean-host', action='store_true', help='build the inputs slab-style (size-1 Ox/Oy/Oz) also at N=1: for 1024^3 on one GPU')
    ap.add_argument('--cpu-sample', type=int, nargs=4, default=[384, 384, 256, 64], help='N1 N2 N3 steps of the oracle sample')
    return ap.parse_args()


def cpu_baseline(args, dt_fn):
    """The oracle (build's own CPU restatement, kind 'port') timed on this host on a bounded
    sample of the same workload: a smaller grid of the same medium/source/sensor construction."""
    from babelbrain_amd import harness as H
    from oracle import oracle as O
    n1, n2, n3, steps = args.cpu_sample
    a, k, info = H.make_problem(args.config, N=(n1, n2, n3), steps=steps, stable_dt_fn=dt_fn, accumulate_all_steps=True)
    cores = O.usable_cpus()
    # the box may grant fewer CPUs than it shows (cgroup quota): probe a few thread counts on 3 steps and
    # time the sample with the best one (measured on the GPU box: 16 threads 645, 64 threads 225, 256 threads 13)
    if 'OMP_NUM_THREADS' in os.environ:
        threads = int(os.environ['OMP_NUM_THREADS'])
    else:
        ap, kp, _ = H.make_problem(args.config, N=(n1, n2, n3), steps=3, stable_dt_fn=dt_fn, accumulate_all_steps=True)
        best = (0.0, 1)
        for cand in (8, 16, 32, 64, 128, 256):
            if cand > cores:
                break
            o = O.StaggeredFDTD_3D_with_relaxation(*ap, nthreads=cand, **kp)
            rate = 1.0 / max(o[-1]['stepLoopSeconds'], 1e-9)
            if rate > best[0]:
                best = (rate, cand)
            elif rate < 0.6 * best[0]:
                break
        threads = best[1]
    out = O.StaggeredFDTD_3D_with_relaxation(*a, nthreads=threads, **k)
    secs = out[-1]['stepLoopSeconds']
    model = ''
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                model = line.split(':', 1)[1].strip()
                break
    except Exception:
        pass
    return {'value': n1 * n2 * n3 * steps / secs / 1e6, 'unit': 'Mvoxel-steps/s', 'cores': threads, 'kind': 'port',
            'sample': '%s medium/source on a %dx%dx%d grid, %d steps, OpenMP float32 oracle (oracle/fdtd_oracle.c), thread count picked by a 3-step probe' % (args.config, n1, n2, n3, steps),
            'cpu_model': model, 'host_cores': cores}


def measured_traffic(args, n1, n2, n3):
    """HBM bytes per half-step from the committed rocprofv3 PMC profile of this workload, if one exists
    (profiles/traffic.json, written from the PMC passes of scripts/pmc_passes.sh)."""
    try:
        t = json.load(open(os.path.join(ROOT, 'profiles', 'traffic.json')))
        key = '%s_%dx%dx%d_variant%d' % (args.config, n1, n2, n3, args.variant)
        return t.get(key, {})
    except Exception:
        return {}


def main():
    args = parse()
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with torch.distributed.run --nproc-per-node %d for --gpus %d' % (args.gpus, args.gpus))
        args.gpus = world
    import torch
    from babelbrain_amd import _engine, harness as H, slab, RayleighAndBHTE
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the HIP engine has no CPU fallback)')
    shared = args.debug_gloo_shared_gpu
    if shared:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if shared:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            # RCCL's stream at high priority: its small send/recv kernel must not queue behind the 16k workgroups of the
            # interior half-step (rocprofv3 timeline: it otherwise starts only when the interior kernel has drained)
            os.environ.setdefault('TORCH_NCCL_HIGH_PRIORITY', '1')
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))

    def dt_fn(ml, f, h, acfl):
        return _engine.stable_dt(ml, f, True, h, acfl)

    cfg = H.CONFIGS[args.config]
    n1, n2, n3 = args.size if args.size else cfg['N']
    N = (n1, n2, n3 * world)                       # weak scaling: one full grid per GPU
    nt = args.steps + args.warmup
    t0 = time.time()
    if world == 1 and not args.lean_host:
        a, k, info = H.make_problem(args.config, N=N, steps=nt, stable_dt_fn=dt_fn, forward=RayleighAndBHTE.ForwardSimple)
        local = None
    else:   # every rank builds only its own Z-slab of the domain
        k0, nk = slab.partition(N[2], world)[rank]
        RayleighAndBHTE._device = local_rank
        a, k, info = H.make_problem(args.config, N=N, steps=nt, stable_dt_fn=dt_fn, zslab=(k0, nk), forward=RayleighAndBHTE.ForwardSimple)
        local = (N[2], k0, nk) + tuple(info['ghost'])
    t_build = time.time() - t0
    # rmsFirstStep=1: the Pressure RMS accumulates in EVERY step (warm-up included), as the 172 B/voxel-step accounting
    # assumes; a production call accumulates only over the last 2 periods (the sensors keep that window here)
    s, sinfo = slab.create_hip_slab(a, k, rank, world, local_rank, kernelVariant=args.variant, local=local, host_staging=shared,
                                    rmsFirstStep=1)
    eng = s.eng
    runner = slab.SlabRunner(s, rank, world, dist, overlap=False if shared else None)
    nvox_rank = float(n1) * n2 * sinfo['nk']

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    runner.run(args.warmup)
    barrier()
    eng.timing_begin(True)
    t0 = time.perf_counter()
    runner.run(args.steps)
    torch.cuda.synchronize()
    barrier()
    wall = time.perf_counter() - t0
    tm = eng.timing_end()
    if world > 1:
        w = torch.tensor([wall], dtype=torch.float64, device='cpu' if shared else 'cuda')
        dist.all_reduce(w, op=dist.ReduceOp.MAX)
        wall = float(w.item())
    total_vox = float(n1) * n2 * n3 * world
    value = total_vox * args.steps / wall / 1e6

    if rank == 0:
        st = tm['stress_ms'] / args.steps * 1e-3          # all launches of one stress half-step
        ve = tm['velocity_ms'] / args.steps * 1e-3        # all launches of one velocity half-step (+ fused Pressure RMS)
        step_dev = tm['total_ms'] / args.steps * 1e-3
        ach_stress = BYTES_STRESS * nvox_rank / st / 1e9
        ach_vel = (BYTES_VELOCITY + BYTES_RMS) * nvox_rank / ve / 1e9
        ach_step = BYTES_STEP * nvox_rank / step_dev / 1e9
        traffic = measured_traffic(args, n1, n2, n3)
        # `roofline` describes the half-step that takes longer; the other one is in `roofline_other`
        halves = {'stress half-step': dict(achieved=ach_stress, frac=ach_stress / HBM_PEAK_GBS, algorithmic_bytes_per_voxel=BYTES_STRESS,
                                           avg_launch_ms=st * 1e3, traffic=traffic.get('stress')),
                  'velocity half-step (+ fused Pressure RMS)': dict(achieved=ach_vel, frac=ach_vel / HBM_PEAK_GBS,
                                                                    algorithmic_bytes_per_voxel=BYTES_VELOCITY + BYTES_RMS,
                                                                    avg_launch_ms=ve * 1e3, traffic=traffic.get('velocity'))}
        dom = max(halves, key=lambda n: halves[n]['avg_launch_ms'])
        oth = [n for n in halves if n != dom][0]
        line = {
            'metric': 'Mvoxel-steps/sec, 512^3 skull FDTD (achieved HBM GB/s in roofline)',
            'value': value, 'unit': 'Mvoxel-steps/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': wall / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': '%s: %dx%dx%d per GPU (%dx%dx%d total), %s medium, %s source, PML 12, %d materials, '
                                   'Pressure RMS accumulated in every step, sensors over the last 2 periods' % (args.config, n1, n2, n3, N[0], N[1], N[2], info['medium'], info['tx'], info['n_mat']),
                       'parallelism': 'z-slab x%d' % world, 'kernel_variant': args.variant, 'dt': info['dt'], 'ppp': info['ppp'],
                       'n_sources': info['n_sources'], 'n_sensors_rank0': int(eng.num_sensors), 'tiles_rank0': eng.tile_counts(),
                       'halo_exchange': 'overlapped' if runner.overlap else ('none' if world == 1 else 'blocking')},
            'roofline': dict(bound='hbm', kernel=dom, peak=HBM_PEAK_GBS, unit='GB/s',
                             note='achieved = algorithmic bytes (SURVEY 8d) / measured time; the tiled kernels move fewer bytes than '
                                  'the algorithmic count (exact fluid-tile shortcuts), so frac can exceed 1; traffic = HBM bytes per '
                                  'launch from rocprofv3 PMC (profiles/), null if no profile matches this grid',
                             **halves[dom]),
            'roofline_other': dict(kernel=oth, **halves[oth]),
            'roofline_step': {'achieved': ach_step, 'frac': ach_step / HBM_PEAK_GBS, 'algorithmic_bytes_per_voxel_step': BYTES_STEP,
                              'device_ms_per_step': step_dev * 1e3, 'other_ms_per_step': tm['other_ms'] / args.steps},
            'device_bytes': int(eng.device_bytes), 'host_build_s': t_build,
        }
        if world == 1 and not args.no_dense_reference and args.variant in (0, 3):
            # the same workload on the dense LDS-tiled kernels (variant 2: every array of every voxel is read and
            # written, no fluid-tile shortcuts) -- the like-for-like figure against the 172 B algorithmic count
            try:
                eng.close()
                s2, _ = slab.create_hip_slab(a, k, 0, 1, local_rank, kernelVariant=2, local=local, rmsFirstStep=1)
                r2 = slab.SlabRunner(s2, 0, 1, None)
                r2.run(args.warmup)
                torch.cuda.synchronize()
                s2.eng.timing_begin(False)
                r2.run(args.steps)
                t2 = s2.eng.timing_end()
                ms2 = t2['total_ms'] / args.steps
                ach2 = BYTES_STEP * nvox_rank / (ms2 * 1e-3) / 1e9
                line['dense_reference'] = {'kernel_variant': 2, 'value': total_vox / (ms2 * 1e-3) / 1e6, 'unit': 'Mvoxel-steps/s',
                                           'device_ms_per_step': ms2, 'roofline_step': {'achieved': ach2, 'frac': ach2 / HBM_PEAK_GBS,
                                                                                        'algorithmic_bytes_per_voxel_step': BYTES_STEP}}
                s2.eng.close()
            except Exception as e:
                line['dense_reference'] = {'value': None, 'error': repr(e)}
        if not args.no_cpu_baseline and world == 1:
            try:
                line['cpu_baseline'] = cpu_baseline(args, dt_fn)
            except Exception as e:   # the baseline is a reported extra; never lose the GPU line over it
                line['cpu_baseline'] = {'value': None, 'unit': 'Mvoxel-steps/s', 'cores': 0, 'kind': 'port', 'sample': 'failed: %r' % (e,)}
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
