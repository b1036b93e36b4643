#!/usr/bin/env python3
"""One-process Z-slab split (bfd_group_*) at bench size: throughput, host time spent queueing the step loop, halo bytes.
On a 1-GPU box every slab sits on device 0 (what the split costs in launches, events and copies); with several GPUs
visible the slabs go to distinct devices (peer copies over xGMI).
  python scripts/r3/group_bench.py [--config C3] [--slabs 1 2 4 8] [--steps 120] [--size N1 N2 N3]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='C3')
    ap.add_argument('--slabs', type=int, nargs='+', default=[1, 2, 4, 8])
    ap.add_argument('--steps', type=int, default=120)
    ap.add_argument('--warmup', type=int, default=40)
    ap.add_argument('--size', type=int, nargs=3, default=None)
    args = ap.parse_args()
    from babelbrain_amd import _engine, harness as H, RayleighAndBHTE
    from babelbrain_amd.PropagationModel import compact_sources
    devs = [d for d, _ in _engine.list_devices()]
    dt_fn = lambda ml, f, h, c: _engine.stable_dt(ml, f, True, h, c)
    N = tuple(args.size) if args.size else H.CONFIGS[args.config]['N']
    nt = args.steps + args.warmup
    a, k, info = H.make_problem(args.config, N=N, steps=nt, stable_dt_fn=dt_fn, zslab=(0, N[2]), full_sensors=False, forward=RayleighAndBHTE.ForwardSimple)
    MaterialMap, ml, f, SourceMap, Pulse, h, T, SensorMap = a
    lin, row, wx, wy, wz = compact_sources(np.asarray(SourceMap), k['Ox'], k['Oy'], k['Oz'])
    vox = float(N[0]) * N[1] * N[2]
    for n in args.slabs:
        devices = [devs[r % len(devs)] for r in range(n)] if len(devs) > 1 else [0] * n
        g = _engine.Group(devices, *N, len(ml), h, k['DT'], f, nt, sensorSub=k['SensorSubSampling'], sensorStart=k['SensorStart'],
                          selRMSorPeak=1, selMapsRMS=['Pressure'], selMapsSensors=['Pressure'], rmsFirstStep=1)
        t0 = time.time()
        g.set_materials(ml, k.get('QCorrection', 1.0))
        g.set_material_map(MaterialMap)
        g.set_sources(lin.astype(np.int64), row, wx, wy, wz, Pulse)
        g.set_sensor_map(SensorMap)
        g.prepare()
        setup = time.time() - t0
        g.run(args.warmup)
        g.sync()
        g.timing_begin()
        g.run(args.steps)
        tm = g.timing_end()
        print(json.dumps({'config': args.config, 'grid': list(N), 'slabs': n, 'devices': devices, 'Gvoxel_steps_per_s': vox * args.steps / tm['total_ms'] / 1e6,
                          'ms_per_step': tm['total_ms'] / args.steps, 'max_device_ms_per_step': tm['max_device_ms'] / args.steps,
                          'host_issue_ms_per_step': tm['host_issue_ms'] / args.steps, 'halo_MB_per_step': tm['halo_bytes_per_step'] / 1e6,
                          'overlapped': tm['overlapped'], 'setup_s': setup}), flush=True)
        g.close()


if __name__ == '__main__':
    main()
