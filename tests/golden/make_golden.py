#!/usr/bin/env python3
"""Generates tests/golden/harness_golden.npz + harness_golden.json by IMPORTING the reference's
caller modules from /root/reference (this container only) and running their own code on small
inputs. The outputs are data (inputs + expected outputs); no reference source is stored.

The FDTD solver itself (BabelViscoFDTD) is absent from the reference tree, so these vectors pin
the caller-side contract (what is fed to / read from the solver), not the kernel arithmetic:
parity with BabelViscoFDTD's numerics stays UNPINNED.

Run:  python tests/golden/make_golden.py
"""
import inspect
import json
import os
import sys
import textwrap
import types
from types import SimpleNamespace
from unittest import mock

import numpy as np

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))


def install_stubs():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    def fake_read(path):
        # MapPichardo.h5 is only used for CT maps, not for anything captured here
        return {'rho': np.linspace(1000, 3000, 8), 'freq': np.linspace(1e5, 1.2e6, 8),
                'MapSoS': np.ones((8, 8)), 'MapAtt': np.ones((8, 8))}
    stub('BabelViscoFDTD')
    stub('BabelViscoFDTD.H5pySimple', ReadFromH5py=fake_read, SaveToH5py=lambda *a, **k: None)
    stub('BabelViscoFDTD.PropagationModel', PropagationModel=mock.MagicMock)
    stub('BabelViscoFDTD.tools')
    stub('BabelViscoFDTD.tools.RayleighAndBHTE', InitCuda=None, InitOpenCL=None, InitMetal=None,
         ForwardSimple=None, SpeedofSoundWater=lambda t: 1482.0, GenerateFocusTx=None)
    for n in ['nibabel', 'SimpleITK', 'h5py', 'pwlf', 'stl', 'trimesh', 'trimesh.creation', 'mkl_fft_absent']:
        sys.modules.setdefault(n, mock.MagicMock())
    stub('linetimer', CodeTimer=mock.MagicMock)
    sys.modules['stl'].mesh = mock.MagicMock()


def block_between(func, start_marker, end_marker, include_end=True):
    """Source lines of `func` from the first line containing start_marker to the first later line
    containing end_marker -- executed in place, never stored."""
    src = inspect.getsource(func).splitlines()
    a = next(i for i, l in enumerate(src) if start_marker in l)
    b = next(i for i in range(a, len(src)) if end_marker in src[i])
    return textwrap.dedent('\n'.join(src[a:b + (1 if include_end else 0)]))


def main():
    install_stubs()
    sys.path.insert(0, REF)
    import matplotlib
    matplotlib.use('Agg')
    from TranscranialModeling import BabelIntegrationBASE as B
    from TranscranialModeling import BabelIntegrationSingle as S

    out = {}
    meta = {'generated_from': 'TranscranialModeling/BabelIntegrationBASE.py, BabelIntegrationSingle.py',
            'cases': {}}

    # (1) material rows, BASE:140-167, and GetSmallestSOS, BASE:170-182
    names = ['Water', 'Cortical', 'Trabecular', 'Skin', 'Brain']
    for f in (500e3, 700e3, 1000e3):
        out['matfreq_%d' % int(f)] = np.array([B.MatFreq[f][n] for n in names], np.float64)
        out['smallest_sos_%d' % int(f)] = np.array([B.GetSmallestSOS(f, bShear=True), B.GetSmallestSOS(f, bShear=False)])
    meta['material_names'] = names

    # (2) PPP snapping, BASE:1809-1828 (block of UpdateConditions run on a stand-in self)
    ppp_code = block_between(B.SimulationConditionsBASE.UpdateConditions, 'self._PPP=np.ceil(', 'TemporalStep=1/self._Frequency/self._PPP')
    rows = []
    for f in (250e3, 500e3, 700e3, 1000e3):
        for dt_ideal in np.concatenate([1 / f / np.array([22.3, 23, 30.5, 31, 33.2, 34, 46.1, 47, 52.7, 58.4, 70.2, 71, 73.5, 74, 78.1, 79, 93.9]),
                                        [2.1e-8, 3.47e-8, 5.05e-8, 7.9e-8]]):
            ns = {'np': np, 'self': SimpleNamespace(_Frequency=f), 'TemporalStep': float(dt_ideal)}
            exec(ppp_code, ns)
            rows.append([f, dt_ideal, ns['self']._PPP, ns['TemporalStep']])
    out['ppp_rule'] = np.array(rows, np.float64)

    # (3) TimeSimulation / nt / SensorSubSampling / SensorStart, BASE:2077-2109
    plan_code = block_between(B.SimulationConditionsBASE.UpdateConditions, 'self._DimDomain=np.zeros((3))', 'self._SensorStart=int(')
    rows = []
    for (N1, N2, N3, f, ppp, sub) in [(128, 128, 128, 500e3, 30, 0), (256, 256, 256, 500e3, 60, 0), (96, 80, 140, 700e3, 35, 0),
                                       (64, 64, 100, 250e3, 25, 5), (128, 100, 90, 1000e3, 48, 0), (70, 70, 70, 500e3, 32, 6)]:
        h = 1102.515 / f / 6
        dt = 1 / f / ppp
        slf = SimpleNamespace(_N1=N1, _N2=N2, _N3=N3, _PMLThickness=12, _TemporalStep=dt, _PPP=float(ppp),
                              _SensorSubSampling=sub, _NumberCyclesToTrackAtEnd=2)
        ns = {'np': np, 'self': slf, 'SpatialStep': h, 'MatArray': np.array([[1000.0, 1500.0, 0, 0, 0]]), 'print': lambda *a, **k: None}
        exec(plan_code, ns)
        rows.append([N1, N2, N3, f, ppp, sub, h, dt, slf._TimeSimulation, ns['ntSteps'], slf._SensorSubSampling, slf._SensorStart])
    out['time_plan'] = np.array(rows, np.float64)

    # (4) CreateSensorMap, BASE:2279-2290
    slf = SimpleNamespace(_N1=30, _N2=28, _N3=40, _PMLThickness=12, _ZSourceLocation=14, _bDisplay=False)
    B.SimulationConditionsBASE.CreateSensorMap(slf)
    out['sensormap_args'] = np.array([30, 28, 40, 12, 14])
    out['sensormap'] = slf._SensorMap
    out['sensormap_back'] = slf._SensorMapBackPropagation

    # (5) CreateSources, Single:313-346
    rng = np.random.default_rng(7)
    N1, N2, N3, pml, zsrc = 34, 32, 40, 12, 15
    plane = np.zeros((N1, N2), np.complex64)
    plane[pml:-pml, pml:-pml] = (rng.normal(size=(N1 - 2 * pml, N2 - 2 * pml)) + 1j * rng.normal(size=(N1 - 2 * pml, N2 - 2 * pml)))
    plane[14, 14] = 0
    f, ppp = 500e3, 30
    dt = 1 / f / ppp
    T = dt * 150
    slf = SimpleNamespace(_TimeSimulation=T, _Frequency=f, _TemporalStep=dt, _N1=N1, _N2=N2, _N3=N3, _ZSourceLocation=zsrc,
                          _SourceMapRayleigh=plane, _bDisplay=False)
    S.SimulationConditions.CreateSources(slf)
    out['sources_plane'] = plane
    out['sources_args'] = np.array([f, dt, T, N3, zsrc])
    out['sources_map'] = slf._SourceMap
    out['sources_pulse'] = slf._PulseSource

    # (6) RUN_SIMULATION, BASE:2299-2458: which arguments the solver receives, and the scaling
    #     applied to what it returns (dispersion correction, sqrt(2))
    calls = []
    N = (26, 26, 30)

    class Recorder:
        def StaggeredFDTD_3D_with_relaxation(self, *a, **k):
            calls.append((a, k))
            sensor = {'time': np.arange(8) * 1e-7, 'Pressure': np.ones((5, 8), np.float32)}
            return sensor, {}, {'Pressure': np.ones(N, np.float32)}, {'IndexSensorMap': np.arange(1, 6, dtype=np.uint32)}
    dt, dtw = 6.0e-8, 1.4e-7
    slf = SimpleNamespace(_MaterialMap=np.zeros(N, np.uint32), _Frequency=500e3, _SourceMap=np.zeros(N, np.uint32),
                          _PulseSource=np.zeros((1, 10)), _SpatialStep=3.6e-4, _TimeSimulation=6e-6, _SensorMap=np.zeros(N, np.uint32),
                          _FactorConvPtoU=1.5e6, _PMLThickness=12, _TemporalStep=dt, _ReflectionLimit=1e-5,
                          _QfactorCorrection=True, _QCorrection=np.ones(3), _SensorSubSampling=6, _SensorStart=11,
                          _SubAirRegions=None, DominantMediumTemporalStep=dtw,
                          _DispersionCorrection=[-2307.53581298, 6875.73903172, -7824.73175146, 4227.49417250, -975.22622721],
                          ReturnArrayMaterial=lambda: np.array([[1000.0, 1500.0, 0, 0, 0]]))
    with mock.patch.object(B, 'PModel', Recorder()), mock.patch('builtins.print'):
        B.SimulationConditionsBASE.RUN_SIMULATION(slf, GPUName='MI355X', COMPUTING_BACKEND=5, bDoRefocusing=False)
    a, k = calls[0]
    meta['run_simulation_positional'] = len(a)
    meta['run_simulation_kwargs'] = {kk: (type(v).__name__ if not isinstance(v, np.ndarray) else 'ndarray%s%s' % (v.dtype, list(v.shape)))
                                     for kk, v in k.items()}
    meta['run_simulation_kwvalues'] = {kk: v for kk, v in k.items() if isinstance(v, (int, float, bool, str))}
    meta['run_simulation_kwvalues']['SelMapsRMSPeakList'] = list(k['SelMapsRMSPeakList'])
    meta['run_simulation_kwvalues']['SelMapsSensorsList'] = list(k['SelMapsSensorsList'])
    out['run_oz_value'] = np.array([float(k['Oz'].reshape(-1)[0]), float(k['Ox'].reshape(-1)[0])])
    out['run_scaling'] = np.array([dt, dtw, float(slf._DictPeakValue['Pressure'].reshape(-1)[0]), float(slf._Sensor['Pressure'].reshape(-1)[0])])

    # (7) CalculatePhaseData, BASE:2460-2560: sensor series -> complex map / peak map
    N1, N2, N3 = 26, 27, 30
    ppp, sub = 30, 3
    dts = sub / 500e3 / ppp
    nTs = 2 * ppp // sub
    sm = np.zeros((N1, N2, N3), np.uint32)
    sm[12:-12, 12:-12, 15:-12] = 1
    lin = np.flatnonzero(sm.transpose(2, 1, 0).ravel()) + 1      # x-fastest, 1-based
    rng = np.random.default_rng(3)
    amp = rng.uniform(0.5, 2, lin.size); ph = rng.uniform(-np.pi, np.pi, lin.size)
    t = np.arange(nTs) * dts
    series = (amp[:, None] * np.sin(2 * np.pi * 500e3 * t[None, :] + ph[:, None])).astype(np.float32)
    slf = SimpleNamespace(_N1=N1, _N2=N2, _N3=N3, _PPP=ppp, _SensorSubSampling=sub, _Frequency=500e3, _PMLThickness=12,
                          _Sensor={'time': t, 'Pressure': series.copy()}, _InputParam=lin.astype(np.uint32),
                          _DictPeakValue={'Pressure': np.zeros((N1, N2, N3), np.float32)})
    with mock.patch('builtins.print'):
        B.SimulationConditionsBASE.CalculatePhaseData(slf, bDoRefocusing=False)
    out['phase_args'] = np.array([N1, N2, N3, ppp, sub, 500e3, dts])
    out['phase_index'] = lin.astype(np.uint32)
    out['phase_series'] = series
    out['phase_fourier'] = slf._PressMapFourier
    out['phase_peak'] = slf._PressMapPeak

    # (8) refocusing: BackPropagationRayleigh + CreateSourcesRefocus of the phased-array integration
    #     (BabelIntegrationCONCAVE_PHASEDARRAY.py:407-484), with ForwardSimple (absent package) replaced by the
    #     float64 Rayleigh sum of oracle/rayleigh_oracle.py -- pins the orchestration arithmetic around it
    from TranscranialModeling import BabelIntegrationCONCAVE_PHASEDARRAY as CC
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import rayleigh_oracle as RO
    rng = np.random.default_rng(11)
    N1, N2, N3, pml, zsrc = 30, 28, 36, 12, 14
    h = 4e-4
    XDim = (np.arange(N1) - N1 / 2) * h
    YDim = (np.arange(N2) - N2 / 2) * h
    ZDim = (np.arange(N3) - zsrc) * h
    smr = np.zeros((N1, N2), np.complex64)
    smr[pml:-pml, pml:-pml] = (rng.normal(size=(N1 - 2 * pml, N2 - 2 * pml)) + 1j * rng.normal(size=(N1 - 2 * pml, N2 - 2 * pml)))
    back = (rng.normal(size=(N1, N2)) + 1j * rng.normal(size=(N1, N2))).astype(np.complex64)
    nElem, edims = 5, 7
    elemc = np.stack([rng.uniform(-4e-3, 4e-3, nElem), rng.uniform(-4e-3, 4e-3, nElem), np.full(nElem, -6e-3)], 1).astype(np.float32)
    cen = np.repeat(elemc, edims, axis=0) + rng.normal(scale=2e-4, size=(nElem * edims, 3)).astype(np.float32)
    Tx = {'elemcenter': elemc, 'center': cen.astype(np.float32), 'ds': np.full((nElem * edims, 1), 1e-7, np.float32),
          'NumberElems': nElem, 'elemdims': edims}
    slf = SimpleNamespace(_SourceMapRayleigh=smr, _PressMapFourierBack=back, _XDim=XDim, _YDim=YDim, _ZDim=ZDim,
                          _ZSourceLocation=zsrc, _SpatialStep=h, _Frequency=500e3, _Tx=Tx, _SourceAmpPa=2.5,
                          BasePhasedArrayProgrammingRefocusing=np.zeros(nElem, np.complex64), _PMLThickness=pml,
                          AdjustWeightAmplitudes=lambda: 1.0, _N1=N1, _N2=N2, _N3=N3, _TemporalStep=1 / 500e3 / 30,
                          _TimeSimulation=150 / 500e3 / 30)
    with mock.patch.object(CC, 'ForwardSimple', lambda k, c, d, u, r, deviceMetal=None: RO.ForwardSimple(k, c, d, u, r)):
        CC.SimulationConditions.BackPropagationRayleigh(slf)
    CC.SimulationConditions.CreateSourcesRefocus(slf)
    out['refocus_args'] = np.array([N1, N2, N3, pml, zsrc, h, 500e3, 2.5, slf._TemporalStep, slf._TimeSimulation])
    out['refocus_source_plane'] = smr
    out['refocus_back_plane'] = back
    out['refocus_elemcenter'] = elemc
    out['refocus_center'] = cen.astype(np.float32)
    out['refocus_ds'] = Tx['ds']
    out['refocus_elem'] = np.array([nElem, edims])
    out['refocus_programming'] = slf.BasePhasedArrayProgrammingRefocusing
    out['refocus_plane'] = slf._SourceMapRayleighRefocus
    out['refocus_pulse'] = slf._PulseSourceRefocus

    # (8b) CreateSources of the phased-array integration (CONCAVE:358-404): besides the plane sources it builds the
    #      point source of the back-propagation call (row a5 of SURVEY 8a): PunctualSource (sine, ramped up AND down)
    #      and SourceMapPunctual (one voxel at the focal spot)
    for tag, nsteps in (('a', 360), ('b', 150)):
        slf2 = SimpleNamespace(_SourceMapRayleigh=smr, _N1=N1, _N2=N2, _N3=N3, _ZSourceLocation=zsrc, _Frequency=500e3,
                               _TemporalStep=1 / 500e3 / 30, _TimeSimulation=nsteps / 500e3 / 30,
                               _FocalSpotLocation=np.array([N1 // 2 + 1, N2 // 2 - 2, 25]),
                               _XSteering=0.0, _YSteering=0.0, _ZSteering=0.0, _SpatialStep=h)
        CC.SimulationConditions.CreateSources(slf2)
        out['punctual_%s_args' % tag] = np.array([500e3, slf2._TemporalStep, slf2._TimeSimulation])
        out['punctual_%s_source' % tag] = slf2._PunctualSource
        out['punctual_%s_voxel' % tag] = np.array(np.nonzero(slf2._SourceMapPunctual)).reshape(-1)
        out['punctual_%s_value' % tag] = np.array([slf2._SourceMapPunctual.max(), slf2._SourceMapPunctual.sum()])
        out['punctual_%s_focal' % tag] = slf2._FocalSpotLocation
        if tag == 'a':
            out['concave_sources_map_plane'] = slf2._SourceMap[:, :, zsrc]
            out['concave_sources_pulse'] = slf2._PulseSource

    # (9) ReturnResults, BASE:2729-2896: crop of the absorbing layer, zeroing up to the source plane, Z flip,
    #     DataForSim dictionary (what Step 3 and the GUI read back from *DataForSim.h5)
    rng = np.random.default_rng(23)
    N1, N2, N3, pml, zsrc = 22, 20, 26, 4, 6
    orig = (12, 11, 15)                                     # shape of the mask volume the results are pasted into
    off = dict(_XLOffset=5, _XROffset=6, _YLOffset=4, _YROffset=6, _ZLOffset=5, _ZROffset=7)
    nx, ny, nz = N1 - 11, N2 - 10, N3 - 12
    shr = dict(_XShrink_L=1, _upperXR=1 + nx, _YShrink_L=0, _upperYR=ny, _ZShrink_L=1, _upperZR=1 + nz)

    def cplx(shape):
        return (rng.normal(size=shape) + 1j * rng.normal(size=shape)).astype(np.complex64)
    mmap = rng.integers(0, 5, size=(N1, N2, N3)).astype(np.uint32)
    fields = dict(_u2RayleighField=cplx((N1, N2, N3)), _InPeakValue=rng.uniform(0, 2, (N1, N2, N3)).astype(np.float32),
                  _PhaseMap=rng.uniform(-3, 3, (N1, N2, N3)).astype(np.float32), _PressMapFourier=cplx((N1, N2, N3)),
                  _InPeakValueRefocus=rng.uniform(0, 2, (N1, N2, N3)).astype(np.float32),
                  _PhaseMapRefocus=rng.uniform(-3, 3, (N1, N2, N3)).astype(np.float32),
                  _PressMapFourierRefocus=cplx((N1, N2, N3)), _PressMapFourierBack=cplx((N1, N2)))
    slf = SimpleNamespace(_ZSourceLocation=zsrc, _SkullMaskDataOrig=rng.integers(0, 5, size=orig).astype(np.uint8),
                          _bSaveStress=False, _bSaveDisplacement=False, _DictPeakValue={}, _DensityCTMap=None,
                          _MaterialMap=mmap.copy(), _FocalSpotLocation=np.array([11, 9, 14]), _SubAirRegions=None,
                          ReturnArrayMaterial=lambda: np.array([[1000.0, 1500.0, 0, 0, 0], [1896.5, 2476.0, 1542.0, 81.0, 164.0]]),
                          _XDim=(np.arange(N1) - N1 / 2) * 4e-4, _YDim=(np.arange(N2) - N2 / 2) * 4e-4, _ZDim=(np.arange(N3) - zsrc) * 4e-4,
                          _SpatialStep=4e-4, _zLengthBeyonFocalPointWhenNarrow=4e-2, _SourceMapRayleigh=cplx((N1, N2)),
                          _PMLThickness=pml, **off, **shr, **{k: v.copy() for k, v in fields.items()})
    for k, v in fields.items():
        out['rr_in' + k] = v
    out['rr_in_MaterialMap'] = mmap
    out['rr_in_SkullMaskDataOrig'] = slf._SkullMaskDataOrig
    out['rr_in_SourceMapRayleigh'] = slf._SourceMapRayleigh
    out['rr_args'] = np.array([N1, N2, N3, pml, zsrc, off['_XLOffset'], off['_XROffset'], off['_YLOffset'], off['_YROffset'],
                               off['_ZLOffset'], off['_ZROffset'], shr['_XShrink_L'], shr['_YShrink_L'], shr['_ZShrink_L'], 11, 9, 14])
    res = B.SimulationConditionsBASE.ReturnResults(slf, bDoRefocusing=True, bUseRayleighForWater=False)
    rnames = ['RayleighWater', 'RayleighWaterOverlay', 'FullSolutionPressure', 'FullSolutionPressureRefocus', 'DataForSim',
              'MaskCalcRegions', 'FullSolutionPhase', 'FullSolutionPhaseRefocus', 'RayleighWaterPhase']
    for nm, v in zip(rnames, res):
        if nm == 'DataForSim':
            meta['data_for_sim_keys'] = {kk: ('%s%s' % (np.asarray(vv).dtype, list(np.shape(vv)))) for kk, vv in v.items()}
            for kk, vv in v.items():
                out['rr_dfs_' + kk] = np.asarray(vv)
        else:
            out['rr_out_' + nm] = np.asarray(v)

    # (10) on-disk layout of a file written by the reference's SaveToH5py: TranscranialModeling/MapPichardo.h5
    #      (read here with the repository's own libhdf5 binding; structure + a few values kept as the fixture)
    from babelbrain_amd import datafile as DF
    meta['h5pysimple_layout'] = DF.describe(os.path.join(REF, 'TranscranialModeling', 'MapPichardo.h5'))
    mp = DF.ReadFromH5py(os.path.join(REF, 'TranscranialModeling', 'MapPichardo.h5'), use_h5py=False)   # h5py is a stub in this process
    out['h5_pichardo_rho_head'] = mp['rho'][:6]
    out['h5_pichardo_sos_corner'] = mp['MapSoS'][:3, :3]

    np.savez_compressed(os.path.join(HERE, 'harness_golden.npz'), **out)
    with open(os.path.join(HERE, 'harness_golden.json'), 'w') as fh:
        json.dump(meta, fh, indent=1, sort_keys=True, default=str)
    print('wrote', len(out), 'arrays')


if __name__ == '__main__':
    main()
