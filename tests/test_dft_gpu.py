"""On-device single-bin DFT of the sensor block against (1) the golden output of the
reference's own CalculatePhaseData (BASE:2460-2560, captured by tests/golden/make_golden.py) and
(2) numpy's FFT of the series the engine itself returns."""
import numpy as np
import pytest

from babelbrain_amd import harness as H
from tests.util import oracle_dt, rel_l2

pytestmark = pytest.mark.gpu


def test_dft_matches_reference_calculate_phase_data(golden):
    from babelbrain_amd import _engine
    g, _ = golden
    N1, N2, N3, ppp, sub, f, dts = g['phase_args']
    F, pk = _engine.dft_series(g['phase_series'], dts, f)
    four, phase, peak = H.phase_maps(F, pk, g['phase_index'], int(N1), int(N2), int(N3))
    assert rel_l2(four.real, g['phase_fourier'].real) < 1e-6 and rel_l2(four.imag, g['phase_fourier'].imag) < 1e-6
    assert np.array_equal(peak, g['phase_peak'])


def test_engine_sensor_dft_matches_host_fft():
    from babelbrain_amd import PropagationModel
    a, k, info = H.make_problem('C2', N=(56, 52, 80), steps=300, stable_dt_fn=oracle_dt)
    k['SelMapsSensorsList'] = ['Pressure', 'Vz']
    out = PropagationModel().StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, ReturnSensorDFT=True, **k)
    S, Inp = out[0], out[-1]
    nTs = S['time'].size
    assert nTs == 2 * info['ppp'] // k['SensorSubSampling']
    freqs = np.fft.fftfreq(nTs, np.diff(S['time']).mean())
    ind = np.argmin(np.abs(freqs - info['freq']))
    for name in ('Pressure', 'Vz'):
        ref = np.fft.fft(S[name].astype(np.float64), axis=1)[:, ind] * 2 / nTs
        got = Inp['SensorDFT'][name]
        assert np.abs(ref).max() > 0
        assert rel_l2(got.real, ref.real) < 1e-6 and rel_l2(got.imag, ref.imag) < 1e-6
        assert np.array_equal(Inp['SensorPeak'][name], S[name].max(axis=1))


def test_in_loop_accumulation_equals_dft_of_stored_series():
    """ReturnSensorSeries=False: re/im/peak are accumulated per sensor while the samples are taken and the series are
    never stored; the result is the DFT of the stored series sample for sample (same arithmetic), device memory shrinks."""
    from babelbrain_amd import PropagationModel
    a, k, info = H.make_problem('C2', N=(56, 52, 80), steps=300, stable_dt_fn=oracle_dt)
    k['SelMapsSensorsList'] = ['Pressure', 'Vz']
    m = PropagationModel()
    full = m.StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, ReturnSensorDFT=True, **k)
    lean = m.StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, ReturnSensorDFT=True, ReturnSensorSeries=False, **k)
    assert set(lean[0].keys()) == {'time'} and np.array_equal(lean[0]['time'], full[0]['time'])
    assert np.array_equal(lean[-1]['IndexSensorMap'], full[-1]['IndexSensorMap'])
    for name in ('Pressure', 'Vz'):
        assert np.abs(full[-1]['SensorDFT'][name]).max() > 0
        assert np.array_equal(lean[-1]['SensorDFT'][name], full[-1]['SensorDFT'][name])
        assert np.array_equal(lean[-1]['SensorPeak'][name], full[-1]['SensorPeak'][name])
    assert np.array_equal(lean[2]['Pressure'], full[2]['Pressure'])
    nS, nTs = full[0]['Pressure'].shape
    assert full[-1]['device_bytes'] - lean[-1]['device_bytes'] == 2 * nS * (4 * nTs - 20)
    with pytest.raises(ValueError):
        m.StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, ReturnSensorSeries=False, **k)


def test_solver_to_data_for_sim_file(tmp_path):
    """The acoustic step end to end on this package: solver call with on-device DFT -> volumes
    (CalculatePhaseData's outputs) -> caller scaling -> DataForSim (BASE:2812-2885) -> HDF5 -> read back."""
    from babelbrain_amd import PropagationModel, results as R, datafile as DF
    try:
        DF.backend()
    except ImportError:
        pytest.skip('no HDF5 library on this machine')
    N1, N2, N3 = 56, 52, 80
    a, k, info = H.make_problem('C2', N=(N1, N2, N3), steps=720, stable_dt_fn=oracle_dt)
    out = PropagationModel().StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, ReturnSensorDFT=True, **k)
    rms, Inp = out[2], out[-1]
    corr = 1.07
    four, phase, peak = H.phase_maps(Inp['SensorDFT']['Pressure'] * corr, Inp['SensorPeak']['Pressure'] * corr,
                                     Inp['IndexSensorMap'], N1, N2, N3)
    in_peak = rms['Pressure'] * np.float32(corr * np.sqrt(2))                 # BASE:2439-2440
    zsrc = info['zsrc']
    crop = R.Crop(12, 12, 12, 12, 12, 12)
    h = a[5]
    focal = (N1 // 2, N2 // 2, N3 - 20)
    mm = a[0]
    d = R.data_for_sim(crop, zsrc, in_peak.copy(), four.copy(), mm, focal, a[1], (np.arange(N1) - N1 / 2) * h,
                       (np.arange(N2) - N2 / 2) * h, (np.arange(N3) - zsrc) * h, h, 4e-2)
    d['bDoRefocusing'] = False
    fn = str(tmp_path / 'case_DataForSim.h5')
    DF.SaveToH5py(d, fn)
    r = DF.ReadFromH5py(fn)
    assert sorted(r) == sorted(d)
    for key in ('p_amp', 'p_complex', 'MaterialMap', 'x_vec', 'z_vec', 'TargetLocation', 'Material'):
        assert np.array_equal(r[key], d[key]) and r[key].dtype == np.asarray(d[key]).dtype, key
    assert r['p_amp'].shape == (N1 - 24, N2 - 24, N3 - 24) and r['p_amp'].dtype == np.float32 and r['p_complex'].dtype == np.complex64
    # amplitude of the DFT map and sqrt(2)*RMS agree where the field is steady (same quantity two ways, BASE:2520 vs 2440)
    amp = np.abs(r['p_complex'])
    sel = r['p_amp'] > 0.2 * r['p_amp'].max()
    assert sel.sum() > 100
    assert np.median(np.abs(amp[sel] / r['p_amp'][sel] - 1)) < 0.02
    # Z is flipped on the way out: nothing at or before the source plane -> zeros at the far end of the file's z axis
    nz_zero = zsrc + 1 - 12
    if nz_zero > 0:
        assert not r['p_amp'][:, :, -nz_zero:].any()
    assert r['bDoRefocusing'] is False
