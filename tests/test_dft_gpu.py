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
