"""NIfTI-1 export of the result volumes (babelbrain_amd/nifti.py; Step10_GetResults, BASE:1459-1598). nibabel and SimpleITK,
which the reference uses, are not installed here: the writer is held to the NIfTI-1 layout itself (field offsets and codes
of the published nifti1.h), to its own reader, and the volume / affine bookkeeping to the reference's formulas."""
import gzip
import os
import struct

import numpy as np
import pytest

from babelbrain_amd import nifti as NI


def _affine(rng, zoom=(0.5, 0.5, 0.5), flip=False):
    # a rotation about a skew axis, scaled, with an offset: what a resampled T1W mask carries
    ax = rng.standard_normal(3); ax /= np.linalg.norm(ax)
    th = 0.4
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    R = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K
    if flip:
        R[:, 0] = -R[:, 0]
    A = np.eye(4); A[:3, :3] = R * np.array(zoom); A[:3, 3] = [-80.0, 35.5, 12.25]
    return A


def test_header_layout_follows_nifti1_h(tmp_path):
    rng = np.random.default_rng(0)
    A = _affine(rng, (0.4, 0.4, 0.4))
    vol = rng.random((7, 5, 3)).astype(np.float32)
    fn = str(tmp_path / 'v.nii')
    NI.save_nifti(fn, vol, A)
    raw = open(fn, 'rb').read()
    assert len(raw) == 352 + vol.size * 4
    assert struct.unpack_from('<i', raw, 0)[0] == 348 and raw[344:348] == b'n+1\0'
    assert struct.unpack_from('<8h', raw, 40) == (3, 7, 5, 3, 1, 1, 1, 1)
    assert struct.unpack_from('<h', raw, 70)[0] == 16 and struct.unpack_from('<h', raw, 72)[0] == 32        # DT_FLOAT32
    np.testing.assert_allclose(struct.unpack_from('<8f', raw, 76)[1:4], [0.4, 0.4, 0.4], rtol=1e-6)
    assert struct.unpack_from('<f', raw, 108)[0] == 352.0 and struct.unpack_from('<f', raw, 112)[0] == 1.0
    assert raw[123] == 10                                                                                     # mm | sec
    assert struct.unpack_from('<2h', raw, 252) == (2, 2)
    np.testing.assert_allclose(np.array([struct.unpack_from('<4f', raw, o) for o in (280, 296, 312)]), A[:3], rtol=1e-6, atol=1e-6)
    # x-fastest voxel order
    assert struct.unpack_from('<f', raw, 352 + 4)[0] == vol[1, 0, 0] and struct.unpack_from('<f', raw, 352 + 4 * 7)[0] == vol[0, 1, 0]


@pytest.mark.parametrize('flip', [False, True])
@pytest.mark.parametrize('dtype', [np.float32, np.uint8, np.complex64, np.float64])
def test_round_trip_and_qform_equals_sform(tmp_path, flip, dtype):
    rng = np.random.default_rng(3)
    A = _affine(rng, (0.75, 0.75, 0.75), flip)
    vol = (rng.random((6, 9, 4)) * 100).astype(dtype)
    fn = str(tmp_path / 'v.nii.gz')
    NI.save_nifti(fn, vol, A)
    data, B, zooms = NI.load_nifti(fn)
    assert data.dtype == np.dtype(dtype) and np.array_equal(data, vol)
    np.testing.assert_allclose(B, A, rtol=1e-6, atol=1e-5)
    np.testing.assert_allclose(zooms, 0.75, rtol=1e-6)
    # the quaternion form describes the same transform: rebuild the affine from qform fields only
    raw = bytearray(gzip.open(fn, 'rb').read())
    struct.pack_into('<h', raw, 254, 0)                       # sform_code = 0: readers fall back to the qform
    fq = str(tmp_path / 'q.nii')
    open(fq, 'wb').write(bytes(raw))
    _, C, _ = NI.load_nifti(fq)
    np.testing.assert_allclose(C, A, rtol=1e-5, atol=1e-4)


def test_enforced_isotropic_spacing(tmp_path):
    """SaveNiftiEnforcedISO: every file the reference keeps has pixdim = round(mean(zooms), 5) on all axes (BASE:754-757)"""
    A = np.diag([0.36751, 0.36749, 0.3675, 1.0])
    fn = str(tmp_path / 'v.nii.gz')
    NI.save_nifti(fn, np.zeros((4, 4, 4), np.float32), A, iso=True)
    _, B, zooms = NI.load_nifti(fn)
    assert np.allclose(zooms, 0.3675, atol=1e-7) and np.allclose(np.diag(B)[:3], 0.3675, atol=1e-7)


def test_step10_volumes_and_files(tmp_path):
    rng = np.random.default_rng(7)
    shape = (24, 20, 30)
    A = _affine(rng, (0.5, 0.5, 0.5))
    press = rng.random(shape).astype(np.float32)
    phase = rng.random(shape).astype(np.float32)
    water = rng.random(shape).astype(np.float32)
    calc = np.zeros(shape, bool); calc[4:18, 3:15, 6:27] = True
    ss = 2
    vols = NI.step10_volumes(A, calc, press, phase, RayleighWater=water, RayleighWaterOverlay=water + 1, subsamplingFactor=ss)
    # BASE:1472-1474 / 1484-1487 / 1499-1512, written out
    aff = A.copy(); aff[0:3, 0:3] = aff[0:3, 0:3] @ (np.eye(3) * ss)
    affSub = A.copy(); affSub[0:3, 3] = (A @ np.array([4, 3, 6, 1.0]))[:3]
    v, a = vols['FullElasticSolution__']
    assert np.array_equal(v, press[::2, ::2, ::2]) and np.allclose(a, aff)
    v, a = vols['FullElasticSolution_Sub__']
    assert np.array_equal(v, press[4:17, 3:14, 6:26]) and np.allclose(a, affSub)        # mx[0]:mx[-1] drops the last plane, as the reference does
    assert set(vols) == {'RayleighFreeWaterWOverlay__', 'RayleighFreeWater__', 'RayleighFreeWater_Sub__', 'FullElasticSolution__',
                         'FullElasticSolutionPhase__', 'FullElasticSolution_Sub__'}
    base = str(tmp_path / 'T_500kHz_6PPW_')
    FILENAMES = {k: base + k + '.nii.gz' for k in ('RayleighFreeWaterWOverlay__', 'RayleighFreeWater__', 'FullElasticSolution__',
                                                    'FullElasticSolutionPhase__', 'FullElasticSolution_Sub__')}
    mask = np.zeros(shape, np.uint8); mask[6:16, 5:13, 10:24] = 4; mask[10, 9, 15] = 5
    written = NI.save_step10(FILENAMES, vols, mask_data=mask, mask_affine=A)
    names = sorted(os.path.basename(w) for w in written)
    assert names == sorted(['T_500kHz_6PPW_' + n + '.nii.gz' for n in ('RayleighFreeWaterWOverlay', 'RayleighFreeWater', 'RayleighFreeWater_Sub',
                            'FullElasticSolution', 'FullElasticSolutionPhase', 'FullElasticSolution_Sub', 'FullElasticSolution_Sub_NORM')])
    norm, An, _ = NI.load_nifti(base + 'FullElasticSolution_Sub_NORM.nii.gz')
    sub = press[4:17, 3:14, 6:26].astype(np.float64)
    inside = mask[4:17, 3:14, 6:26] >= 4
    ref = np.where(inside, sub, 0.0); ref /= ref.max()
    assert np.allclose(norm, ref, atol=1e-6) and norm.max() == 1.0 and np.allclose(An, affSub, atol=1e-4)
