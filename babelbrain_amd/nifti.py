"""NIfTI-1 export of the result volumes: the last stage of Step 2 (`RUN_SIM_BASE.Step10_GetResults`, BASE:1459-1598) for a
pipeline built on this package alone.

The reference writes its volumes with nibabel (`nibabel.Nifti1Image(data, affine=...)`) and then rewrites them with
SimpleITK to force isotropic spacing (`SaveNiftiEnforcedISO`, BASE:737-776). Neither package is needed here: the NIfTI-1
single-file format is a 348-byte header + 4 bytes of extension flag + the voxels in x-fastest order, and the only thing the
second pass changes is `pixdim[1..3]`, which `save_nifti(..., iso=True)` sets to the rounded mean of the zooms itself.
`load_nifti` reads such files back (tests, and `resave_normalized`).

`step10_volumes` restates which volumes Step 10 writes, with which affine and which crop:
  * full volumes subsampled by `ss` with `affine[:3,:3] @ (I * ss)` (BASE:1472-1474, 1499-1503);
  * `_Sub` volumes = the bounding box of `MaskCalcRegions` with the origin moved to its first voxel (BASE:1484-1490,
    1510-1512 -- the reference's half-open `mx[0]:mx[-1]` slices, which drop the last plane of the box, are kept as they are);
  * `resave_normalized` = `ResaveNormalized` (BASE:778-832): `_Sub` volume divided by its maximum inside the mask's brain region.
File names follow `OutputFileNames` (BASE:1039-1067) through the `FILENAMES` dict the caller passes.
"""
import gzip
import struct

import numpy as np

_DTYPES = {np.dtype(np.uint8): (2, 8), np.dtype(np.int16): (4, 16), np.dtype(np.int32): (8, 32), np.dtype(np.float32): (16, 32),
           np.dtype(np.complex64): (32, 64), np.dtype(np.float64): (64, 64), np.dtype(np.int8): (256, 8), np.dtype(np.uint16): (512, 16),
           np.dtype(np.uint32): (768, 32)}
_CODES = {v[0]: k for k, v in _DTYPES.items()}


def _quaternion(R):
    """Rotation matrix (det +1) -> (b, c, d) of the unit quaternion with a >= 0 (NIfTI-1 standard, nifti1_io mat44_to_quatern)."""
    r11, r12, r13, r21, r22, r23, r31, r32, r33 = R.ravel()
    a = r11 + r22 + r33 + 1.0
    if a > 0.5:
        a = 0.5 * np.sqrt(a)
        b = 0.25 * (r32 - r23) / a; c = 0.25 * (r13 - r31) / a; d = 0.25 * (r21 - r12) / a
    else:
        xd, yd, zd = 1.0 + r11 - (r22 + r33), 1.0 + r22 - (r11 + r33), 1.0 + r33 - (r11 + r22)
        if xd > 1.0:
            b = 0.5 * np.sqrt(xd); c = 0.25 * (r12 + r21) / b; d = 0.25 * (r13 + r31) / b; a = 0.25 * (r32 - r23) / b
        elif yd > 1.0:
            c = 0.5 * np.sqrt(yd); b = 0.25 * (r12 + r21) / c; d = 0.25 * (r23 + r32) / c; a = 0.25 * (r13 - r31) / c
        else:
            d = 0.5 * np.sqrt(zd); b = 0.25 * (r13 + r31) / d; c = 0.25 * (r23 + r32) / d; a = 0.25 * (r21 - r12) / d
        if a < 0.0:
            b, c, d = -b, -c, -d
    return float(b), float(c), float(d)


def header_bytes(shape, dtype, affine, iso=False):
    """The 348-byte NIfTI-1 header of a 3-D volume with the given 4 x 4 voxel-to-world affine (mm)."""
    dtype = np.dtype(dtype)
    if dtype not in _DTYPES:
        raise ValueError('no NIfTI-1 datatype for %s' % dtype)
    code, bitpix = _DTYPES[dtype]
    A = np.asarray(affine, np.float64)
    zooms = np.sqrt((A[:3, :3] ** 2).sum(axis=0))
    if iso:           # SaveNiftiEnforcedISO: res = round(mean(zooms), 5) on all three axes (BASE:754-757)
        zooms = np.full(3, float(np.round(zooms.mean(), 5)))
    R = A[:3, :3] / np.where(zooms > 0, np.sqrt((A[:3, :3] ** 2).sum(axis=0)), 1.0)
    qfac = 1.0
    if np.linalg.det(R) < 0:
        R = R.copy(); R[:, 2] = -R[:, 2]; qfac = -1.0
    # nearest orthonormal matrix (polar decomposition), as nifti1_io does before taking the quaternion
    U, _, Vt = np.linalg.svd(R)
    b, c, d = _quaternion(U @ Vt)
    dim = [3, int(shape[0]), int(shape[1]), int(shape[2]), 1, 1, 1, 1]
    pixdim = [qfac, float(zooms[0]), float(zooms[1]), float(zooms[2]), 1.0, 1.0, 1.0, 1.0]
    h = bytearray(348)
    struct.pack_into('<i', h, 0, 348)
    struct.pack_into('<8h', h, 40, *dim)
    struct.pack_into('<h', h, 70, code)
    struct.pack_into('<h', h, 72, bitpix)
    struct.pack_into('<8f', h, 76, *pixdim)
    struct.pack_into('<f', h, 108, 352.0)            # vox_offset
    struct.pack_into('<f', h, 112, 1.0)              # scl_slope
    struct.pack_into('<f', h, 116, 0.0)              # scl_inter
    h[123] = 2 | 8                                   # xyzt_units: millimetres, seconds
    struct.pack_into('<h', h, 252, 2)                # qform_code: aligned
    struct.pack_into('<h', h, 254, 2)                # sform_code: aligned
    struct.pack_into('<3f', h, 256, b, c, d)
    struct.pack_into('<3f', h, 268, A[0, 3], A[1, 3], A[2, 3])
    S = A.copy()
    if iso:           # the rewritten file carries the isotropic spacing along the same directions
        S[:3, :3] = (A[:3, :3] / np.sqrt((A[:3, :3] ** 2).sum(axis=0))) * zooms
    struct.pack_into('<4f', h, 280, *S[0])
    struct.pack_into('<4f', h, 296, *S[1])
    struct.pack_into('<4f', h, 312, *S[2])
    h[344:348] = b'n+1\0'
    return bytes(h)


def save_nifti(path, data, affine, iso=False):
    """Writes `data` (3-D) as a single-file NIfTI-1 volume; `.gz` paths are gzip-compressed. Returns the path."""
    a = np.asarray(data)
    if a.ndim != 3:
        raise ValueError('save_nifti writes 3-D volumes')
    if a.dtype == np.bool_:
        a = a.astype(np.uint8)
    if a.dtype not in _DTYPES:
        a = a.astype(np.float32)
    blob = header_bytes(a.shape, a.dtype, affine, iso) + b'\0\0\0\0' + np.asfortranarray(a).tobytes(order='F')
    with (gzip.open(path, 'wb', compresslevel=1) if str(path).endswith('.gz') else open(path, 'wb')) as f:
        f.write(blob)
    return path


def load_nifti(path):
    """-> (data, affine, zooms) of a single-file NIfTI-1 volume written by save_nifti / nibabel (little endian, 3-D)."""
    with (gzip.open(path, 'rb') if str(path).endswith('.gz') else open(path, 'rb')) as f:
        raw = f.read()
    if struct.unpack_from('<i', raw, 0)[0] != 348 or raw[344:347] != b'n+1':
        raise ValueError('%s is not a little-endian single-file NIfTI-1 volume' % path)
    dim = struct.unpack_from('<8h', raw, 40)
    code = struct.unpack_from('<h', raw, 70)[0]
    pixdim = struct.unpack_from('<8f', raw, 76)
    off = int(struct.unpack_from('<f', raw, 108)[0])
    shape = tuple(dim[1:1 + dim[0]])
    dt = _CODES[code]
    data = np.frombuffer(raw, dt, int(np.prod(shape)), off).reshape(shape, order='F')
    A = np.eye(4)
    if struct.unpack_from('<h', raw, 254)[0] > 0:
        A[0] = struct.unpack_from('<4f', raw, 280); A[1] = struct.unpack_from('<4f', raw, 296); A[2] = struct.unpack_from('<4f', raw, 312)
    else:             # qform only
        b, c, d = struct.unpack_from('<3f', raw, 256)
        a = np.sqrt(max(1.0 - (b * b + c * c + d * d), 0.0))
        R = np.array([[a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
                      [2 * (b * c + a * d), a * a + c * c - b * b - d * d, 2 * (c * d - a * b)],
                      [2 * (b * d - a * c), 2 * (c * d + a * b), a * a + d * d - b * b - c * c]])
        z = np.array(pixdim[1:4]) * np.array([1.0, 1.0, -1.0 if pixdim[0] < 0 else 1.0])
        A[:3, :3] = R * z
        A[:3, 3] = struct.unpack_from('<3f', raw, 268)
    return data, A, np.array(pixdim[1:4])


def step10_volumes(affine_mask, MaskCalcRegions, FullSolutionPressure, FullSolutionPhase, RayleighWater=None, RayleighWaterOverlay=None,
                   FullSolutionPressureRefocus=None, FullSolutionPhaseRefocus=None, subsamplingFactor=1, bMinimalSaving=False):
    """-> {key of FILENAMES: (volume, affine)} exactly as Step10_GetResults assembles them (BASE:1472-1517; the water-only
    variant `bUseRayleighForWater` writes the same volumes under FILENAMESWater and is left to the caller)."""
    ss = int(subsamplingFactor)
    affine = np.array(affine_mask, np.float64)
    affineSub = affine.copy()
    affine[0:3, 0:3] = affine[0:3, 0:3] @ (np.eye(3) * ss)                                     # BASE:1474
    mx, my, mz = np.where(MaskCalcRegions)
    affineSub[0:3, 3] = (affineSub @ np.array([[mx[0], my[0], mz[0], 1]]).T)[0:3, 0]            # BASE:1485-1487
    mx, my, mz = np.unique(mx), np.unique(my), np.unique(mz)
    sub = (slice(mx[0], mx[-1]), slice(my[0], my[-1]), slice(mz[0], mz[-1]))                  # half-open, as BASE:1510
    out = {}
    if RayleighWaterOverlay is not None and not bMinimalSaving:
        out['RayleighFreeWaterWOverlay__'] = (RayleighWaterOverlay[::ss, ::ss, ::ss], affine)
    if RayleighWater is not None:
        out['RayleighFreeWater__'] = (RayleighWater[::ss, ::ss, ::ss], affine)
        out['RayleighFreeWater_Sub__'] = (RayleighWater[sub], affineSub)                       # BASE:1519-1521
    if FullSolutionPressureRefocus is not None:
        out['FullElasticSolutionRefocus__'] = (FullSolutionPressureRefocus[::ss, ::ss, ::ss], affine)
        out['FullElasticSolutionRefocusPhase__'] = (FullSolutionPhaseRefocus[::ss, ::ss, ::ss], affine)
        out['FullElasticSolutionRefocus_Sub__'] = (FullSolutionPressureRefocus[sub], affineSub)
    out['FullElasticSolution__'] = (FullSolutionPressure[::ss, ::ss, ::ss], affine)
    out['FullElasticSolutionPhase__'] = (FullSolutionPhase[::ss, ::ss, ::ss], affine)
    out['FullElasticSolution_Sub__'] = (FullSolutionPressure[sub], affineSub)
    return out


def resave_normalized(sub_path, mask_data, mask_affine):
    """`ResaveNormalized` (BASE:778-832): the `_Sub` volume divided by its maximum over the voxels whose mask value is >= 4
    (brain and target), zero elsewhere, written beside it as `_Sub_NORM`. Voxel correspondence through the two affines."""
    assert '_Sub.nii.gz' in sub_path
    data, A, _ = load_nifti(sub_path)
    data = np.asarray(data, np.float64)
    ii, jj, kk = np.mgrid[0:data.shape[0], 0:data.shape[1], 0:data.shape[2]]
    idx = np.c_[ii.ravel(), jj.ravel(), kk.ravel(), np.ones(kk.size)].T
    im = np.round(np.linalg.inv(mask_affine) @ (A @ idx)).astype(int)
    for ax in range(3):
        im[ax, im[ax] >= mask_data.shape[ax]] = mask_data.shape[ax] - 1
        im[ax, im[ax] < 0] = 0
    inside = (np.asarray(mask_data)[im[0], im[1], im[2]] >= 4).reshape(data.shape)
    out = np.where(inside, data, 0.0)
    if out.max() > 0:
        out = out / out.max()
    return save_nifti(sub_path.replace('_Sub.nii.gz', '_Sub_NORM.nii.gz'), out.astype(np.float32), A)


def save_step10(FILENAMES, volumes, mask_data=None, mask_affine=None):
    """Writes the volumes of step10_volumes under the reference's names: key `X__` of FILENAMES is the temporary name the
    reference gives nibabel; the isotropic file it keeps is the same path without the `__` (BASE:751-759). `_Sub` pressure
    volumes are also re-saved normalised when the mask is given (BASE:1497, 1512)."""
    written = []
    for key, (vol, aff) in volumes.items():
        tmp = FILENAMES[key] if key in FILENAMES else FILENAMES['RayleighFreeWater__'].replace('RayleighFreeWater', 'RayleighFreeWater_Sub')   # BASE:1521
        fn = tmp.split('__.nii.gz')[0] + '.nii.gz'
        written.append(save_nifti(fn, vol, aff, iso=True))
        if mask_data is not None and key in ('FullElasticSolution_Sub__', 'FullElasticSolutionRefocus_Sub__'):
            written.append(resave_normalized(fn, mask_data, mask_affine))
    return written
