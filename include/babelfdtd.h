/*
 * babelfdtd.h -- C ABI of the MI355X-native viscoelastic FDTD engine (libbabelfdtd_hip.so).
 *
 * This is the drop-in boundary for BabelBrain's Step-2 hot path. The reference reaches its
 * solver through two Python methods of a module-global object
 *     PModel = PropagationModel()                                   BabelIntegrationBASE.py:43
 *     PModel.CalculateMatricesForPropagation(...)                   BabelIntegrationBASE.py:1799,1801
 *     PModel.StaggeredFDTD_3D_with_relaxation(...)                  BabelIntegrationBASE.py:2338,2374,2401
 * whose implementation (package BabelViscoFDTD==1.2.4, environment_linux.yml:44) binds a
 * per-backend native module selected by the integer COMPUTING_BACKEND (BabelBrain.py:429-439,
 * SelFiles/SelFiles.py:245-262). The functions below are what such a backend module binds:
 * plain pointers and sizes, int return codes (0 = ok, <0 = error, text via bfd_last_error()),
 * no exceptions, no torch types. babelbrain_amd/PropagationModel.py is the ctypes binding.
 *
 * Conventions
 *  - Volumes handed in by the caller are described by a base pointer and three ELEMENT strides
 *    (s1,s2,s3) for the axes (i,j,k) of an (N1,N2,nk) view, so a C-order numpy array is passed
 *    without a host transpose: element (i,j,k) is base[i*s1 + j*s2 + k*s3]. The byte span
 *    base[0 .. (N1-1)*s1+(N2-1)*s2+(nk-1)*s3] must be readable/writable. Inputs are never modified.
 *  - On the device every volume is "x-fastest": linear index i + N1*(j + N2*k), the order the
 *    reference decodes IndexSensorMap with (BabelIntegrationBASE.py:2508-2511).
 *  - One bfd_sim owns the Z-slab [k0, k0+nk) of the global N1 x N2 x N3 domain (single GPU:
 *    k0=0, nk=N3). Slabs are advanced half-step by half-step; the two ghost planes on each side
 *    are exchanged by the caller between half-steps (bfd_halo_region gives device pointers), or
 *    are zero at the ends of the domain.
 *  - Selectable maps are a bitmask of BFD_MAP_* (names follow the reference's SelMapsRMSPeakList /
 *    SelMapsSensorsList strings, BabelIntegrationBASE.py:1413-1417, 2355-2356).
 */
#ifndef BABELFDTD_H
#define BABELFDTD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BFD_ABI_VERSION 7

enum {
    BFD_MAP_VX = 0, BFD_MAP_VY = 1, BFD_MAP_VZ = 2,
    BFD_MAP_SIGMAXX = 3, BFD_MAP_SIGMAYY = 4, BFD_MAP_SIGMAZZ = 5,
    BFD_MAP_SIGMAXY = 6, BFD_MAP_SIGMAXZ = 7, BFD_MAP_SIGMAYZ = 8,
    BFD_MAP_PRESSURE = 9, BFD_MAP_ALLV = 10, BFD_MAP_COUNT = 11
};

/* which accumulated volume bfd_get_map returns */
enum { BFD_KIND_RMS = 0, BFD_KIND_PEAK = 1, BFD_KIND_LAST = 2 };

/* halo groups exchanged between Z-neighbours */
enum { BFD_HALO_VELOCITY = 0 /* Vx,Vy,Vz: before the stress half-step   */,
       BFD_HALO_STRESS = 1   /* Sxz,Syz,Szz: before the velocity half-step */ };

typedef struct bfd_sim bfd_sim;

typedef struct bfd_config {
    int32_t N1, N2, N3;        /* global domain, voxels, absorbing layer included (BASE:1876-1878) */
    int32_t k0, nk;            /* this slab owns global planes [k0, k0+nk)                        */
    int32_t nMat;              /* rows of MaterialList                                            */
    int32_t NDelta;            /* absorbing-layer thickness in cells (NDelta=12, BASE:2350)       */
    int32_t typeSource;        /* TypeSource: 0 add velocity, 1 set velocity, 2 add stress, 3 set stress (BASE:2327,2332,2393) */
    int32_t sensorSub;         /* SensorSubSampling (BASE:2363)                                   */
    int32_t sensorStart;       /* SensorStart, in sub-sampled steps (BASE:2109,2364)              */
    int32_t nt;                /* total number of time steps of the run (sizes the sensor block)  */
    int32_t selRMSorPeak;      /* SelRMSorPeak: 1 RMS, 2 peak, 3 both (BASE:2357)                 */
    uint32_t selMapsRMS;       /* SelMapsRMSPeakList as BFD_MAP_* bits (BASE:2355)                */
    uint32_t selMapsSensors;   /* SelMapsSensorsList as BFD_MAP_* bits (BASE:2356)                */
    int32_t qfactorCorrection; /* QfactorCorrection (BASE:2361)                                   */
    int32_t device;            /* HIP device ordinal                                              */
    int32_t kernelVariant;     /* 0 = default (= 3), 1 = simple one-thread-per-voxel kernels, 2 = LDS-tiled dense, 3 = LDS-tiled with fluid/solid tile classes, 4 = 3 + fused stress/velocity pass over eligible fluid tiles (whole domains only: keeps two copies of V, Szz, Rzz; on a Z-slab it is 3) */
    int32_t rmsFirstStep;      /* 0 = RMS/peak accumulation starts at step SensorStart*SensorSubSampling (the reference's last-cycles
                                * window, BASE:2108-2109); n > 0 = it starts at step n-1 (bench: every timed step accumulates) */
    int32_t sensorMode;        /* 0 = the sensor series are stored ([nSensors][nTs] per selected map, what the reference returns);
                                * 1 = only their single-frequency content is kept: re/im of the DFT bin nearest to `freq` and the peak
                                * are accumulated per sensor while the samples are taken (what CalculatePhaseData extracts from the
                                * series, BASE:2498-2520) -- 20 B per sensor instead of 4*nTs; bfd_get_sensors is then unavailable */
    double h;                  /* SpatialStep, m (BASE:2344)                                      */
    double dt;                 /* DT, s (BASE:2351)                                               */
    double freq;               /* Frequency, Hz (BASE:2341)                                       */
    double reflectionLimit;    /* ReflectionLimit (BASE:2352)                                     */
} bfd_config;

/* ---- library / device ---- */
int bfd_abi_version(void);
const char *bfd_last_error(void);
int bfd_device_count(void);                                   /* replaces <backend>.ListDevices, SelFiles.py:245-262 */
int bfd_device_name(int device, char *buf, int buflen);

/* ---- CalculateMatricesForPropagation (BASE:1799,1801): stable dt and per-material tables ----
 * matlist: nMat x 5 float64 rows [rho, cL, cS, alphaL, alphaS] (BASE:1712-1729); qcorr: nMat
 * float64 QCorrection factors or NULL (BASE:1290). tables7 (may be NULL) receives 7*nMat
 * float32: AP,BP,AS2,BS2,invMu,tauS,invRho for the given dt. */
double bfd_stable_dt(int32_t nMat, const double *matlist, const double *qcorr, double freq,
                     int32_t qfactorCorrection, double h, double alphaCFL);
int bfd_material_tables(int32_t nMat, const double *matlist, const double *qcorr, double freq,
                        int32_t qfactorCorrection, double h, double dt, float *tables7,
                        float *c1k2, double *cmax);

/* ---- StaggeredFDTD_3D_with_relaxation (BASE:2338-2365), decomposed ---- */
int bfd_create(const bfd_config *cfg, bfd_sim **out);
void bfd_destroy(bfd_sim *sim);

/* launch every kernel of this sim on an existing HIP stream (hipStream_t as void*), e.g. torch's
 * current stream, so the caller's halo exchange orders against it; NULL = the device's default
 * (null) stream. A new sim runs on a private non-blocking stream; bfd_use_private_stream returns to one. */
int bfd_set_stream(bfd_sim *sim, void *hipStream);
int bfd_use_private_stream(bfd_sim *sim);

/* MaterialList + QCorrection (BASE:2340,2362) */
int bfd_set_materials(bfd_sim *sim, const double *matlist, const double *qcorr);
/* MaterialMap slab (BASE:2339). ghostLow/ghostHigh (0..2): planes readable below k=0 / above
 * k=nk-1 of the view (neighbour slab's cells); missing ghost planes replicate the edge plane. */
int bfd_set_material_map(bfd_sim *sim, const uint32_t *map, int64_t s1, int64_t s2, int64_t s3,
                         int32_t ghostLow, int32_t ghostHigh);
/* ReflectorMask slab (BASE:2365); NULL clears it */
int bfd_set_reflector(bfd_sim *sim, const uint32_t *mask, int64_t s1, int64_t s2, int64_t s3);
/* SourceMap/PulseSource/Ox,Oy,Oz (BASE:2342-2349) in compact form: nVox source voxels of this
 * slab, local x-fastest linear index, 0-based PulseSource row, per-voxel weights (NULL = 1),
 * pulse = [nSources][lengthSource] float64 exactly as the caller built it (Single:335-346).
 * A table whose float32 form exceeds 1 GiB (238 k sources x 6.8 k steps at 512^3: 6.4 GB; 1 M x 7 k at 1024^3: 28 GB) is
 * STREAMED: it stays in the caller's memory -- which must then remain valid and unchanged until the last time step has
 * been issued, as it does inside the solver call -- and reaches the device in double-buffered time tiles of 64 steps
 * (float64 -> float32 by a host packer that runs beside the GPU); the device holds 2 tiles instead of 12 bytes per
 * table entry. Smaller tables are converted once and kept resident. Results are identical either way. */
int bfd_set_sources(bfd_sim *sim, int64_t nVox, const uint32_t *localIndex, const uint32_t *row,
                    const float *wx, const float *wy, const float *wz,
                    const double *pulse, int32_t nSources, int32_t lengthSource);
/* SensorMap slab (BASE:2346); returns the number of sensors of this slab in *nSensors */
int bfd_set_sensor_map(bfd_sim *sim, const uint32_t *map, int64_t s1, int64_t s2, int64_t s3,
                       int64_t *nSensors);

/* time stepping. bfd_run = nSteps x (stress half-step, velocity half-step, accumulate, sensors)
 * for a slab without neighbours. With neighbours the caller alternates:
 *   exchange VELOCITY halos -> bfd_half_step_stress -> exchange STRESS halos -> bfd_half_step_velocity */
int bfd_run(bfd_sim *sim, int32_t nSteps);
int bfd_half_step_stress(bfd_sim *sim);
int bfd_half_step_velocity(bfd_sim *sim);   /* also accumulates, records sensors, advances the step counter */
/* the same half-steps in two parts, so a halo exchange can overlap the bulk of the work:
 * part 1 = the tiles of the first and last z-chunk (8-32 planes) of the slab (everything a Z-neighbour reads),
 * part 2 = all other tiles plus the end-of-step work; part 0 = both (the calls above). Per half-step call
 * part 1 then part 2. (kernelVariant 1: part 1 is empty.) */
int bfd_half_step_stress_part(bfd_sim *sim, int32_t part);
int bfd_half_step_velocity_part(bfd_sim *sim, int32_t part);
/* The same launched on a stream of the caller (NULL = the default stream) without any synchronisation: the boundary part
 * can run on a side stream that a halo exchange waits on while the interior part keeps the main stream busy. The caller
 * orders the two streams (events); parts of one half-step are independent of each other, the end-of-step work of
 * velocity part 2 (sensors, non-Pressure accumulators) reads the whole slab. */
int bfd_half_step_stress_part_on(bfd_sim *sim, int32_t part, void *hipStream);
int bfd_half_step_velocity_part_on(bfd_sim *sim, int32_t part, void *hipStream);
int bfd_sync(bfd_sim *sim);
int bfd_current_step(bfd_sim *sim);
/* Everything the first step would otherwise do on entry: per-cell classes, run lists, and the placement of the per-voxel
 * arrays: streams that advance together are 12 % slower when all of them lie in one of the three physical regions of the
 * HBM than when they are spread over two, so a pair probe on the (all-zero) initial state finds the
 * region of every array relative to Vx and the arrays a kernel reads together are made to alternate -- by exchanging
 * buffers, and with fresh allocations where one side is short. BFD_PLACEMENT=0 switches it off. Results do not depend on
 * it. bfd_run and the half-step calls do this by themselves at step 0; call it explicitly BEFORE bfd_halo_region when halo
 * pointers are taken ahead of the first step (pointers handed out pin the arrays). */
int bfd_prepare(bfd_sim *sim);
/* one line on what the placement found and did (valid until the sim is destroyed) */
const char *bfd_placement_note(bfd_sim *sim);
/* Placement policy, before the first step / bfd_prepare. mode 0 = leave the arrays where hipMalloc put them. searchLimitBytes =
 * how much throw-away device memory the search for a buffer in another region may hold at a time (all of it is released
 * before bfd_prepare returns); < 0 = the default rule: nothing at all when the device carries other allocations than this
 * engine's (another process, the other slabs of a group): the engine's own buffers are then only exchanged among themselves;
 * on a device the engine has to itself up to 192 GiB, at most two thirds of what was free on entry and always leaving 48 GiB of it
 * untouched; the search ends at once when allocations that are neither the engine's nor its own appear on the device while it walks, and
 * after 2 s (BABELFDTD_PLACEMENT_SEARCH_SECONDS): the placement is worth a few per cent of one call's run time.
 * bfd_prepare never fails for lack of memory where mode 0 succeeds. */
int bfd_set_placement(bfd_sim *sim, int32_t mode, int64_t searchLimitBytes);
/* The default rule can be moved without code by whoever owns the device: BABELFDTD_PLACEMENT_SEARCH_GIB=<GiB> replaces the 192 GiB
 * (the shared-device rule stays). Buffers a search found in another region are kept when their engine is destroyed and
 * offered to the next engine of this process with arrays of the same size on the same device, so that the two or three solver calls of
 * one RUN_SIMULATION (BabelIntegrationBASE.py:2338, 2374, 2401) pay the search once; at most BABELFDTD_PLACEMENT_CACHE_GIB (default
 * 48, never more than an eighth of the device's memory, 0 = keep nothing) are held between engines; bfd_create evicts buffers of another
 * array size, and any allocation of the library that runs out of memory frees the cache and tries once more.
 * bfd_placement_cache_release frees them now and returns the bytes freed (the Python drop-in calls it when a solver call returns,
 * unless it was built with keepPlacementCache=True / BABELFDTD_PLACEMENT_CACHE_KEEP=1). It also gives back the pinned host pieces
 * (16 MB each, at most 32) the result readbacks keep between them; those are not part of the count. */
int64_t bfd_placement_cache_release(void);

/* device pointer/bytes of a halo region: field f (0..2 within the group), side 0 = low-k face,
 * 1 = high-k face; send = 1: the 2 owned boundary planes, send = 0: the 2 ghost planes.
 * Each region is 2*N1*N2 contiguous float32. */
int bfd_halo_region(bfd_sim *sim, int32_t group, int32_t f, int32_t side, int32_t send,
                    void **devPtr, size_t *bytes);

/* which fields of a halo group this slab READS from its Z-neighbours' planes, as a bit mask (bit f = field f of
 * bfd_halo_region): 7 in general; 4 (Vz / Szz only) for a slab without solid runs under the tiled kernels, whose
 * stencils cross a Z face through those two fields alone. Across an interface each side sends what the other reads. */
int bfd_halo_fields(bfd_sim *sim, int32_t group, uint32_t *mask);

/* timing of the step loop, device time from HIP events on the sim's stream */
int bfd_timing_begin(bfd_sim *sim, int32_t perKernel);
int bfd_timing_end(bfd_sim *sim, double *totalMs, double *stressMs, double *velocityMs,
                   double *otherMs, int64_t *nStressLaunches, int64_t *nVelocityLaunches);

/* Finer timing (bfd_timing_begin(sim, 2)): device ms and launch count per kernel class over the timed window, classes
 * 0 stress fluid tiles, 1 normal stresses of solid tiles, 2 sparse shear stresses, 3 velocity fluid tiles,
 * 4 velocity solid tiles, 5 fused fluid time step. Call after bfd_timing_end. Arrays of 6. */
int bfd_timing_kernels(bfd_sim *sim, double *msPerClass, int64_t *launchesPerClass);
/* ALGORITHMIC bytes one launch of each kernel class has to move (same class order), from the tile-class counts: every
 * field value of the class's per-cell byte table fetched / stored exactly once (float32 4 B, material id 2 B; absorbing-
 * layer memory variables, coefficient tables and halo re-reads excluded). accumulating != 0 adds the
 * Pressure RMS accumulator (8 B per cell outside the absorbing layer) to the velocity classes. DESIGN.md section 6. */
int bfd_algorithmic_bytes(bfd_sim *sim, int32_t accumulating, double *bytesPerClass);
/* back to step 0: fields, absorbing-layer memory, accumulators and the sensor block zeroed; inputs are kept */
int bfd_reset(bfd_sim *sim);

/* results */
int64_t bfd_num_sensors(bfd_sim *sim);
int32_t bfd_num_sensor_steps(bfd_sim *sim);
/* 1-based GLOBAL x-fastest linear index of every sensor of this slab, ascending (BASE:2369,2503) */
int bfd_get_sensor_index(bfd_sim *sim, uint32_t *index);
/* out[nSelSensors][nSensors][nTs] float32, maps in ascending BFD_MAP_* order (BASE:2507: FFT along axis 1). Blocks of 256 MB and
 * more (this one, the maps below) are carried by four host threads of the library through pinned pieces (BFD_D2H_THREADS; the
 * call returns when all of it has arrived); the destination is advised to use huge pages. */
int bfd_get_sensors(bfd_sim *sim, float *out);
/* Single-frequency content of the recorded sensor series, computed on the device: replaces the host
 * FFT + bin pick of CalculatePhaseData (BASE:2498-2520) without moving the (nSensor x nTs) block.
 *   F[q][s] = (2/nTs) sum_n x[q][s][n] exp(-2 pi i bin n / nTs),  bin = argmin |fftfreq(nTs, DT*SensorSubSampling) - freq|
 *   peak[q][s] = max_n x[q][s][n]                                                   (BASE:2518)
 * outReIm: [nSelSensors][nSensors][2] float32, outPeak (may be NULL): [nSelSensors][nSensors].
 * With sensorMode 1 the values come from the in-loop accumulators (same arithmetic, sample by sample; freq must be the
 * sim's own frequency). */
int bfd_get_sensor_dft(bfd_sim *sim, double freq, float *outReIm, float *outPeak);
/* the same transform for a host series [nSensors][nTs] (row-major), sampling period dtSensor */
int bfd_dft_series(int32_t device, int64_t nSensors, int32_t nTs, const float *series, double dtSensor, double freq,
                   float *outReIm, float *outPeak);
/* one accumulated volume of this slab into a strided (N1,N2,nk) float32 view */
int bfd_get_map(bfd_sim *sim, int32_t kind, int32_t map, float *out, int64_t s1, int64_t s2, int64_t s3);
/* raw state array a (0..14: Vx Vy Vz Sxx Syy Szz Sxy Sxz Syz Rxx Ryy Rzz Rxy Rxz Ryz), for tests */
int bfd_get_field(bfd_sim *sim, int32_t a, float *out, int64_t s1, int64_t s2, int64_t s3);
/* number of 64x8x8-voxel sub-tiles per class of the class-specialised kernels (variant 0/3):
 * lossless fluid, lossy fluid, solid, and among the fluid ones how many hold one material only (UNI)
 * and how many touch the absorbing layer (PML) (DESIGN.md "Tile classes"); zeros for variants 1, 2 */
int bfd_tile_counts(bfd_sim *sim, int32_t *nLossless, int32_t *nLossy, int32_t *nSolid, int32_t *nUni, int32_t *nPml);
/* fluid sub-tiles that keep a single copy of their three identical normal stresses: all of them in an all-fluid
 * slab, those without a solid sub-tile beside them in x or y otherwise; 0 when a Sigma** output is selected */
int bfd_tile_count_lean(bfd_sim *sim, int32_t *nLean);
/* fluid sub-tiles advanced by the fused time-step kernel (kernelVariant 4 on a whole domain; 0 otherwise) */
int bfd_tile_count_fused(bfd_sim *sim, int32_t *nFused);
/* Quiet runs (ABI 7). A production call of a whole domain -- the caller's own accumulation window (rmsFirstStep = 0), class-specialised
 * kernels, solid-only values compact; BFD_SKIP_ZERO=0 switches it off -- keeps one activity byte per 64x8x8-voxel sub-tile: set once a kernel
 * has written a non-zero velocity or stress there (the sub-tiles of the source voxels from the start). Ahead of the wave front every field
 * is exactly zero (float32, denormals flushed), and a tile run whose sub-tiles and all their neighbours are clear returns at entry: results
 * are bit-identical, the steps before the front has crossed the domain cost less (the time plan BabelIntegrationBASE.py:2082-2109 gives a
 * call about 1.8 transits of the domain's diagonal). *active = sub-tiles marked so far, *total = sub-tiles of the domain; both 0 when the
 * engine runs every tile in every half-step (Z-slabs, bench.py's timed windows, the other kernel variants). */
int bfd_activity_counts(bfd_sim *sim, int64_t *active, int64_t *total);
/* device memory this sim holds, bytes */
int64_t bfd_device_bytes(bfd_sim *sim);

/* ---- One solver call on several devices of THIS process: Z-slab decomposition behind the drop-in ----
 * The reference's caller is one process making one call (Babel_SingleTx.py:258 spawns a single Process;
 * BabelIntegrationBASE.py:2338-2365 calls PModel.StaggeredFDTD_3D_with_relaxation once, device chosen by
 * DefaultGPUDeviceName :2358). A group owns one slab engine per entry of `devices` (slab r = balanced r-th share of the
 * N3 planes, on HIP device devices[r]; an ordinal may repeat), takes the WHOLE-domain inputs exactly as bfd_set_* take a
 * slab's, runs the step loop in C and moves the 2+2 halo planes per half-step and interface with peer copies ordered by
 * events (hipMemcpyPeerAsync; boundary runs first, interior beside the copies). No second process, no collective.
 * cfg: k0 = 0, nk = N3; cfg->device is ignored. Results equal the single-device run bit for bit. */
typedef struct bfd_group bfd_group;
int bfd_group_create(const bfd_config *cfg, int32_t nSlabs, const int32_t *devices, bfd_group **out);
void bfd_group_destroy(bfd_group *g);
int32_t bfd_group_size(bfd_group *g);
/* slab r: its planes [k0, k0+nk), device and engine (for the per-slab queries above: tile counts, timing, fields) */
int bfd_group_slab(bfd_group *g, int32_t r, int32_t *k0, int32_t *nk, int32_t *device, bfd_sim **sim);
int bfd_group_set_materials(bfd_group *g, const double *matlist, const double *qcorr);
/* whole-domain (N1,N2,N3) views with element strides, as bfd_set_material_map / _reflector / _sensor_map */
int bfd_group_set_material_map(bfd_group *g, const uint32_t *map, int64_t s1, int64_t s2, int64_t s3);
int bfd_group_set_reflector(bfd_group *g, const uint32_t *mask, int64_t s1, int64_t s2, int64_t s3);
/* as bfd_set_sources with GLOBAL x-fastest voxel indices i + N1*(j + N2*k); the pulse table is shared by the slabs */
int bfd_group_set_sources(bfd_group *g, int64_t nVox, const int64_t *globalIndex, const uint32_t *row,
                          const float *wx, const float *wy, const float *wz,
                          const double *pulse, int32_t nSources, int32_t lengthSource);
int bfd_group_set_sensor_map(bfd_group *g, const uint32_t *map, int64_t s1, int64_t s2, int64_t s3, int64_t *nSensors);
int bfd_group_set_placement(bfd_group *g, int32_t mode, int64_t searchLimitBytes);   /* bfd_set_placement of every slab */
int bfd_group_prepare(bfd_group *g);                     /* bfd_prepare of every slab + the halo plan; bfd_group_run does it by itself */
int bfd_group_run(bfd_group *g, int32_t nSteps);         /* queues nSteps time steps on every slab; returns without waiting */
int bfd_group_sync(bfd_group *g);
int bfd_group_reset(bfd_group *g);
/* wall time of the window, the largest device time of a slab (HIP events on its main stream), the host time spent
 * queueing work inside bfd_group_run (exposed launch overhead when it approaches the wall time), halo bytes moved per
 * time step over all interfaces, and whether the overlapped order was used (slabs of >= 64 planes) */
int bfd_group_timing_begin(bfd_group *g);
int bfd_group_timing_end(bfd_group *g, double *wallMs, double *maxDeviceMs, double *hostIssueMs, double *haloBytesPerStep,
                         int32_t *overlapped);
/* How the halo planes travel across each interface (slab r | slab r+1), decided in bfd_group_create -- the reference has no counterpart
 * (one device per call, BabelIntegrationBASE.py:2358); here a silently missing peer path (IOMMU, HIP_VISIBLE_DEVICES, topology) would
 * only show as a bad scaling curve. status[r] & 15 = BFD_PEER_SAME_DEVICE (both slabs on one device: a device copy), BFD_PEER_DIRECT
 * (peer access enabled both ways: hipMemcpyPeerAsync moves the planes device to device) or BFD_PEER_STAGED (not available in at least
 * one direction: the runtime stages the copies through the host). Detail bits: 16 / 32 = can access / enabled from slab r's device to
 * slab r+1's, 64 / 128 = the other way. Returns the number of interfaces (nSlabs - 1); status needs room for that many. */
#define BFD_PEER_SAME_DEVICE 0
#define BFD_PEER_DIRECT 1
#define BFD_PEER_STAGED 2
int bfd_group_peer_status(bfd_group *g, int32_t *status, int32_t n);
/* results over the whole domain: sensors of all slabs in ascending GLOBAL index (BASE:2369, 2503), volumes into a
 * strided (N1,N2,N3) view */
int64_t bfd_group_num_sensors(bfd_group *g);
int32_t bfd_group_num_sensor_steps(bfd_group *g);
int bfd_group_get_sensor_index(bfd_group *g, uint32_t *index);
int bfd_group_get_sensors(bfd_group *g, float *out);                  /* [nSelSensors][nSensors][nTs] */
int bfd_group_get_sensor_dft(bfd_group *g, double freq, float *outReIm, float *outPeak);
int bfd_group_get_map(bfd_group *g, int32_t kind, int32_t map, float *out, int64_t s1, int64_t s2, int64_t s3);
int64_t bfd_group_device_bytes(bfd_group *g);

/* ---- Rayleigh-Sommerfeld integral: replaces BabelViscoFDTD.tools.RayleighAndBHTE.ForwardSimple ----
 * (call sites BabelIntegrationSingle.py:295, BabelIntegrationANNULAR_ARRAY.py:383,411,
 * BabelIntegrationCONCAVE_PHASEDARRAY.py:307,328,425,446).
 *   out[n] = (i k / 2pi) sum_m u0[m] ds[m] exp(-i k R_nm) / R_nm,  k = kReal + i kImag
 * center: nSrc x 3 float32 (m), ds: nSrc float32 (m^2), u0: nSrc x 2 float32 (re,im),
 * rf: nPts x 3 float32 (m), out: nPts x 2 float32 (re,im). kernelMs (may be NULL): device time. */
int bfd_rayleigh_forward(int32_t device, int64_t nSrc, const float *center, const float *ds, const float *u0,
                         double kReal, double kImag, int64_t nPts, const float *rf, float *out, double *kernelMs);

/* ---- Pennes bio-heat equation + CEM43 dose: replaces BabelViscoFDTD.tools.RayleighAndBHTE.BHTE ----
 * (call sites ThermalModeling/CalculateTemperatureEffects.py:365-456, 960). Volumes x-fastest float32, mat = uint8
 * ids into cd/cp (per material: dt k/(rho c dx^2) and dt rho_b c_b w/(6e7 c)); q = temperature increment of one ON
 * step; T and dose are updated in place; faces of the volume keep their temperature. monitorSlice (may be NULL):
 * [N1][N3][ceil(nSteps/nFactorMonitoring)] of plane j = sliceJ; points (may be NULL): [nPoints][nSteps]. */
int bfd_bhte_run(int32_t device, int32_t N1, int32_t N2, int32_t N3, int32_t nMat, const unsigned char *mat,
                 const float *cd, const float *cp, const float *q, float *T, float *dose, float Tcore, double dt,
                 int32_t nSteps, int32_t nStepsOn, int32_t sliceJ, int32_t nFactorMonitoring, float *monitorSlice,
                 int64_t nPoints, const uint32_t *pointIndex, float *points, double *kernelMs);

/* Several pressure fields heating in turn: replaces BabelViscoFDTD.tools.RayleighAndBHTE.BHTEMultiplePressureFields
 * (call sites ThermalModeling/CalculateTemperatureEffects.py:381, 978; schedule built at :715-736). q holds nFields
 * volumes back to back; fieldOfStep[s] in [-1, nFields) names the field that heats during step s (-1 = none). */
int bfd_bhte_run_fields(int32_t device, int32_t N1, int32_t N2, int32_t N3, int32_t nMat, const unsigned char *mat,
                        const float *cd, const float *cp, int32_t nFields, const float *q, float *T, float *dose,
                        float Tcore, double dt, int32_t nSteps, const int32_t *fieldOfStep, int32_t sliceJ,
                        int32_t nFactorMonitoring, float *monitorSlice, int64_t nPoints, const uint32_t *pointIndex,
                        float *points, double *kernelMs);

/* The same run on volumes in the CALLER'S numpy C order, [N1][N2][N3] with the LAST axis fastest (what BHTE() receives at
 * CalculateTemperatureEffects.py:960: no host transposes), and with the heat source computed on the device from the
 * pressure amplitude(s): q = (p p) qf[material], float32; `pressure` holds nFields volumes (Pa), qOut (may be NULL) receives
 * q (the reference returns it as Qarr). The six neighbours are summed axis 0 first. flags bit 0: T holds the initial
 * temperature (else T starts from initT[material] and is output only); bit 1: dose holds the initial dose (else zero).
 * monitorSlice (may be NULL): [N1][N3][ceil(nSteps/nFactorMonitoring)] = T[:, sliceJ, :]; pointIndex: C-order linear
 * indices (i N2 + j) N3 + k. Steps are taken two per launch (one kernel pass moves the 21 B per voxel of a step once for
 * both); the monitors of the intermediate step are computed at the monitored voxels. */
int bfd_bhte_run_volumes(int32_t device, int32_t N1, int32_t N2, int32_t N3, int32_t nMat, const unsigned char *mat,
                         const float *cd, const float *cp, const float *qf, const float *initT, int32_t nFields,
                         const float *pressure, float *qOut, float *T, float *dose, int32_t flags, float Tcore, double dt,
                         int32_t nSteps, const int32_t *fieldOfStep, int32_t sliceJ, int32_t nFactorMonitoring,
                         float *monitorSlice, int64_t nPoints, const uint32_t *pointIndex, float *points, double *kernelMs);

#ifdef __cplusplus
}
#endif
#endif /* BABELFDTD_H */
