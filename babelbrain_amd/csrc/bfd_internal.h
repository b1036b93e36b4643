// Internal structures of libbabelfdtd_hip.so (gfx950 only). Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <future>
#include <string>
#include <vector>
#include "../../include/babelfdtd.h"

#define BFD_REFLECTOR_BIT 0x8000u   // bit 15 of the device material id marks a reflector voxel
#define BFD_MAT_MASK 0x7FFFu
// class byte of a cell (bfd_dev::cls)
#define BFD_CLS_FLUID 1u    // fluid centre (cS = 0), no reflector: Sxx == Syy == Szz, only Szz/Rzz is kept
#define BFD_CLS_NOMEM 2u    // BP == 0 and BS2 == 0: the normal stresses have no memory variable
#define BFD_CLS_EXY 4u      // the xy / xz / yz shear edge of this cell is updated (centre and the 3 other cells solid)
#define BFD_CLS_EXZ 8u
#define BFD_CLS_EYZ 16u
#define BFD_CLS_REFL 32u    // reflector voxel
#define BFD_CLS_MIXED 64u   // an updated shear edge of this cell lies between different materials (coefficients are not the per-material ones)

// CA, CB of the O(4) staggered first derivative (Taylor coefficients) and the matching stability constant
// 1/(CA+CB) of dt <= BFD_STAB h / (sqrt(3) cmax). Overridable for scheme experiments (tests/rayleigh_study.py):
// make TAG=holberg EXTRA='-DBFD_CA=1.1382f -DBFD_CB=0.046414f -DBFD_STAB=(1.0/1.184614)'
#ifndef BFD_CA
#define BFD_CA 1.125f
#define BFD_CB (1.0f / 24.0f)
#endif
#ifndef BFD_STAB
#define BFD_STAB (6.0 / 7.0)
#endif

// tile geometry of the class-specialised kernels (bfd_kernels_v2.hip): 64 x 8 cells per workgroup plane,
// classification in sub-tiles of BFD_SUBZ planes
#define BFD_TILE_X 64
#define BFD_TILE_Y 8
#ifndef BFD_SUBZ
#define BFD_SUBZ 8
#endif
// output rows of a workgroup of the fused time step (bfd_kernels_fused.hip): three tiles of the classification grid
#ifndef BFD_FUSED_ROWS
#define BFD_FUSED_ROWS 24
#endif
// multi-material runs of the fused time step keep AP, BP, 1/rho of every material in LDS: media with more materials fuse only their one-material runs
#define BFD_FUSED_MAX_MATERIALS 1024

// device-side view of one slab; passed by value to kernels
struct bfd_dev {
    int N1, N2, N3;        // global dims
    int k0, nk;            // slab
    int ND, P;             // layer thickness, zone width P = ND+1
    int plane;             // N1*N2
    // state, each (nk+4) planes; pointer addresses local plane 0 (ghost planes at -2,-1,nk,nk+1)
    float *Vx, *Vy, *Vz, *Sxx, *Syy, *Szz, *Sxy, *Sxz, *Syz, *Rxx, *Ryy, *Rzz, *Rxy, *Rxz, *Ryz;
    const uint16_t *mat;   // same ghosting; bit 15 = reflector
    // per-cell class byte (same ghosting), computed at setup from ids, tables and the reflector mask (BFD_CLS_*).
    // Invariant of the tiled kernels (variants 0/3/4): at a cell with BFD_CLS_FLUID only Szz/Rzz of the three identical
    // normal stresses (and memory variables) is maintained -- every reader takes Szz there; a shear stress entry is
    // read only where its edge bit is set (elsewhere it is never updated and stays exactly 0).
    const uint8_t *cls;
    // per-material tables
    const float *AP, *BP, *AS2, *BS2, *invMu, *tauS, *invRho;
    float c1, k2;
    // CPML profiles: [axis][aI,bI,aH,bH]
    const float *axI, *bxI, *axH, *bxH, *ayI, *byI, *ayH, *byH, *azI, *bzI, *azH, *bzH;
    // CPML memory variables (compact zone storage)
    //   x zones: [nk][N2][2P]      y zones: [nk][2P][N1]      z zones: [2P][N2][N1] (global z zone index)
    // stress half-step: 0 dxVx 1 dyVy 2 dzVz 3 dyVx 4 dxVy 5 dzVx 6 dxVz 7 dzVy 8 dyVz
    // velocity half-step: 9 dxSxx 10 dySxy 11 dzSxz 12 dxSxy 13 dySyy 14 dzSyz 15 dxSxz 16 dySyz 17 dzSzz
    float *psi[18];
    int tilesX, tilesY;
    // write targets of the fields that variant 4 keeps in two copies (V, Szz, Rzz); equal to the read pointers in
    // every other variant (in-place update). Kernels read d.X and write d.XW.
    float *VxW, *VyW, *VzW, *SzzW, *RzzW;
    // Compact solid state (round 5; bfd_tiles::css). cssRow != null: the values that exist only at solid cells -- Sxx, Syy, the three
    // shear stresses and the memory variables Rxx, Ryy -- live in the order of the sparse shear list ("listed" cells: solid centre,
    // no reflector) instead of the full-volume arrays, which are then not maintained. cssRow[((kl + 2) * N2 + j) * cssStride + bx] =
    // list index of the first listed cell of row (kl, j) with i >= 64 bx (bx = tilesX: one past the row's last entry); a row of the
    // list is contiguous in x, so the entry of cell i is that base + the number of listed cells of the row in [64 bx, i).
    // 0xFFFFFFFF marks planes without compact values (ghost planes: every value there is 0).
    const unsigned *cssRow; int cssStride;
    float *cSxx, *cSyy, *cSxy, *cSxz, *cSyz, *cRxx, *cRyy, *cRxy, *cRxz, *cRyz;
    // Activity map (round 6; null = every run works in every half-step): one byte per 64 x 8 x BFD_SUBZ sub-tile, 1 = a non-zero V or S value has
    // been written there. Padded by one sub-tile on every side: sub-tile (bx, by, q) at ((q + 1) * actY + by + 1) * actX + bx + 1. A run whose
    // sub-tiles and all their neighbours are clear returns at entry (bfd_kernels_v2.hip, "QUIET runs"). Production calls only (whole domains and Z-slabs).
    unsigned char *act; int actX, actY;
    int actLo, actHi;      // Z-slabs: runs that touch local planes below actLo or from actHi on (the sub-tiles a neighbour's planes reach) always work; whole domains: 0, INT_MAX
};
#define BFD_CSS_NONE 0xFFFFFFFFu
// a cell has compact values ("listed") when its class byte says solid centre, no reflector
static __device__ __forceinline__ bool css_listed(unsigned c) { return (c & (BFD_CLS_FLUID | BFD_CLS_REFL)) == 0u; }
// number of lanes below this one whose predicate is set (all lanes of the wave must get here)
static __device__ __forceinline__ unsigned css_rank(bool listed)
{
    const unsigned long long b = __ballot(listed);
    return __builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, 0u));
}
// list index of owned cell c (local linear index) or -1 when it has no compact values; for the readers outside the marching kernels
// (sensors, maps, sources): walks the class bytes of the row from the tile edge
static __device__ __forceinline__ long css_index(const bfd_dev &d, long c)
{
    if (!css_listed(d.cls[c])) return -1;
    const int kl = (int)(c / d.plane);
    const int r = (int)(c - (long)kl * d.plane);
    const int j = r / d.N1, i = r - j * d.N1;
    const unsigned rb = d.cssRow[((long)(kl + 2) * d.N2 + j) * d.cssStride + (i >> 6)];
    if (rb == BFD_CSS_NONE) return -1;
    unsigned n = 0;
    for (long q = c - (i & 63); q < c; q++) n += css_listed(d.cls[q]) ? 1u : 0u;
    return (long)rb + n;
}

// tile lists of the class-specialised path (variant 3): device array
// runs [fluid boundary | fluid interior | solid boundary | solid interior] (boundary = inside the first/last
// 32 planes); run = (bx + tilesX*by, kbeg | kend<<16, flags: bit0 solid, bit1 lossy, bit2 UNI, bit3 PML, bit4 LEAN, material
// id of UNI runs). n* counters after nSolidB are in 64x8x8 sub-tiles, for reporting.
// kernel classes of the per-kernel timing / byte accounting (bfd_timing_kernels, bfd_algorithmic_bytes)
enum { BFD_K_STRESS_FLUID = 0, BFD_K_STRESS_SOLID = 1, BFD_K_STRESS_SHEAR = 2, BFD_K_VELOCITY_FLUID = 3, BFD_K_VELOCITY_SOLID = 4,
       BFD_K_FUSED = 5, BFD_K_COUNT = 6 };
struct bfd_sim;
// records an event before (end = 0) / after (end = 1) a launch of class cls (bfd_api.hip); t->ktimer != null while class timing is on
void bfd_kmark(bfd_sim *sim, int cls, int end, hipStream_t st);

// maps of bfd_tiles::xmap: fluid stress / fluid velocity / solid stress / solid velocity (plain runs) by part 0, 1, 2; the two pieces of solid
// runs in the absorbing layer; the fused runs
enum { BFD_XM_SF = 0, BFD_XM_VF = 3, BFD_XM_SS = 6, BFD_XM_VS = 9, BFD_XM_VSP_LO = 12, BFD_XM_VSP_HI = 13, BFD_XM_FUSED = 14, BFD_XMAP_COUNT = 15 };
struct bfd_tiles { bfd_sim *ktimer; int nMat; bool merged /* solid runs: normal and shear stresses in one kernel, the sparse list holds the MIXED cells only */; int4 *runs;
                   unsigned *shearCells; float *shearCoef; long nShear, shearLowEnd, shearHighBeg;   /* sparse shear list */
                   unsigned *shearCodes; float *shearTab; long nShearExplicit;   /* per listed cell a byte per edge: 0 inactive, 1 + m = one material around the edge (coefficients from shearTab[8 m], [8 m + 1]; the cell's own AP, BP, AS2, BS2 at [8 m + 4 ..]), 255 = explicit coefficients in shearCoef; number of explicit edges */
                   int4 *runsAll; int nAll, nAllB;   /* compact solid state: every run, fluid and solid, in list order [boundary | interior] -- the stress half-step's one launch of the fluid kernel */
                   /* compact solid state (bfd_dev::cssRow): the row table and where the ten compact arrays (Sxx Syy Sxy Sxz Syz Rxx Ryy Rxy Rxz Ryz, list order) live:
                      cssHosted = inside the full-volume buffers of their own fields (unused otherwise in this mode; the placement has spread those over the memory
                      regions, which the sparse kernel's ten streams need as much as the marching kernels' do: 0.253 against 0.266-0.29 ms), from allocation plane 4 on
                      (the planes a Z-neighbour exchanges stay free); else css = one block [10][cssCap] (solid cells too many for that) */
                   unsigned *cssRow; float *css; long cssCap; bool cssHosted;
                   float *shearR;   /* memory variables Rxy, Rxz, Ryz of the listed cells, [3][nShear] in list order: only the sparse kernel uses them, so they live beside the list (dense, coalesced) instead of in the full-volume arrays, which are filled from here on demand (bfd_get_field) */
                   int nFluid, nFluidB, nSolid, nSolidB, nSolidBP /* leading boundary runs that touch the absorbing layer */, nSolidIP /* trailing interior ones */, nFused /* runs of the fused kernel, after the solid runs */;
                   int nLossless, nLossy, nSolidSub, nUni, nPml, nLean, nFusedSub;
                   /* cost-balanced block -> run maps (round 4): block b of a launch runs on XCD slot b & 7 and takes run seg[slot] + (b >> 3) of
                      the launched range if that is below seg[slot + 1]; the launch has 8 x maxcnt blocks. One map of 10 ints (seg[0..8], maxcnt)
                      per launched range and kernel class, on the device in xmap, on the host in xmapH. */
                   int *xmap; int xmapH[BFD_XMAP_COUNT][10];
                   /* experiment (BFD_CONCURRENT=1): the solid-run kernels of a half-step on side streams beside the fluid kernel (they write
                      disjoint cells); fork / join through events. Null = everything on the engine's stream. */
                   hipStream_t sideStream[2]; hipEvent_t sideFork, sideJoin[2]; };

struct bfd_sim {
    bfd_config cfg;
    bfd_dev d;
    hipStream_t stream;
    bool ownStream;
    int step;
    size_t nloc, nalloc;            // owned voxels, allocated voxels per state array
    float *stateBase[15];           // allocation bases
    uint16_t *matBase;
    uint8_t *clsBase; bool classesReady;
    bool placementDone, haloHandedOut;   // bfd_prepare: the per-voxel arrays may be moved until a halo pointer has been given out
    std::string placementNote;           // what choose_placement found and did (bfd_placement_note)
    std::vector<void *> searched;        // state buffers that a search found in another memory region: kept for the next engine of this process when this one is destroyed
    int placementMode; int64_t placementLimit;   // bfd_set_placement: 0 = off; bytes the search may hold at a time (-1 = default rule)
    float *tables;                  // 7*nMat
    float *profiles;                // 4*(N1+N2+N3)
    std::vector<void *> allocs;     // everything to free
    int64_t devBytes;
    bool haveMaterials, haveMap;
    bfd_tiles tiles; bool tilesReady;   // variants 2, 3, 4
    bool pingpong;                      // variant 4 on a whole domain: second copies of V, Szz, Rzz (ppBase), swapped every step
    float *ppBase[5];
    int zchunk;                         // planes of the longest z-run (8, 16 or 32), chosen per grid
    double cmax;
    // sources
    int64_t nSrcVox, srcLowEnd, srcHighBeg;   // sources sorted by voxel: [0,lowEnd) first z-chunk, [highBeg,n) last z-chunk
    uint32_t *srcLin, *srcRow; float *srcW[3]; float *pulseT; int nSources, lengthSource;
    // streamed source table (large PulseSource): the caller's float64 table stays on the host, time tiles of
    // tileSteps steps are converted to float32 [step][source] by a packer job and uploaded double-buffered
    const double *pulseHost; int tileSteps, nTiles;
    float *tileDev[2], *tilePinned[2]; int tileLoaded[2], tilePacked[2];
    hipEvent_t evTile[2]; bool evTileUsed[2];
    hipEvent_t evRead[2][2]; bool evReadUsed[2][2];      // [buffer][0 = engine stream, 1 = a side stream]: behind the last kernels that read the tile the buffer holds
    std::future<void> packJob[2];
    // sensors
    unsigned char *actBase; size_t actBytes; bool actReady;   // storage of bfd_dev::act; actReady = the map matches the state (cleared by setters and bfd_reset)
    bool sensIsBox; int sensBox[5];    // the sensor set is a dense box of voxels: extents in x, y and its first voxel (i0, j0, local k0); captures then need no index list
    int *sensEnt; bool sensEntValid;   // compact solid state: list entry of every sensor voxel (-1 = none), valid for the current list
    int64_t nSensors; uint32_t *sensLin; float *sensOut; int nTs; int nSelS; int selS[BFD_MAP_COUNT];
    double *dftAcc; float *dftPk; int dftBin;      // sensorMode 1: [nSelS][nSensors][2] running DFT sums, [nSelS][nSensors] running peaks
    // accumulators
    int nSelR; int selR[BFD_MAP_COUNT]; float *acc, *pk;
    int accStart;
    // timing
    bool timing, perKernel;
    hipEvent_t evBegin, evEnd;
    std::vector<hipEvent_t> evStress, evVelocity;  // pairs
    std::vector<hipEvent_t> evK[BFD_K_COUNT];      // pairs per kernel class (perKernel == 2)
    double algBytes[2][BFD_K_COUNT];               // algorithmic bytes per launch and class: [0] no accumulation, [1] Pressure RMS accumulated
    std::vector<hipEvent_t> evPool;
    // optional hipGraph replay of "plain" time steps (no accumulation, no sensor sample, no per-kernel timing) in
    // bfd_run, BFD_USE_GRAPH=1; measured slower than direct launches on ROCm 7.2, so off by default (see bfd_run)
    int *stepDev;                   // device copy of the step counter (sources index their pulse with it inside a graph)
    hipGraphExec_t stepGraph;       // BFD_GRAPH_STEPS time steps
    hipStream_t captureStream;
    int graphState;                 // 0 not tried, 1 ready, -1 unavailable (direct launches are used)
    bool stepDevValid;
};
#define BFD_GRAPH_STEPS 8

void bfd_set_error(const std::string &s);
#define BFD_HIP(call)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            bfd_set_error(std::string(#call) + ": " + hipGetErrorString(e_));                      \
            return -10;                                                                            \
        }                                                                                          \
    } while (0)

// madvise(MADV_HUGEPAGE) on a result buffer of the caller before a large device-to-host copy (bfd_api.hip)
void bfd_advise_result_buffer(void *p, size_t bytes);
// bfd_get_sensors with a row pitch: the series of selected map q at out + q * rowElems (bfd_group.hip: a slab writes into its columns)
int bfd_sensors_into(bfd_sim *s, float *out, int64_t rowElems);
// kernel launchers (bfd_kernels_*.hip)
void bfd_launch_stress_v1(const bfd_dev &d, hipStream_t s);
void bfd_launch_velocity_v1(const bfd_dev &d, hipStream_t s);
void bfd_tile_grid(const bfd_dev &d, int *tilesX, int *tilesY, int *subZ);
// runs [off, off + n) of the fused list (bfd_kernels_fused.hip)
void bfd_launch_fused(const bfd_dev &d, hipStream_t s, float *accP, float *pkP, const bfd_tiles *t, int off, int n);
int bfd_fused_rows(void);
int bfd_fused_max_materials(void);
int bfd_tile_subz(void);
bool bfd_css_supported(void);     // false in the experiment builds whose solid-run kernels have no compact form
void bfd_launch_classify(const bfd_dev &d, hipStream_t s, int *flagsDev, int *tileMatDev);
void bfd_launch_mark_source_subtiles(const bfd_dev &d, hipStream_t s, const uint32_t *lin, long n);
// flags the cells of the sparse shear list: solid, non-reflector centre; mixedOnly: only those with BFD_CLS_MIXED (merged solid stress kernel)
void bfd_launch_mark_solid(const bfd_dev &d, hipStream_t s, unsigned char *flag, long n, bool mixedOnly);
void bfd_launch_shear_order_keys(const bfd_dev &d, hipStream_t s, const unsigned *cells, unsigned long long *keys, long n, int lowPlanes, int hiStart, int mode);
void bfd_launch_shear_coefficients(const bfd_dev &d, hipStream_t s, const unsigned *cells, float *coef, unsigned *codes, float *tab, int nMat, long n);
// copies the list-ordered shear memory variables into the full-volume arrays Rxy, Rxz, Ryz (outputs only)
void bfd_launch_scatter_shear_memory(const bfd_dev &d, hipStream_t s, const bfd_tiles *t);
void bfd_launch_gather_shear_memory(const bfd_dev &d, hipStream_t s, const bfd_tiles *t);
// compact solid state: the row table of a list in order mode 2 (lowPlanes / hiStart as for bfd_launch_shear_order_keys)
void bfd_launch_css_row_table(const bfd_dev &d, hipStream_t s, const unsigned *cells, long n, unsigned *rowTable, int stride, int lowPlanes, int hiStart);
// dstFull (pointer to local plane 0 of a full-volume buffer) [cell] = compact array a [entry], a = 0..9 in the order Sxx Syy Sxy Sxz Syz Rxx Ryy Rxy Rxz Ryz;
// the reverse into dstCompact from a full-volume source. cells / n: the list the entries belong to.
void bfd_launch_css_scatter(hipStream_t s, const unsigned *cells, long n, const float *compact, float *dstFull);
void bfd_launch_css_gather(hipStream_t s, const unsigned *cells, long n, const float *srcFull, float *dstCompact);
void bfd_launch_cell_classes(const bfd_dev &d, hipStream_t s, uint8_t *clsBase, long nalloc);
// counts over the cells of the solid runs: [0] fluid no-memory, [1] fluid with memory, [2] solid no-memory, [3] solid with memory,
// [4] active shear edges, [5] reflector cells
void bfd_launch_count_solid_cells(const bfd_dev &d, hipStream_t s, const int4 *solidRuns, int nSolid, unsigned long long *counts6);
int bfd_tile_zchunk(void);
// placement probe: arrays a and b (pointers to local plane 0) updated in place along all runs of the slab, planes [0, kmax)
void bfd_launch_probe_pair(const bfd_dev &d, hipStream_t s, const bfd_tiles *t, float *a, float *b, int kmax);
// part: 0 = every tile, 1 = boundary tiles, 2 = interior tiles (variant 2 lists every tile as solid)
void bfd_launch_stress_v2(const bfd_dev &d, hipStream_t s, const bfd_tiles *t, int part);
// accP / pkP: Pressure RMS / peak accumulators of this step (slab-local, x-fastest) or nullptr
void bfd_launch_velocity_v2(const bfd_dev &d, hipStream_t s, float *accP, float *pkP, const bfd_tiles *t, int part);
