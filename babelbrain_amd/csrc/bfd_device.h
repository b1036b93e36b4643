// Device-side helpers shared by the tiled kernels (bfd_kernels_v2.hip, bfd_kernels_fused.hip). gfx950 only.
#pragma once
#include "bfd_internal.h"

namespace {

__device__ __forceinline__ float dminus4(float fm2, float fm1, float f0, float fp1)
{
    float t1 = f0 - fm1;
    float t2 = fp1 - fm2;
    return BFD_CA * t1 - BFD_CB * t2;
}
__device__ __forceinline__ float dplus4(float fm1, float f0, float fp1, float fp2)
{
    float t1 = fp1 - f0;
    float t2 = fp2 - fm1;
    return BFD_CA * t1 - BFD_CB * t2;
}
__device__ __forceinline__ float cpml(float *__restrict__ psi, unsigned idx, float a, float b, float D)
{
    float pn = b * psi[idx] + a * D;
    psi[idx] = pn;
    return D + pn;
}


// Element access as (wave-uniform plane base) + (32-bit BYTE offset in a VGPR). Indexing a float* with a 32-bit cell index
// instead makes the compiler build 64-bit addresses in VGPR pairs that stay live (it cannot prove that index*4 stays below
// 2^32), and it reassociates (array + plane) + lane offset into (array + lane offset) + plane, hoisting the first sum out of
// the z loop: a loop-invariant VGPR pair per array.
// uni(): the plane base as an opaque wave-uniform value (SGPR pair); the address of an access is then one v_lshl_add_u64 of
// that pair and the shared offset register, live only until the access. The pointer is rebuilt from integers, so these are
// FLAT accesses; the variant with address_space(1) pointers and saddr-form global loads (scripts/r2/patches/) needs fewer
// registers still but measured 5-6 % slower on the solid-run kernels and equal on the fluid ones.
#ifdef BFD_GLOBAL_ACCESS
// Experiment build (-DBFD_GLOBAL_ACCESS): the same accessors as GLOBAL instructions (address_space(1), saddr form: SGPR base +
// 32-bit VGPR offset). A FLAT load counts on lgkmcnt as well as on vmcnt, so the first wait for an LDS read also waits for every
// load of the next plane issued before it; global loads wait on vmcnt alone. Round 2 measured this form 5-6 % slower on the
// solid-run kernels (one more workgroup per CU in flight); profiles/r4/ has the round-4 numbers.
#define BFD_GLOBAL __attribute__((address_space(1)))
template <typename T>
__device__ __forceinline__ BFD_GLOBAL T *uni(T *p)
{
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (BFD_GLOBAL T *)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ unsigned pinofs(unsigned v) { asm("" : "+v"(v)); return v; }
struct GlobalF4 {
    BFD_GLOBAL float *p;
    __device__ __forceinline__ operator float() const { return *p; }
    __device__ __forceinline__ void operator=(float v) const { *p = v; }
    __device__ __forceinline__ void operator=(const GlobalF4 &o) const { *p = *o.p; }
};
__device__ __forceinline__ GlobalF4 F4(const float *base, unsigned byteOfs) { return GlobalF4{(BFD_GLOBAL float *)((BFD_GLOBAL char *)uni(const_cast<float *>(base)) + pinofs(byteOfs))}; }
__device__ __forceinline__ void ST4(float *base, unsigned byteOfs, float v)
{
#ifndef BFD_NT_STORES_OFF
    __builtin_nontemporal_store(v, (BFD_GLOBAL float *)((BFD_GLOBAL char *)uni(base) + pinofs(byteOfs)));
#else
    *(BFD_GLOBAL float *)((BFD_GLOBAL char *)uni(base) + pinofs(byteOfs)) = v;
#endif
}
__device__ __forceinline__ float LD4(const float *base, unsigned byteOfs)
{
#ifndef BFD_NT_STORES_OFF
    return __builtin_nontemporal_load((BFD_GLOBAL const float *)((BFD_GLOBAL const char *)uni(base) + pinofs(byteOfs)));
#else
    return *(BFD_GLOBAL const float *)((BFD_GLOBAL const char *)uni(base) + pinofs(byteOfs));
#endif
}
template <typename T> __device__ __forceinline__ T LDNT(const T *p)
{
#ifndef BFD_NT_STORES_OFF
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
__device__ __forceinline__ unsigned U2(const uint16_t *base, unsigned byteOfs) { return *(BFD_GLOBAL const uint16_t *)((BFD_GLOBAL const char *)uni(base) + pinofs(byteOfs)); }
__device__ __forceinline__ unsigned U1(const uint8_t *base, unsigned byteOfs) { return *(BFD_GLOBAL const uint8_t *)((BFD_GLOBAL const uint8_t *)uni(base) + pinofs(byteOfs)); }
#else
template <typename T>
__device__ __forceinline__ T *uni(T *p)
{
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (T *)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ float &F4(float *base, unsigned byteOfs) { return *(float *)((char *)uni(base) + byteOfs); }
__device__ __forceinline__ const float &F4(const float *base, unsigned byteOfs) { return *(const float *)((const char *)uni(base) + byteOfs); }
// store of a value nobody reads before the next half-step: non-temporal, so that written lines do not displace halo lines in L2
// (fluid kernels: C3 89.1 -> 89.8 Gvoxel-steps/s on three alternating same-box runs; solid-run kernels: shear medium 63.87 -> 64.10 on
// five; profiles/r3/experiment_nontemporal_stores.txt). -DBFD_NT_STORES_OFF builds the plain stores.
__device__ __forceinline__ void ST4(float *base, unsigned byteOfs, float v)
{
#ifndef BFD_NT_STORES_OFF
    __builtin_nontemporal_store(v, (float *)((char *)uni(base) + byteOfs));
#else
    *(float *)((char *)uni(base) + byteOfs) = v;
#endif
}
// load of a value only this lane reads in this half-step: non-temporal, it need not stay in L2 (fluid stress half-step: Szz / Rzz of
// the own cell, Vz; velocity half-steps: V of the own cell, the RMS sums; the sparse shear kernel's list, coefficients and entries).
// C3 89.7 -> 91.0, shear medium 64.1 -> 66.3 Gvoxel-steps/s on alternating same-box runs; NOT for the own-cell stresses of
// stress_solid (0.300 -> 0.313 ms). profiles/r3/experiment_nontemporal_stores.txt. -DBFD_NT_STORES_OFF builds the plain accesses.
__device__ __forceinline__ float LD4(const float *base, unsigned byteOfs)
{
#ifndef BFD_NT_STORES_OFF
    return __builtin_nontemporal_load((const float *)((const char *)uni(base) + byteOfs));
#else
    return *(const float *)((const char *)uni(base) + byteOfs);
#endif
}
template <typename T> __device__ __forceinline__ T LDNT(const T *p)               // sparse shear kernel: list entries, coefficients, its own S and R entries
{
#ifndef BFD_NT_STORES_OFF
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
__device__ __forceinline__ unsigned U2(const uint16_t *base, unsigned byteOfs) { return *(const uint16_t *)((const char *)uni(base) + byteOfs); }
__device__ __forceinline__ unsigned U1(const uint8_t *base, unsigned byteOfs) { return uni(base)[byteOfs]; }

#endif

// XCD-aware tile order: consecutive block ids land on different XCDs (round robin over 8), so give
// XCD e the e-th contiguous run of tiles.
__device__ __forceinline__ int remap_block(int bid, int nblocks)
{
    const int per = nblocks >> 3;
    if (per == 0 || bid >= (per << 3)) return bid;     // tail blocks keep their id
    return (bid & 7) * per + (bid >> 3);
}

// Cost-balanced form (bfd_tiles::xmap): slot b & 7 = one XCD; its k-th block takes run seg[slot] + k of the launched range, -1 = nothing left
// for this block. The eighths are equal in estimated cost, not in count (profiles/r4/xcd_balance.txt).
__device__ __forceinline__ int xcd_run_index(const int *__restrict__ xmap)
{
    const int slot = __builtin_amdgcn_readfirstlane(blockIdx.x & 7), k = blockIdx.x >> 3;
    const int idx = xmap[slot] + k;
    return idx < xmap[slot + 1] ? idx : -1;
}
// the run of this block: cost-balanced map if the launch has one, otherwise equal counts per XCD; -1 = no run for this block
__device__ __forceinline__ int run_index(int nblocks, const int *__restrict__ xmap)
{
    return xmap ? xcd_run_index(xmap) : remap_block(blockIdx.x, nblocks);
}

}  // namespace
