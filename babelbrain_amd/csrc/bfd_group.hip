// bfd_group_*: ONE solver call split into Z-slabs over several devices of ONE process (include/babelfdtd.h).
//
// The reference's caller is a single process that makes a single call (Babel_SingleTx.py:258 spawns one Process;
// BabelIntegrationBASE.py:2338-2365 calls PModel.StaggeredFDTD_3D_with_relaxation once), so the Z-slab decomposition of
// SURVEY.md 8e has to live behind that call: a group owns one slab engine (bfd_sim) per entry of `devices`, slices the
// caller's whole-domain inputs, runs the step loop here in C and moves the 2+2 halo planes per half-step with peer copies
// (hipMemcpyPeerAsync between devices, a device copy when two slabs share a device) ordered by events -- no collective, no
// second process. The torchrun / RCCL path of babelbrain_amd/slab.py stays for one-process-per-GPU launches (bench.py).
//
// Step order (the one SlabRunner.step uses). Per half-step H and slab r, M = the slab's main stream,
// B = its high-priority side stream:
//     B waits for M  ->  part 1 of H on B (the runs that hold the planes a Z-neighbour reads)  ->  event P1[r]
//     (velocity half-step only: M waits for P1[r]; its end-of-step work reads the whole slab)   ->  part 2 of H on M
//     B waits for P1[neighbours]  ->  copies of the neighbours' fresh boundary planes into r's ghost planes on B  ->  event X[r]
//     M waits for X[r]
// Slabs thinner than BFD_OVERLAP_MIN_PLANES (64), and kernelVariant 1, take the unsplit order (whole half-step on M, then the
// copies on M). A copy into slab d is always issued on d's own streams behind d's previous half-step, and a source slab
// overwrites planes a neighbour copied only after it has waited for an event recorded behind that copy, so neither
// direction needs more than these events.
#include "bfd_internal.h"

#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>

#define GRP_FAIL(code, msg) do { bfd_set_error(msg); return (code); } while (0)

struct bfd_group {
    bfd_config cfg;                       // whole domain
    int n;
    std::vector<bfd_sim *> sim;
    std::vector<int> dev, k0, nk;
    std::vector<hipStream_t> side;        // B streams
    std::vector<hipEvent_t> evMain, evPart1, evPart1b, evHalo;      // evPart1 / evPart1b: even / odd half-steps
    std::vector<uint32_t> reads[2];       // per slab and halo group: bit f set = the slab reads field f of its neighbours' planes
    struct HaloPtr { void *ptr[2][3][2][2]; size_t bytes; };               // [group][field][side][send]
    std::vector<HaloPtr> halo;            // device pointers of every slab's halo regions, taken once in group_prepare
    bool threads;                         // one host thread per slab queues its work (BFD_GROUP_THREADS=0: one thread for all)
    bool prepared, overlap;
    bool forcePeer = false;               // BFD_GROUP_FORCE_PEER_COPY: hipMemcpyPeerAsync also between slabs on one device
    std::vector<int32_t> peer;            // per interface r | r+1: BFD_PEER_* code + detail bits (bfd_group_peer_status)
    std::vector<int64_t> nSens;
    double issueSeconds; int64_t issueSteps;    // host time spent queueing work in bfd_group_run
    double haloBytesPerStep;
    bool timing; std::chrono::steady_clock::time_point t0;
};

namespace {

int partition(int N3, int n, std::vector<int> &k0, std::vector<int> &nk)
{
    if (n < 1 || N3 < 4 * n) return -1;
    const int base = N3 / n, rem = N3 % n;
    k0.resize(n); nk.resize(n);
    int k = 0;
    for (int r = 0; r < n; r++) { nk[r] = base + (r < rem ? 1 : 0); k0[r] = k; k += nk[r]; }
    return 0;
}

// dense host copy of planes [ka, kb) of a strided (N1,N2,*) view, in a layout that keeps the fastest axis of the source:
// returns the element strides of the copy. nullptr copy = the view itself can be used (slab already contiguous enough).
template <typename T>
bool pack_slab(const T *src, int64_t s1, int64_t s2, int64_t s3, int N1, int N2, int ka, int kb, std::vector<T> &buf,
               int64_t &d1, int64_t &d2, int64_t &d3)
{
    const int nks = kb - ka;
    if (s1 == 1 && s2 == N1 && s3 == (int64_t)N1 * N2) return false;       // x-fastest: a Z-slab is one contiguous block
    buf.resize((size_t)N1 * N2 * nks);
    if (s3 == 1) {          // C order (k fastest): runs of nks elements per (i, j)
        d1 = (int64_t)N2 * nks; d2 = nks; d3 = 1;
        for (int i = 0; i < N1; i++)
            for (int j = 0; j < N2; j++)
                memcpy(&buf[(size_t)i * d1 + (size_t)j * d2], src + i * s1 + j * s2 + ka, (size_t)nks * sizeof(T));
    } else {
        d1 = 1; d2 = N1; d3 = (int64_t)N1 * N2;
        for (int k = 0; k < nks; k++)
            for (int j = 0; j < N2; j++)
                for (int i = 0; i < N1; i++) buf[(size_t)k * d3 + (size_t)j * d2 + i] = src[i * s1 + j * s2 + (int64_t)(ka + k) * s3];
    }
    return true;
}

// runs fn(r) for every slab on its own host thread (input slicing / result assembly is memory-bound host work)
template <typename F> int for_slabs(int n, F fn)
{
    std::vector<int> rc(n, 0);
    std::vector<std::string> err(n);
    std::vector<std::thread> th;
    for (int r = 0; r < n; r++) th.emplace_back([&, r]() { rc[r] = fn(r); if (rc[r]) err[r] = bfd_last_error(); });
    for (auto &t : th) t.join();
    for (int r = 0; r < n; r++) if (rc[r]) { bfd_set_error("slab " + std::to_string(r) + ": " + err[r]); return rc[r]; }
    return 0;
}

int group_prepare(bfd_group *g)
{
    if (g->prepared) return 0;
    // Slabs that share a device prepare one after the other (the placement holds transient memory, and a slab that sees its siblings' arrays on
    // the device does not search at all). One slab per device -- the real multi-GPU case -- is every device's only tenant: each slab runs the same
    // bounded search a single-device call runs (choose_placement prices `others` per device), and the slabs prepare SIDE BY SIDE on their host
    // threads, so that eight searches cost the time of one (round 6; BFD_GROUP_PARALLEL_PREPARE=0 / 1 forces either order).
    bool distinct = true;
    for (int r = 0; r < g->n; r++) for (int q = 0; q < r; q++) if (g->dev[r] == g->dev[q]) distinct = false;
    bool side = distinct && g->n > 1;
    if (const char *ev = getenv("BFD_GROUP_PARALLEL_PREPARE")) side = g->n > 1 && atoi(ev) != 0;
    if (side) {
        const int rc = for_slabs(g->n, [&](int r) { return bfd_prepare(g->sim[r]); });
        if (rc) return rc;
    } else for (int r = 0; r < g->n; r++) {
        const int rc = bfd_prepare(g->sim[r]);
        if (rc) return rc;
    }
    int minPlanes = 64;
    if (const char *ev = getenv("BFD_OVERLAP_MIN_PLANES")) minPlanes = atoi(ev);
    g->overlap = g->n > 1 && g->cfg.kernelVariant != 1;
    for (int r = 0; r < g->n; r++) if (g->nk[r] < minPlanes) g->overlap = false;
    if (const char *ev = getenv("BFD_GROUP_OVERLAP")) g->overlap = g->n > 1 && g->cfg.kernelVariant != 1 && atoi(ev) != 0;
    g->haloBytesPerStep = 0;
    for (int r = 0; r < g->n; r++)
        for (int grp = 0; grp < 2; grp++) {
            uint32_t m = 7;
            const int rc = bfd_halo_fields(g->sim[r], grp, &m);
            if (rc) return rc;
            g->reads[grp][r] = m;
            const int nb = (r > 0 ? 1 : 0) + (r + 1 < g->n ? 1 : 0);
            g->haloBytesPerStep += (double)__builtin_popcount(m) * nb * 2.0 * g->cfg.N1 * g->cfg.N2 * 4.0;
        }
    g->halo.assign(g->n, bfd_group::HaloPtr());
    for (int r = 0; r < g->n; r++)
        for (int grp = 0; grp < 2; grp++) for (int f = 0; f < 3; f++) for (int side = 0; side < 2; side++) for (int send = 0; send < 2; send++) {
            const int rc = bfd_halo_region(g->sim[r], grp, f, side, send, &g->halo[r].ptr[grp][f][side][send], &g->halo[r].bytes);
            if (rc) return rc;
        }
    // BFD_GROUP_FORCE_PEER_COPY=1 (tests on a 1-GPU box): the halo planes go through hipMemcpyPeerAsync also between slabs that share a
    // device, so that the call the multi-GPU path makes is exercised with the same arguments
    if (const char *ev = getenv("BFD_GROUP_FORCE_PEER_COPY")) g->forcePeer = atoi(ev) != 0;
    g->threads = g->n > 1;
    if (const char *ev = getenv("BFD_GROUP_THREADS")) g->threads = g->n > 1 && atoi(ev) != 0;
    g->prepared = true;
    return 0;
}

static hipEvent_t part1_event(bfd_group *g, int r, long n) { return (n & 1) ? g->evPart1b[r] : g->evPart1[r]; }

// half-step n (0, 1, 2, ...: even = stress, odd = velocity) of slab r: the kernels. Records the slab's part-1 event of n.
int issue_half(bfd_group *g, int r, long n)
{
    const int half = (int)(n & 1);
    bfd_sim *s = g->sim[r];
    BFD_HIP(hipSetDevice(g->dev[r]));
    hipStream_t M = s->stream, B = g->side[r];
    hipEvent_t p1 = part1_event(g, r, n);
    int rc;
    if (g->overlap) {
        BFD_HIP(hipEventRecord(g->evMain[r], M));
        BFD_HIP(hipStreamWaitEvent(B, g->evMain[r], 0));
        rc = half == 0 ? bfd_half_step_stress_part_on(s, 1, B) : bfd_half_step_velocity_part_on(s, 1, B);
        if (rc) return rc;
        BFD_HIP(hipEventRecord(p1, B));
        if (half == 1) BFD_HIP(hipStreamWaitEvent(M, p1, 0));
        rc = half == 0 ? bfd_half_step_stress_part_on(s, 2, M) : bfd_half_step_velocity_part_on(s, 2, M);
        if (rc) return rc;
    } else {
        rc = half == 0 ? bfd_half_step_stress(s) : bfd_half_step_velocity(s);
        if (rc) return rc;
        BFD_HIP(hipEventRecord(p1, M));
    }
    return 0;
}

// half-step n of slab r: the copies that fill its ghost planes from its neighbours (their part-1 events of n must have
// been recorded by now), on the side stream (overlapped order) or the main stream
int issue_copies(bfd_group *g, int r, long n)
{
    const int grp = (n & 1) == 0 ? BFD_HALO_STRESS : BFD_HALO_VELOCITY;     // the stress half-step produces the STRESS halo group
    BFD_HIP(hipSetDevice(g->dev[r]));
    hipStream_t M = g->sim[r]->stream, X = g->overlap ? g->side[r] : M;
    for (int side = 0; side < 2; side++) {
        const int s = side == 0 ? r - 1 : r + 1;
        if (s < 0 || s >= g->n) continue;
        BFD_HIP(hipStreamWaitEvent(X, part1_event(g, s, n), 0));
        for (int f = 0; f < 3; f++) {
            if (!(g->reads[grp][r] & (1u << f))) continue;
            void *dst = g->halo[r].ptr[grp][f][side][0];                        // my ghost planes on that side
            void *src = g->halo[s].ptr[grp][f][side ^ 1][1];                    // the neighbour's boundary planes facing me
            const size_t nb = g->halo[r].bytes;
            if (g->dev[r] == g->dev[s] && !g->forcePeer) BFD_HIP(hipMemcpyAsync(dst, src, nb, hipMemcpyDeviceToDevice, X));
            else BFD_HIP(hipMemcpyPeerAsync(dst, g->dev[r], src, g->dev[s], nb, X));
        }
    }
    if (g->overlap) {
        BFD_HIP(hipEventRecord(g->evHalo[r], X));
        BFD_HIP(hipStreamWaitEvent(M, g->evHalo[r], 0));
    }
    return 0;
}

// nSteps time steps queued by one host thread
int run_serial(bfd_group *g, int nSteps)
{
    for (long n = 0; n < 2L * nSteps; n++) {
        for (int r = 0; r < g->n; r++) { const int rc = issue_half(g, r, n); if (rc) return rc; }
        for (int r = 0; r < g->n; r++) { const int rc = issue_copies(g, r, n); if (rc) return rc; }
    }
    return 0;
}

// nSteps time steps, one host thread per slab (a step of an 8-way split is ~40 runtime calls per slab: one thread for all
// slabs spends 0.6 ms per step on them, more than a thin slab's GPU time). Threads meet through two counters per slab:
//   recorded[r] = half-steps whose part-1 event slab r has recorded (a neighbour waits for it before it queues
//                 hipStreamWaitEvent on that event: a wait on an event that was not recorded yet is no wait at all);
//   consumed[r] = half-steps whose copies slab r has queued (a neighbour re-records the event of the same parity only
//                 after the waits on its previous record are in the queue).
int run_threaded(bfd_group *g, int nSteps)
{
    const int n = g->n;
    std::vector<std::atomic<long>> recorded(n), consumed(n);
    for (int r = 0; r < n; r++) { recorded[r].store(0); consumed[r].store(0); }
    std::atomic<int> failed(0);
    std::vector<int> rcs(n, 0);
    std::vector<std::string> errs(n);
    auto spin_until = [&](std::atomic<long> &c, long v) {
        int spins = 0;
        while (c.load(std::memory_order_acquire) < v && !failed.load(std::memory_order_relaxed))
            if (++spins > 2000) std::this_thread::yield();
    };
    auto worker = [&](int r) {
        for (long h = 0; h < 2L * nSteps && !failed.load(std::memory_order_relaxed); h++) {
            // the event of this parity was last recorded for half-step h-2: both neighbours must have queued their waits on it
            if (h >= 2) for (int s : {r - 1, r + 1}) if (s >= 0 && s < n) spin_until(consumed[s], h - 1);
            int rc = issue_half(g, r, h);
            if (!rc) {
                recorded[r].store(h + 1, std::memory_order_release);
                for (int s : {r - 1, r + 1}) if (s >= 0 && s < n) spin_until(recorded[s], h + 1);
                if (!failed.load(std::memory_order_relaxed)) rc = issue_copies(g, r, h);
            }
            if (rc) { rcs[r] = rc; errs[r] = bfd_last_error(); failed.store(1); break; }
            consumed[r].store(h + 1, std::memory_order_release);
        }
    };
    std::vector<std::thread> th;
    for (int r = 0; r < n; r++) th.emplace_back(worker, r);
    for (auto &t : th) t.join();
    for (int r = 0; r < n; r++) if (rcs[r]) { bfd_set_error("slab " + std::to_string(r) + ": " + errs[r]); return rcs[r]; }
    return 0;
}

}  // namespace

extern "C" {

int bfd_group_create(const bfd_config *cfg, int32_t nSlabs, const int32_t *devices, bfd_group **out)
{
    if (!cfg || !devices || !out) GRP_FAIL(-1, "bfd_group_create: null argument");
    if (nSlabs < 1 || nSlabs > 64) GRP_FAIL(-2, "bfd_group_create: 1..64 slabs");
    if (cfg->k0 != 0 || cfg->nk != cfg->N3) GRP_FAIL(-2, "bfd_group_create: cfg describes the whole domain (k0 = 0, nk = N3)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) GRP_FAIL(-3, "bfd_group_create: no HIP device available (this engine has no CPU fallback)");
    for (int r = 0; r < nSlabs; r++) if (devices[r] < 0 || devices[r] >= ndev) GRP_FAIL(-3, "bfd_group_create: device ordinal out of range");
    bfd_group *g = new bfd_group();
    g->cfg = *cfg; g->n = nSlabs; g->prepared = false; g->overlap = false; g->issueSeconds = 0; g->issueSteps = 0; g->haloBytesPerStep = 0; g->timing = false;
    if (partition(cfg->N3, nSlabs, g->k0, g->nk)) { delete g; GRP_FAIL(-2, "bfd_group_create: every slab needs at least 4 planes"); }
    g->dev.assign(devices, devices + nSlabs);
    g->sim.assign(nSlabs, nullptr); g->side.assign(nSlabs, nullptr);
    g->evMain.assign(nSlabs, nullptr); g->evPart1.assign(nSlabs, nullptr); g->evPart1b.assign(nSlabs, nullptr); g->evHalo.assign(nSlabs, nullptr);
    g->threads = false;
    g->reads[0].assign(nSlabs, 7u); g->reads[1].assign(nSlabs, 7u); g->nSens.assign(nSlabs, 0);
    // peer access between the devices of neighbouring slabs. Without it hipMemcpyPeerAsync still works, but the runtime stages every halo plane
    // through the host: nothing fails, the curve is just bad -- so what happened is kept per interface (bfd_group_peer_status)
    g->peer.assign(nSlabs > 1 ? nSlabs - 1 : 0, BFD_PEER_SAME_DEVICE);
    for (int r = 0; r + 1 < nSlabs; r++) {
        const int a = devices[r], b = devices[r + 1];
        if (a == b) continue;
        int32_t bits = 0;
        const int from[2] = {a, b}, to[2] = {b, a};
        for (int q = 0; q < 2; q++) {
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, from[q], to[q]) != hipSuccess) { can = 0; (void)hipGetLastError(); }
            if (!can) continue;
            bits |= 16 << (2 * q);
            hipSetDevice(from[q]);
            const hipError_t e = hipDeviceEnablePeerAccess(to[q], 0);
            if (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) bits |= 32 << (2 * q);
            (void)hipGetLastError();
        }
        g->peer[r] = ((bits & 0xF0) == 0xF0 ? BFD_PEER_DIRECT : BFD_PEER_STAGED) | bits;
    }
    int rc = 0;
    for (int r = 0; r < nSlabs && !rc; r++) {
        bfd_config c = *cfg;
        c.k0 = g->k0[r]; c.nk = g->nk[r]; c.device = devices[r];
        rc = bfd_create(&c, &g->sim[r]);
        if (rc) break;
        int lo = 0, hi = 0;
        hipError_t e = hipSetDevice(devices[r]);
        if (e == hipSuccess) e = hipDeviceGetStreamPriorityRange(&lo, &hi);      // hi = numerically lowest = highest priority
        if (e == hipSuccess) e = hipStreamCreateWithPriority(&g->side[r], hipStreamNonBlocking, hi);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&g->evMain[r], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&g->evPart1[r], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&g->evPart1b[r], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&g->evHalo[r], hipEventDisableTiming);
        if (e != hipSuccess) { bfd_set_error(std::string("bfd_group_create: ") + hipGetErrorString(e)); rc = -10; }
    }
    if (rc) { const std::string keep = bfd_last_error(); bfd_group_destroy(g); bfd_set_error(keep); return rc; }
    *out = g;
    return 0;
}

void bfd_group_destroy(bfd_group *g)
{
    if (!g) return;
    for (int r = 0; r < g->n; r++) {
        if (g->sim[r]) { hipSetDevice(g->dev[r]); hipDeviceSynchronize(); }
        if (g->side[r]) hipStreamDestroy(g->side[r]);
        if (g->evMain[r]) hipEventDestroy(g->evMain[r]);
        if (g->evPart1[r]) hipEventDestroy(g->evPart1[r]);
        if (g->evPart1b[r]) hipEventDestroy(g->evPart1b[r]);
        if (g->evHalo[r]) hipEventDestroy(g->evHalo[r]);
        if (g->sim[r]) bfd_destroy(g->sim[r]);
    }
    delete g;
}

int32_t bfd_group_size(bfd_group *g) { return g ? g->n : -1; }

int bfd_group_slab(bfd_group *g, int32_t r, int32_t *k0, int32_t *nk, int32_t *device, bfd_sim **sim)
{
    if (!g || r < 0 || r >= g->n) GRP_FAIL(-1, "bfd_group_slab: bad argument");
    if (k0) *k0 = g->k0[r];
    if (nk) *nk = g->nk[r];
    if (device) *device = g->dev[r];
    if (sim) *sim = g->sim[r];
    return 0;
}

int bfd_group_set_materials(bfd_group *g, const double *matlist, const double *qcorr)
{
    if (!g) GRP_FAIL(-1, "null group");
    for (int r = 0; r < g->n; r++) { const int rc = bfd_set_materials(g->sim[r], matlist, qcorr); if (rc) return rc; }
    g->prepared = false;
    return 0;
}

int bfd_group_set_material_map(bfd_group *g, const uint32_t *map, int64_t s1, int64_t s2, int64_t s3)
{
    if (!g || !map) GRP_FAIL(-1, "bfd_group_set_material_map: null argument");
    if (s1 < 0 || s2 < 0 || s3 < 0) GRP_FAIL(-2, "negative strides are not supported");
    const int N1 = g->cfg.N1, N2 = g->cfg.N2, N3 = g->cfg.N3;
    g->prepared = false;
    return for_slabs(g->n, [&](int r) {
        const int gl = std::min(2, g->k0[r]), gh = std::min(2, N3 - (g->k0[r] + g->nk[r]));
        std::vector<uint32_t> buf; int64_t d1 = s1, d2 = s2, d3 = s3;
        const uint32_t *base = map + (int64_t)(g->k0[r] - gl) * s3;         // first readable plane of the slab's view
        if (pack_slab(map, s1, s2, s3, N1, N2, g->k0[r] - gl, g->k0[r] + g->nk[r] + gh, buf, d1, d2, d3)) base = buf.data();
        return bfd_set_material_map(g->sim[r], base + (int64_t)gl * d3, d1, d2, d3, gl, gh);
    });
}

int bfd_group_set_reflector(bfd_group *g, const uint32_t *mask, int64_t s1, int64_t s2, int64_t s3)
{
    if (!g) GRP_FAIL(-1, "null group");
    g->prepared = false;
    if (!mask) { for (int r = 0; r < g->n; r++) { const int rc = bfd_set_reflector(g->sim[r], nullptr, 0, 0, 0); if (rc) return rc; } return 0; }
    if (s1 < 0 || s2 < 0 || s3 < 0) GRP_FAIL(-2, "negative strides are not supported");
    return for_slabs(g->n, [&](int r) {
        std::vector<uint32_t> buf; int64_t d1 = s1, d2 = s2, d3 = s3;
        const uint32_t *base = mask + (int64_t)g->k0[r] * s3;
        if (pack_slab(mask, s1, s2, s3, g->cfg.N1, g->cfg.N2, g->k0[r], g->k0[r] + g->nk[r], buf, d1, d2, d3)) base = buf.data();
        return bfd_set_reflector(g->sim[r], base, d1, d2, d3);
    });
}

int bfd_group_set_sources(bfd_group *g, int64_t nVox, const int64_t *globalIndex, const uint32_t *row,
                          const float *wx, const float *wy, const float *wz, const double *pulse, int32_t nSources, int32_t lengthSource)
{
    if (!g) GRP_FAIL(-1, "null group");
    if (nVox < 0 || (nVox > 0 && (!globalIndex || !row || !pulse))) GRP_FAIL(-1, "bfd_group_set_sources: null argument");
    const int64_t plane = (int64_t)g->cfg.N1 * g->cfg.N2, total = plane * g->cfg.N3;
    for (int64_t v = 0; v < nVox; v++) if (globalIndex[v] < 0 || globalIndex[v] >= total) GRP_FAIL(-2, "bfd_group_set_sources: voxel index outside the domain");
    for (int r = 0; r < g->n; r++) {
        const int64_t lo = plane * g->k0[r], hi = lo + plane * g->nk[r];
        std::vector<uint32_t> li, rw; std::vector<float> w[3];
        const float *wsrc[3] = {wx, wy, wz};
        for (int64_t v = 0; v < nVox; v++) {
            if (globalIndex[v] < lo || globalIndex[v] >= hi) continue;
            li.push_back((uint32_t)(globalIndex[v] - lo)); rw.push_back(row[v]);
            for (int a = 0; a < 3; a++) if (wsrc[a]) w[a].push_back(wsrc[a][v]);
        }
        const int rc = bfd_set_sources(g->sim[r], (int64_t)li.size(), li.data(), rw.data(), wx ? w[0].data() : nullptr, wy ? w[1].data() : nullptr,
                                       wz ? w[2].data() : nullptr, pulse, nSources, lengthSource);
        if (rc) return rc;
    }
    g->prepared = false;
    return 0;
}

int bfd_group_set_sensor_map(bfd_group *g, const uint32_t *map, int64_t s1, int64_t s2, int64_t s3, int64_t *nSensors)
{
    if (!g || !map) GRP_FAIL(-1, "bfd_group_set_sensor_map: null argument");
    if (s1 < 0 || s2 < 0 || s3 < 0) GRP_FAIL(-2, "negative strides are not supported");
    const int rc = for_slabs(g->n, [&](int r) {
        std::vector<uint32_t> buf; int64_t d1 = s1, d2 = s2, d3 = s3;
        const uint32_t *base = map + (int64_t)g->k0[r] * s3;
        if (pack_slab(map, s1, s2, s3, g->cfg.N1, g->cfg.N2, g->k0[r], g->k0[r] + g->nk[r], buf, d1, d2, d3)) base = buf.data();
        return bfd_set_sensor_map(g->sim[r], base, d1, d2, d3, &g->nSens[r]);
    });
    if (rc) return rc;
    if (nSensors) { *nSensors = 0; for (int r = 0; r < g->n; r++) *nSensors += g->nSens[r]; }
    return 0;
}

int bfd_group_set_placement(bfd_group *g, int32_t mode, int64_t searchLimitBytes)
{
    if (!g) GRP_FAIL(-1, "null group");
    for (int r = 0; r < g->n; r++) { const int rc = bfd_set_placement(g->sim[r], mode, searchLimitBytes); if (rc) return rc; }
    return 0;
}

int bfd_group_prepare(bfd_group *g)
{
    if (!g) GRP_FAIL(-1, "null group");
    return group_prepare(g);
}

int bfd_group_run(bfd_group *g, int32_t nSteps)
{
    if (!g) GRP_FAIL(-1, "null group");
    int rc = group_prepare(g); if (rc) return rc;
    const auto t0 = std::chrono::steady_clock::now();
    if (g->n == 1) rc = bfd_run(g->sim[0], nSteps);
    else rc = g->threads ? run_threaded(g, nSteps) : run_serial(g, nSteps);
    g->issueSeconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    g->issueSteps += nSteps;
    return rc;
}

int bfd_group_sync(bfd_group *g)
{
    if (!g) GRP_FAIL(-1, "null group");
    for (int r = 0; r < g->n; r++) {
        BFD_HIP(hipSetDevice(g->dev[r]));
        BFD_HIP(hipStreamSynchronize(g->side[r]));
        BFD_HIP(hipStreamSynchronize(g->sim[r]->stream));
    }
    return 0;
}

int bfd_group_reset(bfd_group *g)
{
    if (!g) GRP_FAIL(-1, "null group");
    int rc = bfd_group_sync(g); if (rc) return rc;
    for (int r = 0; r < g->n; r++) { rc = bfd_reset(g->sim[r]); if (rc) return rc; }
    return 0;
}

int bfd_group_timing_begin(bfd_group *g)
{
    if (!g) GRP_FAIL(-1, "null group");
    int rc = group_prepare(g); if (rc) return rc;
    rc = bfd_group_sync(g); if (rc) return rc;
    for (int r = 0; r < g->n; r++) { rc = bfd_timing_begin(g->sim[r], 0); if (rc) return rc; }
    g->issueSeconds = 0; g->issueSteps = 0; g->timing = true; g->t0 = std::chrono::steady_clock::now();
    return 0;
}

int bfd_group_peer_status(bfd_group *g, int32_t *status, int32_t n)
{
    if (!g || (!status && n > 0)) GRP_FAIL(-1, "bfd_group_peer_status: null argument");
    if (n < g->n - 1) GRP_FAIL(-2, "bfd_group_peer_status: room for nSlabs - 1 interfaces needed");
    for (int r = 0; r + 1 < g->n; r++) status[r] = g->peer[r];
    return g->n - 1;
}

int bfd_group_timing_end(bfd_group *g, double *wallMs, double *maxDeviceMs, double *hostIssueMs, double *haloBytesPerStep, int32_t *overlapped)
{
    if (!g) GRP_FAIL(-1, "null group");
    if (!g->timing) GRP_FAIL(-6, "bfd_group_timing_end without bfd_group_timing_begin");
    double worst = 0;
    for (int r = 0; r < g->n; r++) {        // the event pair of each slab's main stream; synchronises it
        double tot = 0;
        const int rc = bfd_timing_end(g->sim[r], &tot, nullptr, nullptr, nullptr, nullptr, nullptr);
        if (rc) return rc;
        worst = std::max(worst, tot);
    }
    const int rc = bfd_group_sync(g); if (rc) return rc;
    g->timing = false;
    if (wallMs) *wallMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - g->t0).count();
    if (maxDeviceMs) *maxDeviceMs = worst;
    if (hostIssueMs) *hostIssueMs = g->issueSeconds * 1e3;
    if (haloBytesPerStep) *haloBytesPerStep = g->n > 1 ? g->haloBytesPerStep : 0.0;
    if (overlapped) *overlapped = g->overlap ? 1 : 0;
    return 0;
}

int64_t bfd_group_num_sensors(bfd_group *g)
{
    if (!g) return -1;
    int64_t n = 0;
    for (int r = 0; r < g->n; r++) n += bfd_num_sensors(g->sim[r]);
    return n;
}
int32_t bfd_group_num_sensor_steps(bfd_group *g) { return g ? bfd_num_sensor_steps(g->sim[0]) : -1; }

int bfd_group_get_sensor_index(bfd_group *g, uint32_t *index)
{
    if (!g) GRP_FAIL(-1, "null group");
    int64_t off = 0;
    for (int r = 0; r < g->n; r++) {        // slabs are ordered in k, so the concatenation is ascending in the global index
        const int64_t ns = bfd_num_sensors(g->sim[r]);
        if (ns) { const int rc = bfd_get_sensor_index(g->sim[r], index + off); if (rc) return rc; }
        off += ns;
    }
    return 0;
}

int bfd_group_get_sensors(bfd_group *g, float *out)
{
    if (!g) GRP_FAIL(-1, "null group");
    const int64_t total = bfd_group_num_sensors(g);
    const int nTs = bfd_num_sensor_steps(g->sim[0]);
    if (total == 0 || nTs <= 0) return 0;
    if (!out) GRP_FAIL(-1, "bfd_group_get_sensors: null argument");
    std::vector<int64_t> off(g->n + 1, 0);
    for (int r = 0; r < g->n; r++) off[r + 1] = off[r] + bfd_num_sensors(g->sim[r]);
    return for_slabs(g->n, [&](int r) {
        const int64_t ns = off[r + 1] - off[r];
        if (!ns) return 0;
        return bfd_sensors_into(g->sim[r], out + (size_t)off[r] * nTs, total * (int64_t)nTs);   // straight into this slab's columns
    });
}

int bfd_group_get_sensor_dft(bfd_group *g, double freq, float *outReIm, float *outPeak)
{
    if (!g) GRP_FAIL(-1, "null group");
    const int64_t total = bfd_group_num_sensors(g);
    if (total == 0) return 0;
    if (!outReIm) GRP_FAIL(-1, "bfd_group_get_sensor_dft: null argument");
    const int nSel = g->sim[0]->nSelS;
    int64_t off = 0;
    for (int r = 0; r < g->n; r++) {
        const int64_t ns = bfd_num_sensors(g->sim[r]);
        if (ns) {
            std::vector<float> re((size_t)nSel * ns * 2), pk((size_t)nSel * ns);
            const int rc = bfd_get_sensor_dft(g->sim[r], freq, re.data(), outPeak ? pk.data() : nullptr);
            if (rc) return rc;
            for (int q = 0; q < nSel; q++) {
                memcpy(outReIm + ((size_t)q * total + off) * 2, re.data() + (size_t)q * ns * 2, (size_t)ns * 2 * sizeof(float));
                if (outPeak) memcpy(outPeak + (size_t)q * total + off, pk.data() + (size_t)q * ns, (size_t)ns * sizeof(float));
            }
        }
        off += ns;
    }
    return 0;
}

int bfd_group_get_map(bfd_group *g, int32_t kind, int32_t map, float *out, int64_t s1, int64_t s2, int64_t s3)
{
    if (!g || !out) GRP_FAIL(-1, "bfd_group_get_map: null argument");
    if (s1 < 0 || s2 < 0 || s3 < 0) GRP_FAIL(-2, "negative strides are not supported");
    const int N1 = g->cfg.N1, N2 = g->cfg.N2;
    return for_slabs(g->n, [&](int r) {
        const int nk = g->nk[r];
        float *dst = out + (int64_t)g->k0[r] * s3;
        if (g->n == 1 || (s1 == 1 && s2 == N1 && s3 == (int64_t)N1 * N2)) return bfd_get_map(g->sim[r], kind, map, dst, s1, s2, s3);
        // dense staging copy of the slab, then into the caller's view (its address span covers the other slabs too)
        std::vector<float> tmp((size_t)N1 * N2 * nk);
        int rc;
        if (s3 == 1) {
            rc = bfd_get_map(g->sim[r], kind, map, tmp.data(), (int64_t)N2 * nk, nk, 1);
            if (rc) return rc;
            for (int i = 0; i < N1; i++)
                for (int j = 0; j < N2; j++) memcpy(dst + i * s1 + j * s2, &tmp[((size_t)i * N2 + j) * nk], (size_t)nk * sizeof(float));
        } else {
            rc = bfd_get_map(g->sim[r], kind, map, tmp.data(), 1, N1, (int64_t)N1 * N2);
            if (rc) return rc;
            for (int k = 0; k < nk; k++)
                for (int j = 0; j < N2; j++)
                    for (int i = 0; i < N1; i++) dst[i * s1 + j * s2 + (int64_t)k * s3] = tmp[((size_t)k * N2 + j) * N1 + i];
        }
        return 0;
    });
}

int64_t bfd_group_device_bytes(bfd_group *g)
{
    if (!g) return -1;
    int64_t b = 0;
    for (int r = 0; r < g->n; r++) b += bfd_device_bytes(g->sim[r]);
    return b;
}

}  // extern "C"
