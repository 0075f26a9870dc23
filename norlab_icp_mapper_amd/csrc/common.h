// common.h -- internal declarations shared by the HIP translation units of libicpmi.so.
// Target: gfx950 (MI355X, CDNA4), wave64. Not a public header (the public one is include/icpmi.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <cstring>
#include <string>
#include <vector>
#include <unordered_map>
#include <mutex>
#include <map>
#include <cstdlib>
#include <shared_mutex>

#include "../../include/icpmi.h"

#define ICPMI_NV 32            // values of the minimiser's pair sums (27 of A / b or 16 of the point-to-point sums, sum w, pair count, force2D's b)
#define ICPMI_MAX_OUTLIER 8
#define ICPMI_MAX_SMOOTH 16
#define ICPMI_SEL_BINS 2048     // legacy 11/11/10 selection (stage entry point, chains with > 1 quantile filter)
// Fused selection (one quantile filter in the chain): two 16-bit radix levels of the d^2 bit pattern,
// each level a FINE histogram (65536 bins, direct device atomics: the values spread over hundreds of
// bins) plus a COARSE one over the top 8 bits of the digit (256 bins) that tells a consumer which 256
// fine bins to read.  Level 0's coarse histogram is hot (2-3 exponent values hold everything), so the
// NN workgroups aggregate it in LDS and flush into one of ICPMI_S2_COPIES privatised copies.
#ifndef ICPMI_S2_COPIES
#define ICPMI_S2_COPIES 8   // (r3: 32 -> 8: the level-0 scan of every selection workgroup sums the copies; +1 % on the k = 1 chains, 2 copies -3 %)
#endif
#define ICPMI_S2_C0 4096                                        // word offsets inside d_selhist (legacy bins first)
#define ICPMI_S2_F0 (ICPMI_S2_C0 + ICPMI_S2_COPIES * 256)
#ifndef ICPMI_S2_FCOPIES
#define ICPMI_S2_FCOPIES 1                                       // privatised copies of the level-0 FINE histogram
#endif
#define ICPMI_S2_C1 (ICPMI_S2_F0 + ICPMI_S2_FCOPIES * 65536)
#define ICPMI_S2_F1 (ICPMI_S2_C1 + 256)
// Pair-sum accumulators (r4), behind the histograms so that everything that clears the one clears the other: the ICPMI_NV sums of a
// registration's iteration as signed 64-bit FIXED-POINT device atomics -- integer additions commute, so the total does not depend on the
// order in which the pair-sum workgroups arrive, and what the solver reads is 32 values, not 64 KB of per-workgroup partials to be
// reduced in a fixed order.  A value v is held as two limbs, hi = rint(v 2^-16) and lo = rint((v - hi 2^16) 2^40): together 2^-40
// absolute resolution over +-2^78 with no range analysis, no carries between the limbs (they are summed independently and joined in
// double).  Headroom: |hi| <= 2^62 / adds by the range limit, and one contribution to lo is at most 2^15 x 2^40 = 2^55, so a copy that
// receives ICPMI_ACC_MAX_ADDS = 128 workgroup partials stays below 2^62 -- ONE bit inside int64, not more: loop.hip's acc_cap() clamps the
// number of pair-sum workgroups to ICPMI_ACC_MAX_ADDS x ICPMI_ACC_COPIES whatever ICPMI_ACC_BLOCKS asks for.
// ICPMI_ACC_COPIES privatised copies (workgroup b adds to copy b % COPIES),
// every (copy, value, limb) on its own 128-byte line: device atomics serialise per line.  (r4 kept two parities of them for the solve
// fused into the next NN launch; that path was removed in r5 -- DESIGN 13.3 -- and one set is left.)
#define ICPMI_ACC_COPIES 2   // (8 copies made every reader fetch 64 KB of padded lines -- 100 MB per NN launch once every workgroup reads them; 2: 128 atomics per line)
#define ICPMI_ACC_MAX_ADDS 128 // workgroup partials one copy may receive per iteration (limb headroom, see above)
#define ICPMI_ACC_PAD 16                                               // u64 per slot = one 128-byte line
#define ICPMI_ACC_U64 (ICPMI_ACC_COPIES * ICPMI_NV * 2 * ICPMI_ACC_PAD)   // 2048 u64 = 16 KiB
#define ICPMI_ACC_IDX(copy, i, limb) ((((copy) * ICPMI_NV + (i)) * 2 + (limb)) * ICPMI_ACC_PAD)
#define ICPMI_ACC_FLAG (ICPMI_ACC_IDX(0, 0, 0) + 8)                    // a non-finite partial was met (NaN must reach the solver: ICPMI_ERR_NAN)
#define ICPMI_S2_ACC (ICPMI_S2_F1 + 65536)                             // u32 word offset of the accumulators (128-byte aligned)
// (r5) Speculative level 0 for the k > 1 loop (nnk_wg_kernel builds it, loop.hip: win_lookup reads it).  The 16-bit prefix the selection
// picks moves by a fraction of a bin from one iteration to the next once the registration settles, and the stand-alone level-0 builder
// costs 10.7 us per iteration at knn 6 (116 k device atomics: DESIGN 11.6, 12.8).  So the NN kernel counts its own k x 64 distances into
// ICPMI_WIN_BINS bins around the PREVIOUS iteration's prefix plus "below" and "above" -- nine counts, reduced inside the wave, three
// 64-bit atomics per workgroup (three 21-bit fields each: the host enables the window only below 2^21 matches, so no field can carry) --
// and when the selected rank falls inside the window the builder launch has nothing to do.  A miss (the first iterations, a limit that
// jumps) is the full histogram as before; the counts are exact either way, so the selected element is the same bit pattern.
// Every (copy, word) on its own 128-byte line; the header word holds (lowest window prefix + 1), 0 = no window this iteration.
#define ICPMI_WIN_BINS 7
#ifndef ICPMI_WIN_COPIES
#define ICPMI_WIN_COPIES 8
#endif
#define ICPMI_WIN_PAD 16                                                  // u64 per slot = one 128-byte line
#define ICPMI_WIN_HDR (ICPMI_WIN_COPIES * 3 * ICPMI_WIN_PAD)              // u64 index of the header
#define ICPMI_WIN_U64 (ICPMI_WIN_HDR + ICPMI_WIN_PAD)
#define ICPMI_WIN_MAX_COUNT (1ll << 21)
#define ICPMI_S2_WIN (ICPMI_S2_ACC + 2 * ICPMI_ACC_U64)                   // u32 word offset of the window (128-byte aligned)
#define ICPMI_SELHIST_WORDS (ICPMI_S2_WIN + 2 * ICPMI_WIN_U64)
// Device atomics serialise per cache line, and neighbouring fine bins are hot together: bin b of a fine
// histogram lives at word ((b & 255) << 8) | (b >> 8), i.e. consecutive bins are 1 KiB apart.
#define ICPMI_S2_FIDX(b) ((((b) & 255u) << 8) | ((b) >> 8))

#define ICPMI_MAX_K 32

// Once-touched streams (queries, per-query loop state, matches) with the non-temporal cache policy (`nt`): -DICPMI_NT_STREAMS.  r5
// measurement: an XCD's L2 does not keep ANY line across a kernel boundary on this part (the second of two back-to-back NN
// launches fetches the same 20.6 MB past the L2 as the first), so there is no map slice for these streams to push out between launches;
// inside a launch the hint changed nothing measurable.  Kept as a switch; plain loads / stores by default.
#ifdef __HIPCC__
typedef float icpmi_vf4 __attribute__((ext_vector_type(4)));
#ifdef ICPMI_NT_STREAMS
__device__ __forceinline__ float4 ld_stream(const float4* p) { const icpmi_vf4 v = __builtin_nontemporal_load(reinterpret_cast<const icpmi_vf4*>(p)); return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ float ld_stream(const float* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ int ld_stream(const int* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void st_stream(float4* p, float4 v) { icpmi_vf4 w; w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w; __builtin_nontemporal_store(w, reinterpret_cast<icpmi_vf4*>(p)); }
__device__ __forceinline__ void st_stream(float* p, float v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void st_stream(int* p, int v) { __builtin_nontemporal_store(v, p); }
#else
template <class T> __device__ __forceinline__ T ld_stream(const T* p) { return *p; }
template <class T> __device__ __forceinline__ void st_stream(T* p, T v) { *p = v; }
#endif

// ---- runs of equal keys inside a wave become ONE atomic per run (counting sorts: map_build.hip, the DynamicPoints bucket grid) ----
struct WaveRun { bool head; int rank; int len; int head_lane; };
__device__ __forceinline__ WaveRun wave_run(unsigned key, bool valid)
{
    const int lane = threadIdx.x & 63;
    const unsigned prev = (unsigned)__shfl_up((int)key, 1, 64);
    const bool pvalid = __shfl_up((int)valid, 1, 64) != 0;
    const bool head = valid && (lane == 0 || !pvalid || prev != key);
    const unsigned long long heads = __ballot(head);
    const unsigned long long valids = __ballot(valid);
    WaveRun r;
    r.head = head;
    const unsigned long long below = heads & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
    r.head_lane = below ? 63 - __clzll((long long)below) : lane;
    r.rank = lane - r.head_lane;
    const unsigned long long above = (lane == 63) ? 0ull : (heads & ~((2ull << lane) - 1ull));
    const int nvalid = __popcll(valids); // valid lanes are a prefix of the wave
    const int end = above ? (__ffsll((long long)above) - 1) : nvalid;
    r.len = end - r.head_lane;
    return r;
}
#endif

// ------------------------------------------------------------------------------------------------
// NN grid (built by set_map on the centred map).  Dense uniform grid; cells are x-fastest so the
// three x-neighbours of a cell row are one contiguous run of the cell-sorted point array.
// ------------------------------------------------------------------------------------------------
struct GridParams {
    float ox, oy, oz;      // min corner (centred frame)
    float cell, inv_cell;  // cell edge and its reciprocal
    float slack;           // conservative distance slack for prune / termination tests
    int   nx, ny, nz;
    int   ncells;
};

// Pyramid of grids over the same bounding box: level l has cell edge cell0 * 2^l and its own
// cell-sorted copy of the map.  A query that cannot be decided inside the 3x3x3 block of level l
// (best distance beyond its margin) repeats the identical 27-cell search one level up, where the
// block is twice as wide -- no ring bookkeeping, and always one contiguous x-run per (y, z) row.
#define ICPMI_MAXLEV 8
struct GridLevels {
    int nlev;
    GridParams g[ICPMI_MAXLEV];
    const float4* pts[ICPMI_MAXLEV];    // cell-sorted points of the level: xyz centred, w = original index bits
    const unsigned* cs[ICPMI_MAXLEV];   // cell starts of the level (ncells + 1)
    const unsigned* pos0[ICPMI_MAXLEV]; // level position -> level-0 position (nullptr for level 0)
};

// Batched registration (icpmi_register_batch_dev): B independent readings against the same map go through ONE launch
// of every kernel of the loop, blockIdx.y = reading.  Per-reading arrays are slices of one allocation (slice b starts
// at b * qstride elements; selection histograms, pair-sum partials and the IcpState array have their own fixed
// strides).  A single registration is the batch of one (blockIdx.y == 0, every offset 0): the SAME kernels serve
// both, so a reading registers to the same bits alone or in a batch.
#define ICPMI_MAX_BATCH 16
struct BatchArgs {
    int nscan;
    int qstride;                 // elements between the per-reading slices of the per-query arrays
    int n[ICPMI_MAX_BATCH];      // points of every reading
};
struct BatchSrc { const float4* p[ICPMI_MAX_BATCH]; }; // the readings as handed in (device pointers)
static inline BatchArgs batch_of_one(int64_t n) { BatchArgs b; memset(&b, 0, sizeof b); b.nscan = 1; b.n[0] = (int)n; return b; }

// Device-side description of the ICP chain for one registration (passed by value to kernels).
struct LoopCfg {
    int   k;
    float max_dist;        // un-squared; +inf allowed
    float maxr2;           // max_dist^2 (+inf allowed)
    float inv1e, err2;     // KDTreeMatcher's epsilon when icpmi_config::epsilon_approx asks for it: 1 / (1 + epsilon) and (1 + epsilon)^2; both 1 otherwise
    int   ring_max;        // rings searched on the grid before a query goes to the brute pass
    int   minimizer;
    int   n_out;
    int   out_type[ICPMI_MAX_OUTLIER];
    float out_param[ICPMI_MAX_OUTLIER];
    int   out_iparam[ICPMI_MAX_OUTLIER];
    float out_param2[ICPMI_MAX_OUTLIER];
    float out_param3[ICPMI_MAX_OUTLIER];
    int   ext;             // GenericDescriptor / Robust in the chain: the pair-sum kernel's EXT variant
    int   force_4dof;
    int   force_2d;        // (routes the pair sums through the EXT variant: three more sums, b of the 2-D residual)
    int   is_2d;           // planar clouds: point-to-point solves the in-plane rotation in closed form
    int   max_iter;
    int   use_diff;
    float min_rot, min_trans;
    int   smooth;
    int   use_bound;
    float max_rot, max_trans;
    int   has_read_normals;
    int   sensor_noise;    // the reading carries simpleSensorNoise + normals: getOverlap() is the sensor-noise count (loop.hip: overlap pass)
    const float* read_scalar; // GenericDescriptorOutlierFilter{source: reading}: that descriptor of the reading, caller's order (device)
};

// Device-resident loop state (one per handle).  Everything the iteration needs between kernels
// lives here so that the loop never synchronises with the host.
struct IcpState {
    float T_iter[16];
    int   iter;
    int   done;
    int   error;           // icpmi_status (0 = ok)
    int   stop_reason;
    int   counter;
    int   hist_n;          // number of poses pushed to the differential checker
    double hq[(ICPMI_MAX_SMOOTH + 1) * 4];
    double ht[(ICPMI_MAX_SMOOTH + 1) * 3];
    double hrot[ICPMI_MAX_SMOOTH + 1];   // |angular distance| and translation distance between pose slot i and the pose before it
    double htr[ICPMI_MAX_SMOOTH + 1];
    double init_q[4];
    // quantile selection
    unsigned sel_prefix;
    unsigned sel_rank;
    unsigned n_valid;
    unsigned sel_prefix_l[3];  // fused selection: one slot per radix level (written by level L,
    unsigned sel_rank_l[3];    // read by level L + 1 -- never both in one kernel)
    float limits[ICPMI_MAX_OUTLIER];
    unsigned vt_valid;     // VarTrimmedDist: number of valid matches of this iteration (counted by vt_keys_kernel, cleared by vt_pick_kernel)
    float vt_ratio;        // ... and the ratio optimizeInlierRatio picked (diagnostic)
    float robust_med;      // RobustOutlierFilter{mad}: median of the finite d2 of this iteration
    float robust_scale;    // RobustOutlierFilter::scale, kept between iterations (nbIterationForScale)
    // statistics of the last iteration
    long long pairs;
    double wsum;
    unsigned hard_count;
    unsigned ticket;             // workgroups of the pair-sum kernel that have published their partials (last one solves)
    unsigned long long hard_total;
    unsigned seq;                // registration sequence number (tag of the progress word, see icpmi_ctx::h_progress)
    unsigned long long dbg[24];  // diagnostics: NN phase cycles with -DICPMI_NN_TIMING (scripts/nn_phase.py), [20]/[21] serial solve cycles / calls
    float T_prev[16];            // T_iter BEFORE the last minimisation (kept when LoopCfg::sensor_noise: getOverlap() looks at that step's pairs)
    // result
    float T_out[16];
    // r5: device clocks (wall_clock64, 100 MHz) of the first kernel of the registration's head and of the solve that stopped the loop:
    // stats->loop_ms without a HIP event to wait for (loop_run's fast finish).  t_done == 0: the loop was not stopped by a solve.
    unsigned long long t_start, t_done;
    // r6: device-clock interval of every NN launch, summed over the registration: opened by the NN kernel's first workgroup, closed by the first
    // workgroup of whatever kernel runs next (so one kernel boundary is inside) -- icpmi_stats::nn_ms_avg of a loop that is NOT in profile mode,
    // i.e. bench.py's roofline duration taken from the timed graph replay itself
    unsigned long long t_nn_begin, t_nn_sum;
    unsigned nn_count, pad_nn;
};

#ifdef __HIPCC__
__device__ __forceinline__ void nn_stamp_open(IcpState* st) { st->t_nn_begin = (unsigned long long)wall_clock64(); }
__device__ __forceinline__ void nn_stamp_close(IcpState* st)
{
    const unsigned long long b = st->t_nn_begin;
    if (b) { st->t_nn_sum += (unsigned long long)wall_clock64() - b; ++st->nn_count; st->t_nn_begin = 0; }
}
#endif


// ------------------------------------------------------------------------------------------------
// Device memory through a process-wide block cache (r5).  hipFree unmaps the block behind a device-wide synchronisation and hipMalloc
// maps a new one (~0.1 - 0.3 ms a pair at the sizes of a map's arrays); a mapper grows ~20 arrays by doubling while its map grows, every
// operator call holds a few temporaries, and a replay that builds a fresh mapper pays all of it again (BASELINE config 4: the first two of
// 14 scans were 14 of 37 ms).  dev_free keeps the block (after the same device-wide synchronisation hipFree implies: nothing in flight
// can still touch it), dev_malloc hands out the smallest cached block of at least the size asked for and at most twice that + 1 MiB.
// ICPMI_ALLOC_CACHE_MB (default 1024; 0: plain hipMalloc / hipFree) bounds what is kept; the largest blocks go first.  Blocks are cached per
// device; icpmi_trim_cache() empties it.
// ------------------------------------------------------------------------------------------------
// A device-wide synchronisation and a stream capture in ANOTHER thread do not mix on this runtime (scripts/r5/capture_threads.hip, ROCm 7:
// hipDeviceSynchronize fails with "operation not permitted when stream is capturing" AND invalidates the other thread's thread-local capture;
// so does a synchronous hipMemcpy / hipMemset on the legacy stream; hipMalloc / hipFree / hipStreamCreate / hipHostMalloc / launches do not).
// One handle per thread is the contract of icpmi.h -- a mapper's update thread next to its registration thread (Mapper.cpp:274-288) is exactly
// that -- so: a thread that captures holds this gate shared from BeginCapture to EndCapture (host-side work only), dev_free takes it exclusive
// around its hipDeviceSynchronize, and the library makes no synchronous legacy-stream call outside diagnostics.
inline std::shared_mutex& capture_gate() { static std::shared_mutex* m = new std::shared_mutex; return *m; } // (never destroyed: see dev_block_cache)
struct CaptureGate {
    CaptureGate() { capture_gate().lock_shared(); }
    ~CaptureGate() { capture_gate().unlock_shared(); }
    CaptureGate(const CaptureGate&) = delete;
    CaptureGate& operator=(const CaptureGate&) = delete;
};

struct DevBlockCache {
    struct Live { size_t bytes; int dev; };
    std::mutex mu;
    std::unordered_map<void*, Live> live;                  // every block handed out -> its true size and the device it was mapped on
    std::map<int, std::multimap<size_t, void*>> idle_of;   // cached blocks by size, PER DEVICE (r6, ADVICE r5: icpmi_config::device puts several GPUs in one process)
    size_t idle_bytes = 0, limit = 0;
    long handles = 0;                                      // live top-level handles (icpmi_create / icpmi_destroy; bookkeeping)
    bool on = true;
    DevBlockCache()
    {
        const char* e = getenv("ICPMI_ALLOC_CACHE_MB");
        const long mb = e ? atol(e) : 1024;                // r6: 1 GiB (r5: 4 GiB -- a co-resident allocator sees cached blocks as used memory)
        on = mb > 0; limit = on ? (size_t)mb << 20 : 0;
    }
};
inline DevBlockCache& dev_block_cache() { static DevBlockCache* c = new DevBlockCache; return *c; } // (never destroyed: the runtime may be gone first)

// every cached block of every device back to the runtime (icpmi_trim_cache; the out-of-memory path; the last handle's destruction).
// Caller holds bc.mu.  hipFree synchronises the device it runs on; the blocks are idle, so any device will do.
inline void dev_cache_release_locked(DevBlockCache& bc)
{
    for (auto& dv : bc.idle_of) for (auto& kv : dv.second) (void)hipFree(kv.second);
    bc.idle_of.clear(); bc.idle_bytes = 0;
}
inline void dev_cache_trim()
{
    DevBlockCache& bc = dev_block_cache();
    std::lock_guard<std::mutex> lk(bc.mu);
    dev_cache_release_locked(bc);
}

inline hipError_t dev_malloc(void** p, size_t bytes)
{
    DevBlockCache& bc = dev_block_cache();
    if (!bc.on) return hipMalloc(p, bytes);
    if (bytes == 0) bytes = 1;
    bytes = (bytes + 255) & ~(size_t)255;
    int dev = 0;
    (void)hipGetDevice(&dev); // (a handle is created, used and destroyed with its device current: icpmi_create / CHECK_H set it)
    {
        std::lock_guard<std::mutex> lk(bc.mu);
        auto& idle = bc.idle_of[dev];
        auto it = idle.lower_bound(bytes);
        if (it != idle.end() && it->first <= 2 * bytes + ((size_t)1 << 20)) {
            *p = it->second; bc.live[*p] = {it->first, dev}; bc.idle_bytes -= it->first; idle.erase(it);
            return hipSuccess;
        }
    }
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) { // out of memory: give the cache back and try once more
        std::lock_guard<std::mutex> lk(bc.mu);
        (void)hipGetLastError();
        dev_cache_release_locked(bc);
        e = hipMalloc(p, bytes);
        if (e != hipSuccess) return e;
    }
    std::lock_guard<std::mutex> lk(bc.mu);
    bc.live[*p] = {bytes, dev};
    return hipSuccess;
}

// synced == true: the caller has just synchronised the block's device itself (icpmi_destroy: ONE device-wide wait for its ~80 blocks instead of
// one per block, each under the exclusive capture gate -- ADVICE r5)
inline hipError_t dev_free(void* p, bool synced = false)
{
    if (!p) return hipSuccess;
    DevBlockCache& bc = dev_block_cache();
    if (!bc.on) return hipFree(p);
    DevBlockCache::Live lv{0, 0};
    {
        std::lock_guard<std::mutex> lk(bc.mu);
        auto it = bc.live.find(p);
        if (it == bc.live.end()) return hipFree(p); // not ours (allocated before the cache was switched on)
        lv = it->second; bc.live.erase(it);
    }
    hipError_t e = hipSuccess;
    if (!synced) {
        // what hipFree would have waited for -- on the device the block lives on; not while another thread captures
        std::unique_lock<std::shared_mutex> gate(capture_gate());
        int cur = 0;
        (void)hipGetDevice(&cur);
        if (cur != lv.dev) (void)hipSetDevice(lv.dev);
        e = hipDeviceSynchronize();
        if (cur != lv.dev) (void)hipSetDevice(cur);
    }
    std::lock_guard<std::mutex> lk(bc.mu);
    bc.idle_of[lv.dev].emplace(lv.bytes, p); bc.idle_bytes += lv.bytes;
    while (bc.idle_bytes > bc.limit) { // the largest block of the fullest device goes first
        std::multimap<size_t, void*>* big = nullptr;
        for (auto& dv : bc.idle_of) if (!dv.second.empty() && (!big || std::prev(dv.second.end())->first > std::prev(big->end())->first)) big = &dv.second;
        if (!big) break;
        auto last = std::prev(big->end());
        (void)hipFree(last->second); bc.idle_bytes -= last->first; big->erase(last);
    }
    return e;
}
inline hipError_t dev_sync_for_free() // one device-wide wait under the capture gate, for a run of dev_free(p, true)
{
    std::unique_lock<std::shared_mutex> gate(capture_gate());
    return hipDeviceSynchronize();
}


// Streams through a process-wide pool (r5): hipStreamCreateWithFlags takes 1.7 - 23 ms on this runtime and hipStreamDestroy 2 - 3.5 ms
// (rocprofv3 --hip-runtime-trace on the BASELINE config 4 replay: 10 creations = 50 ms of a run whose 42 scans take 130 ms) -- a mapper
// that is built, fed a trajectory and dropped paid for its handle's stream, the private handle of its map-side operators and the side
// stream of its chain with its first two scans.  A released stream is drained and kept (at most 16).
struct StreamPool { std::mutex mu; std::map<int, std::vector<hipStream_t>> idle_of; }; // per device
inline StreamPool& stream_pool() { static StreamPool* p = new StreamPool; return *p; }
inline hipError_t stream_acquire(hipStream_t* s)
{
    StreamPool& sp = stream_pool();
    int dev = 0;
    (void)hipGetDevice(&dev);
    {
        std::lock_guard<std::mutex> lk(sp.mu);
        auto& idle = sp.idle_of[dev];
        if (!idle.empty()) { *s = idle.back(); idle.pop_back(); return hipSuccess; }
    }
    return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
}
inline void stream_release(hipStream_t s)
{
    if (!s) return;
    (void)hipStreamSynchronize(s);
    StreamPool& sp = stream_pool();
    int dev = 0;
    (void)hipGetDevice(&dev); // (a handle is created, used and destroyed with its device current: icpmi_create / CHECK_H set it)
    {
        std::lock_guard<std::mutex> lk(sp.mu);
        auto& idle = sp.idle_of[dev];
        if (idle.size() < 16) { idle.push_back(s); return; }
    }
    (void)hipStreamDestroy(s);
}

#define ICPMI_SCRATCH_SLOTS 20
struct icpmi_ctx {
    icpmi_config cfg;
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string last_error;
    // Every public entry point holds this for its whole duration: the handle's stream, staging buffers, operator scratch and
    // `temp` handle are shared by all calls, and the reference's callers reach one handle from several threads (online mode:
    // input filters on the caller's thread, the map update on a std::async thread, cell paging on Map::updateThread --
    // Mapper.cpp:274-288, Map.cpp:35-57).  Recursive: entry points call each other.
    std::recursive_mutex mu;

    // map
    int64_t m = 0;
    float mean[3] = {0, 0, 0};
    GridParams grid{};
    int64_t n_occupied = 0;
    float4* d_map_sorted = nullptr;     // xyz centred, w = bits of the original index
    float4* d_normals_sorted = nullptr; // xyz normal, w unused; nullptr if the map has no normals
    // point and normal of level-0 position s side by side (pn[2 s], pn[2 s + 1]: one 64-byte sector): what the pair sums of a k > 1
    // chain gather at random -- two separate 16-byte gathers fetch two sectors (r3: 76.5 MB per launch for 33.6 MB of payload)
    float4* d_map_pn = nullptr; size_t cap_map_pn = 0;
    unsigned* d_cell_start = nullptr;   // ncells + 1
    size_t cap_map = 0, cap_cells = 0, cap_normals = 0;
    bool has_normals = false;
    // coarser pyramid levels (level 0 aliases d_map_sorted / d_cell_start)
    GridLevels levels{};
    uint4* d_lvl_tab = nullptr;       // device copy of `levels` as 4 x uint4 per level (nn.hip reads it into LDS)
    float4* d_lvl_pts[ICPMI_MAXLEV] = {};   size_t cap_lvl_pts[ICPMI_MAXLEV] = {};
    unsigned* d_lvl_cs[ICPMI_MAXLEV] = {};  size_t cap_lvl_cs[ICPMI_MAXLEV] = {};
    unsigned* d_lvl_pos0[ICPMI_MAXLEV] = {}; size_t cap_lvl_pos0[ICPMI_MAXLEV] = {};
    unsigned* d_inv = nullptr; size_t cap_inv = 0; // original index -> level-0 position (kept for the incremental insert, map_build.hip: map_insert)
    // r4: incremental index insert.  A map that grew by an APPEND (Map::updateLocalPointCloud with PointDistanceMapperModule, the merge epoch
    // of the scan-sharded mapper) keeps its grid; its new index is the old one with the delta merged into the cell-sorted arrays of every
    // level -- one streaming pass per level instead of keys + histogram + scan + atomic scatter over all points.  The result is what a
    // full build on the same grid produces up to the order inside a cell (which nothing downstream sees): centred coordinates are
    // recomputed from the raw points with the NEW centroid, bit for bit what map_build writes.
    unsigned* d_lvl_key[ICPMI_MAXLEV] = {}; size_t cap_lvl_key[ICPMI_MAXLEV] = {};   // cell of every sorted position, per level
    float4* d_alt_pts[ICPMI_MAXLEV] = {};   size_t cap_alt_pts[ICPMI_MAXLEV] = {};   // ping-pong set the merge writes into
    unsigned* d_alt_cs[ICPMI_MAXLEV] = {};  size_t cap_alt_cs[ICPMI_MAXLEV] = {};
    unsigned* d_alt_pos0[ICPMI_MAXLEV] = {}; size_t cap_alt_pos0[ICPMI_MAXLEV] = {};
    unsigned* d_alt_key[ICPMI_MAXLEV] = {}; size_t cap_alt_key[ICPMI_MAXLEV] = {};
    float4* d_raw0 = nullptr; size_t cap_raw0 = 0;           // level-0 twin: the RAW points (w = original index) in level-0 sorted order (centred handles)
    float4* d_alt_raw0 = nullptr; size_t cap_alt_raw0 = 0;
    unsigned* d_ins_dstart0 = nullptr; size_t cap_ins_dstart0 = 0; // level 0's delta prefix, kept while the upper levels are merged
    float4* d_alt_nsorted = nullptr; size_t cap_alt_nsorted = 0;
    float4* d_alt_pn = nullptr; size_t cap_alt_pn = 0;
    unsigned* d_ins_key = nullptr; size_t cap_ins_key = 0;   // delta: cell key and rank inside the cell, per level
    unsigned* d_ins_rank = nullptr; size_t cap_ins_rank = 0;
    bool ins_ready = false;            // the current index carries the key / twin arrays the insert needs
    bool ins_normals_changed = false;  // set by a caller that recomputed the normals of the WHOLE cloud before an append-build (ops.hip)
    double sum_raw[3] = {0, 0, 0};     // sum of the raw coordinates of the indexed cloud (the centroid of the grown cloud without a pass over it)
    float lo_raw[3] = {0, 0, 0}, hi_raw[3] = {0, 0, 0}; // its bounding box
    float cell0 = 0.f;                 // level-0 cell edge as chosen by the last full build (the insert keeps it)
    uint64_t raw_epoch = 0;            // bumped whenever d_raw is rewritten other than by an append (private raw index: ops.hip)
    uint64_t temp_raw_epoch = 0; int64_t temp_raw_m = 0;
    uint64_t raw_view_version = 0;     // map_version the raw-frame VIEW of the registration index was set up for (ops.hip: raw_index)
    int64_t raw_view_count = 0;        // PointDistance searches served by the view (diagnostics)
    int64_t ins_count = 0, full_count = 0; // builds served by the insert / by the full path (diagnostics)

    // scratch for set_map
    unsigned* d_keys = nullptr; size_t cap_keys = 0;
    int oct_depth_hint = 0, oct_tag = 0; long oct_respeculated = 0; // octree.hip: tree depth of the previous call on this handle (the sort runs ahead of the depth's arrival), wrong guesses
    unsigned* d_fill = nullptr; size_t cap_fill = 0; bool fill_clean = false; // the count table of the grid builds: zero between builds while fill_clean (map_build.hip: counts_begin)
    unsigned* d_blocksums = nullptr; size_t cap_blocksums = 0;
    double* d_red = nullptr; size_t cap_red = 0;

    // reading-side buffers (capacity in entries)
    float4* d_reading = nullptr; size_t cap_reading = 0;       // centred reading
    float4* d_read_normals = nullptr; size_t cap_read_normals = 0;
    float*  d_read_noise = nullptr; size_t cap_read_noise = 0; int64_t read_noise_n = 0; // simpleSensorNoise of the NEXT reading (one shot)
    float*  d_read_scalar = nullptr; size_t cap_read_scalar = 0; int64_t read_scalar_n = 0; // GenericDescriptor{source: reading} row of the NEXT reading
    bool graph_sorted = false;        // loop state order of the cached graph
    float4* d_qsorted = nullptr; size_t cap_qsorted = 0;       // centred reading sorted by tile (NN locality)
    int*    d_qindex = nullptr; size_t cap_qindex = 0;         // sorted position -> original index
    unsigned* d_qkeys = nullptr; size_t cap_qkeys = 0;
    unsigned* d_qtile = nullptr; size_t cap_qtile = 0;
    int64_t qsorted_n = -1; const float4* qsorted_src = nullptr; // which reading d_qsorted was built from
    float4* d_stage_in = nullptr; size_t cap_stage_in = 0;     // host->device staging
    float*  d_stage_n3 = nullptr; size_t cap_stage_n3 = 0;
    // resident copy of the map as it was handed to set_map (original frame, caller's order): what the
    // device-side map update appends to and rebuilds from
    float4* d_raw = nullptr; size_t cap_raw = 0;
    float*  d_raw_n3 = nullptr; size_t cap_raw_n3 = 0;
    int64_t m_raw = 0; bool raw_has_normals = false;
    // one tracked scalar descriptor of the resident map (`probabilityDynamic` for the shipped chain) and the ping-pong
    // set the map-update chain compacts into (ops.hip: ops_map_update_chain)
    float*  d_raw_s = nullptr; size_t cap_raw_s = 0; bool raw_has_scalar = false;
    int*    d_src = nullptr; size_t cap_src = 0;               // provenance of every point of the map being updated
    float4* d_alt_raw = nullptr; size_t cap_alt_raw = 0;
    float*  d_alt_n3 = nullptr; size_t cap_alt_n3 = 0;
    float*  d_alt_s = nullptr; size_t cap_alt_s = 0;
    int*    d_alt_src = nullptr; size_t cap_alt_src = 0;
    float*  d_stage_s = nullptr; size_t cap_stage_s = 0;
    // scratch of the map-side operators (hash tables, beam buckets, flags): kept between calls -- a hipMalloc / hipFree
    // pair costs more than most of the kernels that use them
    void* scratch[ICPMI_SCRATCH_SLOTS] = {};                  // slots 0..9: operators on the handle's stream; 10..19: the DynamicPoints module (may run on `side`)
    size_t scratch_bytes[ICPMI_SCRATCH_SLOTS] = {};
    hipStream_t side = nullptr;                              // map-update chain: DynamicPoints next to the decimation that follows it (ops.hip)
    hipEvent_t side_fork = nullptr, side_join = nullptr;
    // the scan of the last icpmi_register_prior, in the map frame by its prior (what Mapper::processInput calls `input`)
    float4* d_scan_map = nullptr; size_t cap_scan_map = 0; int64_t scan_map_n = 0;
    float* d_T16 = nullptr;           // a 4x4 for device-side transforms
    icpmi_ctx* temp = nullptr;        // private handle of the map-side operators (indexes arbitrary clouds), created on first use
    // PointDistanceMapperModule builds its kd-tree on the map AS IT IS (PointDistanceMapperModule.cpp:32-36), the ICP matcher on the
    // map minus its centroid: squared distances in the two frames differ by rounding, and a keep decision at d2 ~ minDist^2 with
    // them.  The resident map-update paths therefore decide against this second index of the resident map in its own frame, rebuilt
    // when the map changed (map_version) -- not against the registration index.
    icpmi_ctx* temp_raw = nullptr; uint64_t temp_raw_version = 0;
    uint64_t map_version = 0;         // bumped by every map_build of this handle
    bool single_level = false;        // the raw-frame VIEW of a registration index (ops.hip: raw_index): level 0 of the pyramid is all it searches
    bool keep_raw = true;             // temp handles index clouds they do not own: no resident copy of the input
    bool no_centre = false;           // temp handles of the map-side operators: index raw coordinates (mean = 0)
    bool is_raw_index = false;        // the private raw-frame index of a resident map (ops.hip: raw_index): grows by appends with its owner
    int*    d_sidx = nullptr; size_t cap_sidx = 0;             // k x n sorted-map index (-1 none)
    float*  d_d2 = nullptr; size_t cap_d2 = 0;                 // k x n
    unsigned* d_hard = nullptr; size_t cap_hard = 0;           // hard query list
    unsigned* d_selhist = nullptr; size_t cap_selhist = 0;     // ICPMI_SELHIST_WORDS per reading of a batch
    unsigned* nn_hist0 = nullptr;     // set by the loop when the NN kernel should build the level-0 histogram
    bool nn_builds_hist0 = false;     // set by the NN launcher: true if the launched variant did build it
    bool nn_builds_win = false;       // ... true if it counted the speculative window (ICPMI_S2_WIN: nnk_wg_kernel in a fused-selection loop)
    int nn_iter_hint = 0;             // iteration index of the launch being enqueued (> 0: seeded by the previous match)
    float4* d_match_pt = nullptr; size_t cap_match_pt = 0;     // k = 1 loop: matched map point (xyz, original index bits) per query slot
    float4* nn_match_pt = nullptr;    // set by the loop: keep the loop state (sidx, d2, matched point) in query order
    bool nn_out_sorted = false;       // set by the NN launcher: true if the launched kernel did so
    bool nn_sorted_k = false;         // set by the loop for k > 1: keep the k matches of a query at its slot of the tile-sorted order
    IcpState* d_state = nullptr;                               // ICPMI_MAX_BATCH states (a single registration uses the first)
    IcpState* h_state = nullptr;                               // pinned mirror (ICPMI_MAX_BATCH)
    bool zero_pending = false;                                 // d_state / d_selhist still to be cleared: on the stream the handle really uses, at its first call (zero_state_if_pending)
    unsigned scan_tag = 0;                                     // call number of device_scan_flags_count: the tag its count comes back with (map_build.hip)
    IcpState* d_state_mirror = nullptr;                        // ... and its device address: the solve kernel of a single registration writes the finished state there itself (r5)
    int batch_cur = 1;                                         // readings of the launch sequence being enqueued (set by the loop)
    BatchArgs batch_args{};                                    // their sizes / slice stride
    unsigned* d_nocc_host = nullptr; bool nocc_by_scan = false; // device address of h_nocc; the pending occupancy word is delivered by the build's scan
    unsigned* h_nocc = nullptr; int64_t nocc_m = 0;            // pinned word: occupied cells of the last index build, and that build's point count (map_build)
    unsigned char* h_pin = nullptr;                            // pinned page: [0, ICPMI_PIN_BYTES) small read-backs, behind it the ring of upload_small
    unsigned up_next = 0;
    // Progress word of the running registration in host-mapped pinned memory, written by the solve kernel after every
    // iteration: bit 31 = loop finished, bits 30..12 = registration sequence number, bits 11..0 = iterations completed.
    // A registration with data-dependent length (Differential / Bound checkers) is enqueued eagerly, a bounded number of
    // iterations ahead of this word, instead of stopping the stream for a read-back every few iterations (loop.hip).
    unsigned* h_progress = nullptr; unsigned* d_progress = nullptr; unsigned reg_seq = 0;

    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<hipEvent_t> nn_events;

    // cached graph of one full fixed-count loop
    hipGraphExec_t graph_exec = nullptr;
    int64_t graph_n = -1; int graph_iters = -1; uint64_t graph_sig = 0;
    hipGraphExec_t bgraph_exec = nullptr; uint64_t bgraph_sig = 0; // ... and of one batched registration
    // checked loops (Counter + Differential / Bound: what Mapper::processInput runs) as SEGMENT graphs: [0] = head + the first
    // seg_len iterations, [1] = seg_len further iterations, replayed while the progress word says the loop is still running
    hipGraphExec_t seg_exec[2] = {nullptr, nullptr}; uint64_t seg_sig = 0; int64_t seg_n = -1; int seg_len = 0; bool seg_sorted = false;
    int seg_uses = 0, seg_wasted = 0, graph_uses = 0, graph_wasted = 0; uint64_t eager_sig = 0, map_epoch = 0; int64_t eager_n = -1;   // see drop_loop_graphs (map_epoch: one tick per invalidation)
    // r5: head graphs of OTHER lengths (head + L iterations, L = the iteration count of the handle's previous checked registration): a mapper's
    // registrations stop after about the same number of iterations scan after scan, and a head graph of exactly that length has no dead iterations
    struct SegHead { int len = 0; hipGraphExec_t exec = nullptr; unsigned long used = 0; } seg_heads[4];
    int seg_prev_iters = 0, seg_last_iters = 0; unsigned long seg_clock = 0;

    // buffers of the map-growth epoch (ops.hip: ops_staged_merge_allgather): this rank's accepted points, all ranks' blocks, the merged set
    float4* d_merge_send = nullptr; size_t cap_merge_send = 0;
    float4* d_merge_recv = nullptr; size_t cap_merge_recv = 0;
    float4* d_merged = nullptr; size_t cap_merged = 0;
    // RCCL communicator of the scan-sharded mapping mode (comm.hip); null = a single rank
    void* comm = nullptr; int comm_ranks = 1, comm_rank = 0;
    float comm_loop_shift = 0.f;      // loopback communicator (comm.hip): rank r's block = this rank's, moved r * shift along x
    bool comm_loop_ragged = false;    // ... and cut to unequal sizes (comm.hip: loop_counts_kernel)
    long long* d_comm_cnt = nullptr; size_t cap_comm_cnt = 0; // words of the epoch's count / ready exchanges (allocated by comm_init)
    // r5: the ONE-collective epoch (ops.hip: ops_staged_merge_allgather).  Every rank hands in a block of merge_block + 1 float4: a header
    // {bits(count) | -1, magic} and up to merge_block accepted points; the blocks are sized when the communicator is created (no allocation,
    // hence no `ready` exchange, between the local accept and the collective).  0: the three-collective epoch of r4.
    int64_t merge_block = 0;
    long merge_fast_epochs = 0, merge_slow_epochs = 0; // epochs served by the one-collective path / by the count + ready + points path
    int64_t merged_last_n = 0;        // points of the last epoch's merged set, still in d_merged (icpmi_staged_merged_points)
    // r6 (cells.hip): every merged set binned by icpmi_staged_bin_cells, cell after cell; the host keeps {offset, count} runs per cell id
    float4* d_cell_log = nullptr; size_t cap_cell_log = 0; int64_t cell_log_n = 0;
    // r6 (ops.hip: surface_normals_dev): d^2 of the k-th neighbour of every resident point as the last SurfaceNormal pass over the resident map found it
    // (original order) -- an append recomputes only the normals an appended point can have changed.  Valid for the first dk_m points while
    // dk_epoch == raw_epoch (nobody rewrote the resident copy) and the filter's knn is dk_knn.
    float* d_raw_dk = nullptr; size_t cap_raw_dk = 0; int64_t dk_m = 0; int dk_knn = 0; uint64_t dk_epoch = ~0ull;
    long normals_incremental = 0, normals_full = 0; int64_t normals_last_searched = 0;   // diagnostics (icpmi_debug_counters 20 / 21 / 22)
    bool merged_binned = false;       // the merged set of the last epoch is in the log already
    int cell_bits_hint = 0;           // key bits the previous epoch's cell count needed (the sort is enqueued before the count is known)
    float cell_auto_size = 0.f;       // > 0 (icpmi_cell_log_configure): every epoch enqueues the binning of its merged set itself, behind the merge
    unsigned char* h_cells = nullptr; // pinned: header + cell table of the binning in flight
    int64_t cells_enq_n = 0; int cells_enq_bits = 0; float cells_enq_size = 0.f;   // what is enqueued and not yet collected (n == 0: nothing)
    struct SelfGridCtx* sg = nullptr; // sparse block grid of the self k-NN (selfgrid.hip): tables, work lists and the tuning state of the handle's previous build
    bool counted = false;             // created through icpmi_create (not a private handle): counts towards the allocation cache's lifetime (api.hip)
};

#define HIP_TRY(ctx, expr)                                                                     \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            (ctx)->last_error = std::string(#expr) + ": " + hipGetErrorString(_e);             \
            return ICPMI_ERR_HIP;                                                              \
        }                                                                                      \
    } while (0)

template <typename T>
static inline icpmi_status ensure_cap(icpmi_ctx* c, T** p, size_t* cap, size_t need)
{
    if (need <= *cap && *p) return ICPMI_OK;
    if (*p) { HIP_TRY(c, dev_free(*p)); *p = nullptr; *cap = 0; }
    // doubling: a map that grows by a scan's worth of points per update would otherwise reallocate a few of its ~20 arrays on every
    // update, and a hipFree is a device-wide synchronisation (~0.2 ms each, r3 HIP trace: 1 ms per map update); HBM is not the constraint
    size_t want = 2 * need + 64;
    HIP_TRY(c, dev_malloc((void**)p, want * sizeof(T)));
    *cap = want;
    return ICPMI_OK;
}

// Every cached loop graph (fixed-count, batched, the two segment graphs of a checked loop) holds pointers, grid parameters, the map's mean
// and its normals flag as they were at capture time: a rebuilt index or a changed stream invalidates all of them (only graph_exec
// was dropped, and the signature of the segment graphs does not cover mean / m / has_normals / d_map_pn / the level arrays).
// The loop state and the selection histograms of a new handle are cleared at its FIRST call, on the stream it then has: a private handle adopts its
// owner's stream right after its creation and never touches its own -- clearing there at creation (r5, after the synchronous hipMemset on the
// legacy stream had to go: it breaks another thread's capture) made the runtime build a hardware queue for every such stream, milliseconds
// of the first scans of a mapper.
static inline icpmi_status zero_state_if_pending(icpmi_ctx* c)
{
    if (!c->zero_pending) return ICPMI_OK;
    HIP_TRY(c, hipMemsetAsync(c->d_state, 0, sizeof(IcpState) * ICPMI_MAX_BATCH, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->d_selhist, 0, ICPMI_SELHIST_WORDS * sizeof(unsigned), c->stream));
    c->zero_pending = false;
    return ICPMI_OK;
}

static inline void drop_loop_graphs(icpmi_ctx* c)
{
    if (c->graph_exec) { if (c->graph_uses <= 1) ++c->graph_wasted; else c->graph_wasted = 0; hipGraphExecDestroy(c->graph_exec); c->graph_exec = nullptr; }
    c->graph_n = -1; c->graph_sig = 0; c->graph_uses = 0;
    if (c->bgraph_exec) { hipGraphExecDestroy(c->bgraph_exec); c->bgraph_exec = nullptr; }
    c->bgraph_sig = 0;
    // (r5) segment graphs that served a single registration before the map under them changed were not worth their capture: two such
    // sets in a row and loop_run goes eager until a signature repeats (a mapper rebuilds its map behind every scan)
    if (c->seg_exec[0]) { if (c->seg_uses <= 1) ++c->seg_wasted; else c->seg_wasted = 0; }
    c->seg_uses = 0; ++c->map_epoch;
    for (int g = 0; g < 2; ++g) if (c->seg_exec[g]) { hipGraphExecDestroy(c->seg_exec[g]); c->seg_exec[g] = nullptr; }
    for (auto& hd : c->seg_heads) { if (hd.exec) hipGraphExecDestroy(hd.exec); hd.exec = nullptr; hd.len = 0; }
    c->seg_sig = 0; c->seg_n = -1;
}

// scratch device allocation of one call: freed on every exit path (hipFree waits for work still using it)
template <typename T>
struct DevBuf {
    T* p = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { if (p) dev_free(p); }
    hipError_t alloc(size_t n) { return dev_malloc((void**)&p, (n ? n : 1) * sizeof(T)); }
    operator T*() const { return p; }
};

// Small device -> host read through the pinned page: a copy into pageable memory is staged and blocks for tens of
// microseconds, and a map update makes a dozen of them (counts after every compaction, grid statistics).
#define ICPMI_PIN_BYTES (128 * 1024)
#define ICPMI_PROGRESS_WORDS 512      // size of the host-mapped page in words
#define ICPMI_PROGRESS_HDR_WORD 64    // ... 256 words: the block headers (counts) of a one-collective epoch, one per rank (ops.hip)
#define ICPMI_MERGE_MAGIC 0x49435035u // 'ICP5' in the header's y
#define ICPMI_PROGRESS_OCT_WORD 48  // ... and 8 words for the octree's root cube (octree.hip)
#define ICPMI_PROGRESS_SELF_WORD 56 // ... two words: sum over the A-cells of (points in the cell)^2 of the handle's last self search (selfgrid.hip: the next build tunes its edge with it)
#define ICPMI_PROGRESS_SCAN_WORD 40 // word of the host-mapped progress page (api.hip: h_progress, 64 words) that device_scan_flags_count reports into
static inline icpmi_status read_back2(icpmi_ctx* c, void* dst0, const void* src0, size_t b0, void* dst1, const void* src1, size_t b1)
{
    if (!c->h_pin || b0 + b1 > ICPMI_PIN_BYTES) {
        HIP_TRY(c, hipMemcpyAsync(dst0, src0, b0, hipMemcpyDeviceToHost, c->stream));
        if (b1) HIP_TRY(c, hipMemcpyAsync(dst1, src1, b1, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        return ICPMI_OK;
    }
    HIP_TRY(c, hipMemcpyAsync(c->h_pin, src0, b0, hipMemcpyDeviceToHost, c->stream));
    if (b1) HIP_TRY(c, hipMemcpyAsync(c->h_pin + b0, src1, b1, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    memcpy(dst0, c->h_pin, b0);
    if (b1) memcpy(dst1, c->h_pin + b0, b1);
    return ICPMI_OK;
}
static inline icpmi_status read_back(icpmi_ctx* c, void* dst, const void* src, size_t bytes) { return read_back2(c, dst, src, bytes, nullptr, nullptr, 0); }

// Small host -> device upload (a 4x4, a level table, a counter) through a ring of slots in the pinned page: a copy from PAGEABLE memory is
// staged by the runtime and costs ~40 us of host time each (r3 HIP trace: ~20 of them per map update); from pinned memory it is an async
// DMA of a few microseconds.  The ring lives in the upper half of the page (read-backs use the lower half); 64 slots of 1 KiB: a slot is
// reused only 64 uploads later, and every entry point synchronises its stream long before that.
#define ICPMI_UP_SLOT 1024
#define ICPMI_UP_SLOTS 64
static inline icpmi_status upload_small(icpmi_ctx* c, void* d_dst, const void* h_src, size_t bytes)
{
    if (!c->h_pin || bytes > ICPMI_UP_SLOT) {
        HIP_TRY(c, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, c->stream));
        return ICPMI_OK;
    }
    unsigned char* slot = c->h_pin + ICPMI_PIN_BYTES + (size_t)(c->up_next++ % ICPMI_UP_SLOTS) * ICPMI_UP_SLOT;
    memcpy(slot, h_src, bytes);
    HIP_TRY(c, hipMemcpyAsync(d_dst, slot, bytes, hipMemcpyHostToDevice, c->stream));
    return ICPMI_OK;
}

// slot `k` of the operator scratch, at least `count` entries of T (contents undefined)
template <typename T>
static inline T* scratch_get(icpmi_ctx* c, int k, size_t count)
{
    const size_t need = (count ? count : 1) * sizeof(T);
    if (need > c->scratch_bytes[k] || !c->scratch[k]) {
        if (c->scratch[k]) { (void)hipStreamSynchronize(c->stream); (void)dev_free(c->scratch[k]); c->scratch[k] = nullptr; c->scratch_bytes[k] = 0; }
        const size_t want = 2 * need + 256;
        if (dev_malloc(&c->scratch[k], want) != hipSuccess) { c->scratch[k] = nullptr; c->last_error = "out of device memory (operator scratch)"; return nullptr; }
        c->scratch_bytes[k] = want;
    }
    return (T*)c->scratch[k];
}

// like ensure_cap, but the first `used` entries survive a reallocation
template <typename T>
static inline icpmi_status ensure_cap_keep(icpmi_ctx* c, T** p, size_t* cap, size_t need, size_t used)
{
    if (need <= *cap && *p) return ICPMI_OK;
    const size_t want = 2 * need + 64;
    T* q = nullptr;
    HIP_TRY(c, dev_malloc((void**)&q, want * sizeof(T)));
    if (*p && used) HIP_TRY(c, hipMemcpyAsync(q, *p, used * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
    if (*p) { HIP_TRY(c, hipStreamSynchronize(c->stream)); HIP_TRY(c, dev_free(*p)); }
    *p = q; *cap = want;
    return ICPMI_OK;
}

// ---- device helpers -------------------------------------------------------------------------
__device__ __forceinline__ float3 xf_point(const float* __restrict__ T, float x, float y, float z, float w)
{
    // Same evaluation order as the oracle: column-by-column accumulation with fused multiply-adds.
    float3 o;
    o.x = fmaf(T[12], w, fmaf(T[8], z, fmaf(T[4], y, T[0] * x)));
    o.y = fmaf(T[13], w, fmaf(T[9], z, fmaf(T[5], y, T[1] * x)));
    o.z = fmaf(T[14], w, fmaf(T[10], z, fmaf(T[6], y, T[2] * x)));
    return o;
}

__device__ __forceinline__ float sqdist3(float ax, float ay, float az, float bx, float by, float bz)
{
    const float dx = ax - bx, dy = ay - by, dz = az - bz;
    return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
}

__device__ __forceinline__ unsigned long long pack_key(float d2, unsigned id)
{
    return ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned long long)id;
}

// bounded sorted list of (key = d2 bits << 32 | original index, sorted position): the k best of a k-NN search, in registers
template <int KMAX>
struct KList {
    unsigned long long key[KMAX];
    int sidx[KMAX];
    int k, filled;
    __device__ __forceinline__ void init(int kk)
    {
        k = kk; filled = 0;
#pragma unroll
        for (int i = 0; i < KMAX; ++i) { key[i] = ~0ull; sidx[i] = -1; }
    }
    __device__ __forceinline__ unsigned long long worst() const { return key[KMAX - 1]; }
    // keeps the KMAX smallest; callers read the first k (k <= KMAX), so the list may hold more than
    // k entries -- harmless, and it keeps all indexing static (registers, no scratch).
    __device__ __forceinline__ void insert(unsigned long long kk, int s)
    {
        if (kk >= key[KMAX - 1]) return;
#pragma unroll
        for (int i = KMAX - 1; i >= 0; --i) {
            const unsigned long long prev = i > 0 ? key[i - 1] : 0ull;
            const int prevs = i > 0 ? sidx[i - 1] : -1;
            if (i > 0 && kk < prev) { key[i] = prev; sidx[i] = prevs; }
            else if (kk < key[i]) { key[i] = kk; sidx[i] = s; }
        }
        if (filled < KMAX) ++filled;
    }
};

__device__ inline void quat_from_T(const float* T, double* q)
{
    const double m00 = T[0], m11 = T[5], m22 = T[10];
    double t = m00 + m11 + m22;
    if (t > 0) {
        t = sqrt(t + 1.0);
        q[0] = 0.5 * t; t = 0.5 / t;
        q[1] = ((double)T[4 * 1 + 2] - (double)T[4 * 2 + 1]) * t;
        q[2] = ((double)T[4 * 2 + 0] - (double)T[4 * 0 + 2]) * t;
        q[3] = ((double)T[4 * 0 + 1] - (double)T[4 * 1 + 0]) * t;
    } else {
        int i = 0;
        if (m11 > m00) i = 1;
        if (m22 > (i == 0 ? m00 : m11)) i = 2;
        const int j = (i + 1) % 3, kk = (j + 1) % 3;
        auto M = [&](int r, int c) { return (double)T[4 * c + r]; };
        t = sqrt(M(i, i) - M(j, j) - M(kk, kk) + 1.0);
        __shared__ double v[3]; // (run-time index: LDS instead of scratch memory; one lane per workgroup ever gets here)
        v[i] = 0.5 * t; t = 0.5 / t;
        q[0] = (M(kk, j) - M(j, kk)) * t;
        v[j] = (M(j, i) + M(i, j)) * t;
        v[kk] = (M(kk, i) + M(i, kk)) * t;
        q[1] = v[0]; q[2] = v[1]; q[3] = v[2];
    }
}


// start of a registration: the loop state of one reading (init_state_kernel; the last kernel of the query sort when the head is fused)
__device__ inline void init_state_dev(IcpState* st, const float* T0, unsigned seq, unsigned* progress, const unsigned* seq_src)
{
    // seq_src: the sequence number is read from host-mapped memory at RUN time -- a captured graph must not freeze the number of the
    // registration it was captured for (the host matches the progress word against the number of the registration it is waiting on)
    if (seq_src) seq = __hip_atomic_load(seq_src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    st->seq = seq;
    if (progress) __hip_atomic_store(progress, (seq & 0x7ffffu) << 12, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    for (int i = 0; i < 16; ++i) st->T_iter[i] = T0 ? T0[i] : ((i % 5 == 0) ? 1.f : 0.f);
    st->iter = 0; st->done = 0; st->error = 0; st->stop_reason = 0; st->counter = 0;
    quat_from_T(st->T_iter, st->hq);
    for (int r = 0; r < 3; ++r) st->ht[r] = st->T_iter[12 + r];
    for (int r = 0; r < 4; ++r) st->init_q[r] = st->hq[r];
    st->hist_n = 1;
    st->sel_prefix = 0; st->sel_rank = 0; st->n_valid = 0;
    for (int f = 0; f < ICPMI_MAX_OUTLIER; ++f) st->limits[f] = -1.f;
    st->robust_med = 0.f; st->robust_scale = 1.f; st->vt_valid = 0; st->vt_ratio = -1.f;
    st->pairs = 0; st->wsum = 0; st->hard_count = 0; st->hard_total = 0; st->ticket = 0;
    for (int i = 0; i < 24; ++i) st->dbg[i] = 0;
    st->t_nn_begin = 0; st->t_nn_sum = 0; st->nn_count = 0;
    st->t_done = 0; // (t_start belongs to the first kernel of the head, which runs BEFORE this when the head is folded into the query sort)
}

// The head of a registration folded into the kernels of the query sort (r3: 8 graph nodes -> 3): the first kernel centres the raw
// readings (Mapper.cpp:213's T_refMean_dataIn) while it computes their tile keys, the last one initialises the loop state and clears
// the selection histograms while it scatters.
struct SortHead {
    BatchSrc raw;                 // the readings as handed in; centred into the sort's input array
    float mean[3];
    IcpState* st = nullptr;       // loop state(s) to initialise (one per reading), or nullptr
    unsigned seq = 0;
    unsigned* progress = nullptr;
    const unsigned* seq_src = nullptr;
    unsigned* selhist = nullptr;  // ICPMI_SELHIST_WORDS words per reading to clear, or nullptr
};

// ---- cross-TU host entry points ---------------------------------------------------------------
// keep_prefix: the first keep_prefix points of d_pts are the cloud of the previous build, unchanged and in the same order (an append)
icpmi_status map_build(icpmi_ctx* c, const float4* d_pts, int64_t m, const float* d_normals3, int64_t keep_prefix = 0);
icpmi_status map_patch_normals(icpmi_ctx* c, const unsigned* d_list, int64_t n_list, const float4* d_raw, const float* d_normals3);
icpmi_status upload_level_table(icpmi_ctx* c); // c->levels -> c->d_lvl_tab (the table the NN kernels copy to LDS)
icpmi_status device_exclusive_scan(icpmi_ctx* c, unsigned* data, int n, unsigned total);
icpmi_status device_exclusive_scan_io(icpmi_ctx* c, const unsigned* in, unsigned* out, int n, unsigned total); // in == out allowed
// counts[0..n) -> starts in CURSOR layout (starts[0] = 0, starts[i + 1] = start of cell i, starts[n + 1] = total; n + 2 words): a scatter takes
// its slots with atomicAdd(&starts[key + 1], len) and leaves the plain exclusive scan behind.  zero_counts: counts[0 .. n + 1] end up zero.
icpmi_status device_scan_flags_count(icpmi_ctx* c, const unsigned* flag, unsigned* pos, int n, int64_t* count); // pos = exclusive scan of the 0 / 1 flags, *count = how many are set (one stream wait, no copy)
icpmi_status device_exclusive_scan_sum(icpmi_ctx* c, const unsigned* flag, unsigned* pos, int n, unsigned* d_sum); // ... the count stays on the device
icpmi_status device_exclusive_scan_cursor(icpmi_ctx* c, unsigned* counts, unsigned* starts, int n, unsigned total, bool zero_counts, unsigned* tail_out = nullptr); // tail_out: receives counts[n + 1]
// the same scan on another stream with the caller's own chunk-total words (device_scan_side_words(n) of them); false from
// device_scan_side_ok(n): the table is too large for the two-kernel scan -- stay on the handle's stream
bool device_scan_side_ok(int n);
size_t device_scan_side_words(int n);
icpmi_status device_exclusive_scan_cursor_side(icpmi_ctx* c, hipStream_t stream, unsigned* sums, unsigned* counts, unsigned* starts, int n, unsigned total);
icpmi_status sort_queries(icpmi_ctx* c, const float4* d_pts, int64_t n);
icpmi_status sort_queries_reserve(icpmi_ctx* c, int64_t n, int nscan = 1);
icpmi_status sort_queries_batch(icpmi_ctx* c, const float4* d_pts, const BatchArgs& ba, const SortHead* head = nullptr); // slices of d_pts -> slices of d_qsorted / d_qindex
                                                                                 // (head: d_pts is WRITTEN first, from head->raw minus the mean)
icpmi_status nn_launch_k1(icpmi_ctx* c, const float4* d_reading, int64_t n, const float* d_T, const LoopCfg& lc,
                          int allow_self, int* d_sidx, float* d_d2, IcpState* d_state);
icpmi_status nn_launch_k(icpmi_ctx* c, const float4* d_reading, int64_t n, const float* d_T, const LoopCfg& lc,
                         int allow_self, int* d_sidx, float* d_d2, IcpState* d_state);
icpmi_status nn_ids_to_original(icpmi_ctx* c, const int* d_sidx, int64_t count, int* d_ids);
// self k-NN of a device cloud through a sparse block grid built for the call (selfgrid.hip): rows of d_sidx / d_d2 in the cloud's order, entries =
// positions in c->d_map_sorted (w = original index bits), which the call leaves behind for launch_normals / nn_ids_to_original
// subset search of an appended cloud (selfgrid.hip): in -- m_old, d_dk (d^2 of the k-th neighbour of the first m_old points, original order);
// out -- d_list / n_sel: the searched original indices
struct SelfGridSubset { int64_t m_old = 0; const float* d_dk = nullptr; const unsigned* d_list = nullptr; int64_t n_sel = 0; };
icpmi_status selfgrid_knn(icpmi_ctx* c, const float4* d_pts, int64_t m, int k, int* d_sidx, float* d_d2, SelfGridSubset* sub = nullptr);
void selfgrid_destroy(icpmi_ctx* c);
icpmi_status create_handle(const icpmi_config* cfg, icpmi_handle* out); // icpmi_create without the cache's handle count (private handles)
icpmi_status loop_run(icpmi_ctx* c, const float4* d_scan, const float* d_normals3, int64_t n, const LoopCfg& lc, bool fixed, float T_out[16],
                      icpmi_stats* stats);
icpmi_status loop_sensor_noise_overlap(icpmi_ctx* c, int64_t n, const LoopCfg& lc, bool sorted, float* overlap);
icpmi_status loop_prepare_reading(icpmi_ctx* c, const float4* d_scan, int64_t n, const float* d_normals3, bool* head_done = nullptr);
icpmi_status loop_run_batch(icpmi_ctx* c, int batch, const float* const* d_scans4, const int64_t* n, const LoopCfg& lc, bool fixed,
                            float* T_out, icpmi_stats* stats, icpmi_status* status);
icpmi_status loop_single_step(icpmi_ctx* c, int64_t n, const LoopCfg& lc, const float* T_iter_host, float T_step[16],
                              double sums[32], icpmi_stats* stats);
icpmi_status loop_outlier_weights(icpmi_ctx* c, const LoopCfg& lc, const float* d2, const int32_t* ids, int k, int64_t n,
                                  const float* read_normals3, float* weights, float* limit_out);
LoopCfg make_loop_cfg(const icpmi_ctx* c, int fixed_iterations);

icpmi_status ops_transform(icpmi_ctx* c, const float T[16], const float* in4, int64_t n, float* out4,
                           const float* in_n3, float* out_n3);
icpmi_status ops_map_update_chain(icpmi_ctx* c, const float4* d_scan, int64_t n, const float* d_scan_n3, const float* d_scan_s,
                                  const float to_sensor[16], const float from_sensor[16], const icpmi_map_op* ops, int n_ops, int n_modules,
                                  int32_t* src_out, int64_t src_capacity, int64_t* identity_prefix, int64_t* new_m);
#define ICPMI_MAX_POINT_FILTERS 16
icpmi_status ops_filter_points(icpmi_ctx* c, const float* in4, int64_t n, const icpmi_point_filter* filters, int n_filters, uint8_t* keep);
icpmi_status ops_staged_keep(icpmi_ctx* c, const float correction[16], float min_dist, uint8_t* keep_out, float* placed_out4);
icpmi_status ops_map_scalar(icpmi_ctx* c, const float* set, float* get, int64_t m);
icpmi_status ops_surface_normals(icpmi_ctx* c, const float* pts4, int64_t m, int knn, float* normals3, float* densities = nullptr,
                                 int32_t* matched_ids = nullptr, float* mean_dist = nullptr, float* eig_values = nullptr, float* eig_vectors = nullptr);
icpmi_status ops_dynamic_points_update(icpmi_ctx* c, const icpmi_dynpts_params* prm, const float to_sensor[16], const float* in4, int64_t n,
                                       const float* map4, const float* map_normals3, int64_t m, float* prob);
icpmi_status ops_map_update_point_distance(icpmi_ctx* c, const float* scan4, int64_t n, const float* scan_normals3, float min_dist,
                                           int normals_knn, uint8_t* keep_out, int64_t* appended, int64_t* new_m);
icpmi_status ops_map_update_dev(icpmi_ctx* c, const float4* d_scan, int64_t n, const float* d_scan_n3, float min_dist, int normals_knn,
                                uint8_t* keep_out, int64_t* appended, int64_t* new_m);
icpmi_status ops_transform_dev(icpmi_ctx* c, const float T[16], const float4* d_in, int64_t n, float4* d_out);
icpmi_status ops_get_map(icpmi_ctx* c, float* out4, float* normals3, int64_t capacity, int64_t* m);
icpmi_status ops_voxel_keep_first(icpmi_ctx* c, const float* in4, int64_t n, float edge, int method, uint8_t* keep);
icpmi_status ops_point_distance_keep(icpmi_ctx* c, const float* map4, int64_t m, const float* in4, int64_t n,
                                     float min_dist, uint8_t* keep);
icpmi_status ops_bin_cells(icpmi_ctx* c, const float* pts4, int64_t n, float cell_size, int32_t* ijk3);
icpmi_status comm_unique_id(icpmi_comm_id* id, std::string& err);
icpmi_status comm_init(icpmi_ctx* c, const icpmi_comm_id* id, int n_ranks, int rank);
icpmi_status comm_destroy(icpmi_ctx* c);
icpmi_status comm_info(icpmi_ctx* c, int* n_ranks, int* rank, int* kind);
icpmi_status comm_allgather(icpmi_ctx* c, const void* d_send, void* d_recv, size_t count, bool is_float);
// all-gather of one fixed-size block of `block4` float4 per rank (element 0 = header: x = bits(count) or bits(-1)); the loopback
// communicator builds its simulated ranks' blocks (shifted points, ragged counts) in one launch.  No communicator: d_recv = a copy of d_send.
icpmi_status comm_allgather_blocks(icpmi_ctx* c, const float4* d_send, float4* d_recv, size_t block4);
icpmi_status merge_blocks_reserve(icpmi_ctx* c, int n_ranks); // the exchange buffers of the one-collective epoch (called by comm_init)
icpmi_status ops_staged_merge_allgather(icpmi_ctx* c, const float correction[16], float min_dist, int normals_knn, int64_t* accepted_local,
                                        int64_t* appended_total, int64_t* new_m, float* merged_out4, int64_t merged_capacity, int64_t* merged_n);
icpmi_status ops_staged_merged_points(icpmi_ctx* c, float* out4, int64_t capacity, int64_t* n);
icpmi_status ops_staged_bin_cells(icpmi_ctx* c, float cell_size, int32_t* ijk3, int64_t* offsets, int64_t* counts, int64_t capacity, int64_t* n_cells);
icpmi_status ops_cell_log_read(icpmi_ctx* c, int64_t offset, int64_t count, float* out4, int64_t* log_size);
icpmi_status ops_cell_log_clear(icpmi_ctx* c);
icpmi_status ops_cells_enqueue_in_epoch(icpmi_ctx* c);
icpmi_status ssn_debug_minstd(icpmi_ctx* c, unsigned seed, unsigned n, unsigned* out);
icpmi_status ssn_sample_dev(icpmi_ctx* c, const float4* d_in, int64_t n, float ratio, int knn, float max_box, unsigned seed, int* d_order_out,
                            float* d_normals_out, int64_t* n_out, int method = 0, float* d_mean_out = nullptr, int* d_mstart_out = nullptr,
                            int* d_mcount_out = nullptr, int* d_members_out = nullptr, int64_t* n_members_out = nullptr);
icpmi_status ops_sampling_surface_normal_ex(icpmi_ctx* c, const float* in4, int64_t n, float ratio, int knn, float max_box, int seed, int method,
                                            int32_t* order_out, float* normals3_out, int64_t* n_out, float* mean3_out, int32_t* mstart_out,
                                            int32_t* mcount_out, int32_t* members_out);
icpmi_status ops_sampling_surface_normal(icpmi_ctx* c, const float* in4, int64_t n, float ratio, int knn, float max_box, int seed, int32_t* order_out,
                                         float* normals3_out, int64_t* n_out);
size_t radix_sort_tab_words(int64_t n, int bits);
icpmi_status radix_sort_pairs(icpmi_ctx* c, unsigned long long* d_keys2, unsigned* d_vals2, int64_t n, int bits, unsigned* d_tab, int* result_half);
icpmi_status octree_sample_dev(icpmi_ctx* c, const float4* d_in, int64_t n, float max_size, int max_pts, int method, int* d_order, int* d_leaf_of,
                               int64_t* n_out);
icpmi_status ops_octree_sample(icpmi_ctx* c, const float* in4, int64_t n, float max_size, int max_pts, int method, int32_t* order_out,
                               int32_t* leaf_of_out, int64_t* n_out);
