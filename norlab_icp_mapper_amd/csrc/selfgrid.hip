// selfgrid.hip -- self k-NN of a cloud (SurfaceNormalDataPointsFilter over the whole map: reference norlab_icp_mapper/Map.cpp:523-525 applies
// the `post:` filters to the local map on every update, examples/config.yaml:25-27 `SurfaceNormalDataPointsFilter: knn: 10`; upstream body:
// libpointmatcher DataPointsFilters/SurfaceNormal.cpp -> Nabo::NNS::knn of the cloud against itself) through a SPARSE BLOCK GRID.
//
// Why not the dense single-level grid of r2 - r5 (nn.hip: nnk_self_tiled_kernel + ring kernel + brute pass): a vehicle's lidar map is dense
// along the trajectory and sparse far out (BASELINE config 4: 104 k points in a 283 x 349 x 22 m box, 10-th neighbour at 0.3 m for the median
// point, 2.3 m for the 99-th percentile, 25 m for the last) -- one cell edge is wrong for most of it, the cell table of the bounding box is
// 99.7 % empty (8 M cells: its scan alone was 60 us), 6 - 8 % of the points went through up to six rings and 400 - 500 through a brute pass
// over the whole map: 457 us per update.
//
// Structure (one build per call, 6 launches):
//   * A-cells of edge `cell`, grouped 4 x 4 x 4 into B-blocks; a dense table T over the B-cells in MORTON order (bits of the three axes
//     interleaved, an axis drops out when it runs out of bits: a flat map's z), and for every OCCUPIED block a row of 65 words in F: the
//     starts of its 64 A-cells (Morton order inside the block) + its end.  T costs 1 / 64 of a dense A-table (config 4: 1 M words), F only
//     exists where points are.
//   * the points are sorted by (Morton index of the block, Morton index of the A-cell inside it): EVERY level of the octree over the A-cells
//     -- cells of edge cell * 2^l -- is one contiguous run of the sorted array, found with two look-ups (l <= 2: in F, l > 2: in T).
// Search (2 launches):
//   * sg_tiled_kernel: one wave per occupied block stages the 6 x 6 x 6 A-cells around it into LDS (216 look-ups, then coalesced runs) and
//     every point of the block scans its own 3 x 3 x 3 cells from there.  Decided when the k-th distance fits the margin to the block
//     boundary (faces on the border of the grid do not count: nothing lies beyond them).  Blocks with fewer than `tau` points skip this pass.
//   * sg_wave_kernel: the rest (config 4: ~15 %), one wave per query, level after level (edge x 2): the 3 x 3 x 3 block of level l is visited
//     as its 216 cells of level l - 1, those farther than the k-th distance found one level down are not looked up at all, the candidates of
//     the others are dealt to the lanes 64 at a time.  The top level covers the grid: every query ends here, there is no brute pass.
// Exactness: both kernels return the k smallest keys (d^2 bits << 32 | original index) over candidate sets that contain every point within the
// k-th distance -- the same k points in the same order whatever the grid (the oracle's and r5's bits: tests/test_gpu_parity.py, test_pins.py,
// test_gpu_map_chain.py, test_gpu_configs.py).
#include "common.h"

#include <algorithm>
#include <cmath>
#include <vector>

struct SelfGridCtx {
    unsigned* d_tcnt = nullptr; size_t cap_tcnt = 0; bool tcnt_clean = false; // points per B-cell (zero between builds: the scan cleans what it read)
    unsigned* d_tstart = nullptr; size_t cap_tstart = 0;                      // exclusive scan of d_tcnt, Morton order (+ 2 words: cursor layout while scattering)
    unsigned* d_tbid = nullptr; size_t cap_tbid = 0;                          // B-cell -> block id (valid where the cell is occupied)
    uint4* d_blist = nullptr; size_t cap_blist = 0;                           // block id -> {Morton index, bx, by, bz}
    unsigned* d_f = nullptr; size_t cap_f = 0;                                // SG_F words per block
    float4* d_coarse = nullptr; size_t cap_coarse = 0;                        // the points sorted by block (before the sort inside the blocks)
    uint2* d_queue = nullptr; size_t cap_queue = 0;                           // {sorted position, bits of the bound on the k-th d^2} of the queries the tiled pass left
    float* d_part = nullptr; size_t cap_part = 0;                             // bounding-box partials
    struct SgState* d_state = nullptr;
    // tuning state: the edge of the previous build and what its points saw (size-biased A-cell occupancy, delivered through the mapped page)
    float cell = 0.f; int64_t m = 0; int k = 0;
    unsigned long long seq = 0;
};

struct SgState {
    unsigned nblk;   // occupied blocks (claimed by the key kernel)
    unsigned qcount; // queries queued for the wave kernel
    unsigned bad;    // non-finite coordinates seen by the bounding-box pass
    unsigned pad;
    unsigned long long sq; // sum over the A-cells of (points in the cell)^2
    unsigned long long levels; // diagnostics: sum of the levels at which the wave kernel's queries ended
    unsigned long long tick[16]; // -DSG_TIMING: phase clocks of the search kernels (sampled workgroups)
};

namespace {

constexpr int SG_F = 65;   // words of a block's row of F
constexpr int SG_CH = 384; // candidates staged per chunk (LDS tile)

struct SgGrid {
    float ox, oy, oz, cell, inv_cell, maxabs;
    int na[3];        // A-cells per axis
    int nbits[3];     // bits of a B-cell coordinate per axis
    unsigned mask[3]; // where those bits go in the Morton index
    int tsize;        // entries of T = 1 << (nbits[0] + nbits[1] + nbits[2])
};

__device__ __forceinline__ int sg_cell_of(float v, float o, float inv, int n)
{
    int c = (int)floorf((v - o) * inv);
    return c < 0 ? 0 : (c >= n ? n - 1 : c);
}
// bits of v into the set positions of mask, lowest first (a software pdep; masks carry at most ~10 bits)
__device__ __forceinline__ unsigned sg_dep(unsigned v, unsigned mask)
{
    unsigned r = 0;
    while (mask) {
        const unsigned low = mask & (0u - mask);
        r |= (v & 1u) ? low : 0u;
        v >>= 1; mask ^= low;
    }
    return r;
}
__device__ __forceinline__ unsigned sg_hb(const SgGrid& g, int bx, int by, int bz)
{
    return sg_dep((unsigned)bx, g.mask[0]) | sg_dep((unsigned)by, g.mask[1]) | sg_dep((unsigned)bz, g.mask[2]);
}
// Morton index of an A-cell inside its block (2-bit coordinates; x lowest)
__device__ __forceinline__ unsigned sg_sub(int x, int y, int z)
{
    return (unsigned)((x & 1) | ((y & 1) << 1) | ((z & 1) << 2) | ((x & 2) << 2) | ((y & 2) << 3) | ((z & 2) << 4));
}

// the points of the level-l cell (cx, cy, cz) -- edge cell * 2^l, coordinates in cells of that level -- as a run [s, e) of the sorted array;
// empty outside the grid
__device__ __forceinline__ void sg_cell_range(const SgGrid& g, const unsigned* __restrict__ tstart, const unsigned* __restrict__ tbid,
                                              const unsigned* __restrict__ f, int l, int cx, int cy, int cz, unsigned& s, unsigned& e)
{
    s = 0; e = 0;
    if (cx < 0 || cy < 0 || cz < 0) return;
    if (l <= 2) {
        const int sh = 2 - l;
        const int bx = cx >> sh, by = cy >> sh, bz = cz >> sh;
        if (bx >= (1 << g.nbits[0]) || by >= (1 << g.nbits[1]) || bz >= (1 << g.nbits[2])) return;
        const unsigned hb = sg_hb(g, bx, by, bz);
        const unsigned t0 = tstart[hb], t1 = tstart[hb + 1];
        const unsigned bid = tbid[hb]; // (garbage where the cell is empty: not used then)
        if (t1 == t0) return;
        const int mk = (1 << sh) - 1;
        const unsigned sub0 = sg_sub((cx & mk) << l, (cy & mk) << l, (cz & mk) << l);
        const unsigned* row = f + (size_t)bid * SG_F;
        s = row[sub0]; e = row[sub0 + (1u << (3 * l))];
        return;
    }
    const int sh = l - 2;
    const int c3[3] = {cx, cy, cz};
    int b0[3], gbits = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int bits = g.nbits[a];
        if (sh >= bits) { if (c3[a] != 0) return; b0[a] = 0; gbits += bits; } // the cell spans the whole axis
        else { if (c3[a] >= (1 << (bits - sh))) return; b0[a] = c3[a] << sh; gbits += sh; }
    }
    const unsigned hb0 = sg_hb(g, b0[0], b0[1], b0[2]);
    s = tstart[hb0]; e = tstart[hb0 + (1u << gbits)];
}

// squared radius around a query inside which the 3 x 3 x 3 block of level-l cells around it holds every point of the cloud; +inf: the block
// reaches the border of the grid on every side that matters (nothing lies beyond).  a: the query's A-cell, fa: its position inside it (cells).
__device__ __forceinline__ float sg_margin2(const SgGrid& g, int l, const int a[3], const float fa[3])
{
    const float cl = g.cell * (float)(1u << l);
    float mn = INFINITY;
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        const int c = a[ax] >> l, nl = ((g.na[ax] - 1) >> l) + 1;
        const float fr = ((float)(a[ax] & ((1 << l) - 1)) + fa[ax]) / (float)(1u << l);
        if (c >= 2) mn = fminf(mn, (1.0f + fr) * cl);          // cells below the block exist
        if (c + 2 <= nl - 1) mn = fminf(mn, (2.0f - fr) * cl); // cells above it exist
    }
    if (mn == INFINITY) return INFINITY;
    const float margin = fmaxf(mn - (cl * 1e-3f + g.maxabs * 2e-6f), 0.f);
    return margin * margin;
}

__device__ __forceinline__ unsigned sg_wave_incl_scan(unsigned v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned t = (unsigned)__shfl_up((int)v, o, 64); if (lane >= o) v += t; }
    return v;
}

// ---- build ------------------------------------------------------------------------------------------------------------------------------
constexpr int SG_RB = 256;
__global__ __launch_bounds__(SG_RB) void sg_bbox_kernel(const float4* __restrict__ pts, int64_t m, float* __restrict__ part, SgState* __restrict__ st)
{
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * SG_RB + threadIdx.x; i < m; i += (int64_t)gridDim.x * SG_RB) {
        const float4 p = pts[i];
        bad |= !(fabsf(p.x) <= 3.0e38f) || !(fabsf(p.y) <= 3.0e38f) || !(fabsf(p.z) <= 3.0e38f);
        lo[0] = fminf(lo[0], p.x); lo[1] = fminf(lo[1], p.y); lo[2] = fminf(lo[2], p.z);
        hi[0] = fmaxf(hi[0], p.x); hi[1] = fmaxf(hi[1], p.y); hi[2] = fmaxf(hi[2], p.z);
    }
    __shared__ float sl[3][SG_RB], sh[3][SG_RB];
    const int t = threadIdx.x;
    for (int r = 0; r < 3; ++r) { sl[r][t] = lo[r]; sh[r][t] = hi[r]; }
    __syncthreads();
    for (int s = SG_RB / 2; s > 0; s >>= 1) {
        if (t < s)
            for (int r = 0; r < 3; ++r) { sl[r][t] = fminf(sl[r][t], sl[r][t + s]); sh[r][t] = fmaxf(sh[r][t], sh[r][t + s]); }
        __syncthreads();
    }
    if (t == 0) for (int r = 0; r < 3; ++r) { part[6 * blockIdx.x + r] = sl[r][0]; part[6 * blockIdx.x + 3 + r] = sh[r][0]; }
    if (__ballot(bad) != 0ull && (t & 63) == 0) atomicOr(&st->bad, 1u);
}

__global__ void sg_reset_kernel(SgState* st)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) { st->nblk = 0; st->qcount = 0; st->bad = 0; st->pad = 0; st->sq = 0ull; st->levels = 0ull; for (int i = 0; i < 16; ++i) st->tick[i] = 0ull; }
}

// per point: key = Morton index of its block << 6 | Morton index of its A-cell in the block; points per block counted (one atomic per run of
// equal keys in a wave); the lane that finds a block's counter at zero claims a block id for it
__global__ __launch_bounds__(256) void sg_key_kernel(const float4* __restrict__ pts, int64_t m, SgGrid g, unsigned* __restrict__ keys,
                                                     unsigned* __restrict__ tcnt, unsigned* __restrict__ tbid, uint4* __restrict__ blist,
                                                     SgState* __restrict__ st)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool valid = i < m;
    const float4 p = pts[valid ? i : 0];
    const int ax = sg_cell_of(p.x, g.ox, g.inv_cell, g.na[0]);
    const int ay = sg_cell_of(p.y, g.oy, g.inv_cell, g.na[1]);
    const int az = sg_cell_of(p.z, g.oz, g.inv_cell, g.na[2]);
    const int bx = ax >> 2, by = ay >> 2, bz = az >> 2;
    const unsigned hb = sg_hb(g, bx, by, bz);
    if (valid) keys[i] = (hb << 6) | sg_sub(ax & 3, ay & 3, az & 3);
    const WaveRun r = wave_run(hb, valid);
    const bool first = r.head && atomicAdd(&tcnt[hb], (unsigned)r.len) == 0u;
    // block ids: per wave by ballot, per workgroup in LDS, one global atomic per workgroup
    __shared__ unsigned wg_count, wg_base;
    if (threadIdx.x == 0) wg_count = 0;
    __syncthreads();
    const unsigned long long firsts = __ballot(first);
    const int lane = threadIdx.x & 63;
    unsigned wave_off = 0;
    if (lane == 0 && firsts) wave_off = atomicAdd(&wg_count, (unsigned)__popcll(firsts));
    wave_off = (unsigned)__shfl((int)wave_off, 0, 64);
    __syncthreads();
    if (threadIdx.x == 0 && wg_count) wg_base = atomicAdd(&st->nblk, wg_count);
    __syncthreads();
    if (first) {
        const unsigned bid = wg_base + wave_off + (unsigned)__popcll(firsts & ((1ull << lane) - 1ull));
        tbid[hb] = bid;
        blist[bid] = make_uint4(hb, (unsigned)bx, (unsigned)by, (unsigned)bz);
    }
}

// points into block order (any order inside a block): cursor scatter on T (tstart + 1: common.h, device_exclusive_scan_cursor)
__global__ __launch_bounds__(256) void sg_scatter_kernel(const float4* __restrict__ pts, int64_t m, const unsigned* __restrict__ keys,
                                                         unsigned* __restrict__ cursor, float4* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool valid = i < m;
    const float4 p = pts[valid ? i : 0];
    const unsigned hb = keys[valid ? i : 0] >> 6;
    const WaveRun r = wave_run(hb, valid);
    unsigned base = 0;
    if (r.head) base = atomicAdd(&cursor[hb], (unsigned)r.len);
    base = (unsigned)__shfl((int)base, r.head_lane, 64);
    if (!valid) return;
    out[base + (unsigned)r.rank] = make_float4(p.x, p.y, p.z, __uint_as_float((unsigned)i));
}

// one wave per occupied block: its points into the Morton order of their A-cells (a counting sort in LDS; the order inside a cell is whatever
// the atomics give -- every comparison downstream is on (d^2, original index)), its row of F, its share of the sum of squared cell counts
__global__ __launch_bounds__(64) void sg_block_sort_kernel(const float4* __restrict__ in, float4* __restrict__ out, SgGrid g,
                                                           const unsigned* __restrict__ tstart, const uint4* __restrict__ blist,
                                                           unsigned* __restrict__ f, SgState* __restrict__ st)
{
    __shared__ unsigned cnt[64], cur[64];
    const int lane = threadIdx.x;
    const unsigned nblk = st->nblk;
    unsigned long long sq = 0;
    for (unsigned b = blockIdx.x; b < nblk; b += gridDim.x) {
        const unsigned hb = blist[b].x;
        const unsigned s0 = tstart[hb], e0 = tstart[hb + 1], n = e0 - s0;
        cnt[lane] = 0;
        __syncthreads();
        float4 p0 = make_float4(0.f, 0.f, 0.f, 0.f);
        unsigned sub0 = 0;
        for (unsigned i = lane; i < n; i += 64) {
            const float4 p = in[s0 + i];
            const unsigned sub = sg_sub(sg_cell_of(p.x, g.ox, g.inv_cell, g.na[0]) & 3, sg_cell_of(p.y, g.oy, g.inv_cell, g.na[1]) & 3,
                                        sg_cell_of(p.z, g.oz, g.inv_cell, g.na[2]) & 3);
            if (i < 64) { p0 = p; sub0 = sub; }
            atomicAdd(&cnt[sub], 1u);
        }
        __syncthreads();
        const unsigned c = cnt[lane];
        const unsigned off = sg_wave_incl_scan(c) - c;
        f[(size_t)b * SG_F + lane] = s0 + off;
        if (lane == 0) f[(size_t)b * SG_F + 64] = e0;
        cur[lane] = off;
        sq += (unsigned long long)c * c;
        __syncthreads();
        for (unsigned i = lane; i < n; i += 64) {
            float4 p = p0;
            unsigned sub = sub0;
            if (i >= 64) {
                p = in[s0 + i];
                sub = sg_sub(sg_cell_of(p.x, g.ox, g.inv_cell, g.na[0]) & 3, sg_cell_of(p.y, g.oy, g.inv_cell, g.na[1]) & 3,
                             sg_cell_of(p.z, g.oz, g.inv_cell, g.na[2]) & 3);
            }
            const unsigned r = atomicAdd(&cur[sub], 1u);
            out[s0 + r] = p;
        }
        __syncthreads();
    }
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
    if (lane == 0 && sq) atomicAdd(&st->sq, sq);
}

// ---- search -----------------------------------------------------------------------------------------------------------------------------
// A query is its sorted position; what leaves the tiled pass undecided is queued with the k-th distance it did find (a bound on the answer)
__device__ __forceinline__ void sg_queue_push(SgState* st, uint2* __restrict__ queue, unsigned pos, unsigned bound_bits, bool push)
{
    const unsigned long long who = __ballot(push);
    if (!who) return;
    const int lane = threadIdx.x & 63;
    unsigned base = 0;
    if (lane == __ffsll((long long)who) - 1) base = atomicAdd(&st->qcount, (unsigned)__popcll(who));
    base = (unsigned)__shfl((int)base, __ffsll((long long)who) - 1, 64);
    if (push) queue[base + (unsigned)__popcll(who & ((1ull << lane) - 1ull))] = make_uint2(pos, bound_bits);
}

template <int KMAX, int SELF_Q>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(KMAX <= 10 ? 4 : 1))) void sg_tiled_kernel(
    SgGrid g, const float4* __restrict__ map, const unsigned* __restrict__ tstart, const unsigned* __restrict__ tbid, const unsigned* __restrict__ f,
    const uint4* __restrict__ blist, SgState* __restrict__ st, int k, unsigned tau, int* __restrict__ out_sidx, float* __restrict__ out_d2,
    uint2* __restrict__ queue, unsigned long long* __restrict__ sq_mapped)
{
    __shared__ int nb_bid[27];
    __shared__ unsigned cstart[216], coff[217];
    __shared__ float4 tile[SG_CH];
    __shared__ unsigned short qslot[SELF_Q][64];
    const int lane = threadIdx.x;
    const unsigned nblk = st->nblk;
    // what the points of this build see (the block sort is done): to the host-mapped page, where the NEXT build of this handle reads it
    if (blockIdx.x == 0 && lane == 0 && sq_mapped) *sq_mapped = st->sq;
    const unsigned INF_BITS = 0x7f800000u;
#ifdef SG_TIMING
    long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev = clock64();
#define SG_TICK(i) do { const long long t_now = clock64(); tk[i] += t_now - t_prev; t_prev = t_now; } while (0)
#else
#define SG_TICK(i) do { } while (0)
#endif
    for (unsigned b = blockIdx.x; b < nblk; b += gridDim.x) {
        __syncthreads(); // (LDS of the previous block fully consumed)
        SG_TICK(7);
        const uint4 bi = blist[b];
        const int bx = (int)bi.y, by = (int)bi.z, bz = (int)bi.w;
        const unsigned qs = tstart[bi.x], qe = tstart[bi.x + 1];
        if (qe - qs < tau) { // a sparse block: its points are not decided by cells this small -- straight to the levels
            for (unsigned q0 = qs; q0 < qe; q0 += 64) sg_queue_push(st, queue, q0 + lane, INF_BITS, q0 + lane < qe);
            continue;
        }
        if (lane < 27) {
            const int nx = bx + lane % 3 - 1, ny = by + (lane / 3) % 3 - 1, nz = bz + lane / 9 - 1;
            int bid = -1;
            if (nx >= 0 && ny >= 0 && nz >= 0 && nx < (1 << g.nbits[0]) && ny < (1 << g.nbits[1]) && nz < (1 << g.nbits[2])) {
                const unsigned hb = sg_hb(g, nx, ny, nz);
                const unsigned t0 = tstart[hb], t1 = tstart[hb + 1];
                const unsigned v = tbid[hb];
                if (t1 > t0) bid = (int)v;
            }
            nb_bid[lane] = bid;
        }
        __syncthreads();
        // the 6 x 6 x 6 A-cells around the block, x fastest: start and count of each, then the flat offsets
        unsigned total = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int idx = lane + 64 * j;
            unsigned cn = 0, cs0 = 0;
            if (idx < 216) {
                const int lx = idx % 6, ly = (idx / 6) % 6, lz = idx / 36;
                const int bidn = nb_bid[((lx + 3) >> 2) + 3 * ((ly + 3) >> 2) + 9 * ((lz + 3) >> 2)];
                if (bidn >= 0) {
                    const unsigned sub = sg_sub((4 * bx - 1 + lx) & 3, (4 * by - 1 + ly) & 3, (4 * bz - 1 + lz) & 3);
                    const unsigned* row = f + (size_t)bidn * SG_F;
                    cs0 = row[sub]; cn = row[sub + 1] - cs0;
                }
            }
            const unsigned incl = sg_wave_incl_scan(cn);
            if (idx < 216) { cstart[idx] = cs0; coff[idx + 1] = total + incl; }
            total += (unsigned)__shfl((int)incl, 63, 64);
        }
        if (lane == 0) coff[0] = 0;
        __syncthreads();
        SG_TICK(0);
        const unsigned ncand = total;
        if (ncand < (unsigned)k) { // fewer than k points in the whole neighbourhood
            for (unsigned q0 = qs; q0 < qe; q0 += 64) sg_queue_push(st, queue, q0 + lane, INF_BITS, q0 + lane < qe);
            continue;
        }
        auto stage = [&](unsigned c0, unsigned cn) {
            for (unsigned i = lane; i < cn; i += 64) {
                const unsigned fl = c0 + i;
                int lo = 0, hi = 216; // the cell of flat candidate fl: the last offset <= fl
#pragma unroll
                for (int it = 0; it < 8; ++it) { const int mid = (lo + hi) >> 1; if (coff[mid] <= fl) lo = mid; else hi = mid; }
                tile[i] = map[cstart[lo] + (fl - coff[lo])];
            }
        };
        const bool single = ncand <= (unsigned)SG_CH;
        if (single) { stage(0u, ncand); __syncthreads(); }
        SG_TICK(1);
        for (unsigned q0 = qs; q0 < qe; q0 += 64) {
            const unsigned qi = q0 + lane;
            const bool active = qi < qe;
            const float4 me = map[active ? qi : qs];
            const float fx = (me.x - g.ox) * g.inv_cell, fy = (me.y - g.oy) * g.inv_cell, fz = (me.z - g.oz) * g.inv_cell;
            const int a3[3] = {sg_cell_of(me.x, g.ox, g.inv_cell, g.na[0]), sg_cell_of(me.y, g.oy, g.inv_cell, g.na[1]), sg_cell_of(me.z, g.oz, g.inv_cell, g.na[2])};
            const float fa[3] = {fminf(fmaxf(fx - (float)a3[0], 0.f), 1.f), fminf(fmaxf(fy - (float)a3[1], 0.f), 1.f), fminf(fmaxf(fz - (float)a3[2], 0.f), 1.f)};
            const int lx = a3[0] - (4 * bx - 1), ly = a3[1] - (4 * by - 1), lz = a3[2] - (4 * bz - 1); // 1 .. 4
            KList<KMAX> L; L.init(k);
            for (unsigned c0 = 0; c0 < ncand; c0 += SG_CH) {
                const unsigned cn = min((unsigned)SG_CH, ncand - c0);
                if (!single) { __syncthreads(); stage(c0, cn); __syncthreads(); SG_TICK(1); }
                if (active) {
                    // a candidate that beats the lane's k-th best is only QUEUED (its tile slot); the sorted insertion -- ~70 instructions the
                    // whole wave pays whenever one lane inserts -- runs for all lanes together when a queue is full and at the end of the chunk
                    int qn = 0;
                    auto drain = [&]() {
                        for (int t = 0; t < SELF_Q; ++t) {
                            if (__ballot(t < qn) == 0ull) break;
#ifdef SG_TIMING
                            ++tk[4];
#endif
                            if (t < qn) {
                                const unsigned ti = qslot[t][lane];
                                const float4 q = tile[ti];
                                const float d2 = sqdist3(me.x, me.y, me.z, q.x, q.y, q.z);
                                L.insert(pack_key(d2, __float_as_uint(q.w)), (int)(c0 + ti)); // flat index; a map position below
                            }
                        }
                        qn = 0;
                    };
#pragma unroll 1
                    for (int r = 0; r < 9; ++r) {
                        const int rb = ((lz + r / 3 - 1) * 6 + (ly + r % 3 - 1)) * 6 + (lx - 1);
                        const unsigned lo = max(coff[rb], c0), hi = min(coff[rb + 3], c0 + cn);
                        for (unsigned i = lo; i < hi; ++i) {
#ifdef SG_TIMING
                            ++tk[5];
#endif
                            const float4 q = tile[i - c0];
                            const float d2 = sqdist3(me.x, me.y, me.z, q.x, q.y, q.z);
                            if (pack_key(d2, __float_as_uint(q.w)) < L.worst()) { qslot[qn][lane] = (unsigned short)(i - c0); ++qn; }
                            if (__ballot(qn == SELF_Q) != 0ull) drain();
                        }
                    }
                    drain();
                }
                SG_TICK(2);
            }
            // exactness: every point within the margin of the query lies in its 3 x 3 x 3 cells
            const float m2 = sg_margin2(g, 0, a3, fa);
            unsigned long long kth = ~0ull;
#pragma unroll
            for (int i = 0; i < KMAX; ++i) if (i == k - 1) kth = L.key[i];
            const float kd2 = __uint_as_float((unsigned)(kth >> 32));
            const bool decided = active && (m2 == INFINITY || (kth != ~0ull && kd2 <= m2));
            if (decided) {
                const unsigned orig = __float_as_uint(me.w);
#pragma unroll
                for (int j = 0; j < KMAX; ++j)
                    if (j < k) {
                        int pos = -1;
                        if (L.sidx[j] >= 0) {
                            const unsigned fl = (unsigned)L.sidx[j];
                            int lo = 0, hi = 216;
#pragma unroll
                            for (int it = 0; it < 8; ++it) { const int mid = (lo + hi) >> 1; if (coff[mid] <= fl) lo = mid; else hi = mid; }
                            pos = (int)(cstart[lo] + (fl - coff[lo]));
                        }
                        out_sidx[(size_t)k * orig + j] = pos;
                        out_d2[(size_t)k * orig + j] = pos < 0 ? INFINITY : __uint_as_float((unsigned)(L.key[j] >> 32));
                    }
            }
            sg_queue_push(st, queue, qi, kth != ~0ull ? (unsigned)(kth >> 32) : INF_BITS, active && !decided);
            SG_TICK(3);
        }
    }
#ifdef SG_TIMING
    if (lane == 0 && (blockIdx.x % 16) == 0) { for (int i = 0; i < 8; ++i) atomicAdd(&st->tick[i], (unsigned long long)tk[i]); }
#endif
#undef SG_TICK
}

// One wave per queued query, level after level.  At level l the 3 x 3 x 3 block of cells of edge cell * 2^l around the query is visited as
// its 6 x 6 x 6 cells of level l - 1: the lanes take four of them each, drop those farther than the bound (the k-th distance one level down,
// or the tiled pass's), look the others up, and the candidates of all runs are dealt to the lanes 64 at a time (a flat index over the runs'
// prefix sums, in LDS).  Every lane keeps the KMAX best of its share; k rounds of "extract the wave minimum" merge them.  The list starts
// over at every level (no point is seen twice inside one) and only admits d^2 <= bound.
template <int KMAX>
__global__ __launch_bounds__(64) void sg_wave_kernel(SgGrid g, const float4* __restrict__ map, const unsigned* __restrict__ tstart,
                                                     const unsigned* __restrict__ tbid, const unsigned* __restrict__ f, SgState* __restrict__ st,
                                                     int k, int* __restrict__ out_sidx, float* __restrict__ out_d2, const uint2* __restrict__ queue)
{
    __shared__ unsigned rs[256], rp[257];
    const int lane = threadIdx.x;
    const unsigned count = st->qcount;
    unsigned long long lev_sum = 0;
    for (unsigned w = blockIdx.x; w < count; w += gridDim.x) {
        const uint2 qe = queue[w];
        const float4 me = map[qe.x];
        float bound = __uint_as_float(qe.y);
        const float fx = (me.x - g.ox) * g.inv_cell, fy = (me.y - g.oy) * g.inv_cell, fz = (me.z - g.oz) * g.inv_cell;
        const int a3[3] = {sg_cell_of(me.x, g.ox, g.inv_cell, g.na[0]), sg_cell_of(me.y, g.oy, g.inv_cell, g.na[1]), sg_cell_of(me.z, g.oz, g.inv_cell, g.na[2])};
        const float fa[3] = {fminf(fmaxf(fx - (float)a3[0], 0.f), 1.f), fminf(fmaxf(fy - (float)a3[1], 0.f), 1.f), fminf(fmaxf(fz - (float)a3[2], 0.f), 1.f)};
        KList<KMAX> Gl; Gl.init(k);
        int l = 1;
        for (;; ++l) {
            const float m2 = sg_margin2(g, l, a3, fa);
            const float cs = g.cell * (float)(1u << (l - 1));      // edge of the cells looked up at this level
            const float slack = cs * 2e-3f + g.maxabs * 2e-6f;
            const int s0x = ((a3[0] >> l) - 1) * 2, s0y = ((a3[1] >> l) - 1) * 2, s0z = ((a3[2] >> l) - 1) * 2;
            __syncthreads(); // (rs / rp of the previous level consumed)
            unsigned total = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int idx = lane + 64 * j;
                unsigned s = 0, e = 0;
                if (idx < 216) {
                    const int sx = s0x + idx % 6, sy = s0y + (idx / 6) % 6, sz = s0z + idx / 36;
                    // distance from the query to the cell's box, every gap shortened by the slack (a point may sit a rounding outside its cell)
                    const float lox = g.ox + (float)sx * cs, loy = g.oy + (float)sy * cs, loz = g.oz + (float)sz * cs;
                    const float gx = fmaxf(fmaxf(lox - me.x, me.x - (lox + cs)) - slack, 0.f);
                    const float gy = fmaxf(fmaxf(loy - me.y, me.y - (loy + cs)) - slack, 0.f);
                    const float gz = fmaxf(fmaxf(loz - me.z, me.z - (loz + cs)) - slack, 0.f);
                    if (!(fmaf(gz, gz, fmaf(gy, gy, gx * gx)) > bound)) sg_cell_range(g, tstart, tbid, f, l - 1, sx, sy, sz, s, e);
                }
                const unsigned len = e - s;
                const unsigned incl = sg_wave_incl_scan(len);
                if (idx < 216) { rs[idx] = s; rp[idx + 1] = total + incl; }
                total += (unsigned)__shfl((int)incl, 63, 64);
            }
            if (lane == 0) rp[0] = 0;
            __syncthreads();
            KList<KMAX> Lc; Lc.init(k);
            for (unsigned base = 0; base < total; base += 128u) {
                float4 q[2];
                unsigned pos[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const unsigned fl = base + 64u * u + (unsigned)lane;
                    pos[u] = ~0u;
                    if (fl < total) {
                        int lo = 0, hi = 216;
#pragma unroll
                        for (int it = 0; it < 8; ++it) { const int mid = (lo + hi) >> 1; if (rp[mid] <= fl) lo = mid; else hi = mid; }
                        pos[u] = rs[lo] + (fl - rp[lo]);
                    }
                    q[u] = map[pos[u] != ~0u ? pos[u] : qe.x];
                }
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    if (pos[u] != ~0u) {
                        const float d2 = sqdist3(me.x, me.y, me.z, q[u].x, q[u].y, q[u].z);
                        if (d2 <= bound) Lc.insert(pack_key(d2, __float_as_uint(q[u].w)), (int)pos[u]);
                    }
            }
            // merge: k rounds of the wave minimum over the heads of the lanes' lists (keys are unique: one lane gives up its head per round)
            Gl.init(k);
#pragma unroll
            for (int j = 0; j < KMAX; ++j) {
                const unsigned long long head = Lc.key[0];
                unsigned long long mk = head;
                int ms = Lc.sidx[0];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const unsigned long long ok = __shfl_xor(mk, off, 64);
                    const int os = __shfl_xor(ms, off, 64);
                    if (ok < mk) { mk = ok; ms = os; }
                }
                if (j < k) { Gl.key[j] = mk; Gl.sidx[j] = mk != ~0ull ? ms : -1; }
                if (head == mk && mk != ~0ull) {
#pragma unroll
                    for (int i = 0; i + 1 < KMAX; ++i) { Lc.key[i] = Lc.key[i + 1]; Lc.sidx[i] = Lc.sidx[i + 1]; }
                    Lc.key[KMAX - 1] = ~0ull; Lc.sidx[KMAX - 1] = -1;
                }
            }
            unsigned long long kth = ~0ull;
#pragma unroll
            for (int i = 0; i < KMAX; ++i) if (i == k - 1) kth = Gl.key[i];
            const float kd2 = __uint_as_float((unsigned)(kth >> 32));
            if (m2 == INFINITY || (kth != ~0ull && kd2 <= m2) || l >= 31) break;
            if (kth != ~0ull) bound = kd2; // k real points within kd2: the answer's k-th is no farther
        }
        lev_sum += (unsigned long long)l;
        if (lane == 0) {
            const unsigned orig = __float_as_uint(me.w);
#pragma unroll
            for (int j = 0; j < KMAX; ++j)
                if (j < k) {
                    const int sx = Gl.sidx[j];
                    out_sidx[(size_t)k * orig + j] = sx;
                    out_d2[(size_t)k * orig + j] = sx < 0 ? INFINITY : __uint_as_float((unsigned)(Gl.key[j] >> 32));
                }
        }
    }
    if (lane == 0 && lev_sum) atomicAdd(&st->levels, lev_sum);
}

template <typename T>
icpmi_status sg_cap(icpmi_ctx* c, T** p, size_t* cap, size_t need, bool* fresh = nullptr)
{
    if (need <= *cap && *p) return ICPMI_OK;
    if (*p) { HIP_TRY(c, hipStreamSynchronize(c->stream)); HIP_TRY(c, dev_free(*p)); *p = nullptr; *cap = 0; }
    const size_t want = need + need / 4 + 64;
    HIP_TRY(c, dev_malloc((void**)p, want * sizeof(T)));
    *cap = want;
    if (fresh) *fresh = true;
    return ICPMI_OK;
}

template <int KMAX>
void sg_launch_search(icpmi_ctx* c, SelfGridCtx* sg, const SgGrid& g, int k, unsigned tau, unsigned blocks_max, int* d_sidx, float* d_d2,
                      unsigned long long* sq_mapped)
{
    const unsigned g1 = std::max(1u, std::min(blocks_max, 16384u));
    hipLaunchKernelGGL((sg_tiled_kernel<KMAX, 8>), dim3(g1), dim3(64), 0, c->stream, g, (const float4*)c->d_map_sorted, (const unsigned*)sg->d_tstart,
                       (const unsigned*)sg->d_tbid, (const unsigned*)sg->d_f, (const uint4*)sg->d_blist, sg->d_state, k, tau, d_sidx, d_d2, sg->d_queue,
                       sq_mapped);
    hipLaunchKernelGGL(sg_wave_kernel<KMAX>, dim3(4096), dim3(64), 0, c->stream, g, (const float4*)c->d_map_sorted, (const unsigned*)sg->d_tstart,
                       (const unsigned*)sg->d_tbid, (const unsigned*)sg->d_f, sg->d_state, k, d_sidx, d_d2, (const uint2*)sg->d_queue);
}

} // namespace

void selfgrid_destroy(icpmi_ctx* c)
{
    SelfGridCtx* sg = c->sg;
    if (!sg) return;
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    dev_free(sg->d_tcnt); dev_free(sg->d_tstart); dev_free(sg->d_tbid); dev_free(sg->d_blist); dev_free(sg->d_f); dev_free(sg->d_coarse);
    dev_free(sg->d_queue); dev_free(sg->d_part); dev_free(sg->d_state);
    delete sg;
    c->sg = nullptr;
}

icpmi_status selfgrid_knn(icpmi_ctx* c, const float4* d_pts, int64_t m, int k, int* d_sidx, float* d_d2)
{
    if (m <= 0) return ICPMI_OK;
    if (k < 1 || k > ICPMI_MAX_K) { c->last_error = "self knn: k must be in [1, 32]"; return ICPMI_ERR_INVALID_ARG; }
    if (m > 0x7fffff00ll) { c->last_error = "self knn: cloud too large"; return ICPMI_ERR_UNSUPPORTED; }
    if (!c->sg) { c->sg = new (std::nothrow) SelfGridCtx(); if (!c->sg) { c->last_error = "out of host memory"; return ICPMI_ERR_HIP; } }
    SelfGridCtx* sg = c->sg;
    if (!sg->d_state) HIP_TRY(c, dev_malloc((void**)&sg->d_state, sizeof(SgState)));
    // ---- bounding box (one read-back: the table sizes depend on it) ----
    const int rblocks = (int)std::min<int64_t>((m + SG_RB - 1) / SG_RB, 256);
    if (sg_cap(c, &sg->d_part, &sg->cap_part, (size_t)rblocks * 6) != ICPMI_OK) return ICPMI_ERR_HIP;
    hipLaunchKernelGGL(sg_reset_kernel, dim3(1), dim3(64), 0, c->stream, sg->d_state);
    hipLaunchKernelGGL(sg_bbox_kernel, dim3(rblocks), dim3(SG_RB), 0, c->stream, d_pts, m, sg->d_part, sg->d_state);
    HIP_TRY(c, hipGetLastError());
    std::vector<float> part((size_t)rblocks * 6);
    unsigned bad = 0;
    if (read_back2(c, part.data(), sg->d_part, part.size() * sizeof(float), &bad, &sg->d_state->bad, sizeof(unsigned)) != ICPMI_OK) return ICPMI_ERR_HIP;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}, maxabs = 0.f;
    for (int b = 0; b < rblocks; ++b)
        for (int r = 0; r < 3; ++r) { lo[r] = fminf(lo[r], part[(size_t)b * 6 + r]); hi[r] = fmaxf(hi[r], part[(size_t)b * 6 + 3 + r]); }
    for (int r = 0; r < 3; ++r) {
        if (bad || !(lo[r] <= hi[r]) || !std::isfinite(lo[r]) || !std::isfinite(hi[r])) { c->last_error = "set_map: non-finite coordinates in the map cloud"; return ICPMI_ERR_INVALID_ARG; }
        maxabs = fmaxf(maxabs, fmaxf(fabsf(lo[r]), fabsf(hi[r])));
    }
    const double ext[3] = {(double)hi[0] - lo[0], (double)hi[1] - lo[1], (double)hi[2] - lo[2]};
    const double max_ext = std::max(ext[0], std::max(ext[1], ext[2]));

    // ---- the edge of an A-cell.  What the search pays for is the number of points an average POINT finds in its cell (size-biased occupancy:
    //      sum of squared cell counts / points); the previous build of this handle left that number in the host-mapped page ----
    static const double target_cfg = [] { const char* e = getenv("ICPMI_SG_TARGET"); return e ? atof(e) : 0.0; }();
    const double target = target_cfg > 0.0 ? target_cfg : std::max(2.0, 0.8 * (double)k);
    volatile unsigned long long* h_sq = c->h_progress ? reinterpret_cast<volatile unsigned long long*>(c->h_progress + ICPMI_PROGRESS_SELF_WORD) : nullptr;
    unsigned long long* d_sq = c->d_progress ? reinterpret_cast<unsigned long long*>(c->d_progress + ICPMI_PROGRESS_SELF_WORD) : nullptr;
    auto make = [&](double cell_d) {
        SgGrid g;
        double cell = std::max(cell_d, std::max(max_ext * 1e-6, 1e-6));
        for (int it = 0; it < 96; ++it) { // keep T within 2^24 entries
            int bits = 0;
            for (int r = 0; r < 3; ++r) { const long long nb = ((long long)floor(ext[r] / cell) + 1 + 3) / 4; int b = 0; while ((1ll << b) < nb) ++b; bits += b; }
            if (bits <= 24) break;
            cell *= 1.26;
        }
        g.ox = lo[0]; g.oy = lo[1]; g.oz = lo[2]; g.cell = (float)cell; g.inv_cell = 1.0f / g.cell; g.maxabs = maxabs;
        int pos = 0, maxb = 0;
        for (int r = 0; r < 3; ++r) {
            g.na[r] = (int)floorf((hi[r] - lo[r]) * g.inv_cell) + 1;
            const int nb = (g.na[r] + 3) / 4;
            int b = 0; while ((1 << b) < nb) ++b;
            g.nbits[r] = b; g.mask[r] = 0; maxb = std::max(maxb, b);
        }
        for (int j = 0; j < maxb; ++j) for (int r = 0; r < 3; ++r) if (j < g.nbits[r]) g.mask[r] |= 1u << pos++;
        g.tsize = 1 << pos;
        return g;
    };
    const bool like_before = sg->cell > 0.f && sg->m > 0 && sg->k == k && (double)m > 0.7 * (double)sg->m && (double)m < 1.4 * (double)sg->m;
    double cell = 0.0;
    int trials = 1;
    if (like_before) {
        cell = sg->cell;
        const double sb = h_sq ? (double)*h_sq / (double)sg->m : 0.0;
        if (sb > 0.0 && !(sb > 0.85 * target && sb < 1.18 * target)) cell *= std::min(2.0, std::max(0.5, sqrt(target / sb))); // surfaces: occupancy ~ edge^2
    } else {
        const double vol = std::max(ext[0], 1e-3) * std::max(ext[1], 1e-3) * std::max(ext[2], 1e-3);
        cell = cbrt(vol / (double)m) * 1.2;
        trials = 5; // a cloud this handle has not seen the like of: build, look at the occupancy, correct
    }
    static const int diag = [] { const char* e = getenv("ICPMI_SELF_DIAG"); return e ? atoi(e) : 0; }();
    static const unsigned tau = [] { const char* e = getenv("ICPMI_SG_TAU"); return e ? (unsigned)atoi(e) : 8u; }();
    const int blocks256 = (int)((m + 255) / 256);
    SgGrid g{};
    unsigned blocks_max = 0;
    for (int trial = 0; trial < trials; ++trial) {
        g = make(cell);
        blocks_max = (unsigned)std::min<int64_t>(m, (int64_t)g.tsize);
        bool fresh = false;
        if (sg_cap(c, &sg->d_tcnt, &sg->cap_tcnt, (size_t)g.tsize + 2, &fresh) != ICPMI_OK) return ICPMI_ERR_HIP;
        if (fresh || !sg->tcnt_clean) HIP_TRY(c, hipMemsetAsync(sg->d_tcnt, 0, sg->cap_tcnt * sizeof(unsigned), c->stream));
        sg->tcnt_clean = false;
        if (sg_cap(c, &sg->d_tstart, &sg->cap_tstart, (size_t)g.tsize + 2) != ICPMI_OK || sg_cap(c, &sg->d_tbid, &sg->cap_tbid, (size_t)g.tsize + 1) != ICPMI_OK ||
            sg_cap(c, &sg->d_blist, &sg->cap_blist, (size_t)blocks_max + 1) != ICPMI_OK || sg_cap(c, &sg->d_f, &sg->cap_f, (size_t)blocks_max * SG_F + 1) != ICPMI_OK ||
            sg_cap(c, &sg->d_coarse, &sg->cap_coarse, (size_t)m + 16) != ICPMI_OK || sg_cap(c, &sg->d_queue, &sg->cap_queue, (size_t)m + 1) != ICPMI_OK ||
            ensure_cap(c, &c->d_keys, &c->cap_keys, (size_t)m) != ICPMI_OK || ensure_cap(c, &c->d_map_sorted, &c->cap_map, (size_t)m + 16) != ICPMI_OK)
            return ICPMI_ERR_HIP;
        if (trial > 0) hipLaunchKernelGGL(sg_reset_kernel, dim3(1), dim3(64), 0, c->stream, sg->d_state);
        hipLaunchKernelGGL(sg_key_kernel, dim3(blocks256), dim3(256), 0, c->stream, d_pts, m, g, c->d_keys, sg->d_tcnt, sg->d_tbid, sg->d_blist, sg->d_state);
        HIP_TRY(c, hipGetLastError());
        if (device_exclusive_scan_cursor(c, sg->d_tcnt, sg->d_tstart, g.tsize, (unsigned)m, true) != ICPMI_OK) return ICPMI_ERR_HIP;
        sg->tcnt_clean = true;
        hipLaunchKernelGGL(sg_scatter_kernel, dim3(blocks256), dim3(256), 0, c->stream, d_pts, m, (const unsigned*)c->d_keys, sg->d_tstart + 1, sg->d_coarse);
        hipLaunchKernelGGL(sg_block_sort_kernel, dim3(std::max(1u, std::min(blocks_max, 8192u))), dim3(64), 0, c->stream, (const float4*)sg->d_coarse,
                           c->d_map_sorted, g, (const unsigned*)sg->d_tstart, (const uint4*)sg->d_blist, sg->d_f, sg->d_state);
        HIP_TRY(c, hipGetLastError());
        if (trial + 1 >= trials) break;
        unsigned long long sq = 0;
        if (read_back(c, &sq, &sg->d_state->sq, sizeof sq) != ICPMI_OK) return ICPMI_ERR_HIP;
        const double sb = (double)sq / (double)m;
        if (sb > 0.85 * target && sb < 1.18 * target) break;
        const double next = (double)g.cell * std::min(4.0, std::max(0.25, sqrt(target / sb)));
        if (fabs(next - (double)g.cell) < 0.05 * (double)g.cell) break;
        cell = next;
    }
    sg->cell = g.cell; sg->m = m; sg->k = k; ++sg->seq;

    // ---- search ----
    if (k <= 4) sg_launch_search<4>(c, sg, g, k, tau, blocks_max, d_sidx, d_d2, d_sq);
    else if (k <= 8) sg_launch_search<8>(c, sg, g, k, tau, blocks_max, d_sidx, d_d2, d_sq);
    else if (k <= 10) sg_launch_search<10>(c, sg, g, k, tau, blocks_max, d_sidx, d_d2, d_sq); // the shipped post filter (examples/config.yaml:26-27)
    else if (k <= 16) sg_launch_search<16>(c, sg, g, k, tau, blocks_max, d_sidx, d_d2, d_sq);
    else sg_launch_search<32>(c, sg, g, k, tau, blocks_max, d_sidx, d_d2, d_sq);
    HIP_TRY(c, hipGetLastError());
    if (diag) {
        SgState hs{};
        if (read_back(c, &hs, sg->d_state, sizeof hs) == ICPMI_OK)
            fprintf(stderr, "[icpmi self-knn] m %lld k %d: A-cell %.3f (%d x %d x %d), T %d entries, %u blocks, size-biased occupancy %.1f (target %.1f); "
                            "%u queries (%.2f %%) through the levels, mean end level %.2f\n",
                    (long long)m, k, (double)g.cell, g.na[0], g.na[1], g.na[2], g.tsize, hs.nblk, (double)hs.sq / (double)m, target, hs.qcount,
                    100.0 * hs.qcount / (double)m, hs.qcount ? (double)hs.levels / (double)hs.qcount : 0.0);
#ifdef SG_TIMING
        fprintf(stderr, "[icpmi self-knn] tiled kernel, sampled workgroups (cycles x 1e3): setup %.0f  stage %.0f  scan+insert %.0f  decide+write %.0f  loop-top %.0f | drain rounds %llu  scan iterations %llu\n",
                hs.tick[0] * 1e-3, hs.tick[1] * 1e-3, hs.tick[2] * 1e-3, hs.tick[3] * 1e-3, hs.tick[7] * 1e-3, hs.tick[4], hs.tick[5]);
#endif
    }
    return ICPMI_OK;
}
