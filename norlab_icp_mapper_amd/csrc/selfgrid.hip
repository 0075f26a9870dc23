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
// Search (2 launches; selection ACROSS the lanes of a wave -- see the kernels):
//   * sg_cell_kernel: one wave per 16 consecutive sorted points.  Per A-cell: its 27 neighbours looked up (one lane each), their points into
//     registers; per query: distances, one 64-key bitonic sort, further batches merged only when they can matter.  Decided when the k-th
//     distance fits the margin to the border of the 3 x 3 x 3 cells (faces on the border of the grid do not count: nothing lies beyond them).
//   * sg_level_kernel: the rest (config 4: ~12 %), one wave per query, level after level (edge x 2): the 3 x 3 x 3 block of level l is visited
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
    unsigned* d_inv = nullptr; size_t cap_inv = 0;                            // original index -> sorted position
    float* d_part = nullptr; size_t cap_part = 0;                             // bounding-box partials
    unsigned* d_dirty = nullptr; size_t cap_dirty = 0;                        // subset search: bitmaps of the cells appended points fell into, per level
    float4* d_prev = nullptr; size_t cap_prev = 0; int64_t prev_m = 0; float prev_lo[3] = {0, 0, 0}, prev_hi[3] = {0, 0, 0};   // the sorted copy (and its bounding box) of the last TRACKED build (+ room for the next append): the next build of the grown cloud reads it instead of the cloud in the caller's order -- nearly sorted input, coalesced scatter
    unsigned char* d_sel = nullptr; size_t cap_sel = 0;                       // ... selected queries by sorted position (m) and by original index (m)
    struct SgState* d_state = nullptr;
    // tuning state: the edge of the previous build and what its points saw (size-biased A-cell occupancy, delivered through the mapped page)
    float cell = 0.f; int64_t m = 0; int k = 0;
    unsigned long long seq = 0;
};

struct SgState {
    unsigned nblk;   // occupied blocks (claimed by the key kernel)
    unsigned bad;    // non-finite coordinates seen by the bounding-box pass
    unsigned long long sq;     // sum over the A-cells of (points in the cell)^2 (summed from sqpart by the cell kernel)
    unsigned long long levels; // diagnostics: sum of the levels at which the level kernel's queries ended
    unsigned qcount[64];       // queries queued for the level kernel, per sub-queue
    unsigned long long sqpart[64]; // partial sums of sq, by workgroup % 64 (thousands of atomics on ONE address serialise at 10 - 25 ns each)
};

namespace {

constexpr int SG_F = 65;   // words of a block's row of F

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
// the set positions of mask gathered from v, lowest first (a software pext: the inverse of sg_dep)
__device__ __forceinline__ unsigned sg_ext(unsigned v, unsigned mask)
{
    unsigned r = 0, bit = 1u;
    while (mask) {
        const unsigned low = mask & (0u - mask);
        r |= (v & low) ? bit : 0u;
        bit <<= 1; mask ^= low;
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

// a square root that never under-estimates (one v_sqrt_f32, 1 ulp, nudged up)
__device__ __forceinline__ float sqrt_up_sg(float x) { return __builtin_amdgcn_sqrtf(x) * 1.0000005f; }

// inclusive scan over the 64 lanes: four DPP row shifts inside the rows of 16, two row broadcasts across them
__device__ __forceinline__ unsigned sg_wave_incl_scan(unsigned v)
{
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true); // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true); // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true); // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true); // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, true); // row_bcast:15 -> rows 1, 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, true); // row_bcast:31 -> rows 2, 3
    return (unsigned)x;
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
    if (blockIdx.x == 0) {
        if (threadIdx.x == 0) { st->nblk = 0; st->bad = 0; st->sq = 0ull; st->levels = 0ull; }
        if (threadIdx.x < 64) { st->qcount[threadIdx.x] = 0; st->sqpart[threadIdx.x] = 0ull; }
    }
}

// per point: key = Morton index of its block << 6 | Morton index of its A-cell in the block; points per block counted (one atomic per run of
// equal keys in a wave); the lane that finds a block's counter at zero claims a block id for it.  A workgroup works through `tiles` consecutive
// tiles of 256 points and fetches the ids of ALL the blocks it claimed with ONE atomic on the id counter (r6: one per tile was 39 k atomics on one
// address at 10 M points -- 0.4 of the kernel's 0.46 ms)
constexpr int SG_KEY_TILES = 16;
__global__ __launch_bounds__(256) void sg_key_kernel(const float4* __restrict__ pts, int64_t m, SgGrid g, unsigned* __restrict__ keys,
                                                     unsigned* __restrict__ tcnt, unsigned* __restrict__ tbid, uint4* __restrict__ blist,
                                                     SgState* __restrict__ st, int tiles)
{
    __shared__ unsigned wave_cnt[SG_KEY_TILES * 4 + 1], wg_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned hbs[SG_KEY_TILES];
    unsigned fm = 0; // bit t: this lane claimed its block in tile t
#pragma unroll
    for (int t = 0; t < SG_KEY_TILES; ++t) {
        hbs[t] = 0;
        if (t >= tiles) continue; // (uniform)
        const int64_t i = ((int64_t)blockIdx.x * tiles + t) * 256 + threadIdx.x;
        const bool valid = i < m;
        const float4 p = pts[valid ? i : 0];
        const int ax = sg_cell_of(p.x, g.ox, g.inv_cell, g.na[0]);
        const int ay = sg_cell_of(p.y, g.oy, g.inv_cell, g.na[1]);
        const int az = sg_cell_of(p.z, g.oz, g.inv_cell, g.na[2]);
        const unsigned hb = sg_hb(g, ax >> 2, ay >> 2, az >> 2);
        if (valid) keys[i] = (hb << 6) | sg_sub(ax & 3, ay & 3, az & 3);
        const WaveRun r = wave_run(hb, valid);
        const bool first = r.head && atomicAdd(&tcnt[hb], (unsigned)r.len) == 0u;
        hbs[t] = hb;
        fm |= first ? 1u << t : 0u;
        const unsigned long long firsts = __ballot(first);
        if (lane == 0) wave_cnt[t * 4 + wave] = (unsigned)__popcll(firsts);
    }
    __syncthreads();
    if (threadIdx.x == 0) { // exclusive prefix over (tile, wave), one atomic for the workgroup
        unsigned run = 0;
        for (int e = 0; e < tiles * 4; ++e) { const unsigned c = wave_cnt[e]; wave_cnt[e] = run; run += c; }
        wg_base = run ? atomicAdd(&st->nblk, run) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < SG_KEY_TILES; ++t) {
        if (t >= tiles) continue;
        const bool first = (fm >> t) & 1u;
        const unsigned long long firsts = __ballot(first);
        if (first) {
            const unsigned bid = wg_base + wave_cnt[t * 4 + wave] + (unsigned)__popcll(firsts & ((1ull << lane) - 1ull));
            const unsigned hb = hbs[t];
            tbid[hb] = bid;
            blist[bid] = make_uint4(hb, sg_ext(hb, g.mask[0]), sg_ext(hb, g.mask[1]), sg_ext(hb, g.mask[2]));
        }
    }
}

// points into block order (any order inside a block): cursor scatter on T (tstart + 1: common.h, device_exclusive_scan_cursor)
// keep_w: the input already carries the original index in w (the previous build's sorted copy + the appended tail) instead of being in original order
__global__ __launch_bounds__(256) void sg_scatter_kernel(const float4* __restrict__ pts, int64_t m, const unsigned* __restrict__ keys,
                                                         unsigned* __restrict__ cursor, float4* __restrict__ out, int keep_w)
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
    out[base + (unsigned)r.rank] = make_float4(p.x, p.y, p.z, keep_w ? p.w : __uint_as_float((unsigned)i));
}

// the appended points behind the previous build's sorted copy, w = original index
__global__ __launch_bounds__(256) void sg_tail_kernel(const float4* __restrict__ pts, int64_t m_old, int64_t m, float4* __restrict__ prev)
{
    const int64_t i = m_old + (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const float4 p = pts[i];
    prev[i] = make_float4(p.x, p.y, p.z, __uint_as_float((unsigned)i));
}

// one wave per occupied block: its points into the Morton order of their A-cells (a counting sort in LDS; the order inside a cell is whatever
// the atomics give -- every comparison downstream is on (d^2, original index)), its row of F, its share of the sum of squared cell counts
__global__ __launch_bounds__(64) void sg_block_sort_kernel(const float4* __restrict__ in, float4* __restrict__ out, SgGrid g,
                                                           const unsigned* __restrict__ tstart, const uint4* __restrict__ blist,
                                                           unsigned* __restrict__ f, unsigned* __restrict__ inv, SgState* __restrict__ st)
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
            inv[__float_as_uint(p.w)] = s0 + r;
        }
        __syncthreads();
    }
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
    if (lane == 0 && sq) atomicAdd(&st->sqpart[blockIdx.x & 63], sq);
}

// ---- search -----------------------------------------------------------------------------------------------------------------------------
// Both kernels select ACROSS the lanes of a wave: every lane holds the key of one candidate -- d^2 bits << 32 | payload, payload ordered by the
// original index --, a bitonic network sorts the 64 keys (21 compare-exchange steps, each two lane permutes and a 64-bit compare), and further
// batches of 64 are sorted and merged only when one of their keys beats the k-th so far.  No per-lane lists: r2 - r5 (and this file's first
// version) kept a sorted list of k keys per QUERY lane, ~120 instructions per insertion paid by the whole wave whenever one lane inserted, and
// ran at 2.5 - 5 ns per query on a lidar map; here a query costs one sort (~130 instructions) whatever its neighbours do.
constexpr int SGQ = 16; // consecutive sorted queries per wave of the cell kernel (they share cells: the candidates stay in registers)
constexpr int SG_R = 4; // candidate registers per lane: neighbourhoods of up to 256 points are kept across the queries of a cell

__device__ const signed char SG_ORD[27][3] = { // the 27 cells of a neighbourhood, nearest first: centre, faces, edges, corners
    {0, 0, 0},
    {-1, 0, 0}, {1, 0, 0}, {0, -1, 0}, {0, 1, 0}, {0, 0, -1}, {0, 0, 1},
    {-1, -1, 0}, {1, -1, 0}, {-1, 1, 0}, {1, 1, 0}, {-1, 0, -1}, {1, 0, -1}, {-1, 0, 1}, {1, 0, 1}, {0, -1, -1}, {0, 1, -1}, {0, -1, 1}, {0, 1, 1},
    {-1, -1, -1}, {1, -1, -1}, {-1, 1, -1}, {1, 1, -1}, {-1, -1, 1}, {1, -1, 1}, {-1, 1, 1}, {1, 1, 1}};

// value of lane (lane ^ J), without the LDS crossbar: DPP quad permutes / row shifts / row rotate inside a row of 16 lanes, gfx950's
// v_permlane16_swap / v_permlane32_swap across rows (a sort written with __shfl_xor is 42 ds_bpermute_b32 round trips, one behind the other)
template <int J>
__device__ __forceinline__ unsigned sg_lane_xor(unsigned v)
{
    const int x = (int)v;
    if constexpr (J == 1) return (unsigned)__builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false);      // quad_perm [1, 0, 3, 2]
    else if constexpr (J == 2) return (unsigned)__builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false); // quad_perm [2, 3, 0, 1]
    else if constexpr (J == 4) {
        const int t = __builtin_amdgcn_update_dpp(x, x, 0x104, 0xF, 0x5, false);                            // row_shl:4 into banks 0, 2
        return (unsigned)__builtin_amdgcn_update_dpp(t, x, 0x114, 0xF, 0xA, false);                         // row_shr:4 into banks 1, 3
    } else if constexpr (J == 8) return (unsigned)__builtin_amdgcn_update_dpp(x, x, 0x128, 0xF, 0xF, false); // row_ror:8
    else if constexpr (J == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false); // r[0]: odd rows <- even rows; r[1]: even rows <- odd rows
        return (threadIdx.x & 16) ? r[0] : r[1];
    } else {
        const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
        return (threadIdx.x & 32) ? r[0] : r[1];
    }
}
template <int J>
__device__ __forceinline__ unsigned long long sg_lane_xor64(unsigned long long v)
{
    return ((unsigned long long)sg_lane_xor<J>((unsigned)(v >> 32)) << 32) | (unsigned long long)sg_lane_xor<J>((unsigned)v);
}
__device__ __forceinline__ unsigned long long sg_readlane64(unsigned long long v, int l /* uniform */)
{
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l) << 32) |
           (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l);
}

// bitonic sort across the 64 lanes (keys are unique, or ~0 = nothing): ascending, or DESC: descending.  n (uniform): only lanes [0, n) hold
// real keys -- the phases that merge halves of nothing are skipped (n <= 16: 10 steps, n <= 32: 15, else 21)
#define SG_CE(K2, J) { const unsigned long long other = sg_lane_xor64<J>(key); \
                       const bool take_min = ((((K2) >= 64) || (lane & (K2)) == 0) == ((lane & (J)) == 0)) != DESC; \
                       key = (take_min == (other < key)) ? other : key; }
template <bool DESC>
__device__ __forceinline__ void sg_sort64(unsigned long long& key, unsigned n = 64u)
{
    const int lane = threadIdx.x & 63;
    SG_CE(2, 1)
    SG_CE(4, 2) SG_CE(4, 1)
    SG_CE(8, 4) SG_CE(8, 2) SG_CE(8, 1)
    SG_CE(16, 8) SG_CE(16, 4) SG_CE(16, 2) SG_CE(16, 1)
    if (DESC || n > 16u) { SG_CE(32, 16) SG_CE(32, 8) SG_CE(32, 4) SG_CE(32, 2) SG_CE(32, 1) }
    if (DESC || n > 32u) { SG_CE(64, 32) SG_CE(64, 16) SG_CE(64, 8) SG_CE(64, 4) SG_CE(64, 2) SG_CE(64, 1) }
}
// best: ascending over the lanes, b: DESCENDING -> best = the 64 smallest of both, ascending
__device__ __forceinline__ void sg_merge64(unsigned long long& best, unsigned long long b)
{
    const int lane = threadIdx.x & 63;
    constexpr bool DESC = false;
    unsigned long long key = b < best ? b : best; // bitonic: the low half of the merge
    SG_CE(64, 32) SG_CE(64, 16) SG_CE(64, 8) SG_CE(64, 4) SG_CE(64, 2) SG_CE(64, 1)
    best = key;
}
#undef SG_CE
// one key (uniform) into `best` (ascending over the lanes): the lanes holding larger keys take their left neighbour's, the first of them x
__device__ __forceinline__ void sg_insert64(unsigned long long& best, unsigned long long x)
{
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(best >> 32), 0x138, 0xF, 0xF, true); // wave_shr:1 (lane 0 reads 0)
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)best, 0x138, 0xF, 0xF, true);
    const unsigned long long left = ((unsigned long long)hi << 32) | lo;
    best = best > x ? (left > x ? left : x) : best;
}
// a further batch of keys (one per lane, ~0 = nothing) into `best`: nothing when none beats the k-th so far, the few that do one by one,
// a sort and a merge when they are many
__device__ __forceinline__ void sg_absorb(unsigned long long& best, unsigned long long key, int k)
{
    unsigned long long kth = sg_readlane64(best, k - 1);
    unsigned long long mask = __ballot(key < kth);
    if (!mask) return;
    if (__popcll(mask) > 12) { sg_sort64<true>(key); sg_merge64(best, key); return; }
    while (mask) {
        const int l = __ffsll((long long)mask) - 1;
        mask &= mask - 1ull;
        const unsigned long long x = sg_readlane64(key, l);
        if (x < kth) { sg_insert64(best, x); kth = sg_readlane64(best, k - 1); }
    }
}

constexpr int SG_NQ = 64; // sub-queues between the two search kernels (a wave appends to queue blockIdx % SG_NQ: ~100 atomics per counter
                          // instead of thousands on one address -- same-address device atomics serialise at 10 - 25 ns each)

// One wave per SGQ consecutive points of the sorted array (= a few A-cells).  Per cell: 27 look-ups (one lane each, nearest cells first), the
// neighbourhood's points into registers (lane j holds candidates j, j + 64, ...); per query: distances, one sort of the first 64, the rest
// absorbed.  A query is decided when its k-th distance fits the margin to the border of its 3 x 3 x 3 cells; the others are queued for the
// level kernel with the k-th distance they did find.  packed_ok (cloud below 2^24 points): the payload is original index << 8 | candidate
// slot, and the slot finds the sorted position in LDS; otherwise (and for neighbourhoods beyond 256 points) the payload is the original
// index and `inv` maps it back.
__global__ __launch_bounds__(64) void sg_cell_kernel(SgGrid g, const float4* __restrict__ map, const unsigned* __restrict__ tstart,
                                                     const unsigned* __restrict__ tbid, const unsigned* __restrict__ f,
                                                     const unsigned* __restrict__ inv, SgState* __restrict__ st, int k, unsigned m, int packed_ok,
                                                     int* __restrict__ out_sidx, float* __restrict__ out_d2, uint2* __restrict__ queue, unsigned qcap,
                                                     unsigned long long* __restrict__ sq_mapped, const unsigned char* __restrict__ qsel /* by sorted position, or nullptr: all */,
                                                     const unsigned* __restrict__ wlist /* subset search: the groups of SGQ sorted positions that hold a selected query, [0] = how many */)
{
    __shared__ unsigned pre[28], cst[27], cpos[64 * SG_R];
    __shared__ uint2 undq[SGQ];
    const int lane = threadIdx.x;
    // what the points of this build see (the block sort is done): to the host-mapped page, where the NEXT build of this handle reads it
    if (blockIdx.x == 0 && sq_mapped) {
        unsigned long long sq = st->sqpart[lane];
        for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
        if (lane == 0) { *sq_mapped = sq; st->sq = sq; }
    }
    const unsigned INF_BITS = 0x7f800000u;
    unsigned grp = blockIdx.x;
    if (wlist) { if (blockIdx.x >= wlist[0]) return; grp = wlist[1 + blockIdx.x]; }
    const unsigned q0 = grp * SGQ, q1 = min(q0 + (unsigned)SGQ, m);
    // a subset search (an appended cloud: only the points whose neighbourhood may have changed): a wave none of whose queries is selected is done
    int msel = 1;
    if (qsel) {
        msel = (q0 + (unsigned)lane < q1) ? (int)qsel[q0 + (unsigned)lane] : 0;
        if (__ballot(msel != 0) == 0ull) return;
    }
    // the wave's queries, one trip; what a query needs besides its coordinates is computed by its lane and read with v_readlane below
    const float4 mine = map[min(q0 + (unsigned)lane, m - 1)];
    const float mfx = (mine.x - g.ox) * g.inv_cell, mfy = (mine.y - g.oy) * g.inv_cell, mfz = (mine.z - g.oz) * g.inv_cell;
    const int ma3[3] = {sg_cell_of(mine.x, g.ox, g.inv_cell, g.na[0]), sg_cell_of(mine.y, g.oy, g.inv_cell, g.na[1]), sg_cell_of(mine.z, g.oz, g.inv_cell, g.na[2])};
    const float mfa[3] = {fminf(fmaxf(mfx - (float)ma3[0], 0.f), 1.f), fminf(fmaxf(mfy - (float)ma3[1], 0.f), 1.f), fminf(fmaxf(mfz - (float)ma3[2], 0.f), 1.f)};
    const float mm2 = sg_margin2(g, 0, ma3, mfa);
    int pa0 = -1, pa1 = -1, pa2 = -1;
    unsigned N = 0, nund = 0;
    bool big = false;
    float4 cand[SG_R];
    unsigned cpay[SG_R];
    bool cval[SG_R];
#pragma unroll
    for (int r = 0; r < SG_R; ++r) { cand[r] = make_float4(0.f, 0.f, 0.f, 0.f); cpay[r] = 0; cval[r] = false; }
    for (unsigned qi = q0; qi < q1; ++qi) {
        const int src = (int)(qi - q0);
        if (!__builtin_amdgcn_readlane(msel, src)) continue; // (uniform)
        const float mex = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine.x), src)), mey = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine.y), src)),
                    mez = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine.z), src));
        const unsigned orig_q = (unsigned)__builtin_amdgcn_readlane(__float_as_int(mine.w), src);
        const int a0 = __builtin_amdgcn_readlane(ma3[0], src), a1 = __builtin_amdgcn_readlane(ma3[1], src), a2 = __builtin_amdgcn_readlane(ma3[2], src);
        const float m2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mm2), src));
        if (a0 != pa0 || a1 != pa1 || a2 != pa2) { // (wave-uniform) a new cell: its neighbourhood
            pa0 = a0; pa1 = a1; pa2 = a2;
            __syncthreads();
            unsigned s = 0, e = 0;
            if (lane < 27) sg_cell_range(g, tstart, tbid, f, 0, a0 + SG_ORD[lane][0], a1 + SG_ORD[lane][1], a2 + SG_ORD[lane][2], s, e);
            const unsigned incl = sg_wave_incl_scan(e - s);
            if (lane < 27) { pre[lane + 1] = incl; cst[lane] = s; }
            if (lane == 0) pre[0] = 0;
            N = (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
            __syncthreads();
            big = !packed_ok || N > (unsigned)(64 * SG_R);
#pragma unroll
            for (int r = 0; r < SG_R; ++r) {
                const unsigned fl = (unsigned)(lane + 64 * r);
                unsigned pos = ~0u;
                if (fl < N) {
                    int lo = 0, hi = 27; // the cell of flat candidate fl: the last offset <= fl
#pragma unroll
                    for (int it = 0; it < 5; ++it) { const int mid = (lo + hi) >> 1; if (pre[mid] <= fl) lo = mid; else hi = mid; }
                    pos = cst[lo] + (fl - pre[lo]);
                }
                cval[r] = pos != ~0u;
                cand[r] = map[cval[r] ? pos : qi];
                cpos[fl] = pos;
                const unsigned orig = __float_as_uint(cand[r].w);
                cpay[r] = big ? orig : ((orig << 8) | fl);
            }
            __syncthreads();
        }
        unsigned long long best = ~0ull;
        if (N >= (unsigned)k || m2 == INFINITY) { // (fewer than k points around it: the levels decide)
#pragma unroll
            for (int r = 0; r < SG_R; ++r) {
                if ((unsigned)(64 * r) >= N) break; // (uniform)
                const float d2 = sqdist3(mex, mey, mez, cand[r].x, cand[r].y, cand[r].z);
                const unsigned long long key = cval[r] ? pack_key(d2, cpay[r]) : ~0ull;
                if (r == 0) { best = key; sg_sort64<false>(best, N); }
                else sg_absorb(best, key, k);
            }
            for (unsigned c0 = 64u * SG_R; c0 < N; c0 += 64u) { // a neighbourhood beyond the registers (big mode): fetched per query
                const unsigned fl = c0 + (unsigned)lane;
                unsigned long long key = ~0ull;
                if (fl < N) {
                    int lo = 0, hi = 27;
#pragma unroll
                    for (int it = 0; it < 5; ++it) { const int mid = (lo + hi) >> 1; if (pre[mid] <= fl) lo = mid; else hi = mid; }
                    const float4 q = map[cst[lo] + (fl - pre[lo])];
                    key = pack_key(sqdist3(mex, mey, mez, q.x, q.y, q.z), __float_as_uint(q.w));
                }
                sg_absorb(best, key, k);
            }
        }
        const unsigned long long kth = sg_readlane64(best, k - 1);
        const float kd2 = __uint_as_float((unsigned)(kth >> 32));
        const bool decided = m2 == INFINITY || (kth != ~0ull && kd2 <= m2);
        if (decided) {
            if (lane < k) {
                const bool valid = best != ~0ull;
                int pos = -1;
                if (valid) pos = big ? (int)inv[(unsigned)best] : (int)cpos[(unsigned)best & 0xffu];
                out_sidx[(size_t)k * orig_q + lane] = pos;
                out_d2[(size_t)k * orig_q + lane] = valid ? __uint_as_float((unsigned)(best >> 32)) : INFINITY;
            }
        } else {
            if (lane == 0) undq[nund] = make_uint2(qi, kth != ~0ull ? (unsigned)(kth >> 32) : INF_BITS);
            ++nund;
        }
    }
    if (nund) { // (uniform) one atomic per wave, on one of SG_NQ counters
        __syncthreads();
        const unsigned qn = blockIdx.x % SG_NQ;
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(&st->qcount[qn], nund);
        base = (unsigned)__builtin_amdgcn_readlane((int)base, 0);
        if ((unsigned)lane < nund) queue[(size_t)qn * qcap + base + lane] = undq[lane];
    }
}

// One wave per queued query.  Phase A (only while fewer than k points have been seen -- a query the cell kernel found k candidates for skips
// it): level after level (edge x 2), the 27 cells around the query, each one contiguous run of the sorted array, until k points are in hand
// or the block is the whole grid.  Their k-th distance b bounds the answer, so phase B is ONE ball query: the cells of the level whose edge is
// at least b / 2 that the box [p - b, p + b] touches (at most 6 per axis, 216 in all: four per lane), those whose box is farther than b
// dropped, the points of the others dealt to the lanes 64 at a time (a flat index over the runs' prefix sums, in LDS), four batches
// requested before any is looked at; keys beyond b are dropped at once, the first batch that holds anything is sorted, the rest absorbed.
// Every point within b of the query lies in those cells (a point's cell index is monotone in its coordinate), k of them exist: exact, and
// done -- r6's first version climbed the levels to the end and a query in the sparse periphery of a lidar map streamed the thousands of
// points of a 50 m block past one wave.
__device__ __forceinline__ void sg_flat_locate(const unsigned* __restrict__ rp, const unsigned* __restrict__ rs, int n_runs_pow2_steps, int n_runs,
                                               unsigned fl, unsigned& pos)
{
    int lo = 0, hi = n_runs; // the run of flat candidate fl: the last offset <= fl
    for (int it = 0; it < n_runs_pow2_steps; ++it) { const int mid = (lo + hi) >> 1; if (rp[mid] <= fl) lo = mid; else hi = mid; }
    pos = rs[lo] + (fl - rp[lo]);
}

__global__ __launch_bounds__(64) void sg_level_kernel(SgGrid g, const float4* __restrict__ map, const unsigned* __restrict__ tstart,
                                                      const unsigned* __restrict__ tbid, const unsigned* __restrict__ f,
                                                      const unsigned* __restrict__ inv, SgState* __restrict__ st, int k,
                                                      int* __restrict__ out_sidx, float* __restrict__ out_d2, const uint2* __restrict__ queue, unsigned qcap,
                                                      int diag)
{
    __shared__ unsigned rs[256], rp[257];
    const int lane = threadIdx.x;
    const unsigned qn = blockIdx.x % SG_NQ, stride = gridDim.x / SG_NQ; // (the grid is a multiple of SG_NQ)
    const unsigned count = st->qcount[qn];
    unsigned long long lev_sum = 0;
    for (unsigned w = blockIdx.x / SG_NQ; w < count; w += stride) {
        const uint2 qe = queue[(size_t)qn * qcap + w];
        const float4 me = map[qe.x];
        float bound = __uint_as_float(qe.y);
        const float fx = (me.x - g.ox) * g.inv_cell, fy = (me.y - g.oy) * g.inv_cell, fz = (me.z - g.oz) * g.inv_cell;
        const int a3[3] = {sg_cell_of(me.x, g.ox, g.inv_cell, g.na[0]), sg_cell_of(me.y, g.oy, g.inv_cell, g.na[1]), sg_cell_of(me.z, g.oz, g.inv_cell, g.na[2])};
        const float fa[3] = {fminf(fmaxf(fx - (float)a3[0], 0.f), 1.f), fminf(fmaxf(fy - (float)a3[1], 0.f), 1.f), fminf(fmaxf(fz - (float)a3[2], 0.f), 1.f)};
        unsigned long long best = ~0ull;
        bool done = false;
        int l = 0;
        // ---- phase A: any k points ----
        while (bound == INFINITY) {
            ++l;
            const float m2 = sg_margin2(g, l, a3, fa);
            __syncthreads(); // (rs / rp of the previous level consumed)
            unsigned s = 0, e = 0;
            if (lane < 27) sg_cell_range(g, tstart, tbid, f, l, (a3[0] >> l) + SG_ORD[lane][0], (a3[1] >> l) + SG_ORD[lane][1], (a3[2] >> l) + SG_ORD[lane][2], s, e);
            const unsigned incl = sg_wave_incl_scan(e - s);
            if (lane < 27) { rp[lane + 1] = incl; rs[lane] = s; }
            if (lane == 0) rp[0] = 0;
            const unsigned total = (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
            __syncthreads();
            best = ~0ull;
            bool first = true;
            for (unsigned base = 0; base < total; base += 256u) {
                float4 q[4];
                bool val[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const unsigned fl = base + 64u * u + (unsigned)lane;
                    unsigned pos = qe.x;
                    val[u] = fl < total;
                    if (val[u]) sg_flat_locate(rp, rs, 5, 27, fl, pos);
                    q[u] = map[pos];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (base + 64u * u >= total) break; // (uniform)
                    const unsigned long long key = val[u] ? pack_key(sqdist3(me.x, me.y, me.z, q[u].x, q[u].y, q[u].z), __float_as_uint(q[u].w)) : ~0ull;
                    if (first) { best = key; sg_sort64<false>(best, total); first = false; }
                    else sg_absorb(best, key, k);
                }
            }
            const unsigned long long kth = sg_readlane64(best, k - 1);
            const float kd2 = __uint_as_float((unsigned)(kth >> 32));
            if (m2 == INFINITY || (kth != ~0ull && kd2 <= m2) || l >= 31) { done = true; break; } // the block holds every point that matters
            if (kth != ~0ull) bound = kd2; // k real points within kd2: the answer's k-th is no farther
        }
        // ---- phase B: the ball of the bound ----
        if (!done) {
            const float b = sqrt_up_sg(bound);
            int sl = 0;
            while (sl < 30 && g.cell * (float)(1u << sl) < 0.5f * b) ++sl;
            const float reach = b * 1.00001f + (g.cell * (float)(1u << sl) * 1e-3f + g.maxabs * 4e-6f);
            int c0[3], nc[3];
            const float pme[3] = {me.x, me.y, me.z}, org[3] = {g.ox, g.oy, g.oz};
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                const int lo_c = sg_cell_of(pme[ax] - reach, org[ax], g.inv_cell, g.na[ax]) >> sl, hi_c = sg_cell_of(pme[ax] + reach, org[ax], g.inv_cell, g.na[ax]) >> sl;
                c0[ax] = lo_c; nc[ax] = min(hi_c - lo_c + 1, 6); // (at most 6 by the choice of sl; the clamp only guards the arithmetic)
            }
            const float cs = g.cell * (float)(1u << sl);
            const float slack = cs * 2e-3f + g.maxabs * 2e-6f;
            __syncthreads();
            unsigned total = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int idx = lane + 64 * j;
                unsigned s = 0, e = 0;
                const int ix = idx % 6, iy = (idx / 6) % 6, iz = idx / 36;
                if (idx < 216 && ix < nc[0] && iy < nc[1] && iz < nc[2]) {
                    const int sx = c0[0] + ix, sy = c0[1] + iy, sz = c0[2] + iz;
                    // distance from the query to the cell's box, every gap shortened by the slack (a point may sit a rounding outside its cell)
                    const float lox = g.ox + (float)sx * cs, loy = g.oy + (float)sy * cs, loz = g.oz + (float)sz * cs;
                    const float gx = fmaxf(fmaxf(lox - me.x, me.x - (lox + cs)) - slack, 0.f);
                    const float gy = fmaxf(fmaxf(loy - me.y, me.y - (loy + cs)) - slack, 0.f);
                    const float gz = fmaxf(fmaxf(loz - me.z, me.z - (loz + cs)) - slack, 0.f);
                    if (!(fmaf(gz, gz, fmaf(gy, gy, gx * gx)) > bound)) sg_cell_range(g, tstart, tbid, f, sl, sx, sy, sz, s, e);
                }
                const unsigned incl = sg_wave_incl_scan(e - s);
                if (idx < 216) { rs[idx] = s; rp[idx + 1] = total + incl; }
                total += (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
            }
            if (lane == 0) rp[0] = 0;
            __syncthreads();
            best = ~0ull;
            bool first = true;
            for (unsigned base = 0; base < total; base += 256u) {
                float4 q[4];
                bool val[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const unsigned fl = base + 64u * u + (unsigned)lane;
                    unsigned pos = qe.x;
                    val[u] = fl < total;
                    if (val[u]) sg_flat_locate(rp, rs, 8, 216, fl, pos);
                    q[u] = map[pos];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (base + 64u * u >= total) break; // (uniform)
                    const float d2 = sqdist3(me.x, me.y, me.z, q[u].x, q[u].y, q[u].z);
                    const unsigned long long key = (val[u] && d2 <= bound) ? pack_key(d2, __float_as_uint(q[u].w)) : ~0ull;
                    if (first) {
                        if (__ballot(key != ~0ull) != 0ull) { best = key; sg_sort64<false>(best); first = false; }
                    } else sg_absorb(best, key, k);
                }
            }
        }
        lev_sum += (unsigned long long)l;
        if (lane < k) {
            const bool valid = best != ~0ull;
            const unsigned orig = __float_as_uint(me.w);
            out_sidx[(size_t)k * orig + lane] = valid ? (int)inv[(unsigned)best] : -1;
            out_d2[(size_t)k * orig + lane] = valid ? __uint_as_float((unsigned)(best >> 32)) : INFINITY;
        }
    }
    if (diag && lane == 0 && lev_sum) atomicAdd(&st->levels, lev_sum); // (ICPMI_SELF_DIAG only: thousands of atomics on one address)
}

// ---- r6: the subset search of an APPENDED cloud (exact incremental SurfaceNormal: ops.hip surface_normals_dev) ----------------------------------
// The first m_old points of the cloud are the cloud of the handle's previous call and each of them remembers the d^2 of its k-th neighbour
// (dk).  A point's k nearest change only if an appended point lies strictly inside that ball (an appended point has the larger index: a tie
// loses).  Conservative test on the grid of THIS build: level l = the first whose cells (edge cell * 2^l) are at least the ball's radius wide
// -- then every point of the ball lies in the 3 x 3 x 3 cells of level l around the old point's, and those 27 are looked up in a bitmap of the
// cells appended points fell into.  A false positive is searched again and finds what it had.
struct SgLevels {
    int L, lmin;                    // levels lmin .. L - 1 have a bitmap (the top one is a single cell); a ball narrower than level lmin is tested there
    unsigned long long off[26];     // first bit of level l (cells per axis at level l: ((na - 1) >> l) + 1)
};

// one lane per (appended point, level): blockIdx.y = level.  The lanes of a wave that hit the same word (at the coarse levels: all of them)
// send ONE atomic -- thousands of atomics on one address serialise at 10 - 25 ns each.
// (the level table lives in device memory: a by-value struct indexed by a run-time level would be copied to scratch)
__global__ __launch_bounds__(256) void sg_dirty_kernel(const float4* __restrict__ pts, int64_t m_old, int64_t m, SgGrid g, const SgLevels* __restrict__ lvp, unsigned* __restrict__ bits)
{
    const int64_t i = m_old + (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool valid = i < m;
    const int l = lvp->lmin + (int)blockIdx.y;
    unsigned long long b = 0;
    if (valid) {
        const float4 p = pts[i];
        const int a0 = sg_cell_of(p.x, g.ox, g.inv_cell, g.na[0]) >> l, a1 = sg_cell_of(p.y, g.oy, g.inv_cell, g.na[1]) >> l, a2 = sg_cell_of(p.z, g.oz, g.inv_cell, g.na[2]) >> l;
        const int n0 = ((g.na[0] - 1) >> l) + 1, n1 = ((g.na[1] - 1) >> l) + 1;
        b = lvp->off[l] + ((unsigned long long)a2 * n1 + (unsigned long long)a1) * n0 + (unsigned long long)a0;
    }
    const unsigned word = (unsigned)(b >> 5);
    unsigned mask = valid ? 1u << (unsigned)(b & 31u) : 0u;
    unsigned long long todo = __ballot(valid);
    const int lane = threadIdx.x & 63;
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const unsigned lw = (unsigned)__builtin_amdgcn_readlane((int)word, leader);
        const bool same = valid && word == lw;
        const unsigned long long grp = __ballot(same);
        unsigned mm = same ? mask : 0u;
        for (int o = 32; o > 0; o >>= 1) mm |= (unsigned)__shfl_xor((int)mm, o, 64);
        if (lane == leader) atomicOr(&bits[lw], mm);
        todo &= ~grp;
    }
}

// one lane per SORTED position (neighbouring lanes look up the same words of the bitmaps): a selected point marks its sorted position for the
// cell kernel and joins the list the normals are solved from (original indices; order irrelevant: one row per point)
__global__ __launch_bounds__(256) void sg_affected_kernel(const float4* __restrict__ sorted, int64_t m, unsigned m_old, const float* __restrict__ dk, SgGrid g,
                                                          const SgLevels* __restrict__ lvp, const unsigned* __restrict__ bits,
                                                          unsigned char* __restrict__ sel_sorted, unsigned* __restrict__ list, unsigned* __restrict__ n_sel)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int lv_lmin = lvp->lmin, lv_L = lvp->L;
    bool hit = false;
    unsigned orig = 0;
    if (i < m) {
        const float4 p = sorted[i];
        orig = __float_as_uint(p.w);
        hit = orig >= m_old;
        if (!hit) {
            const float d2 = dk[orig];
            if (!(d2 < INFINITY)) hit = true; // fewer than k points so far: every appended point enters
            else {
                const float r = sqrt_up_sg(d2) * 1.00001f + (g.cell * 1e-3f + g.maxabs * 4e-6f);
                int l = lv_lmin;
                while (l < lv_L - 1 && g.cell * (float)(1u << l) < r) ++l;
                const int n0 = ((g.na[0] - 1) >> l) + 1, n1 = ((g.na[1] - 1) >> l) + 1, n2 = ((g.na[2] - 1) >> l) + 1;
                const unsigned long long off = lvp->off[l];
                const int c0 = sg_cell_of(p.x, g.ox, g.inv_cell, g.na[0]) >> l, c1 = sg_cell_of(p.y, g.oy, g.inv_cell, g.na[1]) >> l,
                          c2 = sg_cell_of(p.z, g.oz, g.inv_cell, g.na[2]) >> l;
                for (int dz = -1; dz <= 1 && !hit; ++dz)
                    for (int dy = -1; dy <= 1 && !hit; ++dy) {
                        const int z = c2 + dz, y = c1 + dy;
                        if (z < 0 || y < 0 || z >= n2 || y >= n1) continue;
                        const unsigned long long row = off + ((unsigned long long)z * n1 + (unsigned long long)y) * n0;
                        for (int dx = -1; dx <= 1; ++dx) {
                            const int x = c0 + dx;
                            if (x < 0 || x >= n0) continue;
                            const unsigned long long b = row + (unsigned long long)x;
                            if (bits[b >> 5] & (1u << (b & 31u))) { hit = true; break; }
                        }
                    }
            }
        }
        sel_sorted[i] = hit ? 1 : 0;
    }
    const unsigned long long bal = __ballot(hit);
    if (bal) {
        const int lane = threadIdx.x & 63;
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(n_sel, (unsigned)__popcll(bal));
        base = (unsigned)__builtin_amdgcn_readlane((int)base, 0);
        if (hit) list[base + (unsigned)__popcll(bal & ((1ull << lane) - 1ull))] = orig;
    }
}

// the groups of SGQ consecutive sorted positions (one wave of the cell kernel each) that hold a selected query: wlist[0] = count, then the groups
__global__ __launch_bounds__(256) void sg_wavelist_kernel(const unsigned char* __restrict__ sel_sorted, unsigned m, unsigned* __restrict__ wlist)
{
    static_assert(SGQ == 16, "one uint4 of flags per group");
    const unsigned gq = blockIdx.x * 256 + threadIdx.x, groups = (m + SGQ - 1) / SGQ;
    bool any = false;
    if (gq < groups) {
        if ((gq + 1) * SGQ <= m) { const uint4 v = reinterpret_cast<const uint4*>(sel_sorted)[gq]; any = (v.x | v.y | v.z | v.w) != 0u; }
        else for (unsigned i = gq * SGQ; i < m; ++i) any |= sel_sorted[i] != 0;
    }
    const unsigned long long bal = __ballot(any);
    if (bal) {
        const int lane = threadIdx.x & 63;
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(&wlist[0], (unsigned)__popcll(bal));
        base = (unsigned)__builtin_amdgcn_readlane((int)base, 0);
        if (any) wlist[1 + base + (unsigned)__popcll(bal & ((1ull << lane) - 1ull))] = gq;
    }
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

unsigned sg_queue_cap(int64_t m) { const int64_t waves = (m + SGQ - 1) / SGQ; return (unsigned)(((waves + SG_NQ - 1) / SG_NQ) * SGQ); }

void sg_launch_search(icpmi_ctx* c, SelfGridCtx* sg, const SgGrid& g, int k, int64_t m, int* d_sidx, float* d_d2, unsigned long long* sq_mapped, int diag,
                      const unsigned char* qsel = nullptr, const unsigned* wlist = nullptr, int64_t n_sel = 0)
{
    unsigned g1 = (unsigned)((m + SGQ - 1) / SGQ);
    const unsigned qcap = sg_queue_cap(m);
    if (wlist) g1 = (unsigned)std::max<int64_t>(1, std::min<int64_t>(g1, n_sel)); // (at most one group per selected query; the kernel reads the real count)
    hipLaunchKernelGGL(sg_cell_kernel, dim3(g1), dim3(64), 0, c->stream, g, (const float4*)c->d_map_sorted, (const unsigned*)sg->d_tstart,
                       (const unsigned*)sg->d_tbid, (const unsigned*)sg->d_f, (const unsigned*)sg->d_inv, sg->d_state, k, (unsigned)m,
                       m < (1ll << 24) ? 1 : 0, d_sidx, d_d2, sg->d_queue, qcap, sq_mapped, qsel, wlist);
    hipLaunchKernelGGL(sg_level_kernel, dim3(8192), dim3(64), 0, c->stream, g, (const float4*)c->d_map_sorted, (const unsigned*)sg->d_tstart,
                       (const unsigned*)sg->d_tbid, (const unsigned*)sg->d_f, (const unsigned*)sg->d_inv, sg->d_state, k, d_sidx, d_d2,
                       (const uint2*)sg->d_queue, qcap, diag);
}

} // namespace

void selfgrid_destroy(icpmi_ctx* c)
{
    SelfGridCtx* sg = c->sg;
    if (!sg) return;
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    dev_free(sg->d_tcnt); dev_free(sg->d_tstart); dev_free(sg->d_tbid); dev_free(sg->d_blist); dev_free(sg->d_f); dev_free(sg->d_coarse);
    dev_free(sg->d_queue); dev_free(sg->d_inv); dev_free(sg->d_part); dev_free(sg->d_state); dev_free(sg->d_dirty); dev_free(sg->d_sel); dev_free(sg->d_prev);
    delete sg;
    c->sg = nullptr;
}

// sub (may be null): d_pts[0 .. m_old) is the cloud of an earlier call whose k-th neighbour distances are in d_dk (original order): only the
// appended points and the old points an appended point may have entered the neighbourhood of are searched; their rows of d_sidx / d_d2 are
// written; sub->d_list (original indices, owned by the handle's grid) says which, sub->n_sel how many.
icpmi_status selfgrid_knn(icpmi_ctx* c, const float4* d_pts, int64_t m, int k, int* d_sidx, float* d_d2, SelfGridSubset* sub)
{
    if (sub) { sub->d_list = nullptr; sub->n_sel = 0; }
    if (m <= 0) return ICPMI_OK;
    if (k < 1 || k > ICPMI_MAX_K) { c->last_error = "self knn: k must be in [1, 32]"; return ICPMI_ERR_INVALID_ARG; }
    if (m > 0x7fffff00ll) { c->last_error = "self knn: cloud too large"; return ICPMI_ERR_UNSUPPORTED; }
    if (!c->sg) { c->sg = new (std::nothrow) SelfGridCtx(); if (!c->sg) { c->last_error = "out of host memory"; return ICPMI_ERR_HIP; } }
    SelfGridCtx* sg = c->sg;
    if (!sg->d_state) HIP_TRY(c, dev_malloc((void**)&sg->d_state, sizeof(SgState)));
    // ---- r6: an appended cloud whose previous sorted copy this grid still holds is built from THAT copy + the appended tail: the points arrive
    //      nearly in the order they leave in, so the counting sort's atomics and its scatter are coalesced (10 M points: sg_key 0.47 -> , sg_scatter 0.68 -> ms)
    const float4* src = d_pts;
    int keep_w = 0;
    if (sub && sub->m_old > 0 && sub->m_old < m && sg->d_prev && sg->prev_m == sub->m_old) {
        if ((size_t)m + 16 > sg->cap_prev) { // grow, keeping the sorted copy
            const size_t want = (size_t)m + (size_t)m / 4 + 64;
            float4* q = nullptr;
            HIP_TRY(c, dev_malloc((void**)&q, want * sizeof(float4)));
            HIP_TRY(c, hipMemcpyAsync(q, sg->d_prev, (size_t)sub->m_old * sizeof(float4), hipMemcpyDeviceToDevice, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            HIP_TRY(c, dev_free(sg->d_prev));
            sg->d_prev = q; sg->cap_prev = want;
        }
        hipLaunchKernelGGL(sg_tail_kernel, dim3((int)((m - sub->m_old + 255) / 256)), dim3(256), 0, c->stream, d_pts, sub->m_old, m, sg->d_prev);
        src = sg->d_prev; keep_w = 1;
    }
    sg->prev_m = 0; // (valid again once this build has left its copy behind)
    // ---- bounding box (one read-back: the table sizes depend on it) ----
    // (an appended cloud built from its previous copy: the box of the appended points, joined with the box that copy had)
    const int64_t bb_n = keep_w ? m - sub->m_old : m;
    const float4* bb_src = keep_w ? d_pts + sub->m_old : src;
    const int rblocks = (int)std::min<int64_t>((bb_n + SG_RB - 1) / SG_RB, 256);
    if (sg_cap(c, &sg->d_part, &sg->cap_part, (size_t)rblocks * 6) != ICPMI_OK) return ICPMI_ERR_HIP;
    hipLaunchKernelGGL(sg_reset_kernel, dim3(1), dim3(64), 0, c->stream, sg->d_state);
    hipLaunchKernelGGL(sg_bbox_kernel, dim3(rblocks), dim3(SG_RB), 0, c->stream, bb_src, bb_n, sg->d_part, sg->d_state);
    HIP_TRY(c, hipGetLastError());
    std::vector<float> part((size_t)rblocks * 6);
    unsigned bad = 0;
    if (read_back2(c, part.data(), sg->d_part, part.size() * sizeof(float), &bad, &sg->d_state->bad, sizeof(unsigned)) != ICPMI_OK) return ICPMI_ERR_HIP;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}, maxabs = 0.f;
    if (keep_w) for (int r = 0; r < 3; ++r) { lo[r] = sg->prev_lo[r]; hi[r] = sg->prev_hi[r]; }
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
    const double target = std::max(2.0, 0.8 * (double)k);
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
            sg_cap(c, &sg->d_coarse, &sg->cap_coarse, (size_t)m + 16) != ICPMI_OK || sg_cap(c, &sg->d_queue, &sg->cap_queue, (size_t)sg_queue_cap(m) * SG_NQ + 1) != ICPMI_OK || sg_cap(c, &sg->d_inv, &sg->cap_inv, (size_t)m + 1) != ICPMI_OK ||
            ensure_cap(c, &c->d_keys, &c->cap_keys, (size_t)m) != ICPMI_OK || ensure_cap(c, &c->d_map_sorted, &c->cap_map, (size_t)m + 16) != ICPMI_OK)
            return ICPMI_ERR_HIP;
        if (trial > 0) hipLaunchKernelGGL(sg_reset_kernel, dim3(1), dim3(64), 0, c->stream, sg->d_state);
        // (tiles per workgroup: enough workgroups to fill the chip, as few atomics on the block-id counter as that allows)
        const int key_tiles = (int)std::max<int64_t>(1, std::min<int64_t>(SG_KEY_TILES, blocks256 / 2048));
        hipLaunchKernelGGL(sg_key_kernel, dim3((blocks256 + key_tiles - 1) / key_tiles), dim3(256), 0, c->stream, src, m, g, c->d_keys, sg->d_tcnt, sg->d_tbid, sg->d_blist,
                           sg->d_state, key_tiles);
        HIP_TRY(c, hipGetLastError());
        if (device_exclusive_scan_cursor(c, sg->d_tcnt, sg->d_tstart, g.tsize, (unsigned)m, true) != ICPMI_OK) return ICPMI_ERR_HIP;
        sg->tcnt_clean = true;
        hipLaunchKernelGGL(sg_scatter_kernel, dim3(blocks256), dim3(256), 0, c->stream, src, m, (const unsigned*)c->d_keys, sg->d_tstart + 1, sg->d_coarse, keep_w);
        hipLaunchKernelGGL(sg_block_sort_kernel, dim3(std::max(1u, std::min(blocks_max, 8192u))), dim3(64), 0, c->stream, (const float4*)sg->d_coarse,
                           c->d_map_sorted, g, (const unsigned*)sg->d_tstart, (const uint4*)sg->d_blist, sg->d_f, sg->d_inv, sg->d_state);
        HIP_TRY(c, hipGetLastError());
        if (trial + 1 >= trials) break;
        unsigned long long sqp[64], sq = 0;
        if (read_back(c, sqp, sg->d_state->sqpart, sizeof sqp) != ICPMI_OK) return ICPMI_ERR_HIP;
        for (int i = 0; i < 64; ++i) sq += sqp[i];
        const double sb = (double)sq / (double)m;
        if (sb > 0.85 * target && sb < 1.18 * target) break;
        const double next = (double)g.cell * std::min(4.0, std::max(0.25, sqrt(target / sb)));
        if (fabs(next - (double)g.cell) < 0.05 * (double)g.cell) break;
        cell = next;
    }
    sg->cell = g.cell; sg->m = m; sg->k = k; ++sg->seq;
    if (sub) { // a tracked cloud (the resident map of append-only updates): leave the sorted copy behind for the build of the next append
        if (sg_cap(c, &sg->d_prev, &sg->cap_prev, (size_t)m + 16) != ICPMI_OK) return ICPMI_ERR_HIP; // (a reallocation drops the old content: it is rewritten below)
        HIP_TRY(c, hipMemcpyAsync(sg->d_prev, c->d_map_sorted, (size_t)m * sizeof(float4), hipMemcpyDeviceToDevice, c->stream));
        sg->prev_m = m;
        for (int r = 0; r < 3; ++r) { sg->prev_lo[r] = lo[r]; sg->prev_hi[r] = hi[r]; }
    }

    // ---- the subset of an appended cloud ----
    const unsigned char* qsel = nullptr;
    const unsigned* wlist = nullptr;
    if (sub && sub->m_old > 0 && sub->m_old < m && sub->d_dk) {
        SgLevels lv{};
        unsigned long long bits_total = 0;
        auto cells_at = [&](int l) { return (unsigned long long)(((g.na[0] - 1) >> l) + 1) * (unsigned long long)(((g.na[1] - 1) >> l) + 1) * (unsigned long long)(((g.na[2] - 1) >> l) + 1); };
        int L = 0;
        for (; L < 26; ++L) if (cells_at(L) == 1) { ++L; break; }
        lv.L = L;
        // fine levels of a huge sparse box are folded into the first level whose bitmaps fit 2^31 bits (256 MB): conservative, rarely needed
        for (lv.lmin = 0; lv.lmin < L - 1; ++lv.lmin) {
            bits_total = 0;
            for (int l = lv.lmin; l < L; ++l) bits_total += cells_at(l);
            if (bits_total <= (1ull << 31)) break;
        }
        bits_total = 0;
        for (int l = lv.lmin; l < L; ++l) { lv.off[l] = bits_total; bits_total += cells_at(l); }
        const size_t words = (size_t)((bits_total + 31) / 32) + 128; // + 64 words: the counter of the selection, + 64: the level table
        // d_sel: [0, m) by sorted position, then (4-byte aligned) the list of the selected original indices
        const size_t list_at = ((size_t)m + 3) / 4 * 4;
        const size_t groups = (size_t)((m + SGQ - 1) / SGQ);
        if (sg_cap(c, &sg->d_dirty, &sg->cap_dirty, words) != ICPMI_OK || sg_cap(c, &sg->d_sel, &sg->cap_sel, list_at + (size_t)4 * (m + groups + 2) + 64) != ICPMI_OK) return ICPMI_ERR_HIP;
        static_assert(sizeof(SgLevels) <= 64 * sizeof(unsigned), "level table");
        HIP_TRY(c, hipMemsetAsync(sg->d_dirty, 0, words * sizeof(unsigned), c->stream));
        unsigned* n_sel = sg->d_dirty + (words - 128);
        const SgLevels* d_lv = reinterpret_cast<const SgLevels*>(sg->d_dirty + (words - 64));
        { const icpmi_status us = upload_small(c, sg->d_dirty + (words - 64), &lv, sizeof lv); if (us != ICPMI_OK) return us; }
        unsigned* list = reinterpret_cast<unsigned*>(sg->d_sel + list_at);
        const int64_t added = m - sub->m_old;
        hipLaunchKernelGGL(sg_dirty_kernel, dim3((int)((added + 255) / 256), lv.L - lv.lmin), dim3(256), 0, c->stream, d_pts, sub->m_old, m, g, d_lv, sg->d_dirty);
        hipLaunchKernelGGL(sg_affected_kernel, dim3(blocks256), dim3(256), 0, c->stream, (const float4*)c->d_map_sorted, m, (unsigned)sub->m_old, sub->d_dk, g, d_lv,
                           (const unsigned*)sg->d_dirty, sg->d_sel, list, n_sel);
        HIP_TRY(c, hipGetLastError());
        unsigned* d_wl = list + m; // [0] = count (cleared here), then the groups
        HIP_TRY(c, hipMemsetAsync(d_wl, 0, sizeof(unsigned), c->stream));
        hipLaunchKernelGGL(sg_wavelist_kernel, dim3((int)((groups + 255) / 256)), dim3(256), 0, c->stream, (const unsigned char*)sg->d_sel, (unsigned)m, d_wl);
        HIP_TRY(c, hipGetLastError());
        qsel = sg->d_sel; wlist = d_wl;
        sub->d_list = list;
        {   // the size of the list: the normals are solved from it (one small read-back)
            unsigned cnt = 0;
            if (read_back(c, &cnt, n_sel, sizeof cnt) != ICPMI_OK) return ICPMI_ERR_HIP;
            sub->n_sel = (int64_t)cnt;
            if (diag) fprintf(stderr, "[icpmi self-knn] subset: %lld of %lld points searched (%lld appended), levels %d..%d, %llu bitmap bits\n", (long long)sub->n_sel,
                              (long long)m, (long long)added, lv.lmin, lv.L - 1, bits_total);
        }
    }
    // ---- search ----
    sg_launch_search(c, sg, g, k, m, d_sidx, d_d2, d_sq, diag, qsel, wlist, sub ? sub->n_sel : 0);
    HIP_TRY(c, hipGetLastError());
    if (diag) {
        SgState hs{};
        unsigned qtot = 0;
        if (read_back(c, &hs, sg->d_state, sizeof hs) == ICPMI_OK) for (int i = 0; i < 64; ++i) qtot += hs.qcount[i];
        if (qtot || hs.nblk)
            fprintf(stderr, "[icpmi self-knn] m %lld k %d: A-cell %.3f (%d x %d x %d), T %d entries, %u blocks, size-biased occupancy %.1f (target %.1f); "
                            "%u queries (%.2f %%) through the levels, mean end level %.2f\n",
                    (long long)m, k, (double)g.cell, g.na[0], g.na[1], g.na[2], g.tsize, hs.nblk, (double)hs.sq / (double)m, target, qtot,
                    100.0 * qtot / (double)m, qtot ? (double)hs.levels / (double)qtot : 0.0);
    }
    return ICPMI_OK;
}
