// map_build.hip -- device side of PM::ICPSequence::setMap (reference call sites
// norlab_icp_mapper/Map.cpp:111,178,528,581; semantics SURVEY.md B.1): copy the cloud, centre it on
// its centroid and build the nearest-neighbour index.  Where upstream builds a libnabo kd-tree
// (O(M log M), single-threaded) this builds a dense uniform grid by counting sort:
//   stats (sum, bbox)  ->  cell keys + histogram  ->  exclusive scan  ->  scatter.
// HBM traffic per build (M points): read 16 M (stats) + read 16 M / write 4 M (keys) + read 20 M /
// write 16 M (scatter) (+ 32 M for normals) and two passes over the cell table.
#include "common.h"
#include <cstring>

namespace {

constexpr int RB = 256; // reduction block

// ---- pass 1: sum (double) and bounding box of the raw cloud ----------------------------------
__global__ __launch_bounds__(RB) void stats_kernel(const float4* __restrict__ pts, int64_t m, double* __restrict__ part)
{
    double sx = 0, sy = 0, sz = 0;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = (int64_t)blockIdx.x * RB + threadIdx.x; i < m; i += (int64_t)gridDim.x * RB) {
        const float4 p = pts[i];
        sx += p.x; sy += p.y; sz += p.z;
        lo[0] = fminf(lo[0], p.x); lo[1] = fminf(lo[1], p.y); lo[2] = fminf(lo[2], p.z);
        hi[0] = fmaxf(hi[0], p.x); hi[1] = fmaxf(hi[1], p.y); hi[2] = fmaxf(hi[2], p.z);
    }
    __shared__ double sh[RB * 3];
    __shared__ float shlo[RB * 3], shhi[RB * 3];
    const int t = threadIdx.x;
    sh[t] = sx; sh[RB + t] = sy; sh[2 * RB + t] = sz;
    for (int r = 0; r < 3; ++r) { shlo[r * RB + t] = lo[r]; shhi[r * RB + t] = hi[r]; }
    __syncthreads();
    for (int s = RB / 2; s > 0; s >>= 1) {
        if (t < s) {
            for (int r = 0; r < 3; ++r) {
                sh[r * RB + t] += sh[r * RB + t + s];
                shlo[r * RB + t] = fminf(shlo[r * RB + t], shlo[r * RB + t + s]);
                shhi[r * RB + t] = fmaxf(shhi[r * RB + t], shhi[r * RB + t + s]);
            }
        }
        __syncthreads();
    }
    if (t == 0) {
        double* o = part + (size_t)blockIdx.x * 9;
        for (int r = 0; r < 3; ++r) { o[r] = sh[r * RB]; o[3 + r] = shlo[r * RB]; o[6 + r] = shhi[r * RB]; }
    }
}

__device__ __forceinline__ int cell_of(float v, float o, float inv, int n)
{
    int c = (int)floorf((v - o) * inv);
    return c < 0 ? 0 : (c >= n ? n - 1 : c);
}

// ---- runs of equal keys inside a wave become ONE atomic per run --------------------------------
// The coarser pyramid levels are built from the cell-sorted level-0 array (10^3..10^6 points per coarse cell: one atomic per
// point serialises at ~6.5 ns each on the same address -- 11 ms per kernel at 1 M points), and the cloud handed to set_map is
// often spatially ordered too (the octree filter leaves the map in Morton order; a lidar scan is ordered along its beams):
// level 0 uses the same trick (r2: 98 -> 4x fewer device atomics on an octree-ordered 0.9 M-point map).
// (WaveRun / wave_run: common.h -- the bucket grid of the DynamicPoints module uses them too)

// ---- pass 2: keys + per-cell histogram -------------------------------------------------------
__global__ __launch_bounds__(256) void key_kernel(const float4* __restrict__ pts, int64_t m, float mx, float my, float mz,
                                                  GridParams g, unsigned* __restrict__ keys, unsigned* __restrict__ count,
                                                  unsigned* __restrict__ n_occ, int run_atomics)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool valid = i < m;
    const float4 p = pts[valid ? i : 0];
    const int cx = cell_of(p.x - mx, g.ox, g.inv_cell, g.nx);
    const int cy = cell_of(p.y - my, g.oy, g.inv_cell, g.ny);
    const int cz = cell_of(p.z - mz, g.oz, g.inv_cell, g.nz);
    const unsigned key = (unsigned)((cz * g.ny + cy) * g.nx + cx);
    if (valid) keys[i] = key;
    // number of occupied cells = lanes that found their cell's counter at zero.  One same-address device atomic per such lane
    // serialises at ~1.3 ns each (115 k occupied cells: 150 us of a kernel that otherwise takes 25): counted per wave by
    // ballot, per workgroup in LDS, one global atomic per workgroup.
    bool first;
    if (!run_atomics) first = valid && atomicAdd(&count[key], 1u) == 0u;
    else {
        const WaveRun r = wave_run(key, valid);
        first = r.head && atomicAdd(&count[key], (unsigned)r.len) == 0u;
    }
    __shared__ unsigned occ;
    if (threadIdx.x == 0) occ = 0;
    __syncthreads();
    const unsigned long long firsts = __ballot(first);
    if ((threadIdx.x & 63) == 0 && firsts) atomicAdd(&occ, (unsigned)__popcll(firsts));
    __syncthreads();
    if (threadIdx.x == 0 && occ) atomicAdd(n_occ, occ);
}

// ---- exclusive scan of the cell histogram (3 kernels) ----------------------------------------
constexpr int SCAN_T = 256, SCAN_E = 8, SCAN_CHUNK = SCAN_T * SCAN_E;

__device__ __forceinline__ unsigned block_exclusive_scan(unsigned v, unsigned* sh, unsigned* total)
{
    // sh: SCAN_T entries
    const int t = threadIdx.x;
    sh[t] = v;
    __syncthreads();
    for (int off = 1; off < SCAN_T; off <<= 1) {
        unsigned add = t >= off ? sh[t - off] : 0u;
        __syncthreads();
        sh[t] += add;
        __syncthreads();
    }
    const unsigned incl = sh[t];
    if (total) *total = sh[SCAN_T - 1];
    __syncthreads();
    return incl - v;
}

__global__ __launch_bounds__(SCAN_T) void scan_sums_kernel(const unsigned* __restrict__ in, int n, unsigned* __restrict__ sums)
{
    __shared__ unsigned sh[SCAN_T];
    const int base = blockIdx.x * SCAN_CHUNK + threadIdx.x * SCAN_E;
    unsigned s = 0;
    for (int e = 0; e < SCAN_E; ++e) if (base + e < n) s += in[base + e];
    unsigned tot;
    block_exclusive_scan(s, sh, &tot);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(SCAN_T) void scan_top_kernel(unsigned* __restrict__ sums, int nb)
{
    __shared__ unsigned sh[SCAN_T];
    __shared__ unsigned carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base0 = 0; base0 < nb; base0 += SCAN_CHUNK) {
        const int base = base0 + threadIdx.x * SCAN_E;
        unsigned v[SCAN_E], s = 0;
        for (int e = 0; e < SCAN_E; ++e) { v[e] = base + e < nb ? sums[base + e] : 0u; s += v[e]; }
        unsigned tot;
        unsigned ex = block_exclusive_scan(s, sh, &tot) + carry;
        for (int e = 0; e < SCAN_E; ++e) { if (base + e < nb) sums[base + e] = ex; ex += v[e]; }
        __syncthreads();
        if (threadIdx.x == 0) carry += tot;
        __syncthreads();
    }
}

// in-place: count[] -> start[]; start[n] = total
__global__ __launch_bounds__(SCAN_T) void scan_final_kernel(unsigned* __restrict__ data, int n, const unsigned* __restrict__ sums,
                                                           unsigned total)
{
    __shared__ unsigned sh[SCAN_T];
    const int base = blockIdx.x * SCAN_CHUNK + threadIdx.x * SCAN_E;
    unsigned v[SCAN_E], s = 0;
    for (int e = 0; e < SCAN_E; ++e) { v[e] = base + e < n ? data[base + e] : 0u; s += v[e]; }
    unsigned ex = block_exclusive_scan(s, sh, nullptr) + sums[blockIdx.x];
    for (int e = 0; e < SCAN_E; ++e) { if (base + e < n) data[base + e] = ex; ex += v[e]; }
    if (blockIdx.x == 0 && threadIdx.x == 0) data[n] = total;
}

// ---- r5: the same scan in TWO kernels (n <= SCAN2_MAX_NB chunks).  The offset of a chunk is the sum of the totals of the chunks before it;
// every workgroup of the second kernel adds those up itself (at most 8 192 words that sit in the L2) instead of waiting for a one-workgroup
// launch that scans them: one launch (~4.5 us behind a 3 us kernel) less for each of the ~7 scans of a map update, and `in` may differ from
// `out` (the flag -> position scans of the compactions copied their input first).  Integer sums: the result is the three-kernel scan's.
constexpr int SCAN2_MAX_NB = 8192;

__device__ __forceinline__ unsigned wave_incl_scan_u32(unsigned v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned t = (unsigned)__shfl_up((int)v, o, 64); if (lane >= o) v += t; }
    return v;
}
// exclusive scan over the SCAN_T threads of a workgroup; sh: SCAN_T / 64 words; *total (optional): the workgroup's sum
__device__ __forceinline__ unsigned block_exclusive_scan_fast(unsigned v, unsigned* sh, unsigned* total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned incl = wave_incl_scan_u32(v);
    if (lane == 63) sh[wave] = incl;
    __syncthreads();
    unsigned base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SCAN_T / 64; ++w) { const unsigned t = sh[w]; if (w < wave) base += t; tot += t; }
    if (total) *total = tot;
    __syncthreads();
    return base + incl - v;
}

__global__ __launch_bounds__(SCAN_T) void scan2_sums_kernel(const unsigned* __restrict__ in, int n, unsigned* __restrict__ sums)
{
    __shared__ unsigned sh[SCAN_T / 64];
    const int base = blockIdx.x * SCAN_CHUNK + threadIdx.x * SCAN_E;
    unsigned s = 0;
#pragma unroll
    for (int e = 0; e < SCAN_E; ++e) if (base + e < n) s += in[base + e];
    unsigned tot;
    block_exclusive_scan_fast(s, sh, &tot);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

// shift = 1: out[0] = 0, out[i + 1] = exclusive sum of in[0..i), out[n + 1] = total -- the layout the cursor scatters below turn into the
// plain starts.  zero_in: in[0 .. n + 1] is left zero (the count table cleans itself for its next user: no memset in front of a build).
// shift / zero_in need in != out.
__global__ __launch_bounds__(SCAN_T) void scan2_final_kernel(const unsigned* __restrict__ in, unsigned* __restrict__ out, int n,
                                                             const unsigned* __restrict__ sums, unsigned total, int shift, unsigned* __restrict__ zero_in,
                                                             unsigned* __restrict__ sum_out = nullptr /* the sum of in[0..n): e.g. a word of host-mapped memory */,
                                                             unsigned* __restrict__ tail_out = nullptr /* receives in[n + 1] (a grid build's occupancy word) before it is zeroed */,
                                                             unsigned sum_tag = 0u /* != 0: sum_out is a 64-bit slot of host-mapped memory and receives tag << 32 | sum with a system-scope release (a host that spins on the tag) */)
{
    __shared__ unsigned sh[SCAN_T / 64];
    const int base = blockIdx.x * SCAN_CHUNK + threadIdx.x * SCAN_E;
    unsigned v[SCAN_E], s = 0;
#pragma unroll
    for (int e = 0; e < SCAN_E; ++e) { v[e] = base + e < n ? in[base + e] : 0u; s += v[e]; }
    unsigned part = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += SCAN_T) part += sums[b];
    unsigned off;
    block_exclusive_scan_fast(part, sh, &off);
    unsigned wg_tot;
    unsigned ex = block_exclusive_scan_fast(s, sh, &wg_tot) + off;
    if (sum_out && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        if (sum_tag) __hip_atomic_store(reinterpret_cast<unsigned long long*>(sum_out), ((unsigned long long)sum_tag << 32) | (unsigned long long)(off + wg_tot), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        else *sum_out = off + wg_tot;
    }
#pragma unroll
    for (int e = 0; e < SCAN_E; ++e) { if (base + e < n) { out[base + e + shift] = ex; if (zero_in && v[e]) zero_in[base + e] = 0u; } ex += v[e]; }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        out[n + shift] = total;
        if (shift) out[0] = 0u;
        if (tail_out) *tail_out = in[n + 1];
        if (zero_in) { zero_in[n] = 0u; zero_in[n + 1] = 0u; }
    }
}

// ---- pass 3: scatter into cell order ---------------------------------------------------------
// Order inside a cell is whatever the atomics give; nothing downstream depends on it because every
// nearest-neighbour comparison is on the pair (d^2, original index).
__global__ __launch_bounds__(256) void scatter_kernel(const float4* __restrict__ pts, const float* __restrict__ normals3, int64_t m,
                                                      float mx, float my, float mz, const unsigned* __restrict__ keys,
                                                      unsigned* __restrict__ cursor /* = starts + 1: cursor[key] = next free slot of cell `key` */,
                                                      float4* __restrict__ out, float4* __restrict__ out_n, int run_atomics,
                                                      unsigned* __restrict__ okey, float4* __restrict__ twin)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool valid = i < m;
    const float4 p = pts[valid ? i : 0];
    const unsigned key = keys[valid ? i : 0];
    unsigned pos;
    if (!run_atomics) { if (!valid) return; pos = atomicAdd(&cursor[key], 1u); }
    else {
        const WaveRun r = wave_run(key, valid);
        unsigned base = 0;
        if (r.head) base = atomicAdd(&cursor[key], (unsigned)r.len);
        base = (unsigned)__shfl((int)base, r.head_lane, 64);
        if (!valid) return;
        pos = base + (unsigned)r.rank;
    }
    out[pos] = make_float4(p.x - mx, p.y - my, p.z - mz, __uint_as_float((unsigned)i));
    if (out_n) out_n[pos] = make_float4(normals3[3 * i], normals3[3 * i + 1], normals3[3 * i + 2], 0.f);
    if (okey) okey[pos] = key;                                                   // (what an incremental insert of the next append needs,
    if (twin) twin[pos] = make_float4(p.x, p.y, p.z, __uint_as_float((unsigned)i)); //  map_insert: the cell of every position, the raw points in sorted order)
}

__global__ __launch_bounds__(256) void lvl_key_kernel(const float4* __restrict__ pts0, int64_t m, GridParams g,
                                                      unsigned* __restrict__ keys, unsigned* __restrict__ count)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool valid = i < m;
    unsigned key = 0xffffffffu;
    if (valid) {
        const float4 p = pts0[i];
        const int cx = cell_of(p.x, g.ox, g.inv_cell, g.nx);
        const int cy = cell_of(p.y, g.oy, g.inv_cell, g.ny);
        const int cz = cell_of(p.z, g.oz, g.inv_cell, g.nz);
        key = (unsigned)((cz * g.ny + cy) * g.nx + cx);
        keys[i] = key;
    }
    const WaveRun r = wave_run(key, valid);
    if (r.head) atomicAdd(&count[key], (unsigned)r.len);
}

// point and normal of every level-0 position side by side (common.h: d_map_pn)
__global__ __launch_bounds__(256) void pn_kernel(const float4* __restrict__ pts, const float4* __restrict__ nrm, int64_t m, float4* __restrict__ pn)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    pn[2 * i] = pts[i];
    pn[2 * i + 1] = nrm[i];
}

__global__ __launch_bounds__(256) void lvl_scatter_kernel(const float4* __restrict__ pts0, int64_t m, const unsigned* __restrict__ keys,
                                                          unsigned* __restrict__ cursor /* = starts + 1 */,
                                                          float4* __restrict__ out, unsigned* __restrict__ pos0, unsigned* __restrict__ okey)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool valid = i < m;
    const unsigned key = valid ? keys[i] : 0xffffffffu;
    const WaveRun r = wave_run(key, valid);
    unsigned base = 0;
    if (r.head) base = atomicAdd(&cursor[key], (unsigned)r.len);
    base = (unsigned)__shfl((int)base, r.head_lane, 64);
    if (!valid) return;
    const unsigned pos = base + (unsigned)r.rank;
    out[pos] = pts0[i];
    pos0[pos] = (unsigned)i;
    if (okey) okey[pos] = key;
}

// ---- incremental insert (common.h: d_lvl_key ...) -------------------------------------------------------------------------------
// delta point j (original index m0 + j): its cell at this level from the coordinates the index will store, its rank among the delta
// points of that cell (arrival order of the atomics: the order inside a cell is free, see scatter_kernel)
__global__ __launch_bounds__(256) void ins_key_kernel(const float4* __restrict__ pts, int64_t m0, int64_t n, float mx, float my, float mz, GridParams g,
                                                      unsigned* __restrict__ dkey, unsigned* __restrict__ drank, unsigned* __restrict__ dcount)
{
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const float4 p = pts[m0 + j];
    const int cx = cell_of(p.x - mx, g.ox, g.inv_cell, g.nx);
    const int cy = cell_of(p.y - my, g.oy, g.inv_cell, g.ny);
    const int cz = cell_of(p.z - mz, g.oz, g.inv_cell, g.nz);
    const unsigned key = (unsigned)((cz * g.ny + cy) * g.nx + cx);
    dkey[j] = key;
    drank[j] = atomicAdd(&dcount[key], 1u);
}

// Level 0: old sorted position s -> s + (delta points in cells before its cell).  The centred coordinates are recomputed from the RAW
// point with the new centroid (exactly scatter_kernel's arithmetic); the raw points travel in a twin array in the same order
// (d_raw0: streamed, not gathered by original index -- a random 16-byte gather per point was 370 us of this kernel at 10 M points).
// Handles that index raw coordinates (centroid 0) have no twin: their points are copied.
__global__ __launch_bounds__(256) void ins_move0_kernel(const float4* __restrict__ pts_old, const float4* __restrict__ twin_old,
                                                        const unsigned* __restrict__ key_old, int64_t m0, const unsigned* __restrict__ dstart,
                                                        float mx, float my, float mz, float4* __restrict__ pts_new, float4* __restrict__ twin_new,
                                                        unsigned* __restrict__ key_new, const float* __restrict__ normals3,
                                                        const float4* __restrict__ nrm_old, float4* __restrict__ nrm_new, float4* __restrict__ pn_new)
{
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= m0) return;
    const unsigned c = key_old[s];
    const unsigned np = (unsigned)s + dstart[c];
    float4 p;
    if (twin_old) {
        const float4 r = twin_old[s];
        twin_new[np] = r;
        p = make_float4(r.x - mx, r.y - my, r.z - mz, r.w);
    } else p = pts_old[s];
    pts_new[np] = p;
    key_new[np] = c;
    if (nrm_new) {
        // (normals3 != nullptr: the caller recomputed the whole field -- gathered by original index; else the old sorted normals move along)
        float4 nn;
        if (normals3) { const unsigned o = __float_as_uint(p.w); nn = make_float4(normals3[3 * (size_t)o], normals3[3 * (size_t)o + 1], normals3[3 * (size_t)o + 2], 0.f); }
        else nn = nrm_old[s];
        nrm_new[np] = nn;
        if (pn_new) { pn_new[2 * (size_t)np] = p; pn_new[2 * (size_t)np + 1] = nn; }
    }
}

// Level l > 0: the entry moves by the delta points in cells before its cell OF THIS LEVEL; its level-0 position moves by the delta
// points before its level-0 cell, and its coordinates are read from the NEW level-0 array at that position (both orders are spatial:
// the gathers stay inside a few contiguous runs) -- what lvl_scatter_kernel copies in a full build.
__global__ __launch_bounds__(256) void ins_movel_kernel(const unsigned* __restrict__ key_old, const unsigned* __restrict__ pos0_old, int64_t m0,
                                                        const unsigned* __restrict__ dstart, const unsigned* __restrict__ key0_old,
                                                        const unsigned* __restrict__ dstart0, const float4* __restrict__ pts0_new,
                                                        float4* __restrict__ pts_new, unsigned* __restrict__ key_new, unsigned* __restrict__ pos0_new)
{
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= m0) return;
    const unsigned c = key_old[s];
    const unsigned np = (unsigned)s + dstart[c];
    const unsigned p0 = pos0_old[s];
    const unsigned p0n = p0 + dstart0[key0_old[p0]];
    pts_new[np] = pts0_new[p0n];
    key_new[np] = c;
    pos0_new[np] = p0n;
}

// delta point j goes behind the old points of its cell
__global__ __launch_bounds__(256) void ins_delta_kernel(const float4* __restrict__ pts, int64_t m0, int64_t n, float mx, float my, float mz,
                                                        const unsigned* __restrict__ dkey, const unsigned* __restrict__ drank,
                                                        const unsigned* __restrict__ cs_old, const unsigned* __restrict__ dstart,
                                                        float4* __restrict__ pts_new, unsigned* __restrict__ key_new, unsigned* __restrict__ inv,
                                                        unsigned* __restrict__ pos0_new, const float* __restrict__ normals3, float4* __restrict__ nrm_new,
                                                        float4* __restrict__ pn_new, float4* __restrict__ twin_new)
{
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const float4 r = pts[m0 + j];
    const unsigned c = dkey[j];
    const unsigned o = (unsigned)(m0 + j);
    const unsigned np = cs_old[c + 1] + dstart[c] + drank[j];
    const float4 p = make_float4(r.x - mx, r.y - my, r.z - mz, __uint_as_float(o));
    pts_new[np] = p;
    key_new[np] = c;
    if (twin_new) twin_new[np] = make_float4(r.x, r.y, r.z, __uint_as_float(o));
    if (pos0_new) pos0_new[np] = inv[j]; // (inv: level-0 position of delta point j, written by this kernel's level-0 launch)
    else {
        inv[j] = np;
        if (nrm_new) {
            const float4 nn = make_float4(normals3[3 * (size_t)o], normals3[3 * (size_t)o + 1], normals3[3 * (size_t)o + 2], 0.f);
            nrm_new[np] = nn;
            if (pn_new) { pn_new[2 * (size_t)np] = p; pn_new[2 * (size_t)np + 1] = nn; }
        }
    }
}

// newly != nullptr (level 0): also counts the cells the delta occupies for the first time (icpmi_get_grid_info stays true after an insert)
__global__ __launch_bounds__(256) void ins_cs_kernel(const unsigned* __restrict__ cs_old, const unsigned* __restrict__ dstart, int ncells1,
                                                     unsigned* __restrict__ cs_new, unsigned* __restrict__ newly)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < ncells1) cs_new[c] = cs_old[c] + dstart[c];
    if (newly) {
        const bool fresh = c + 1 < ncells1 && cs_old[c + 1] == cs_old[c] && dstart[c + 1] > dstart[c];
        const unsigned long long b = __ballot(fresh);
        if ((threadIdx.x & 63) == 0 && b) atomicAdd(newly, (unsigned)__popcll(b));
    }
}

// ---- query (reading) sort by super-tile ----------------------------------------------------------
// super-tile = STX x STY x STZ grid cells (long in x: the cell-sorted map is x-fastest).  Queries
// outside the grid clamp to the border super-tile.  The k = 1 loop keeps its per-query state and runs
// its pair sums in this order, so the order must be a function of the input alone: a STABLE LSD
// radix sort on the super-tile key (6-bit digits, 1024 elements per workgroup, two kernels per pass):
//   qhist: per-workgroup digit histogram -> count[digit][workgroup], total[digit]
//   qpass: stable ranks (wave ballots + a 16-entry LDS prefix per digit), workgroup base from
//          count / total (summed by the workgroup itself: 64 x nwg words), scatter.
// Equal keys keep ascending original index.  The last pass scatters the points themselves.
constexpr int STX = 16, STY = 4, STZ = 4;
constexpr int QS_BITS = 6, QS_BINS = 1 << QS_BITS, QS_EPB = 1024;

__device__ __forceinline__ unsigned st_key(const float4 p, const GridParams& g, int tx, int ty)
{
    const int cx = (int)fminf(fmaxf(floorf((p.x - g.ox) * g.inv_cell), 0.f), (float)(g.nx - 1));
    const int cy = (int)fminf(fmaxf(floorf((p.y - g.oy) * g.inv_cell), 0.f), (float)(g.ny - 1));
    const int cz = (int)fminf(fmaxf(floorf((p.z - g.oz) * g.inv_cell), 0.f), (float)(g.nz - 1));
    return (unsigned)(((cz / STZ) * ty + (cy / STY)) * tx + (cx / STX));
}

// lanes of the wave (among `valid` ones) holding the same 6-bit digit as this lane
__device__ __forceinline__ unsigned long long match_digit(unsigned d, bool valid)
{
    unsigned long long m = __ballot(valid);
#pragma unroll
    for (int b = 0; b < QS_BITS; ++b) {
        const bool bit = (d >> b) & 1u;
        const unsigned long long bal = __ballot(bit);
        m &= bit ? bal : ~bal;
    }
    return m;
}

// (blockIdx.y = reading of a batch: slices of the per-query arrays qs elements apart, one count table of `tabstride`
// words per reading)
// First kernel of the sort: tile keys and the digit counts of pass 0, per workgroup.  CENTRE (the registration head, r3): the points
// are read from the readings as handed in and moved by -mean through the fmaf chain of a transform (what centre_kernel did in a
// launch of its own) into `pts`, the sort's input.  It also clears the count tables of the later passes, which the scatter kernels
// fill with atomics.  r2 cleared a `total` table with a memset node and ran a histogram kernel per pass.
template <bool CENTRE>
__global__ __launch_bounds__(256) void qfirst_kernel(BatchSrc raw, float mx, float my, float mz, float4* __restrict__ pts, BatchArgs ba, GridParams g,
                                                     int tx, int ty, unsigned* __restrict__ keys_out, int nwg, unsigned* __restrict__ tables,
                                                     int later_words, int tab, int qs, int tabstride, IcpState* __restrict__ st_stamp)
{
    if (st_stamp && blockIdx.x == 0 && threadIdx.x == 0) st_stamp[blockIdx.y].t_start = (unsigned long long)wall_clock64(); // the registration's first kernel
    const int n = ba.n[blockIdx.y];
    const float4* __restrict__ src = CENTRE ? raw.p[blockIdx.y] : nullptr;
    pts += (size_t)blockIdx.y * qs;
    keys_out += (size_t)blockIdx.y * 4 * qs;
    tables += (size_t)blockIdx.y * tabstride;
    unsigned* count = tables;
    __shared__ unsigned h[QS_BINS];
    const int t = threadIdx.x, lane = t & 63;
    if (t < QS_BINS) h[t] = 0;
    for (int i = blockIdx.x * 256 + t; i < later_words; i += nwg * 256) tables[tab + i] = 0;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < QS_EPB / 256; ++r) {
        const int e = blockIdx.x * QS_EPB + r * 256 + t;
        const bool ok = e < n;
        unsigned key = 0;
        if (ok) {
            float4 p;
            if (CENTRE) {
                const float4 q = src[e];
                p = make_float4(fmaf(-mx, q.w, q.x), fmaf(-my, q.w, q.y), fmaf(-mz, q.w, q.z), q.w);
                pts[e] = p;
            } else p = pts[e];
            key = st_key(p, g, tx, ty);
            keys_out[e] = key;
        }
        const unsigned d = key & (QS_BINS - 1);
        const unsigned long long m = match_digit(d, ok);
        if (ok && (m & ((1ull << lane) - 1ull)) == 0ull) atomicAdd(&h[d], (unsigned)__popcll(m));
    }
    __syncthreads();
    if (t < QS_BINS) count[t * nwg + blockIdx.x] = h[t];
}

// digit counts of a later pass, per workgroup, from the keys as the previous pass left them (every entry of the table is written:
// nothing to clear)
__global__ __launch_bounds__(256) void qcount_kernel(const unsigned* __restrict__ keys_in, BatchArgs ba, int shift, int nwg,
                                                     unsigned* __restrict__ count, int qs, int tabstride)
{
    const int n = ba.n[blockIdx.y];
    keys_in += (size_t)blockIdx.y * 4 * qs;
    count += (size_t)blockIdx.y * tabstride;
    __shared__ unsigned h[QS_BINS];
    const int t = threadIdx.x, lane = t & 63;
    if (t < QS_BINS) h[t] = 0;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < QS_EPB / 256; ++r) {
        const int e = blockIdx.x * QS_EPB + r * 256 + t;
        const bool ok = e < n;
        const unsigned d = ((ok ? keys_in[e] : 0u) >> shift) & (QS_BINS - 1);
        const unsigned long long m = match_digit(d, ok);
        if (ok && (m & ((1ull << lane) - 1ull)) == 0ull) atomicAdd(&h[d], (unsigned)__popcll(m));
    }
    __syncthreads();
    if (t < QS_BINS) count[t * nwg + blockIdx.x] = h[t];
}

// One pass: stable scatter by the 6-bit digit at `shift`.  A workgroup's base per digit comes from the per-workgroup counts (elements
// of smaller digits anywhere + the same digit in earlier workgroups -- the digit totals are summed from the same table, no atomics).
// (Building the NEXT pass's counts here, one global atomic per element on count_next[digit'][destination workgroup], was tried: the
// scatter went 7.4 -> 21 us, 76 us with the 10 M map's three passes -- qcount_kernel does it from LDS in 5 us.)
// FINAL: points and original indices land in the sorted arrays; with a head, workgroup 0 of every reading also initialises that
// reading's loop state and all workgroups clear its selection histograms (init_state_kernel and a memset node in r2).
template <bool FINAL>
__global__ __launch_bounds__(256) void qpass_kernel(const unsigned* __restrict__ keys_in, const unsigned* __restrict__ vals_in, BatchArgs ba,
                                                    int shift, int nwg, const unsigned* __restrict__ count, unsigned* __restrict__ count_next,
                                                    unsigned* __restrict__ keys_out, unsigned* __restrict__ vals_out,
                                                    const float4* __restrict__ pts, float4* __restrict__ out_pts, int* __restrict__ out_index,
                                                    int qs, int tabstride, IcpState* st_init, unsigned seq, unsigned* progress,
                                                    const unsigned* seq_src, unsigned* __restrict__ selhist)
{
    const int n = ba.n[blockIdx.y];
    keys_in += (size_t)blockIdx.y * 4 * qs;
    if (vals_in) vals_in += (size_t)blockIdx.y * 4 * qs;
    if (keys_out) keys_out += (size_t)blockIdx.y * 4 * qs;
    if (vals_out) vals_out += (size_t)blockIdx.y * 4 * qs;
    count += (size_t)blockIdx.y * tabstride;
    if (count_next) count_next += (size_t)blockIdx.y * tabstride;
    pts += (size_t)blockIdx.y * qs;
    if (out_pts) out_pts += (size_t)blockIdx.y * qs;
    if (out_index) out_index += (size_t)blockIdx.y * qs;
    constexpr int R = QS_EPB / 256;
    __shared__ unsigned wc[R][4][QS_BINS]; // [round][wave][digit]: count, then exclusive prefix in element order
    __shared__ unsigned part[4][QS_BINS], whole[4][QS_BINS];
    __shared__ unsigned base[QS_BINS];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (FINAL) {
        if (st_init && blockIdx.x == 0 && t == 0) init_state_dev(st_init + blockIdx.y, nullptr, seq, progress ? progress + blockIdx.y : nullptr, seq_src);
        if (selhist) {
            unsigned* hs = selhist + (size_t)blockIdx.y * ICPMI_SELHIST_WORDS;
            for (int i = blockIdx.x * 256 + t; i < ICPMI_SELHIST_WORDS; i += nwg * 256) hs[i] = 0;
        }
    }
    for (int i = t; i < R * 4 * QS_BINS; i += 256) (&wc[0][0][0])[i] = 0;
    {
        const int d = t & (QS_BINS - 1), q = t >> QS_BITS; // 4 lane-quarters share the sums over the workgroups
        unsigned before = 0, all = 0;
        const unsigned* __restrict__ cp = count + (size_t)d * nwg;
        for (int b = q; b < nwg; b += 64) { // (r5) sixteen counts per trip (100 k queries: 49 workgroups -- one trip instead of twelve dependent ones)
            unsigned v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) { const int bb = b + 4 * u; v[u] = bb < nwg ? cp[bb] : 0u; }
#pragma unroll
            for (int u = 0; u < 16; ++u) { all += v[u]; before += b + 4 * u < (int)blockIdx.x ? v[u] : 0u; }
        }
        part[q][d] = before; whole[q][d] = all;
    }
    unsigned key[R], val[R], rank[R], dig[R];
    bool ok[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int e = blockIdx.x * QS_EPB + r * 256 + t;
        ok[r] = e < n;
        key[r] = ok[r] ? keys_in[e] : 0u;
        val[r] = ok[r] ? (vals_in ? vals_in[e] : (unsigned)e) : 0u;
        dig[r] = (key[r] >> shift) & (QS_BINS - 1);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const unsigned long long m = match_digit(dig[r], ok[r]);
        rank[r] = (unsigned)__popcll(m & ((1ull << lane) - 1ull));
        if (ok[r] && rank[r] == 0) wc[r][w][dig[r]] = (unsigned)__popcll(m);
    }
    __syncthreads();
    if (t < QS_BINS) {
        unsigned run = 0;
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) { const unsigned cnt = wc[r][ww][t]; wc[r][ww][t] = run; run += cnt; }
        // exclusive scan of the digit totals over the 64 lanes of wave 0
        const unsigned tot = whole[0][t] + whole[1][t] + whole[2][t] + whole[3][t];
        unsigned incl = tot;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        base[t] = incl - tot + part[0][t] + part[1][t] + part[2][t] + part[3][t];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (!ok[r]) continue;
        const unsigned pos = base[dig[r]] + wc[r][w][dig[r]] + rank[r];
        if (FINAL) { out_pts[pos] = pts[val[r]]; out_index[pos] = (int)val[r]; }
        else {
            keys_out[pos] = key[r]; vals_out[pos] = val[r];
        }
    }
}

} // namespace

// capacity of everything sort_queries touches (so that the sort itself allocates nothing and can be
// captured into a graph); nscan slices of n elements each
static void sort_dims(const icpmi_ctx* c, int64_t n, int& tx, int& ty, int& passes, int& nwg, size_t& tab)
{
    const GridParams& g = c->grid;
    tx = (g.nx + STX - 1) / STX; ty = (g.ny + STY - 1) / STY;
    const int tz = (g.nz + STZ - 1) / STZ;
    const int nst = tx * ty * tz;
    int bits = 0;
    while ((1ll << bits) < (long long)nst) ++bits;
    passes = bits <= QS_BITS ? 1 : (bits + QS_BITS - 1) / QS_BITS;
    nwg = (int)((n + QS_EPB - 1) / QS_EPB);
    tab = (size_t)QS_BINS * (nwg > 0 ? nwg : 1) + QS_BINS; // count[digit][workgroup] + total[digit], per pass
}

icpmi_status sort_queries_reserve(icpmi_ctx* c, int64_t n, int nscan)
{
    int tx, ty, passes, nwg; size_t tab;
    sort_dims(c, n, tx, ty, passes, nwg, tab);
    const size_t tot = (size_t)n * nscan;
    if (ensure_cap(c, &c->d_qsorted, &c->cap_qsorted, tot + 1) != ICPMI_OK) return ICPMI_ERR_HIP;
    if (ensure_cap(c, &c->d_qindex, &c->cap_qindex, tot + 1) != ICPMI_OK) return ICPMI_ERR_HIP;
    if (ensure_cap(c, &c->d_qkeys, &c->cap_qkeys, (size_t)4 * tot + 4) != ICPMI_OK) return ICPMI_ERR_HIP; // keys / values, ping-pong
    if (ensure_cap(c, &c->d_qtile, &c->cap_qtile, tab * passes * nscan) != ICPMI_OK) return ICPMI_ERR_HIP;
    return ICPMI_OK;
}

// stable radix sort of every reading's points by super-tile: slice b of d_pts (ba.n[b] points, slices qs elements apart;
// a batch of one: qs = n) -> slice b of d_qsorted / d_qindex.  2 passes kernels (first pass: keys + counts in one), no memset node.  head != nullptr: see SortHead.
icpmi_status sort_queries_batch(icpmi_ctx* c, const float4* d_pts, const BatchArgs& ba, const SortHead* head)
{
    const GridParams& g = c->grid;
    int nmax = 0;
    for (int b = 0; b < ba.nscan; ++b) nmax = ba.n[b] > nmax ? ba.n[b] : nmax;
    const int qs = ba.nscan == 1 ? nmax : ba.qstride;
    int tx, ty, passes, nwg; size_t tab;
    sort_dims(c, nmax, tx, ty, passes, nwg, tab);
    if (nwg == 0) { if (head) { c->last_error = "sort_queries: a fused head needs points"; return ICPMI_ERR_INVALID_ARG; } return ICPMI_OK; }
    if (sort_queries_reserve(c, qs, ba.nscan) != ICPMI_OK) return ICPMI_ERR_HIP;
    const size_t tab_all = tab * passes; // one reading's tables
    unsigned* kbuf[2] = {c->d_qkeys, c->d_qkeys + qs};
    unsigned* vbuf[2] = {c->d_qkeys + 2 * (size_t)qs, c->d_qkeys + 3 * (size_t)qs};
    const dim3 grid(nwg, ba.nscan);
    const int later = 0; // (the tables of the later passes are written in full by qcount_kernel)
    if (head)
        hipLaunchKernelGGL(qfirst_kernel<true>, grid, dim3(256), 0, c->stream, head->raw, head->mean[0], head->mean[1], head->mean[2],
                           const_cast<float4*>(d_pts), ba, g, tx, ty, kbuf[0], nwg, c->d_qtile, later, (int)tab, qs, (int)tab_all, head->st);
    else {
        BatchSrc none; memset(&none, 0, sizeof none);
        hipLaunchKernelGGL(qfirst_kernel<false>, grid, dim3(256), 0, c->stream, none, 0.f, 0.f, 0.f, const_cast<float4*>(d_pts), ba, g, tx, ty,
                           kbuf[0], nwg, c->d_qtile, later, (int)tab, qs, (int)tab_all, (IcpState*)nullptr);
    }
    for (int ps = 0; ps < passes; ++ps) {
        unsigned* count = c->d_qtile + tab * ps;
        const int shift = ps * QS_BITS;
        const int in = ps & 1, out = in ^ 1;
        const unsigned* vin = ps == 0 ? nullptr : vbuf[in];
        if (ps > 0) hipLaunchKernelGGL(qcount_kernel, grid, dim3(256), 0, c->stream, (const unsigned*)kbuf[in], ba, shift, nwg, count, qs, (int)tab_all);
        if (ps == passes - 1)
            hipLaunchKernelGGL(qpass_kernel<true>, grid, dim3(256), 0, c->stream, (const unsigned*)kbuf[in], vin, ba, shift, nwg,
                               (const unsigned*)count, (unsigned*)nullptr, (unsigned*)nullptr, (unsigned*)nullptr, d_pts, c->d_qsorted,
                               c->d_qindex, qs, (int)tab_all, head ? head->st : (IcpState*)nullptr, head ? head->seq : 0u,
                               head ? head->progress : (unsigned*)nullptr, head ? head->seq_src : (const unsigned*)nullptr,
                               head ? head->selhist : (unsigned*)nullptr);
        else
            hipLaunchKernelGGL(qpass_kernel<false>, grid, dim3(256), 0, c->stream, (const unsigned*)kbuf[in], vin, ba, shift, nwg,
                               (const unsigned*)count, count + tab, kbuf[out], vbuf[out], d_pts, (float4*)nullptr, (int*)nullptr,
                               qs, (int)tab_all, (IcpState*)nullptr, 0u, (unsigned*)nullptr, (const unsigned*)nullptr, (unsigned*)nullptr);
    }
    HIP_TRY(c, hipGetLastError());
    c->qsorted_n = ba.nscan == 1 ? nmax : -1; c->qsorted_src = ba.nscan == 1 ? d_pts : nullptr;
    return ICPMI_OK;
}

icpmi_status sort_queries(icpmi_ctx* c, const float4* d_pts, int64_t n) { return sort_queries_batch(c, d_pts, batch_of_one(n)); }

static int run_atomics_cfg()
{
    return 1; // (one atomic per run of equal keys in a wave: r2; the per-point variant stays in the kernels for reference)
}

// in-place exclusive scan of data[0..n) (counts -> starts); data[n] = total
icpmi_status device_exclusive_scan(icpmi_ctx* c, unsigned* data, int n, unsigned total) { return device_exclusive_scan_io(c, data, data, n, total); }

static int scan2_enabled()
{
    return 1; // (the two-kernel scan, r5; tables beyond SCAN2_MAX_NB chunks take the three-kernel one)
}

// The count table of a grid build -> the cell starts, in the layout a CURSOR scatter wants (r5): counts[0..n) (+ the occupancy word at
// counts[n + 1]) -> starts[0] = 0, starts[i + 1] = start of cell i, starts[n + 1] = total.  A scatter then takes its slots with
// atomicAdd(&starts[key + 1], len): when every point is placed, starts[i + 1] has grown to the start of cell i + 1 -- the array IS the plain
// exclusive scan, with no second table of fill cursors to clear (r4: two memsets per grid, three to four launches).  The counts are left
// ZERO (c->fill_clean): the next build counts into them as they are.  starts needs n + 2 words.
icpmi_status device_exclusive_scan_cursor(icpmi_ctx* c, unsigned* counts, unsigned* starts, int n, unsigned total, bool zero_counts, unsigned* tail_out)
{
    const int nb = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
    if (ensure_cap(c, &c->d_blocksums, &c->cap_blocksums, (size_t)nb + 1) != ICPMI_OK) return ICPMI_ERR_HIP;
    if (scan2_enabled() && nb <= SCAN2_MAX_NB) {
        hipLaunchKernelGGL(scan2_sums_kernel, dim3(nb > 0 ? nb : 1), dim3(SCAN_T), 0, c->stream, (const unsigned*)counts, n, c->d_blocksums);
        hipLaunchKernelGGL(scan2_final_kernel, dim3(nb > 0 ? nb : 1), dim3(SCAN_T), 0, c->stream, (const unsigned*)counts, starts, n, (const unsigned*)c->d_blocksums, total, 1,
                           zero_counts ? counts : (unsigned*)nullptr, (unsigned*)nullptr, tail_out);
        HIP_TRY(c, hipGetLastError());
        return ICPMI_OK;
    }
    // very large tables: the three-kernel scan on a copy, the zero word in front, the counts cleared by a memset
    if (tail_out) HIP_TRY(c, hipMemcpyAsync(tail_out, counts + n + 1, sizeof(unsigned), hipMemcpyDefault, c->stream));
    HIP_TRY(c, hipMemcpyAsync(starts + 1, counts, (size_t)n * sizeof(unsigned), hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(c, hipMemsetAsync(starts, 0, sizeof(unsigned), c->stream));
    const icpmi_status s = device_exclusive_scan_io(c, starts + 1, starts + 1, n, total);
    if (s != ICPMI_OK) return s;
    if (zero_counts) HIP_TRY(c, hipMemsetAsync(counts, 0, ((size_t)n + 2) * sizeof(unsigned), c->stream));
    return ICPMI_OK;
}

bool device_scan_side_ok(int n) { return scan2_enabled() && (n + SCAN_CHUNK - 1) / SCAN_CHUNK <= SCAN2_MAX_NB; }
size_t device_scan_side_words(int n) { return (size_t)(n + SCAN_CHUNK - 1) / SCAN_CHUNK + 1; }
icpmi_status device_exclusive_scan_cursor_side(icpmi_ctx* c, hipStream_t stream, unsigned* sums, unsigned* counts, unsigned* starts, int n, unsigned total)
{
    if (!device_scan_side_ok(n)) return ICPMI_ERR_UNSUPPORTED;
    const int nb = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
    hipLaunchKernelGGL(scan2_sums_kernel, dim3(nb > 0 ? nb : 1), dim3(SCAN_T), 0, stream, (const unsigned*)counts, n, sums);
    hipLaunchKernelGGL(scan2_final_kernel, dim3(nb > 0 ? nb : 1), dim3(SCAN_T), 0, stream, (const unsigned*)counts, starts, n, (const unsigned*)sums, total, 1,
                       (unsigned*)nullptr, (unsigned*)nullptr, (unsigned*)nullptr);
    HIP_TRY(c, hipGetLastError());
    return ICPMI_OK;
}

static icpmi_status device_scan_counts_to_cursors(icpmi_ctx* c, unsigned* counts, unsigned* starts, int n, unsigned total, unsigned* tail_out = nullptr)
{
    const icpmi_status s = device_exclusive_scan_cursor(c, counts, starts, n, total, true, tail_out);
    if (s == ICPMI_OK) c->fill_clean = true;
    return s;
}

// the handle's count table (c->d_fill), `words` zero words: cleared only when a previous user left it dirty or it was reallocated
static icpmi_status counts_begin(icpmi_ctx* c, size_t words)
{
    const size_t cap_before = c->cap_fill;
    if (ensure_cap(c, &c->d_fill, &c->cap_fill, words) != ICPMI_OK) return ICPMI_ERR_HIP;
    if (c->cap_fill != cap_before) c->fill_clean = false; // a fresh allocation (the allocator may hand back the same address)
    if (!c->fill_clean) HIP_TRY(c, hipMemsetAsync(c->d_fill, 0, c->cap_fill * sizeof(unsigned), c->stream));
    c->fill_clean = false; // (about to be counted into; device_scan_counts_to_cursors hands it back clean)
    return ICPMI_OK;
}

// out[0..n) = exclusive scan of flag[0..n) (out[n] = 0) and the NUMBER of set flags on the host: the last workgroup of the scan writes the sum
// into host-mapped memory, the caller's wait for the stream is the read-back (r4: two copy launches into the pinned page and the same wait)
icpmi_status device_scan_flags_count(icpmi_ctx* c, const unsigned* flag, unsigned* pos, int n, int64_t* count)
{
    *count = 0;
    if (n <= 0) return ICPMI_OK;
    const int nb = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
    if (c->d_progress && c->h_progress && scan2_enabled() && nb <= SCAN2_MAX_NB) {
        if (ensure_cap(c, &c->d_blocksums, &c->cap_blocksums, (size_t)nb + 1) != ICPMI_OK) return ICPMI_ERR_HIP;
        unsigned* d_word = c->d_progress + ICPMI_PROGRESS_SCAN_WORD;
        volatile unsigned* h_word = c->h_progress + ICPMI_PROGRESS_SCAN_WORD;
        // r5: the count arrives TAGGED (words 40 / 41 as one 64-bit slot: call number << 32 | count, system-scope release) and the host spins on
        // the tag instead of draining the stream: a drained stream costs the completion signal and the restart of an empty queue (~20 us,
        // DESIGN 13.6e) -- five to seven times per map update --, the word is here ~2 us after the kernel's last workgroup wrote it, and what the
        // caller enqueues next queues up behind a GPU that never went idle.
        const unsigned tag = ++c->scan_tag ? c->scan_tag : ++c->scan_tag;
        hipLaunchKernelGGL(scan2_sums_kernel, dim3(nb), dim3(SCAN_T), 0, c->stream, flag, n, c->d_blocksums);
        hipLaunchKernelGGL(scan2_final_kernel, dim3(nb), dim3(SCAN_T), 0, c->stream, flag, pos, n, (const unsigned*)c->d_blocksums, 0u, 0, (unsigned*)nullptr, d_word,
                           (unsigned*)nullptr, tag);
        HIP_TRY(c, hipGetLastError());
        if (tag) {
            const volatile unsigned long long* h64 = reinterpret_cast<const volatile unsigned long long*>(c->h_progress + ICPMI_PROGRESS_SCAN_WORD);
            for (unsigned spins = 1;; ++spins) {
                const unsigned long long v = __atomic_load_n(h64, __ATOMIC_ACQUIRE);
                if ((unsigned)(v >> 32) == tag) { *count = (int64_t)(unsigned)(v & 0xffffffffull); return ICPMI_OK; }
                if ((spins & 1023u) != 0) continue;
                const hipError_t qe = hipStreamQuery(c->stream);
                if (qe == hipErrorNotReady) continue;
                HIP_TRY(c, qe); // (a stream in an error state: nothing will ever write the word)
                const unsigned long long w = __atomic_load_n(h64, __ATOMIC_ACQUIRE);
                if ((unsigned)(w >> 32) == tag) { *count = (int64_t)(unsigned)(w & 0xffffffffull); return ICPMI_OK; }
                c->last_error = "device_scan_flags_count: the count never arrived"; return ICPMI_ERR_HIP;
            }
        }
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        *count = (int64_t)*h_word;
        return ICPMI_OK;
    }
    const icpmi_status s = device_exclusive_scan_io(c, flag, pos, n, 0u);
    if (s != ICPMI_OK) return s;
    unsigned lp = 0, lf = 0;
    if (read_back2(c, &lp, pos + (n - 1), sizeof(unsigned), &lf, flag + (n - 1), sizeof(unsigned)) != ICPMI_OK) return ICPMI_ERR_HIP;
    *count = (int64_t)lp + lf;
    return ICPMI_OK;
}

__global__ void scan_tail_sum_kernel(const unsigned* __restrict__ pos, const unsigned* __restrict__ flag, int n, unsigned* __restrict__ out)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) *out = n > 0 ? pos[n - 1] + flag[n - 1] : 0u;
}

// pos = exclusive scan of flag[0..n) (pos[n] = 0) and the number of set flags in the DEVICE word *d_sum -- nothing waits, nothing is read back
icpmi_status device_exclusive_scan_sum(icpmi_ctx* c, const unsigned* flag, unsigned* pos, int n, unsigned* d_sum)
{
    const int nb = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
    if (n > 0 && scan2_enabled() && nb <= SCAN2_MAX_NB) {
        if (ensure_cap(c, &c->d_blocksums, &c->cap_blocksums, (size_t)nb + 1) != ICPMI_OK) return ICPMI_ERR_HIP;
        hipLaunchKernelGGL(scan2_sums_kernel, dim3(nb), dim3(SCAN_T), 0, c->stream, flag, n, c->d_blocksums);
        hipLaunchKernelGGL(scan2_final_kernel, dim3(nb), dim3(SCAN_T), 0, c->stream, flag, pos, n, (const unsigned*)c->d_blocksums, 0u, 0, (unsigned*)nullptr, d_sum);
        HIP_TRY(c, hipGetLastError());
        return ICPMI_OK;
    }
    if (n > 0) { const icpmi_status s = device_exclusive_scan_io(c, flag, pos, n, 0u); if (s != ICPMI_OK) return s; }
    hipLaunchKernelGGL(scan_tail_sum_kernel, dim3(1), dim3(64), 0, c->stream, (const unsigned*)pos, flag, n, d_sum);
    HIP_TRY(c, hipGetLastError());
    return ICPMI_OK;
}

// out[0..n) = exclusive scan of in[0..n), out[n] = total; in == out allowed
icpmi_status device_exclusive_scan_io(icpmi_ctx* c, const unsigned* in, unsigned* data, int n, unsigned total)
{
    const int nb = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
    if (ensure_cap(c, &c->d_blocksums, &c->cap_blocksums, (size_t)nb + 1) != ICPMI_OK) return ICPMI_ERR_HIP;
    if (scan2_enabled() && nb <= SCAN2_MAX_NB) {
        hipLaunchKernelGGL(scan2_sums_kernel, dim3(nb > 0 ? nb : 1), dim3(SCAN_T), 0, c->stream, in, n, c->d_blocksums);
        hipLaunchKernelGGL(scan2_final_kernel, dim3(nb > 0 ? nb : 1), dim3(SCAN_T), 0, c->stream, in, data, n, (const unsigned*)c->d_blocksums, total, 0, (unsigned*)nullptr);
        HIP_TRY(c, hipGetLastError());
        return ICPMI_OK;
    }
    if (in != data) HIP_TRY(c, hipMemcpyAsync(data, in, (size_t)n * sizeof(unsigned), hipMemcpyDeviceToDevice, c->stream));
    hipLaunchKernelGGL(scan_sums_kernel, dim3(nb), dim3(SCAN_T), 0, c->stream, data, n, c->d_blocksums);
    hipLaunchKernelGGL(scan_top_kernel, dim3(1), dim3(SCAN_T), 0, c->stream, c->d_blocksums, nb);
    hipLaunchKernelGGL(scan_final_kernel, dim3(nb), dim3(SCAN_T), 0, c->stream, data, n, c->d_blocksums, total);
    HIP_TRY(c, hipGetLastError());
    return ICPMI_OK;
}

// h_nocc == nullptr: the occupied-cell count is not waited for -- it is copied to the handle's pinned word behind the kernel and read
// at the NEXT build (whose bounding-box read-back has synchronised the stream by then)
static icpmi_status grid_count(icpmi_ctx* c, const float4* d_pts, int64_t m, const GridParams& g, unsigned* h_nocc)
{
    if (ensure_cap(c, &c->d_cell_start, &c->cap_cells, (size_t)g.ncells + 2) != ICPMI_OK) return ICPMI_ERR_HIP;
    // r5: the counts go to the handle's self-cleaning count table (c->d_fill), the starts are written in full by the scan: no memset of either
    if (counts_begin(c, (size_t)g.ncells + 2) != ICPMI_OK) return ICPMI_ERR_HIP;
    unsigned* d_nocc = c->d_fill + g.ncells + 1; // spare word behind the counts (cleared with them)
    const int blocks = (int)((m + 255) / 256);
    hipLaunchKernelGGL(key_kernel, dim3(blocks), dim3(256), 0, c->stream, d_pts, m, c->mean[0], c->mean[1], c->mean[2], g,
                       c->d_keys, c->d_fill, d_nocc, run_atomics_cfg());
    HIP_TRY(c, hipGetLastError());
    if (!h_nocc) { // not waited for: the scan that turns these counts into starts writes the word to the pinned page (no copy launch)
        if (c->d_nocc_host) c->nocc_by_scan = true;
        else HIP_TRY(c, hipMemcpyAsync(c->h_nocc, d_nocc, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
        return ICPMI_OK;
    }
    if (read_back(c, h_nocc, d_nocc, sizeof(unsigned)) != ICPMI_OK) return ICPMI_ERR_HIP;
    return ICPMI_OK;
}


static GridParams make_grid(const float lo[3], const float hi[3], float cell, float maxabs)
{
    GridParams g;
    g.ox = lo[0]; g.oy = lo[1]; g.oz = lo[2];
    g.cell = cell; g.inv_cell = 1.0f / cell;
    g.slack = cell * 1e-3f + maxabs * 2e-6f;
    g.nx = (int)floorf((hi[0] - lo[0]) * g.inv_cell) + 1;
    g.ny = (int)floorf((hi[1] - lo[1]) * g.inv_cell) + 1;
    g.nz = (int)floorf((hi[2] - lo[2]) * g.inv_cell) + 1;
    g.ncells = g.nx * g.ny * g.nz;
    return g;
}

icpmi_status upload_level_table(icpmi_ctx* c)
{
    // level table for the NN kernels: per level [ox oy oz cell][inv_cell slack nx ny][nz ncells pts][cs pos0]
    GridLevels& L = c->levels;
    uint32_t tab[ICPMI_MAXLEV * 16] = {0};
    auto fbits = [](float f) { uint32_t u; memcpy(&u, &f, 4); return u; };
    for (int l = 0; l < L.nlev; ++l) {
        uint32_t* t = tab + 16 * l;
        const GridParams& gl = L.g[l];
        t[0] = fbits(gl.ox); t[1] = fbits(gl.oy); t[2] = fbits(gl.oz); t[3] = fbits(gl.cell);
        t[4] = fbits(gl.inv_cell); t[5] = fbits(gl.slack); t[6] = (uint32_t)gl.nx; t[7] = (uint32_t)gl.ny;
        t[8] = (uint32_t)gl.nz; t[9] = (uint32_t)gl.ncells;
        const uint64_t pp = (uint64_t)(uintptr_t)L.pts[l], pc = (uint64_t)(uintptr_t)L.cs[l], p0 = (uint64_t)(uintptr_t)L.pos0[l];
        t[10] = (uint32_t)pp; t[11] = (uint32_t)(pp >> 32);
        t[12] = (uint32_t)pc; t[13] = (uint32_t)(pc >> 32); t[14] = (uint32_t)p0; t[15] = (uint32_t)(p0 >> 32);
    }
    if (!c->d_lvl_tab) HIP_TRY(c, dev_malloc((void**)&c->d_lvl_tab, sizeof tab));
    return upload_small(c, c->d_lvl_tab, tab, sizeof tab);
}

// The index of [previous cloud ; delta] from the index of the previous cloud (common.h: "incremental index insert").  *done = false:
// not applicable (the delta leaves the bounding box, the grid would change, the arrays of the previous build lack the keys ...) -- the
// caller builds from scratch.
static icpmi_status map_insert(icpmi_ctx* c, const float4* d_pts, int64_t m0, int64_t m1, const float* d_normals3, bool normals_changed, bool* done)
{
    *done = false;
    static int on = -1, min_m = -1;
    if (on < 0) { on = 1; const char* e = getenv("ICPMI_INSERT_MIN"); min_m = e ? atoi(e) : 0; }
    // (measured, one map-growth epoch = two index updates, A/B in one call: 1 M points 0.72 -> 0.56 ms, 10 M points 3.65 -> 0.97 ms; the first
    //  version -- raw points gathered by original index -- lost at 1 M: 0.74 ms.  ICPMI_INSERT_MIN: smallest cloud the insert serves.)
    const int64_t n = m1 - m0;
    GridLevels& L = c->levels;
    if (!on || !c->ins_ready || m0 != c->m || m0 <= 0 || n <= 0 || n > m0 / 2 || m1 < min_m || (d_normals3 != nullptr) != c->has_normals || c->cfg.grid_cell > 0.f) return ICPMI_OK;
    if (c->keep_raw && (d_pts != c->d_raw || (d_normals3 && d_normals3 != c->d_raw_n3))) return ICPMI_OK; // (an owner's index is built from its resident copy)
    // ---- the delta's sum and bounding box (one read-back, like the full build's)
    const int rblocks = (int)std::min<int64_t>((n + RB - 1) / RB, 256);
    if (ensure_cap(c, &c->d_red, &c->cap_red, (size_t)rblocks * 9) != ICPMI_OK) return ICPMI_ERR_HIP;
    hipLaunchKernelGGL(stats_kernel, dim3(rblocks), dim3(RB), 0, c->stream, d_pts + m0, n, c->d_red);
    HIP_TRY(c, hipGetLastError());
    std::vector<double> part((size_t)rblocks * 9);
    if (read_back(c, part.data(), c->d_red, part.size() * sizeof(double)) != ICPMI_OK) return ICPMI_ERR_HIP;
    double sum[3] = {c->sum_raw[0], c->sum_raw[1], c->sum_raw[2]};
    for (int b = 0; b < rblocks; ++b)
        for (int r = 0; r < 3; ++r) {
            sum[r] += part[(size_t)b * 9 + r];
            const float lo = (float)part[(size_t)b * 9 + 3 + r], hi = (float)part[(size_t)b * 9 + 6 + r];
            // New points may overhang the box the grid was laid over by a couple of cells: cell_of clamps them into the border cells,
            // where every search finds them (a point's clamped cell holds its projection onto the box, so the cell is never farther
            // from a query than the point is: row / cell pruning stays conservative, and a border block is a superset of the virtual
            // one the exactness margin speaks of) -- the noise tail of a scan on a wall that already bounds the map.  Farther out
            // (the robot has moved on): a fresh grid.
            const float over = 2.0f * c->levels.g[0].cell;
            if (!(lo >= c->lo_raw[r] - over) || !(hi <= c->hi_raw[r] + over)) return ICPMI_OK;
        }
    float mean[3], clo[3], chi[3], maxabs = 0.f;
    for (int r = 0; r < 3; ++r) {
        mean[r] = c->no_centre ? 0.f : (float)(sum[r] / (double)m1);
        clo[r] = c->lo_raw[r] - mean[r]; chi[r] = c->hi_raw[r] - mean[r];
        maxabs = fmaxf(maxabs, fmaxf(fabsf(clo[r]), fabsf(chi[r])));
    }
    GridParams gnew[ICPMI_MAXLEV];
    for (int l = 0; l < L.nlev; ++l) {
        gnew[l] = make_grid(clo, chi, L.g[l].cell, maxabs);
        if (gnew[l].nx != L.g[l].nx || gnew[l].ny != L.g[l].ny || gnew[l].nz != L.g[l].nz) return ICPMI_OK; // (the box's extent rounds differently under the new centroid)
    }
    // ---- buffers (all allocations before anything is written)
    const bool with_n = d_normals3 != nullptr;
    const bool with_pn = with_n && c->d_map_pn != nullptr && !c->single_level && c->keep_raw;
    if (ensure_cap(c, &c->d_ins_key, &c->cap_ins_key, (size_t)n + 1) != ICPMI_OK || ensure_cap(c, &c->d_ins_rank, &c->cap_ins_rank, (size_t)n + 1) != ICPMI_OK) return ICPMI_ERR_HIP;
    const bool twin = !c->no_centre; // (centroid 0: the sorted points ARE the raw points)
    if (twin && !c->d_raw0) return ICPMI_OK;
    if (ensure_cap(c, &c->d_inv, &c->cap_inv, (size_t)n + 1) != ICPMI_OK) return ICPMI_ERR_HIP; // level-0 position of every delta point
    if (ensure_cap(c, &c->d_fill, &c->cap_fill, (size_t)L.g[0].ncells + 2) != ICPMI_OK) return ICPMI_ERR_HIP;
    c->fill_clean = false; // (used below as the delta's cell starts of the coarser levels: a full build clears it first)
    if (ensure_cap(c, &c->d_ins_dstart0, &c->cap_ins_dstart0, (size_t)L.g[0].ncells + 2) != ICPMI_OK) return ICPMI_ERR_HIP;
    if (twin && ensure_cap(c, &c->d_alt_raw0, &c->cap_alt_raw0, (size_t)m1 + 16) != ICPMI_OK) return ICPMI_ERR_HIP;
    for (int l = 0; l < L.nlev; ++l) {
        if (ensure_cap(c, &c->d_alt_pts[l], &c->cap_alt_pts[l], (size_t)m1 + 16) != ICPMI_OK) return ICPMI_ERR_HIP;
        if (ensure_cap(c, &c->d_alt_key[l], &c->cap_alt_key[l], (size_t)m1 + 1) != ICPMI_OK) return ICPMI_ERR_HIP;
        if (ensure_cap(c, &c->d_alt_cs[l], &c->cap_alt_cs[l], (size_t)L.g[l].ncells + 2) != ICPMI_OK) return ICPMI_ERR_HIP;
        if (l > 0 && ensure_cap(c, &c->d_alt_pos0[l], &c->cap_alt_pos0[l], (size_t)m1 + 1) != ICPMI_OK) return ICPMI_ERR_HIP;
    }
    if (with_n && ensure_cap(c, &c->d_alt_nsorted, &c->cap_alt_nsorted, (size_t)m1 + 1) != ICPMI_OK) return ICPMI_ERR_HIP;
    if (with_pn && ensure_cap(c, &c->d_alt_pn, &c->cap_alt_pn, 2 * (size_t)m1 + 2) != ICPMI_OK) return ICPMI_ERR_HIP;
    // ---- per level: delta keys + counts -> prefix -> move the old points, place the delta, new cell starts
    const int gb0 = (int)((m0 + 255) / 256), gbn = (int)((n + 255) / 256);
    for (int l = 0; l < L.nlev; ++l) {
        const GridParams& g = gnew[l];
        float4* pts_old = l == 0 ? c->d_map_sorted : c->d_lvl_pts[l];
        unsigned* cs_old = l == 0 ? c->d_cell_start : c->d_lvl_cs[l];
        unsigned* key_old = c->d_lvl_key[l];
        unsigned* dstart = l == 0 ? c->d_ins_dstart0 : c->d_fill;
        HIP_TRY(c, hipMemsetAsync(dstart, 0, ((size_t)g.ncells + 2) * sizeof(unsigned), c->stream));
        hipLaunchKernelGGL(ins_key_kernel, dim3(gbn), dim3(256), 0, c->stream, d_pts, m0, n, mean[0], mean[1], mean[2], g, c->d_ins_key, c->d_ins_rank, dstart);
        if (device_exclusive_scan(c, dstart, g.ncells, (unsigned)n) != ICPMI_OK) return ICPMI_ERR_HIP;
        if (l == 0)
            hipLaunchKernelGGL(ins_move0_kernel, dim3(gb0), dim3(256), 0, c->stream, (const float4*)pts_old, twin ? (const float4*)c->d_raw0 : (const float4*)nullptr,
                               (const unsigned*)key_old, m0, (const unsigned*)dstart, mean[0], mean[1], mean[2], c->d_alt_pts[0],
                               twin ? c->d_alt_raw0 : (float4*)nullptr, c->d_alt_key[0], normals_changed ? d_normals3 : (const float*)nullptr,
                               (const float4*)c->d_normals_sorted, with_n ? c->d_alt_nsorted : (float4*)nullptr, with_pn ? c->d_alt_pn : (float4*)nullptr);
        else
            hipLaunchKernelGGL(ins_movel_kernel, dim3(gb0), dim3(256), 0, c->stream, (const unsigned*)key_old, (const unsigned*)c->d_lvl_pos0[l], m0,
                               (const unsigned*)dstart, (const unsigned*)c->d_lvl_key[0], (const unsigned*)c->d_ins_dstart0, (const float4*)c->d_alt_pts[0],
                               c->d_alt_pts[l], c->d_alt_key[l], c->d_alt_pos0[l]);
        hipLaunchKernelGGL(ins_delta_kernel, dim3(gbn), dim3(256), 0, c->stream, d_pts, m0, n, mean[0], mean[1], mean[2], (const unsigned*)c->d_ins_key,
                           (const unsigned*)c->d_ins_rank, (const unsigned*)cs_old, (const unsigned*)dstart, c->d_alt_pts[l], c->d_alt_key[l], c->d_inv,
                           l == 0 ? (unsigned*)nullptr : c->d_alt_pos0[l], d_normals3, (l == 0 && with_n) ? c->d_alt_nsorted : (float4*)nullptr,
                           (l == 0 && with_pn) ? c->d_alt_pn : (float4*)nullptr, (l == 0 && twin) ? c->d_alt_raw0 : (float4*)nullptr);
        // (level 0: dstart[ncells + 1] -- cleared with the table, not touched by the scan -- counts the newly occupied cells)
        hipLaunchKernelGGL(ins_cs_kernel, dim3((g.ncells + 1 + 255) / 256), dim3(256), 0, c->stream, (const unsigned*)cs_old, (const unsigned*)dstart, g.ncells + 1, c->d_alt_cs[l],
                           l == 0 ? dstart + g.ncells + 1 : (unsigned*)nullptr);
        HIP_TRY(c, hipGetLastError());
        if (l == 0 && c->h_nocc) HIP_TRY(c, hipMemcpyAsync(c->h_nocc + 1, dstart + g.ncells + 1, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    }
    // ---- level table and completion against the ALT set, BEFORE anything of the live index is replaced: a failure up to here leaves the
    //      handle on the index of m0 points, consistent
    GridLevels Lnew = L;
    for (int l = 0; l < L.nlev; ++l) {
        Lnew.g[l] = gnew[l];
        Lnew.pts[l] = c->d_alt_pts[l];
        Lnew.cs[l] = c->d_alt_cs[l];
        Lnew.pos0[l] = l == 0 ? nullptr : c->d_alt_pos0[l];
    }
    {
        const GridLevels Lold = L;
        L = Lnew; // (upload_level_table reads c->levels)
        const icpmi_status us = upload_level_table(c);
        const hipError_t se = us == ICPMI_OK ? hipStreamSynchronize(c->stream) : hipSuccess;
        if (us != ICPMI_OK || se != hipSuccess) {
            // the device copy of the level table may describe either set now: no index until a full build has run
            L = Lold; c->ins_ready = false; c->m = 0; c->m_raw = 0; drop_loop_graphs(c);
            if (us != ICPMI_OK) return us;
            HIP_TRY(c, se);
        }
    }
    // ---- the written set becomes the index
    std::swap(c->d_map_sorted, c->d_alt_pts[0]); std::swap(c->cap_map, c->cap_alt_pts[0]);
    std::swap(c->d_cell_start, c->d_alt_cs[0]); std::swap(c->cap_cells, c->cap_alt_cs[0]);
    for (int l = 0; l < L.nlev; ++l) { std::swap(c->d_lvl_key[l], c->d_alt_key[l]); std::swap(c->cap_lvl_key[l], c->cap_alt_key[l]); }
    for (int l = 1; l < L.nlev; ++l) {
        std::swap(c->d_lvl_pts[l], c->d_alt_pts[l]); std::swap(c->cap_lvl_pts[l], c->cap_alt_pts[l]);
        std::swap(c->d_lvl_cs[l], c->d_alt_cs[l]); std::swap(c->cap_lvl_cs[l], c->cap_alt_cs[l]);
        std::swap(c->d_lvl_pos0[l], c->d_alt_pos0[l]); std::swap(c->cap_lvl_pos0[l], c->cap_alt_pos0[l]);
    }
    if (twin) { std::swap(c->d_raw0, c->d_alt_raw0); std::swap(c->cap_raw0, c->cap_alt_raw0); }
    if (with_n) { std::swap(c->d_normals_sorted, c->d_alt_nsorted); std::swap(c->cap_normals, c->cap_alt_nsorted); }
    if (with_pn) { std::swap(c->d_map_pn, c->d_alt_pn); std::swap(c->cap_map_pn, c->cap_alt_pn); }
    for (int l = 0; l < L.nlev; ++l) {
        L.g[l] = gnew[l];
        L.pts[l] = l == 0 ? c->d_map_sorted : c->d_lvl_pts[l];
        L.cs[l] = l == 0 ? c->d_cell_start : c->d_lvl_cs[l];
        L.pos0[l] = l == 0 ? nullptr : c->d_lvl_pos0[l];
    }
    c->grid = gnew[0];
    for (int r = 0; r < 3; ++r) { c->mean[r] = mean[r]; c->sum_raw[r] = sum[r]; }
    if (c->h_nocc) { c->n_occupied += c->h_nocc[1]; *c->h_nocc = (unsigned)c->n_occupied; c->nocc_m = m1; } // (the stream is idle: the copy has landed)
    c->m = m1;
    c->m_raw = c->keep_raw ? m1 : 0; c->raw_has_normals = c->keep_raw && with_n;
    c->qsorted_n = -1; c->qsorted_src = nullptr;
    drop_loop_graphs(c);
    ++c->ins_count;
    *done = true;
    return ICPMI_OK;
}

// r6: the normals of a FEW resident points changed (ops.hip: surface_normals_dev after an append) -- their sorted copies are found through the
// point's level-0 cell (the key of key_kernel from the raw coordinates and the index's mean) and rewritten, instead of gathering the whole field again
__global__ __launch_bounds__(256) void patch_normals_kernel(const unsigned* __restrict__ list, int64_t n_list, const float4* __restrict__ raw,
                                                            float mx, float my, float mz, GridParams g, const unsigned* __restrict__ cs,
                                                            const float4* __restrict__ pts0, const float* __restrict__ normals3,
                                                            float4* __restrict__ nrm, float4* __restrict__ pn)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_list) return;
    const unsigned orig = list[t];
    const float4 p = raw[orig];
    const int cx = cell_of(p.x - mx, g.ox, g.inv_cell, g.nx), cy = cell_of(p.y - my, g.oy, g.inv_cell, g.ny), cz = cell_of(p.z - mz, g.oz, g.inv_cell, g.nz);
    const unsigned key = (unsigned)((cz * g.ny + cy) * g.nx + cx);
    const unsigned e = cs[key + 1];
    for (unsigned np = cs[key]; np < e; ++np)
        if (__float_as_uint(pts0[np].w) == orig) {
            const float4 nn = make_float4(normals3[3 * (size_t)orig], normals3[3 * (size_t)orig + 1], normals3[3 * (size_t)orig + 2], 0.f);
            nrm[np] = nn;
            if (pn) pn[2 * (size_t)np + 1] = nn;
            return;
        }
    // (not reached: the key is the build's own -- tests/test_gpu_incremental_normals.py registers against the patched index and against a fresh one)
}

icpmi_status map_patch_normals(icpmi_ctx* c, const unsigned* d_list, int64_t n_list, const float4* d_raw, const float* d_normals3)
{
    if (n_list <= 0 || !c->d_normals_sorted || c->m <= 0) return ICPMI_OK;
    const bool with_pn = c->d_map_pn != nullptr && !c->single_level && c->keep_raw;
    hipLaunchKernelGGL(patch_normals_kernel, dim3((int)((n_list + 255) / 256)), dim3(256), 0, c->stream, d_list, n_list, d_raw, c->mean[0], c->mean[1], c->mean[2],
                       c->levels.g[0], (const unsigned*)c->d_cell_start, (const float4*)c->d_map_sorted, d_normals3, c->d_normals_sorted,
                       with_pn ? c->d_map_pn : (float4*)nullptr);
    HIP_TRY(c, hipGetLastError());
    return ICPMI_OK;
}

icpmi_status map_build(icpmi_ctx* c, const float4* d_pts, int64_t m, const float* d_normals3, int64_t keep_prefix)
{
    ++c->map_version;
    if (keep_prefix > 0 && keep_prefix < m) {
        bool done = false;
        // (normals of the kept prefix: unchanged unless the caller says otherwise through c->ins_normals_changed -- a recomputed field)
        const icpmi_status is = map_insert(c, d_pts, keep_prefix, m, d_normals3, c->ins_normals_changed, &done);
        if (is != ICPMI_OK || done) return is;
    }
    ++c->full_count;
    // ---- stats ----
    const int rblocks = (int)std::min<int64_t>((m + RB - 1) / RB, 1024);
    if (ensure_cap(c, &c->d_red, &c->cap_red, (size_t)rblocks * 9) != ICPMI_OK) return ICPMI_ERR_HIP;
    hipLaunchKernelGGL(stats_kernel, dim3(rblocks), dim3(RB), 0, c->stream, d_pts, m, c->d_red);
    HIP_TRY(c, hipGetLastError());
    std::vector<double> part((size_t)rblocks * 9);
    if (read_back(c, part.data(), c->d_red, part.size() * sizeof(double)) != ICPMI_OK) return ICPMI_ERR_HIP;
    double sum[3] = {0, 0, 0};
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int b = 0; b < rblocks; ++b)
        for (int r = 0; r < 3; ++r) {
            sum[r] += part[(size_t)b * 9 + r];
            lo[r] = fminf(lo[r], (float)part[(size_t)b * 9 + 3 + r]);
            hi[r] = fmaxf(hi[r], (float)part[(size_t)b * 9 + 6 + r]);
        }
    for (int r = 0; r < 3; ++r) {
        if (!(lo[r] <= hi[r]) || !std::isfinite(lo[r]) || !std::isfinite(hi[r])) {
            c->last_error = "set_map: non-finite coordinates in the map cloud";
            return ICPMI_ERR_INVALID_ARG;
        }
        // ICPSequence::setMap centres on the centroid; the map-side operators (libnabo on the raw cloud in the
        // reference) index the coordinates as they are
        c->mean[r] = c->no_centre ? 0.f : (float)(sum[r] / (double)m);
        c->sum_raw[r] = sum[r]; c->lo_raw[r] = lo[r]; c->hi_raw[r] = hi[r];
    }
    // bbox of the centred cloud: x -> x - mean is monotone in float, so the extrema commute
    float clo[3], chi[3], maxabs = 0.f;
    for (int r = 0; r < 3; ++r) {
        clo[r] = lo[r] - c->mean[r]; chi[r] = hi[r] - c->mean[r];
        maxabs = fmaxf(maxabs, fmaxf(fabsf(clo[r]), fabsf(chi[r])));
    }

    if (ensure_cap(c, &c->d_keys, &c->cap_keys, (size_t)m) != ICPMI_OK) return ICPMI_ERR_HIP;

    // ---- choose the cell edge ----
    const double ext[3] = {(double)chi[0] - clo[0], (double)chi[1] - clo[1], (double)chi[2] - clo[2]};
    const double max_ext = std::max(ext[0], std::max(ext[1], ext[2]));
    const double MAX_CELLS = (double)(1 << 26);
    auto clamp_cell = [&](double cell) {
        // keep the dense table within MAX_CELLS and the cell strictly positive
        double lo_cell = std::max(max_ext * 1e-6, 1e-6);
        if (cell < lo_cell) cell = lo_cell;
        for (int it = 0; it < 64; ++it) {
            const double n = (floor(ext[0] / cell) + 1) * (floor(ext[1] / cell) + 1) * (floor(ext[2] / cell) + 1);
            if (n <= MAX_CELLS) break;
            cell *= 1.26;
        }
        return (float)cell;
    };
    GridParams g;
    unsigned n_occ = 0;
    if (c->cfg.grid_cell > 0.f) {
        g = make_grid(clo, chi, clamp_cell(c->cfg.grid_cell), maxabs);
        if (grid_count(c, d_pts, m, g, &n_occ) != ICPMI_OK) return ICPMI_ERR_HIP;
    } else {
        // trial edge from the bounding volume, then one correction assuming the points sample
        // surfaces (occupied cells ~ area / cell^2): aim at TARGET points per occupied cell.
        const double TARGET = 8.0;
        // A handle that rebuilds the index of a slowly growing map (every map update, twice) does not wait for the occupancy of THIS
        // build: it corrects the edge with the count of the previous one, which arrived in pinned memory long ago (r3: one stream
        // synchronisation less per build).  The edge only steers speed: the search is exact.
        if (c->h_nocc && c->nocc_m > 0 && c->grid.cell > 0.f && c->m > 0 && (double)m > 0.7 * (double)c->m && (double)m < 1.4 * (double)c->m) {
            const double occ = (double)c->nocc_m / (double)std::max(1u, *c->h_nocc);
            double cell = c->grid.cell;
            if (!(occ > TARGET * 0.6 && occ < TARGET * 1.6)) cell = cell * sqrt(TARGET / occ);
            g = make_grid(clo, chi, clamp_cell(cell), maxabs);
            n_occ = *c->h_nocc;
            if (grid_count(c, d_pts, m, g, nullptr) != ICPMI_OK) return ICPMI_ERR_HIP;
            c->nocc_m = m;
            goto grid_chosen;
        }
        double vol = std::max(ext[0], 1e-3) * std::max(ext[1], 1e-3) * std::max(ext[2], 1e-3);
        double cell = cbrt(vol / (double)m) * 1.2;
        // a handle that indexed a cloud of about this size before (the map of the previous update, the private handle of a
        // map-side operator) starts from that edge: the trial count is then usually the only one
        if (c->grid.cell > 0.f && c->m > 0 && (double)m > 0.7 * (double)c->m && (double)m < 1.4 * (double)c->m) cell = c->grid.cell;
        g = make_grid(clo, chi, clamp_cell(cell), maxabs);
        if (grid_count(c, d_pts, m, g, &n_occ) != ICPMI_OK) return ICPMI_ERR_HIP;
        for (int it = 0; it < 2; ++it) {
            const double occ = (double)m / (double)std::max(1u, n_occ);
            if (occ > TARGET * 0.6 && occ < TARGET * 1.6) break;
            const float c2 = clamp_cell(g.cell * sqrt(TARGET / occ));
            if (fabsf(c2 - g.cell) < 0.05f * g.cell) break;
            g = make_grid(clo, chi, c2, maxabs);
            if (grid_count(c, d_pts, m, g, &n_occ) != ICPMI_OK) return ICPMI_ERR_HIP;
        }
    }
    if (c->h_nocc) { *c->h_nocc = n_occ; c->nocc_m = m; }
grid_chosen:
    c->grid = g;
    c->n_occupied = n_occ;

    // ---- exclusive scan of the histogram: counts (c->d_fill, left zero) -> cell starts in cursor layout ----
    if (device_scan_counts_to_cursors(c, c->d_fill, c->d_cell_start, g.ncells, (unsigned)m, (c->nocc_by_scan && c->d_nocc_host) ? c->d_nocc_host : nullptr) != ICPMI_OK) return ICPMI_ERR_HIP;
    c->nocc_by_scan = false;

    // ---- scatter ----
    if (ensure_cap(c, &c->d_map_sorted, &c->cap_map, (size_t)m + 16) != ICPMI_OK) return ICPMI_ERR_HIP;
    if (d_normals3 && ensure_cap(c, &c->d_normals_sorted, &c->cap_normals, (size_t)m) != ICPMI_OK) return ICPMI_ERR_HIP;
    c->has_normals = d_normals3 != nullptr;
    // resident raw copy (skipped when the caller IS the raw copy: the device-side map update)
    if (d_pts != c->d_raw && c->keep_raw) {
        ++c->raw_epoch; // the resident copy is replaced, not appended to: a private raw-frame index starts over
        c->raw_has_scalar = false; // a map handed in from outside: its scalar channel comes through icpmi_set_map_scalar
        if (ensure_cap(c, &c->d_raw, &c->cap_raw, (size_t)m) != ICPMI_OK) return ICPMI_ERR_HIP;
        HIP_TRY(c, hipMemcpyAsync(c->d_raw, d_pts, (size_t)m * sizeof(float4), hipMemcpyDeviceToDevice, c->stream));
    }
    if (d_normals3 && d_normals3 != c->d_raw_n3 && c->keep_raw) {
        if (ensure_cap(c, &c->d_raw_n3, &c->cap_raw_n3, (size_t)m * 3) != ICPMI_OK) return ICPMI_ERR_HIP;
        HIP_TRY(c, hipMemcpyAsync(c->d_raw_n3, d_normals3, (size_t)m * 3 * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    }
    c->m_raw = c->keep_raw ? m : 0; c->raw_has_normals = c->keep_raw && d_normals3 != nullptr;
    const int blocks = (int)((m + 255) / 256);
    // handles whose cloud grows by appends (the owner of a resident map, its private raw-frame index) keep what map_insert needs: the
    // cell of every sorted position per level and original index -> level-0 position (+4 bytes per point and level written here)
    const bool want_ins = (c->keep_raw || c->is_raw_index) && !c->single_level && !(c->cfg.grid_cell > 0.f);
    const bool want_twin = want_ins && !c->no_centre;
    if (want_ins) {
        if (ensure_cap(c, &c->d_lvl_key[0], &c->cap_lvl_key[0], (size_t)m + 1) != ICPMI_OK) return ICPMI_ERR_HIP;
        if (want_twin && ensure_cap(c, &c->d_raw0, &c->cap_raw0, (size_t)m + 16) != ICPMI_OK) return ICPMI_ERR_HIP;
    }
    c->ins_ready = false;
    hipLaunchKernelGGL(scatter_kernel, dim3(blocks), dim3(256), 0, c->stream, d_pts, d_normals3, m, c->mean[0], c->mean[1], c->mean[2],
                       c->d_keys, c->d_cell_start + 1, c->d_map_sorted, d_normals3 ? c->d_normals_sorted : nullptr, run_atomics_cfg(),
                       want_ins ? c->d_lvl_key[0] : (unsigned*)nullptr, want_twin ? c->d_raw0 : (float4*)nullptr);
    HIP_TRY(c, hipGetLastError());
    if (d_normals3 && !c->single_level && c->keep_raw) { // (the handles of the map-side operators never run pair sums)
        if (ensure_cap(c, &c->d_map_pn, &c->cap_map_pn, 2 * (size_t)m + 2) != ICPMI_OK) return ICPMI_ERR_HIP;
        hipLaunchKernelGGL(pn_kernel, dim3(blocks), dim3(256), 0, c->stream, (const float4*)c->d_map_sorted, (const float4*)c->d_normals_sorted, m, c->d_map_pn);
        HIP_TRY(c, hipGetLastError());
    }

    // ---- coarser pyramid levels: cell edge doubles until one 3x3x3 block reaches past maxDist (or
    //      the grid is at most 2 cells wide, where a block always covers it) ----
    GridLevels& L = c->levels;
    L.nlev = 1;
    L.g[0] = g; L.pts[0] = c->d_map_sorted; L.cs[0] = c->d_cell_start; L.pos0[0] = nullptr;
    for (int l = 1; l < ICPMI_MAXLEV && !c->single_level; ++l) {
        const GridParams& prev = L.g[l - 1];
        const bool reaches = std::isfinite(c->cfg.max_dist) && (prev.cell - prev.slack) > c->cfg.max_dist;
        const bool tiny = prev.nx <= 2 && prev.ny <= 2 && prev.nz <= 2;
        if (reaches || tiny) break;
        GridParams gl = make_grid(clo, chi, prev.cell * 2.0f, maxabs);
        if (ensure_cap(c, &c->d_lvl_cs[l], &c->cap_lvl_cs[l], (size_t)gl.ncells + 2) != ICPMI_OK) return ICPMI_ERR_HIP;
        if (ensure_cap(c, &c->d_lvl_pts[l], &c->cap_lvl_pts[l], (size_t)m + 16) != ICPMI_OK) return ICPMI_ERR_HIP;
        if (ensure_cap(c, &c->d_lvl_pos0[l], &c->cap_lvl_pos0[l], (size_t)m) != ICPMI_OK) return ICPMI_ERR_HIP;
        if (counts_begin(c, (size_t)gl.ncells + 2) != ICPMI_OK) return ICPMI_ERR_HIP; // (the scan of the level below left the table zero)
        hipLaunchKernelGGL(lvl_key_kernel, dim3(blocks), dim3(256), 0, c->stream, c->d_map_sorted, m, gl, c->d_keys, c->d_fill);
        if (device_scan_counts_to_cursors(c, c->d_fill, c->d_lvl_cs[l], gl.ncells, (unsigned)m) != ICPMI_OK) return ICPMI_ERR_HIP;
        if (want_ins && ensure_cap(c, &c->d_lvl_key[l], &c->cap_lvl_key[l], (size_t)m + 1) != ICPMI_OK) return ICPMI_ERR_HIP;
        hipLaunchKernelGGL(lvl_scatter_kernel, dim3(blocks), dim3(256), 0, c->stream, c->d_map_sorted, m, c->d_keys, c->d_lvl_cs[l] + 1,
                           c->d_lvl_pts[l], c->d_lvl_pos0[l], want_ins ? c->d_lvl_key[l] : (unsigned*)nullptr);
        HIP_TRY(c, hipGetLastError());
        L.g[l] = gl; L.pts[l] = c->d_lvl_pts[l]; L.cs[l] = c->d_lvl_cs[l]; L.pos0[l] = c->d_lvl_pos0[l];
        L.nlev = l + 1;
    }
    { const icpmi_status us = upload_level_table(c); if (us != ICPMI_OK) return us; }
    c->ins_ready = want_ins;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    // (a deferred build steered its edge with the PREVIOUS build's occupancy; its own count has arrived in the pinned word by now:
    //  icpmi_get_grid_info reports the grid it describes -- ADVICE r3)
    if (c->h_nocc && c->nocc_m == m) c->n_occupied = *c->h_nocc;
    c->m = m;
    c->qsorted_n = -1; c->qsorted_src = nullptr; // tiles are defined on the grid of the map
    // any cached loop graph captured pointers / grid parameters of the previous map
    drop_loop_graphs(c);
    return ICPMI_OK;
}
