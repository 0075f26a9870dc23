// octree.hip -- OctreeGridDataPointsFilter on the device (behind OctreeMapperModule, reference
// norlab_icp_mapper/MapperModules/OctreeMapperModule.cpp:8-12,35-39: `map.concatenate(input); octreeFilter->inPlaceFilter(map)`;
// upstream libpointmatcher 1.4.x octree.hpp / DataPointsFilters/OctreeGrid.cpp).
//
// Upstream builds the tree recursively: the root is the bounding CUBE of the cloud, a node is a leaf when its edge is at most
// maxSizeByNode or it holds at most maxPointByNode points, otherwise its points go to 8 children (bit r of the child index set
// iff p_r > centre_r); the sampler then visits the leaves depth first and keeps one point of each, compacting the cloud in
// visiting order.  Here, without recursion:
//   1. root cube from a min / max reduction; D = the depth at which the edge falls to maxSizeByNode (at most 21);
//   2. every point walks down D levels on its own, with the float centre arithmetic of the recursion (centre +- radius / 2,
//      radius / 2), and packs its child indices into a 3 D-bit path, root split in the top bits: nodes are prefixes, the
//      depth-first visiting order is the order of the paths;
//   3. stable LSD radix sort of (path, original index) -- 6-bit digits, ceil(3 D / 6) passes: every node is now one
//      contiguous run, in list order (smallest original index first);
//   4. a node of depth d holds more than k = maxPointByNode points iff some window of k + 1 consecutive sorted points
//      containing the point shares a prefix of >= d levels, so with L = the longest common prefix over those windows the
//      leaf of a point is its ancestor of depth min(L + 1, D); a leaf starts where that prefix changes;
//   5. one representative per leaf (samplingMethod 0: its smallest original index = the first of upstream's list; 1: the
//      smallest fmix32(original index), the reproducible stand-in for upstream's random pick), written in leaf order.
// The CPU oracle restates the recursion itself (oracle/icp_oracle.c: orc_octree_sample); the two agree index for index.
#include "common.h"

namespace {

struct OctRoot { float cx, cy, cz, radius; int depth; int pad[3]; };

constexpr int OB = 256;

__global__ __launch_bounds__(OB) void oct_bbox_kernel(const float4* __restrict__ pts, int64_t n, float* __restrict__ part)
{
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = (int64_t)blockIdx.x * OB + threadIdx.x; i < n; i += (int64_t)gridDim.x * OB) {
        const float4 p = pts[i];
        lo[0] = fminf(lo[0], p.x); lo[1] = fminf(lo[1], p.y); lo[2] = fminf(lo[2], p.z);
        hi[0] = fmaxf(hi[0], p.x); hi[1] = fmaxf(hi[1], p.y); hi[2] = fmaxf(hi[2], p.z);
    }
    __shared__ float sl[3][OB], sh[3][OB];
    const int t = threadIdx.x;
    for (int r = 0; r < 3; ++r) { sl[r][t] = lo[r]; sh[r][t] = hi[r]; }
    __syncthreads();
    for (int s = OB / 2; s > 0; s >>= 1) {
        if (t < s)
            for (int r = 0; r < 3; ++r) { sl[r][t] = fminf(sl[r][t], sl[r][t + s]); sh[r][t] = fmaxf(sh[r][t], sh[r][t + s]); }
        __syncthreads();
    }
    if (t == 0) for (int r = 0; r < 3; ++r) { part[6 * blockIdx.x + r] = sl[r][0]; part[6 * blockIdx.x + 3 + r] = sh[r][0]; }
}

__global__ __launch_bounds__(OB) void oct_root_kernel(const float* __restrict__ part, int nb, float max_size, OctRoot* __restrict__ root,
                                                      OctRoot* __restrict__ root_host /* host-mapped copy, may be null */, int tag)
{
    __shared__ float sl[3][OB], sh[3][OB];
    const int t = threadIdx.x;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int b = t; b < nb; b += OB)
        for (int r = 0; r < 3; ++r) { lo[r] = fminf(lo[r], part[6 * b + r]); hi[r] = fmaxf(hi[r], part[6 * b + 3 + r]); }
    for (int r = 0; r < 3; ++r) { sl[r][t] = lo[r]; sh[r][t] = hi[r]; }
    __syncthreads();
    for (int s = OB / 2; s > 0; s >>= 1) {
        if (t < s)
            for (int r = 0; r < 3; ++r) { sl[r][t] = fminf(sl[r][t], sl[r][t + s]); sh[r][t] = fmaxf(sh[r][t], sh[r][t + s]); }
        __syncthreads();
    }
    if (t == 0) {
        float c[3], radius = 0.f;
        // upstream's Octree_::build: radii = max - min; centre = min + radii * 0.5; maxRadius = max(radii) * 0.5 (the oracle's rule, r3)
        for (int r = 0; r < 3; ++r) { const float ext = sh[r][0] - sl[r][0]; c[r] = sl[r][0] + ext * 0.5f; const float rr = ext * 0.5f; if (rr > radius) radius = rr; }
        int d = 0;
        float rad = radius;
        while (d < 21 && !(rad * 2.f <= max_size)) { rad *= 0.5f; ++d; } // the first depth whose edge is <= maxSizeByNode
        root->cx = c[0]; root->cy = c[1]; root->cz = c[2]; root->radius = radius; root->depth = d;
        if (root_host) {
            // the host reads the depth WITHOUT draining the stream (it spins on the count of a later scan): published with the call's tag behind a
            // system-scope release, checked on the host (r6, ADVICE r5: a plain store was visible in practice, not by contract)
            root_host->cx = c[0]; root_host->cy = c[1]; root_host->cz = c[2]; root_host->radius = radius; root_host->depth = d;
            __hip_atomic_store(&root_host->pad[0], tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

__global__ __launch_bounds__(256) void oct_path_kernel(const float4* __restrict__ pts, int64_t n, const OctRoot* __restrict__ root,
                                                       unsigned long long* __restrict__ keys, unsigned* __restrict__ vals)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 p = pts[i];
    float cx = root->cx, cy = root->cy, cz = root->cz, half = root->radius;
    const int D = root->depth;
    unsigned long long code = 0;
    for (int d = 0; d < D; ++d) {
        const unsigned o = (p.x > cx ? 1u : 0u) | (p.y > cy ? 2u : 0u) | (p.z > cz ? 4u : 0u);
        code = (code << 3) | o;
        half *= 0.5f;
        cx += (o & 1u) ? half : -half;
        cy += (o & 2u) ? half : -half;
        cz += (o & 4u) ? half : -half;
    }
    keys[i] = code;
    vals[i] = (unsigned)i;
}

// ---- stable LSD radix sort of (u64 key, u32 value), 6-bit digits: histogram per workgroup, then ranked scatter ----
#ifndef ICPMI_RS_EPB
#define ICPMI_RS_EPB 2048
#endif
constexpr int RS_BITS = 6, RS_BINS = 64, RS_EPB = ICPMI_RS_EPB; // elements per workgroup

__device__ __forceinline__ unsigned long long match6(unsigned d, bool valid)
{
    unsigned long long m = __ballot(valid);
#pragma unroll
    for (int b = 0; b < RS_BITS; ++b) {
        const bool bit = (d >> b) & 1u;
        const unsigned long long bal = __ballot(bit);
        m &= bit ? bal : ~bal;
    }
    return m;
}

__global__ __launch_bounds__(256) void rs_hist_kernel(const unsigned long long* __restrict__ keys, int64_t n, int shift, int nwg,
                                                      unsigned* __restrict__ count, unsigned* __restrict__ total)
{
    __shared__ unsigned h[RS_BINS];
    const int t = threadIdx.x, lane = t & 63;
    if (t < RS_BINS) h[t] = 0;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_EPB / 256; ++r) {
        const int64_t e = (int64_t)blockIdx.x * RS_EPB + r * 256 + t;
        const bool ok = e < n;
        const unsigned d = ok ? (unsigned)((keys[e] >> shift) & (RS_BINS - 1)) : 0u;
        const unsigned long long m = match6(d, ok);
        if (ok && (m & ((1ull << lane) - 1ull)) == 0ull) atomicAdd(&h[d], (unsigned)__popcll(m));
    }
    __syncthreads();
    if (t < RS_BINS) {
        count[t * nwg + blockIdx.x] = h[t];
        if (h[t]) atomicAdd(&total[t], h[t]);
    }
}

__global__ __launch_bounds__(256) void rs_scatter_kernel(const unsigned long long* __restrict__ keys_in, const unsigned* __restrict__ vals_in,
                                                         int64_t n, int shift, int nwg, const unsigned* __restrict__ count,
                                                         const unsigned* __restrict__ total, unsigned long long* __restrict__ keys_out,
                                                         unsigned* __restrict__ vals_out)
{
    constexpr int R = RS_EPB / 256;
    __shared__ unsigned wc[R][4][RS_BINS]; // [round][wave][digit]: count, then exclusive prefix in element order
    __shared__ unsigned part[4][RS_BINS];
    __shared__ unsigned base[RS_BINS];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    for (int i = t; i < R * 4 * RS_BINS; i += 256) (&wc[0][0][0])[i] = 0;
    {
        const int d = t & (RS_BINS - 1), q = t >> RS_BITS; // 4 quarters share the sum over the earlier workgroups
        // (r5) sixteen counts requested per trip: the loop over the earlier workgroups was one dependent load after the other -- 110 round
        // trips for the last workgroup of a 0.9 M-element pass, most of the kernel's 27 us
        unsigned s = 0;
        const unsigned* __restrict__ cp = count + (size_t)d * nwg;
        for (int b = q; b < (int)blockIdx.x; b += 64) {
            unsigned v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) { const int bb = b + 4 * u; v[u] = bb < (int)blockIdx.x ? cp[bb] : 0u; }
#pragma unroll
            for (int u = 0; u < 16; ++u) s += v[u];
        }
        part[q][d] = s;
    }
    unsigned long long key[R];
    unsigned val[R], rank[R], dig[R];
    bool ok[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t e = (int64_t)blockIdx.x * RS_EPB + r * 256 + t;
        ok[r] = e < n;
        key[r] = ok[r] ? keys_in[e] : 0ull;
        val[r] = ok[r] ? vals_in[e] : 0u;
        dig[r] = (unsigned)((key[r] >> shift) & (RS_BINS - 1));
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const unsigned long long m = match6(dig[r], ok[r]);
        rank[r] = (unsigned)__popcll(m & ((1ull << lane) - 1ull));
        if (ok[r] && rank[r] == 0) wc[r][w][dig[r]] = (unsigned)__popcll(m);
    }
    __syncthreads();
    if (t < RS_BINS) {
        unsigned run = 0;
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) { const unsigned cnt = wc[r][ww][t]; wc[r][ww][t] = run; run += cnt; }
        const unsigned tot = total[t];
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
        keys_out[pos] = key[r];
        vals_out[pos] = val[r];
    }
}

// common prefix of two paths, in levels
__device__ __forceinline__ int cpl_levels(unsigned long long a, unsigned long long b, int D)
{
    const unsigned long long x = a ^ b;
    if (x == 0ull) return D;
    const int p = 63 - __clzll((long long)x); // highest differing bit
    return D - 1 - p / 3;
}

// leaf of every sorted point: flag[j] = 1 where a leaf starts
__global__ __launch_bounds__(256) void oct_leaf_kernel(const unsigned long long* __restrict__ keys, int64_t n, const OctRoot* __restrict__ root,
                                                       int k, unsigned* __restrict__ flag)
{
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const int D = root->depth;
    const unsigned long long me = keys[j];
    int L = -1; // longest prefix shared by some k + 1 consecutive points that include j
    for (int64_t i = j - k; i <= j; ++i) {
        if (i < 0 || i + k >= n) continue;
        const int c = cpl_levels(keys[i], keys[i + k], D);
        L = c > L ? c : L;
    }
    const int dl = L + 1 < D ? L + 1 : D; // depth of this point's leaf
    bool start = j == 0;
    if (!start) start = cpl_levels(keys[j - 1], me, D) < dl;
    flag[j] = start ? 1u : 0u;
}

__device__ __forceinline__ unsigned fmix32_dev(unsigned h) { h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h; }

// ordinal[j] = exclusive scan of flag: leaf number of the leaf starting at j, or of the next one; leaf of j = ordinal[j] + flag[j] - 1
__global__ __launch_bounds__(256) void oct_pick_kernel(const unsigned* __restrict__ vals, const unsigned* __restrict__ flag,
                                                       const unsigned* __restrict__ ordinal, int64_t n, int method,
                                                       unsigned long long* __restrict__ best, int* __restrict__ order_out,
                                                       int* __restrict__ leaf_of)
{
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const unsigned leaf = ordinal[j] + flag[j] - 1u;
    if (leaf_of) leaf_of[vals[j]] = (int)leaf;
    // the representative of a leaf: the smallest original index (a leaf shallower than D holds points of different paths, so
    // the first of its sorted run is not necessarily the first of upstream's list), or the smallest hash of it
    const unsigned long long rank = method == 0 ? 0ull : (unsigned long long)fmix32_dev(vals[j]);
    atomicMin(&best[leaf], (rank << 32) | (unsigned long long)vals[j]);
}

__global__ __launch_bounds__(256) void oct_best_kernel(const unsigned long long* __restrict__ best, int64_t leaves, int* __restrict__ order_out)
{
    const int64_t l = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (l < leaves) order_out[l] = (int)(best[l] & 0xffffffffull);
}

} // namespace

// Leaves of the octree over the n points of d_in, one representative each: d_order[0 .. *n_out) = their original indices in
// leaf-visiting order (device array of at least n ints); d_leaf_of (may be null) = leaf ordinal of every input point.
// Uses operator scratch slots 0-4; stream-ordered on c->stream except for two small read-backs (depth, leaf count).
// Stable LSD radix sort of (u64 key, u32 value) pairs on the low `bits` bits of the key, 6 bits per pass.  d_keys2 / d_vals2
// hold 2 n elements (two ping-pong halves, the input in the first); d_tab holds radix_sort_tab_words(n, bits) words.
// *result_half tells which half holds the sorted pairs.  Everything is enqueued on c->stream, nothing is read back.
size_t radix_sort_tab_words(int64_t n, int bits)
{
    const int nwg = (int)((n + RS_EPB - 1) / RS_EPB);
    return ((size_t)RS_BINS * nwg + RS_BINS) * (size_t)((bits + RS_BITS - 1) / RS_BITS) + 16;
}

icpmi_status radix_sort_pairs(icpmi_ctx* c, unsigned long long* d_keys2, unsigned* d_vals2, int64_t n, int bits, unsigned* d_tab, int* result_half)
{
    const int nwg = (int)((n + RS_EPB - 1) / RS_EPB);
    const size_t tab = (size_t)RS_BINS * nwg + RS_BINS;
    const int passes = (bits + RS_BITS - 1) / RS_BITS;
    unsigned long long* kb[2] = {d_keys2, d_keys2 + n};
    unsigned* vb[2] = {d_vals2, d_vals2 + n};
    if (passes > 0) HIP_TRY(c, hipMemsetAsync(d_tab, 0, tab * passes * sizeof(unsigned), c->stream));
    int cur = 0;
    for (int ps = 0; ps < passes; ++ps) {
        unsigned* count = d_tab + tab * ps;
        unsigned* total = count + (size_t)RS_BINS * nwg;
        hipLaunchKernelGGL(rs_hist_kernel, dim3(nwg), dim3(256), 0, c->stream, (const unsigned long long*)kb[cur], n, ps * RS_BITS, nwg, count, total);
        hipLaunchKernelGGL(rs_scatter_kernel, dim3(nwg), dim3(256), 0, c->stream, (const unsigned long long*)kb[cur], (const unsigned*)vb[cur], n,
                           ps * RS_BITS, nwg, (const unsigned*)count, (const unsigned*)total, kb[cur ^ 1], vb[cur ^ 1]);
        cur ^= 1;
    }
    HIP_TRY(c, hipGetLastError());
    *result_half = cur;
    return ICPMI_OK;
}

icpmi_status octree_sample_dev(icpmi_ctx* c, const float4* d_in, int64_t n, float max_size, int max_pts, int method, int* d_order, int* d_leaf_of,
                               int64_t* n_out)
{
    *n_out = 0;
    if (n == 0) return ICPMI_OK;
    if (n > 0x7ffffff0ll) { c->last_error = "octree: too many points"; return ICPMI_ERR_UNSUPPORTED; }
    if (max_pts < 1) max_pts = 1;
    if (max_pts > 64) { c->last_error = "octree: maxPointByNode above 64 is not supported on the device"; return ICPMI_ERR_UNSUPPORTED; }
    if (method != 0 && method != 1) { c->last_error = "octree: samplingMethod must be 0 (first) or 1 (random)"; return ICPMI_ERR_INVALID_ARG; }
    const int blocks = (int)((n + 255) / 256);
    const int rb = blocks < 1024 ? blocks : 1024;
    const int nwg = (int)((n + RS_EPB - 1) / RS_EPB);
    const size_t tab = (size_t)RS_BINS * nwg + RS_BINS;
    unsigned long long* d_keys = scratch_get<unsigned long long>(c, 0, (size_t)2 * n + 2);
    unsigned* d_vals = scratch_get<unsigned>(c, 1, (size_t)2 * n + 2);
    unsigned* d_tab = scratch_get<unsigned>(c, 2, tab * 11 + 16);
    float* d_part = scratch_get<float>(c, 3, (size_t)6 * rb + sizeof(OctRoot) / sizeof(float) + 8);
    unsigned* d_flag = scratch_get<unsigned>(c, 4, (size_t)2 * n + 4);
    if (!d_keys || !d_vals || !d_tab || !d_part || !d_flag) return ICPMI_ERR_HIP;
    OctRoot* d_root = reinterpret_cast<OctRoot*>(d_part + 6 * rb);
    // The depth of the tree (= the number of sort passes) is a function of the bounding cube, which the host does not know.  r4 waited for
    // it; r5 sorts with the depth of the handle's PREVIOUS call (a map that grows by a scan keeps its cube to within a factor of two
    // almost always) and checks the real one when it waits for the leaf count anyway -- the root travels through host-mapped memory, no
    // copy launch.  A wrong guess repeats paths + sort with the right depth (tests/test_gpu_octree.py forces one).
    OctRoot* d_root_host = c->d_progress ? reinterpret_cast<OctRoot*>(c->d_progress + ICPMI_PROGRESS_OCT_WORD) : nullptr;
    const volatile OctRoot* h_root = c->h_progress ? reinterpret_cast<const volatile OctRoot*>(c->h_progress + ICPMI_PROGRESS_OCT_WORD) : nullptr;
    hipLaunchKernelGGL(oct_bbox_kernel, dim3(rb), dim3(OB), 0, c->stream, d_in, n, d_part);
    const int root_tag = ++c->oct_tag ? c->oct_tag : ++c->oct_tag;
    hipLaunchKernelGGL(oct_root_kernel, dim3(1), dim3(OB), 0, c->stream, (const float*)d_part, rb, max_size, d_root, d_root_host, root_tag);
    HIP_TRY(c, hipGetLastError());
    int depth = (h_root && c->oct_depth_hint > 0) ? c->oct_depth_hint : -1;
    if (depth < 0) {
        OctRoot root;
        if (read_back(c, &root, d_root, sizeof root) != ICPMI_OK) return ICPMI_ERR_HIP;
        depth = root.depth;
    }
    unsigned long long* kb[2] = {d_keys, d_keys + n};
    unsigned* vb[2] = {d_vals, d_vals + n};
    unsigned* d_ord = d_flag + n + 2; // exclusive scan of the leaf-start flags
    int cur = 0;
    int64_t leaves = 0;
    for (int attempt = 0;; ++attempt) {
        hipLaunchKernelGGL(oct_path_kernel, dim3(blocks), dim3(256), 0, c->stream, d_in, n, (const OctRoot*)d_root, d_keys, d_vals);
        HIP_TRY(c, hipGetLastError());
        {
            const icpmi_status ss = radix_sort_pairs(c, d_keys, d_vals, n, 3 * depth, d_tab, &cur);
            if (ss != ICPMI_OK) return ss;
        }
        hipLaunchKernelGGL(oct_leaf_kernel, dim3(blocks), dim3(256), 0, c->stream, (const unsigned long long*)kb[cur], n, (const OctRoot*)d_root, max_pts,
                           d_flag);
        const icpmi_status s = device_scan_flags_count(c, d_flag, d_ord, (int)n, &leaves); // (spins on the count's tag; the stream is NOT drained)
        if (s != ICPMI_OK) return s;
        int real_depth = depth;
        if (h_root) {
            // the root kernel ran long before the scan whose count just arrived; its tag says that its stores are visible here
            bool seen = false;
            for (int spins = 0; spins < (1 << 20) && !seen; ++spins) seen = __atomic_load_n(const_cast<const int*>(&h_root->pad[0]), __ATOMIC_ACQUIRE) == root_tag;
            if (seen) real_depth = h_root->depth;
            else { OctRoot root; if (read_back(c, &root, d_root, sizeof root) != ICPMI_OK) return ICPMI_ERR_HIP; real_depth = root.depth; }
        }
        c->oct_depth_hint = real_depth;
        if (real_depth <= depth || attempt > 0) break; // (a guess that was too deep sorted on a few zero bits more: still the order of the paths)
        depth = real_depth; // the guess was too shallow: once more with the cube's own depth
        ++c->oct_respeculated;
    }
    unsigned long long* d_best = kb[cur ^ 1]; // the other key buffer is free now
    HIP_TRY(c, hipMemsetAsync(d_best, 0xff, (size_t)leaves * sizeof(unsigned long long), c->stream));
    hipLaunchKernelGGL(oct_pick_kernel, dim3(blocks), dim3(256), 0, c->stream, (const unsigned*)vb[cur], (const unsigned*)d_flag, (const unsigned*)d_ord, n,
                       method, d_best, d_order, d_leaf_of);
    hipLaunchKernelGGL(oct_best_kernel, dim3((int)((leaves + 255) / 256)), dim3(256), 0, c->stream, (const unsigned long long*)d_best, leaves, d_order);
    HIP_TRY(c, hipGetLastError());
    *n_out = leaves;
    return ICPMI_OK;
}

// host-pointer entry (icpmi_octree_sample)
icpmi_status ops_octree_sample(icpmi_ctx* c, const float* in4, int64_t n, float max_size, int max_pts, int method, int32_t* order_out,
                               int32_t* leaf_of_out, int64_t* n_out)
{
    if (n_out) *n_out = 0;
    if (n == 0) return ICPMI_OK;
    DevBuf<float4> d_in; DevBuf<int> d_order, d_leaf;
    HIP_TRY(c, d_in.alloc((size_t)n));
    HIP_TRY(c, d_order.alloc((size_t)n));
    if (leaf_of_out) HIP_TRY(c, d_leaf.alloc((size_t)n));
    HIP_TRY(c, hipMemcpyAsync(d_in, in4, (size_t)n * sizeof(float4), hipMemcpyHostToDevice, c->stream));
    int64_t m = 0;
    icpmi_status s = octree_sample_dev(c, d_in, n, max_size, max_pts, method, d_order, leaf_of_out ? d_leaf.p : nullptr, &m);
    if (s != ICPMI_OK) return s;
    if (order_out) HIP_TRY(c, hipMemcpyAsync(order_out, d_order, (size_t)m * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    if (leaf_of_out) HIP_TRY(c, hipMemcpyAsync(leaf_of_out, d_leaf, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (n_out) *n_out = m;
    return ICPMI_OK;
}
