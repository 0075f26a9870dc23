// ssn.hip -- SamplingSurfaceNormalDataPointsFilter on the device.
//
// PM::ICPSequence::setDefault() (the chain of a configuration without an `icp:` key, Mapper.cpp:74-78) puts this
// filter on the REFERENCE: at every `icp.setMap` (Map.cpp:111,178,528,581) the whole map is cut by median splits of the widest box
// dimension until a box holds <= knn points, every box gets one normal (PCA of its points), and its points survive with probability
// `ratio`.  r2 ran the recursion on one host thread (seconds for a million points); this is the same partition, the same normals and the
// same std::minstd_rand stream as the oracle's orc_sampling_surface_normal, produced level by level:
//
//   * the shape of the tree depends on n and knn alone (a node of c points splits into c - c/2 and c/2), so every node is a fixed range of
//     positions; per level ONE stable radix sort of all points on the key (node start << 32 | order-preserving bits of the coordinate along
//     the node's split dimension), taken from the IDENTITY order, leaves every node sorted by (coordinate, index) -- the oracle's order;
//   * a second kernel hands every point its child range and every split node its children's boxes (the parent's box cut at the median);
//   * after the last level one more sort (all nodes are leaves: key = node start) leaves every box in index order; one lane per box does
//     the covariance and a double-precision Jacobi sweep; the position of a point in the depth-first stream of surviving boxes is a prefix
//     sum, and its random number is the minstd state a^k x0 mod (2^31 - 1) by modular exponentiation -- no sequential generator.
#include "common.h"

#include <algorithm>

namespace {

struct SsnBox { float lo[3], hi[3]; };

__device__ __forceinline__ unsigned ordered_bits(float x)
{
    if (x == 0.f) x = 0.f; // -0 and +0 compare equal on the host: one key
    const unsigned u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ int widest_dim(const SsnBox& b)
{
    int dim = 0;
    for (int r = 1; r < 3; ++r) if (b.hi[r] - b.lo[r] > b.hi[dim] - b.lo[dim]) dim = r;
    return dim;
}

__global__ __launch_bounds__(256) void ssn_bbox_kernel(const float4* __restrict__ p, int64_t n, float* __restrict__ part)
{
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float4 q = p[i];
        lo[0] = fminf(lo[0], q.x); lo[1] = fminf(lo[1], q.y); lo[2] = fminf(lo[2], q.z);
        hi[0] = fmaxf(hi[0], q.x); hi[1] = fmaxf(hi[1], q.y); hi[2] = fmaxf(hi[2], q.z);
    }
    __shared__ float sh[6][256];
    for (int r = 0; r < 3; ++r) { sh[r][threadIdx.x] = lo[r]; sh[3 + r][threadIdx.x] = hi[r]; }
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off)
            for (int r = 0; r < 3; ++r) {
                sh[r][threadIdx.x] = fminf(sh[r][threadIdx.x], sh[r][threadIdx.x + off]);
                sh[3 + r][threadIdx.x] = fmaxf(sh[3 + r][threadIdx.x], sh[3 + r][threadIdx.x + off]);
            }
        __syncthreads();
    }
    if (threadIdx.x < 6) part[6 * blockIdx.x + threadIdx.x] = sh[threadIdx.x][0];
}

__global__ __launch_bounds__(256) void ssn_init_kernel(int64_t n, unsigned* __restrict__ nstart, unsigned* __restrict__ ncnt)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { nstart[i] = 0u; ncnt[i] = (unsigned)n; }
}

// keys of one level, from the identity order: (node start << 32) | coordinate bits along the node's split dimension (0 for a finished box)
__global__ __launch_bounds__(256) void ssn_key_kernel(const float4* __restrict__ p, int64_t n, int knn, const unsigned* __restrict__ nstart,
                                                      const unsigned* __restrict__ ncnt, const SsnBox* __restrict__ box,
                                                      unsigned long long* __restrict__ keys, unsigned* __restrict__ vals)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned s = nstart[i];
    unsigned lowbits = 0u;
    if (ncnt[i] > (unsigned)knn) {
        const SsnBox b = box[s];
        const int dim = widest_dim(b);
        const float4 q = p[i];
        lowbits = ordered_bits(dim == 0 ? q.x : (dim == 1 ? q.y : q.z));
    }
    keys[i] = ((unsigned long long)s << 32) | lowbits;
    vals[i] = (unsigned)i;
}

// after the sort: position -> child range of its point; the first position of a split node writes the children's boxes
__global__ __launch_bounds__(256) void ssn_split_kernel(const float4* __restrict__ p, int64_t n, int knn, const unsigned* __restrict__ order,
                                                        unsigned* __restrict__ nstart, unsigned* __restrict__ ncnt, const SsnBox* __restrict__ box_in,
                                                        SsnBox* __restrict__ box_out)
{
    const int64_t pos = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (pos >= n) return;
    const unsigned i = order[pos];
    const unsigned s = nstart[i], cnt = ncnt[i];
    if (cnt <= (unsigned)knn) return;
    const unsigned right = cnt / 2u, left = cnt - right;
    if ((unsigned)pos == s) {
        const SsnBox b = box_in[s];
        const int dim = widest_dim(b);
        const float4 q = p[order[s + left]];
        const float cut = dim == 0 ? q.x : (dim == 1 ? q.y : q.z);
        SsnBox l = b, r = b;
        l.hi[dim] = cut; r.lo[dim] = cut;
        box_out[s] = l; box_out[s + left] = r;
    }
    if ((unsigned)pos - s < left) ncnt[i] = left;
    else { nstart[i] = s + left; ncnt[i] = right; }
}

__device__ void ssn_jacobi3(double* A, double* w, double* Q) // the oracle's cyclic Jacobi (icp_oracle.c: jacobi_eig_sym), n = 3, column-major
{
    for (int i = 0; i < 9; ++i) Q[i] = (i % 4 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0, dg = 0;
        for (int a = 0; a < 3; ++a) for (int b = a + 1; b < 3; ++b) off += A[3 * b + a] * A[3 * b + a];
        for (int a = 0; a < 3; ++a) dg += A[3 * a + a] * A[3 * a + a];
        if (off <= 1e-32 * dg || off < 1e-300) break;
        for (int a = 0; a < 2; ++a)
            for (int b = a + 1; b < 3; ++b) {
                const double apq = A[3 * b + a];
                if (apq == 0.0) continue;
                const double theta = (A[3 * b + b] - A[3 * a + a]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) { const double x = A[3 * a + k], y = A[3 * b + k]; A[3 * a + k] = c * x - s * y; A[3 * b + k] = s * x + c * y; }
                for (int k = 0; k < 3; ++k) { const double x = A[3 * k + a], y = A[3 * k + b]; A[3 * k + a] = c * x - s * y; A[3 * k + b] = s * x + c * y; }
                for (int k = 0; k < 3; ++k) { const double x = Q[3 * a + k], y = Q[3 * b + k]; Q[3 * a + k] = c * x - s * y; Q[3 * b + k] = s * x + c * y; }
            }
    }
    for (int i = 0; i < 3; ++i) w[i] = A[3 * i + i];
}

// one lane per box (its first position): bounding box, mean, covariance in double, normal = eigenvector of the smallest eigenvalue;
// boxes wider than maxBoxDim or of rank < 2 are dropped.  draws[pos] = 1 for every point of a surviving box (it consumes one random number).
__global__ __launch_bounds__(128) void ssn_fuse_kernel(const float4* __restrict__ p, int64_t n, const unsigned* __restrict__ order,
                                                       const unsigned* __restrict__ nstart, const unsigned* __restrict__ ncnt, float max_box,
                                                       float* __restrict__ box_normal, unsigned* __restrict__ draws, float* __restrict__ box_mean)
{
    const int64_t pos = (int64_t)blockIdx.x * 128 + threadIdx.x;
    if (pos >= n) return;
    const unsigned i0 = order[pos];
    if (nstart[i0] != (unsigned)pos) return;
    const unsigned cnt = ncnt[i0];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    double mean[3] = {0, 0, 0};
    for (unsigned k = 0; k < cnt; ++k) {
        const float4 q = p[order[pos + k]];
        const float v[3] = {q.x, q.y, q.z};
        for (int r = 0; r < 3; ++r) { lo[r] = fminf(lo[r], v[r]); hi[r] = fmaxf(hi[r], v[r]); mean[r] += (double)v[r]; }
    }
    bool ok = fmaxf(hi[0] - lo[0], fmaxf(hi[1] - lo[1], hi[2] - lo[2])) <= max_box;
    float nrm[3] = {0.f, 0.f, 0.f};
    if (ok) {
        for (int r = 0; r < 3; ++r) mean[r] /= (double)cnt;
        double C[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (unsigned k = 0; k < cnt; ++k) {
            const float4 q = p[order[pos + k]];
            const double v[3] = {(double)q.x - mean[0], (double)q.y - mean[1], (double)q.z - mean[2]};
            for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) C[3 * c + r] += v[r] * v[c];
        }
        double w[3], Q[9];
        ssn_jacobi3(C, w, Q);
        const double wmax = fmax(fabs(w[0]), fmax(fabs(w[1]), fabs(w[2])));
        int rank = 0;
        for (int e = 0; e < 3; ++e) if (wmax > 0 && fabs(w[e]) > 3.0 * 1.1920928955078125e-07 * wmax) ++rank;
        ok = rank >= 2;
        int e = 0;
        if (w[1] < w[e]) e = 1;
        if (w[2] < w[e]) e = 2;
        for (int r = 0; r < 3; ++r) nrm[r] = (float)Q[3 * e + r];
    }
    for (int r = 0; r < 3; ++r) box_normal[3 * pos + r] = nrm[r];
    if (box_mean) for (int r = 0; r < 3; ++r) box_mean[3 * pos + r] = (float)mean[r]; // (samplingMethod 1: the box's new point; divided above when the box survives)
    for (unsigned k = 0; k < cnt; ++k) draws[pos + k] = ok ? 1u : 0u;
}

__device__ __forceinline__ unsigned minstd_mulmod(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) % 2147483647ull); }

// the n-th value (n >= 1) of std::minstd_rand seeded with `seed`: x0 48271^n mod (2^31 - 1) by square-and-multiply -- no sequential generator.
// [rand.predef]: the 10 000th value of a default-constructed (seed 1) minstd_rand is 399268537 (tests/test_gpu_pins.py holds this function to it).
__device__ __forceinline__ unsigned minstd_nth(unsigned seed, unsigned n)
{
    unsigned x = seed % 2147483647u;
    if (x == 0u) x = 1u;
    unsigned e = n, base = 48271u, acc = 1u;
    while (e) { if (e & 1u) acc = minstd_mulmod(acc, base); base = minstd_mulmod(base, base); e >>= 1; }
    return minstd_mulmod(acc, x);
}
__global__ void minstd_nth_kernel(unsigned seed, unsigned n, unsigned* __restrict__ out) { if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = minstd_nth(seed, n); }

// keep[pos] = the point's random number (the (rank + 1)-th of std::minstd_rand seeded with `seed`) is below ratio
__global__ __launch_bounds__(256) void ssn_draw_kernel(int64_t n, const unsigned* __restrict__ draws, const unsigned* __restrict__ rank, float ratio,
                                                       unsigned seed, unsigned* __restrict__ keep)
{
    const int64_t pos = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (pos >= n) return;
    unsigned k = 0u;
    if (draws[pos]) {
        const unsigned x = minstd_nth(seed, rank[pos] + 1u);
        k = ((float)x / 2147483645.0f) < ratio ? 1u : 0u;
    }
    keep[pos] = k;
}

__global__ __launch_bounds__(256) void ssn_emit_kernel(int64_t n, const unsigned* __restrict__ order, const unsigned* __restrict__ nstart,
                                                       const unsigned* __restrict__ keep, const unsigned* __restrict__ outpos,
                                                       const float* __restrict__ box_normal, int* __restrict__ order_out, float* __restrict__ normals_out)
{
    const int64_t pos = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (pos >= n || !keep[pos]) return;
    const unsigned i = order[pos], o = outpos[pos], s = nstart[i];
    order_out[o] = (int)i;
    for (int r = 0; r < 3; ++r) normals_out[3 * (size_t)o + r] = box_normal[3 * (size_t)s + r];
}

// samplingMethod 1: one output per surviving box.  first[pos] = pos is the first position of a surviving box
__global__ __launch_bounds__(256) void ssn_first_kernel(int64_t n, const unsigned* __restrict__ order, const unsigned* __restrict__ nstart,
                                                        const unsigned* __restrict__ draws, unsigned* __restrict__ first)
{
    const int64_t pos = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (pos >= n) return;
    first[pos] = (draws[pos] && nstart[order[pos]] == (unsigned)pos) ? 1u : 0u;
}

// ... box j (= outpos of its first position): smallest member index (the box is in index order), normal, mean, its run in the member list;
// every point of a surviving box goes to the member list at its rank among such points
__global__ __launch_bounds__(256) void ssn_emit_box_kernel(int64_t n, const unsigned* __restrict__ order, const unsigned* __restrict__ ncnt,
                                                           const unsigned* __restrict__ draws, const unsigned* __restrict__ rank,
                                                           const unsigned* __restrict__ first, const unsigned* __restrict__ outpos,
                                                           const float* __restrict__ box_normal, const float* __restrict__ box_mean,
                                                           int* __restrict__ order_out, float* __restrict__ normals_out, float* __restrict__ mean_out,
                                                           int* __restrict__ mstart_out, int* __restrict__ mcount_out, int* __restrict__ members_out)
{
    const int64_t pos = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (pos >= n || !draws[pos]) return;
    const unsigned i = order[pos];
    if (members_out) members_out[rank[pos]] = (int)i;
    if (!first[pos]) return;
    const unsigned o = outpos[pos];
    order_out[o] = (int)i;
    for (int r = 0; r < 3; ++r) normals_out[3 * (size_t)o + r] = box_normal[3 * (size_t)pos + r];
    if (mean_out) for (int r = 0; r < 3; ++r) mean_out[3 * (size_t)o + r] = box_mean[3 * (size_t)pos + r];
    if (mstart_out) mstart_out[o] = (int)rank[pos];
    if (mcount_out) mcount_out[o] = (int)ncnt[i];
}

} // namespace

// test seam (icpmi_debug_minstd_nth): the value ssn_draw_kernel's skip-ahead computes for the n-th number of the stream
icpmi_status ssn_debug_minstd(icpmi_ctx* c, unsigned seed, unsigned n, unsigned* out)
{
    DevBuf<unsigned> d; HIP_TRY(c, d.alloc(1));
    hipLaunchKernelGGL(minstd_nth_kernel, dim3(1), dim3(64), 0, c->stream, seed, n, d.p);
    HIP_TRY(c, hipGetLastError());
    if (read_back(c, out, d.p, sizeof(unsigned)) != ICPMI_OK) return ICPMI_ERR_HIP;
    return ICPMI_OK;
}

// device pointers in, device pointers out (d_order_out / d_normals_out: capacity n / 3 n); *n_out read back once
// method 1 (samplingMethod 1): one output per surviving box -- its smallest index, its normal, d_mean_out (3 per box) its new position,
// d_members_out[d_mstart_out[j] .. + d_mcount_out[j]) its members in index order (*n_members_out of them in all); the extras may be null.
icpmi_status ssn_sample_dev(icpmi_ctx* c, const float4* d_in, int64_t n, float ratio, int knn, float max_box, unsigned seed, int* d_order_out,
                            float* d_normals_out, int64_t* n_out, int method, float* d_mean_out, int* d_mstart_out, int* d_mcount_out,
                            int* d_members_out, int64_t* n_members_out)
{
    *n_out = 0;
    if (n_members_out) *n_members_out = 0;
    if (n == 0) return ICPMI_OK;
    if (n > 0x7ffffff0ll) { c->last_error = "SamplingSurfaceNormal: too many points"; return ICPMI_ERR_UNSUPPORTED; }
    if (knn < 3) { c->last_error = "InvalidParameter: SamplingSurfaceNormalDataPointsFilter knn must be >= 3"; return ICPMI_ERR_INVALID_ARG; }
    const int blocks = (int)((n + 255) / 256);
    int nbits = 1;
    while ((1ll << nbits) <= n) ++nbits;
    const int bits = 32 + nbits;
    DevBuf<unsigned long long> d_keys; DevBuf<unsigned> d_vals, d_nstart, d_ncnt, d_tab, d_draws, d_rank, d_keep, d_outpos;
    DevBuf<SsnBox> d_box[2]; DevBuf<float> d_part, d_bnrm, d_bmean;
    HIP_TRY(c, d_keys.alloc(2 * (size_t)n)); HIP_TRY(c, d_vals.alloc(2 * (size_t)n));
    HIP_TRY(c, d_nstart.alloc((size_t)n)); HIP_TRY(c, d_ncnt.alloc((size_t)n));
    HIP_TRY(c, d_tab.alloc(radix_sort_tab_words(n, bits)));
    HIP_TRY(c, d_box[0].alloc((size_t)n + 1)); HIP_TRY(c, d_box[1].alloc((size_t)n + 1));
    HIP_TRY(c, d_draws.alloc((size_t)n + 2)); HIP_TRY(c, d_rank.alloc((size_t)n + 2)); HIP_TRY(c, d_keep.alloc((size_t)n + 2)); HIP_TRY(c, d_outpos.alloc((size_t)n + 2));
    HIP_TRY(c, d_bnrm.alloc(3 * (size_t)n));
    if (method == 1) HIP_TRY(c, d_bmean.alloc(3 * (size_t)n));
    constexpr int RB = 64;
    HIP_TRY(c, d_part.alloc(6 * RB));
    // root box
    hipLaunchKernelGGL(ssn_bbox_kernel, dim3(RB), dim3(256), 0, c->stream, d_in, n, d_part.p);
    float hp[6 * RB];
    if (read_back(c, hp, d_part.p, sizeof hp) != ICPMI_OK) return ICPMI_ERR_HIP;
    SsnBox root;
    for (int r = 0; r < 3; ++r) { root.lo[r] = INFINITY; root.hi[r] = -INFINITY; }
    for (int b = 0; b < RB; ++b)
        for (int r = 0; r < 3; ++r) { root.lo[r] = std::min(root.lo[r], hp[6 * b + r]); root.hi[r] = std::max(root.hi[r], hp[6 * b + 3 + r]); }
    { const icpmi_status us = upload_small(c, d_box[0].p, &root, sizeof root); if (us != ICPMI_OK) return us; }
    hipLaunchKernelGGL(ssn_init_kernel, dim3(blocks), dim3(256), 0, c->stream, n, d_nstart.p, d_ncnt.p);
    // levels: the largest node of level L + 1 holds ceil(size / 2) points
    int levels = 0;
    for (int64_t sz = n; sz > knn; sz = sz - sz / 2) ++levels;
    int cur_box = 0, half = 0;
    for (int lv = 0; lv <= levels; ++lv) { // the last round sorts finished boxes only: index order inside every box
        hipLaunchKernelGGL(ssn_key_kernel, dim3(blocks), dim3(256), 0, c->stream, d_in, n, knn, (const unsigned*)d_nstart.p, (const unsigned*)d_ncnt.p,
                           (const SsnBox*)d_box[cur_box].p, d_keys.p, d_vals.p);
        const icpmi_status ss = radix_sort_pairs(c, d_keys.p, d_vals.p, n, bits, d_tab.p, &half);
        if (ss != ICPMI_OK) return ss;
        if (lv == levels) break;
        hipLaunchKernelGGL(ssn_split_kernel, dim3(blocks), dim3(256), 0, c->stream, d_in, n, knn, (const unsigned*)(d_vals.p + (half ? n : 0)), d_nstart.p, d_ncnt.p,
                           (const SsnBox*)d_box[cur_box].p, d_box[cur_box ^ 1].p);
        cur_box ^= 1;
    }
    const unsigned* d_order = d_vals.p + (half ? n : 0);
    hipLaunchKernelGGL(ssn_fuse_kernel, dim3((int)((n + 127) / 128)), dim3(128), 0, c->stream, d_in, n, d_order, (const unsigned*)d_nstart.p, (const unsigned*)d_ncnt.p,
                       max_box, d_bnrm.p, d_draws.p, method == 1 ? d_bmean.p : nullptr);
    icpmi_status s = device_exclusive_scan_io(c, d_draws.p, d_rank.p, (int)n, 0u);
    if (s != ICPMI_OK) return s;
    if (method == 1) { // no random number: every surviving box gives one point
        hipLaunchKernelGGL(ssn_first_kernel, dim3(blocks), dim3(256), 0, c->stream, n, d_order, (const unsigned*)d_nstart.p, (const unsigned*)d_draws.p, d_keep.p);
        s = device_exclusive_scan_io(c, d_keep.p, d_outpos.p, (int)n, 0u);
        if (s != ICPMI_OK) return s;
        unsigned lp = 0, lk = 0, lr = 0, ld = 0;
        if (read_back2(c, &lp, d_outpos.p + (n - 1), sizeof(unsigned), &lk, d_keep.p + (n - 1), sizeof(unsigned)) != ICPMI_OK) return ICPMI_ERR_HIP;
        if (read_back2(c, &lr, d_rank.p + (n - 1), sizeof(unsigned), &ld, d_draws.p + (n - 1), sizeof(unsigned)) != ICPMI_OK) return ICPMI_ERR_HIP;
        hipLaunchKernelGGL(ssn_emit_box_kernel, dim3(blocks), dim3(256), 0, c->stream, n, d_order, (const unsigned*)d_ncnt.p, (const unsigned*)d_draws.p,
                           (const unsigned*)d_rank.p, (const unsigned*)d_keep.p, (const unsigned*)d_outpos.p, (const float*)d_bnrm.p, (const float*)d_bmean.p,
                           d_order_out, d_normals_out, d_mean_out, d_mstart_out, d_mcount_out, d_members_out);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipStreamSynchronize(c->stream)); // the scratch of this call is freed on return
        *n_out = (int64_t)lp + lk;
        if (n_members_out) *n_members_out = (int64_t)lr + ld;
        return ICPMI_OK;
    }
    hipLaunchKernelGGL(ssn_draw_kernel, dim3(blocks), dim3(256), 0, c->stream, n, (const unsigned*)d_draws.p, (const unsigned*)d_rank.p, ratio, seed, d_keep.p);
    s = device_exclusive_scan_io(c, d_keep.p, d_outpos.p, (int)n, 0u);
    if (s != ICPMI_OK) return s;
    unsigned lp = 0, lk = 0;
    if (read_back2(c, &lp, d_outpos.p + (n - 1), sizeof(unsigned), &lk, d_keep.p + (n - 1), sizeof(unsigned)) != ICPMI_OK) return ICPMI_ERR_HIP;
    hipLaunchKernelGGL(ssn_emit_kernel, dim3(blocks), dim3(256), 0, c->stream, n, d_order, (const unsigned*)d_nstart.p, (const unsigned*)d_keep.p,
                       (const unsigned*)d_outpos.p, (const float*)d_bnrm.p, d_order_out, d_normals_out);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream)); // the scratch of this call is freed on return
    *n_out = (int64_t)lp + lk;
    return ICPMI_OK;
}

// host-pointer entries (icpmi_sampling_surface_normal, icpmi_sampling_surface_normal_ex)
icpmi_status ops_sampling_surface_normal_ex(icpmi_ctx* c, const float* in4, int64_t n, float ratio, int knn, float max_box, int seed, int method,
                                            int32_t* order_out, float* normals3_out, int64_t* n_out, float* mean3_out, int32_t* mstart_out,
                                            int32_t* mcount_out, int32_t* members_out)
{
    if (n_out) *n_out = 0;
    if (n == 0) return ICPMI_OK;
    DevBuf<float4> d_in; DevBuf<int> d_order, d_ms, d_mc, d_mem; DevBuf<float> d_nrm, d_mean;
    HIP_TRY(c, d_in.alloc((size_t)n)); HIP_TRY(c, d_order.alloc((size_t)n)); HIP_TRY(c, d_nrm.alloc(3 * (size_t)n));
    if (method == 1) { HIP_TRY(c, d_mean.alloc(3 * (size_t)n)); HIP_TRY(c, d_ms.alloc((size_t)n)); HIP_TRY(c, d_mc.alloc((size_t)n)); HIP_TRY(c, d_mem.alloc((size_t)n)); }
    HIP_TRY(c, hipMemcpyAsync(d_in.p, in4, (size_t)n * sizeof(float4), hipMemcpyHostToDevice, c->stream));
    int64_t kept = 0, members = 0;
    const icpmi_status s = ssn_sample_dev(c, d_in.p, n, ratio, knn, max_box, (unsigned)seed, d_order.p, d_nrm.p, &kept, method, d_mean.p, d_ms.p, d_mc.p, d_mem.p, &members);
    if (s != ICPMI_OK) return s;
    if (kept > 0) {
        if (order_out) HIP_TRY(c, hipMemcpyAsync(order_out, d_order.p, (size_t)kept * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        if (normals3_out) HIP_TRY(c, hipMemcpyAsync(normals3_out, d_nrm.p, 3 * (size_t)kept * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        if (method == 1) {
            if (mean3_out) HIP_TRY(c, hipMemcpyAsync(mean3_out, d_mean.p, 3 * (size_t)kept * sizeof(float), hipMemcpyDeviceToHost, c->stream));
            if (mstart_out) HIP_TRY(c, hipMemcpyAsync(mstart_out, d_ms.p, (size_t)kept * sizeof(int), hipMemcpyDeviceToHost, c->stream));
            if (mcount_out) HIP_TRY(c, hipMemcpyAsync(mcount_out, d_mc.p, (size_t)kept * sizeof(int), hipMemcpyDeviceToHost, c->stream));
            if (members_out && members > 0) HIP_TRY(c, hipMemcpyAsync(members_out, d_mem.p, (size_t)members * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        }
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    if (n_out) *n_out = kept;
    return ICPMI_OK;
}

icpmi_status ops_sampling_surface_normal(icpmi_ctx* c, const float* in4, int64_t n, float ratio, int knn, float max_box, int seed, int32_t* order_out,
                                         float* normals3_out, int64_t* n_out)
{
    return ops_sampling_surface_normal_ex(c, in4, n, ratio, knn, max_box, seed, 0, order_out, normals3_out, n_out, nullptr, nullptr, nullptr, nullptr);
}
