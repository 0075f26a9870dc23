// ops.hip -- the map-side operators of the path that are not part of the ICP iteration itself:
//   RigidTransformation::compute            (Mapper.cpp:197,221; Map.cpp:523,525)
//   SurfaceNormalDataPointsFilter           (Map.cpp:524 through examples/config.yaml:26-27)
//   PointDistanceMapperModule keep mask     (MapperModules/PointDistanceMapperModule.cpp:28-50)
//   Map::unloadCells cell binning           (Map.cpp:206-209,232-235)
#include "common.h"
#include <chrono>
#include <utility>
#include <cstring>

namespace {

// `dists(i) >= std::pow(minDistNewPoint, 2)` (PointDistanceMapperModule.cpp:42): std::pow(float, int) is evaluated in
// double and the float distance is promoted for the comparison.  The product of two floats is exact in double.
inline double pd_limit(float min_dist) { return (double)min_dist * (double)min_dist; }
// squared search radius that is guaranteed to return every neighbour with d2 < limit (the smallest float >= limit)
inline float pd_radius2(double lim) { float r = (float)lim; if ((double)r < lim) r = nextafterf(r, INFINITY); return r; }

// A 4 x 4 (col-major) as a kernel ARGUMENT: the matrix rides in the kernarg segment (scalar loads), no device copy of it and no upload
// launch in front of the kernel that reads it (r5: the map-update chain uploaded four matrices per update, one copy kernel each).
struct Mat16 { float v[16]; };
static inline Mat16 mat16(const float T[16]) { Mat16 m; memcpy(m.v, T, sizeof m.v); return m; }

__global__ __launch_bounds__(256) void transform_kernel(const float4* __restrict__ in, int64_t n, Mat16 M, float4* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float* T = M.v;
    const float4 p = in[i];
    const float3 o = xf_point(T, p.x, p.y, p.z, p.w);
    const float w = fmaf(T[15], p.w, fmaf(T[11], p.z, fmaf(T[7], p.y, T[3] * p.x)));
    out[i] = make_float4(o.x, o.y, o.z, w);
}

__global__ __launch_bounds__(256) void rotate3_kernel(const float* __restrict__ in3, int64_t n, Mat16 M, float* __restrict__ out3)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float* T = M.v;
    const float x = in3[3 * i], y = in3[3 * i + 1], z = in3[3 * i + 2];
    out3[3 * i] = fmaf(T[8], z, fmaf(T[4], y, T[0] * x));
    out3[3 * i + 1] = fmaf(T[9], z, fmaf(T[5], y, T[1] * x));
    out3[3 * i + 2] = fmaf(T[10], z, fmaf(T[6], y, T[2] * x));
}

// RigidTransformation::compute on a whole resident cloud, in place: features by T and -- n3 != nullptr -- normals by its rotation, one
// launch (the arithmetic of transform_kernel and rotate3_kernel, element for element)
__global__ __launch_bounds__(256) void move_kernel(float4* __restrict__ pts, float* __restrict__ n3, int64_t n, Mat16 M)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float* T = M.v;
    const float4 p = pts[i];
    const float3 o = xf_point(T, p.x, p.y, p.z, p.w);
    const float w = fmaf(T[15], p.w, fmaf(T[11], p.z, fmaf(T[7], p.y, T[3] * p.x)));
    pts[i] = make_float4(o.x, o.y, o.z, w);
    if (n3) {
        const float x = n3[3 * i], y = n3[3 * i + 1], z = n3[3 * i + 2];
        n3[3 * i] = fmaf(T[8], z, fmaf(T[4], y, T[0] * x));
        n3[3 * i + 1] = fmaf(T[9], z, fmaf(T[5], y, T[1] * x));
        n3[3 * i + 2] = fmaf(T[10], z, fmaf(T[6], y, T[2] * x));
    }
}

__global__ __launch_bounds__(256) void bin_kernel(const float4* __restrict__ in, int64_t n, float cell, int* __restrict__ ijk)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 p = in[i];
    ijk[3 * i] = (int)floorf(p.x / cell);
    ijk[3 * i + 1] = (int)floorf(p.y / cell);
    ijk[3 * i + 2] = (int)floorf(p.z / cell);
}

__global__ __launch_bounds__(256) void keep_kernel(const float* __restrict__ d2, int64_t n, double lim, uint8_t* __restrict__ keep)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    keep[i] = (double)d2[i] >= lim ? 1 : 0; // `dists(i) >= std::pow(minDistNewPoint, 2)`: a double comparison (PointDistanceMapperModule.cpp:42)
}

// one lane per point: mean + covariance of the kNN set in double, smallest eigenvector by Jacobi
// densities (may be null): `keepDensities` of the filter -- points per volume of the sphere that holds the neighbourhood around
// its centroid: k / (4/3 pi r^3), r = the largest distance of a neighbour from the centroid (computeDensity)
// mean_dist (may be null): `keepMeanDist` -- distance from the point to the mean of its neighbours.  The point is read as its own first
// neighbour (d2 = 0 sorts first; a duplicate that wins the index tie has the same coordinates), which keeps it in the index's centred frame.
// KMAX > 0 (k <= KMAX; r5): the row of neighbour ids and the k neighbours are requested up front -- two round trips -- and both passes (mean,
// scatter matrix) run from registers in the same order; the generic variant (KMAX = 0: any k) walks them one dependent load after the other,
// twice.  Same sums in the same order: same bits.
template <int KMAX>
__global__ __launch_bounds__(128) void normals_kernel(const float4* __restrict__ map, const int* __restrict__ sidx, int64_t m, int k,
                                                      float* __restrict__ normals3, float* __restrict__ densities, int dim2,
                                                      float* __restrict__ mean_dist = nullptr, float* __restrict__ eig_values = nullptr,
                                                      float* __restrict__ eig_vectors = nullptr, const unsigned* __restrict__ list = nullptr)
{
    // list (r6): an appended cloud -- m entries, the points whose neighbourhood was searched again (surface_normals_dev); otherwise all m points
    const int64_t t = (int64_t)blockIdx.x * 128 + threadIdx.x;
    if (t >= m) return;
    const int64_t i = list ? (int64_t)list[t] : t;
    // (r2: walking the points in cell-sorted order instead -- coherent neighbour gathers, scattered row reads and normal
    // writes -- measured 108 -> 164 us on the octree-ordered 0.9 M-point map and within noise on an append-ordered one: kept
    // in the caller's order)
    double mean[3] = {0, 0, 0};
    int real = 0;
    double c00 = 0, c01 = 0, c02 = 0, c11 = 0, c12 = 0, c22 = 0, rmax2 = 0;
    float4 p0 = make_float4(0.f, 0.f, 0.f, 0.f); // the point itself (its own first neighbour), or map[0] where the row starts with "none"
    if constexpr (KMAX > 0) {
        int sv[KMAX];
        float4 qv[KMAX];
#pragma unroll
        for (int j = 0; j < KMAX; ++j) sv[j] = j < k ? sidx[(size_t)k * i + j] : -1;
#pragma unroll
        for (int j = 0; j < KMAX; ++j) qv[j] = map[sv[j] < 0 ? 0 : sv[j]];
#pragma unroll
        for (int j = 0; j < KMAX; ++j)
            if (sv[j] >= 0) { mean[0] += qv[j].x; mean[1] += qv[j].y; mean[2] += qv[j].z; ++real; }
        const double inv = 1.0 / (real > 0 ? real : 1);
        mean[0] *= inv; mean[1] *= inv; mean[2] *= inv;
#pragma unroll
        for (int j = 0; j < KMAX; ++j)
            if (sv[j] >= 0) {
                const double x = qv[j].x - mean[0], y = qv[j].y - mean[1], z = qv[j].z - mean[2];
                c00 += x * x; c01 += x * y; c02 += x * z; c11 += y * y; c12 += y * z; c22 += z * z;
                const double r2 = x * x + y * y + z * z;
                rmax2 = r2 > rmax2 ? r2 : rmax2;
            }
        p0 = qv[0];
    } else {
        for (int j = 0; j < k; ++j) {
            const int s = sidx[(size_t)k * i + j];
            if (s < 0) continue;
            const float4 q = map[s];
            mean[0] += q.x; mean[1] += q.y; mean[2] += q.z; ++real;
        }
        const double inv = 1.0 / (real > 0 ? real : 1);
        mean[0] *= inv; mean[1] *= inv; mean[2] *= inv;
        for (int j = 0; j < k; ++j) {
            const int s = sidx[(size_t)k * i + j];
            if (s < 0) continue;
            const float4 q = map[s];
            const double x = q.x - mean[0], y = q.y - mean[1], z = q.z - mean[2];
            c00 += x * x; c01 += x * y; c02 += x * z; c11 += y * y; c12 += y * z; c22 += z * z;
            const double r2 = x * x + y * y + z * z;
            rmax2 = r2 > rmax2 ? r2 : rmax2;
        }
        if (mean_dist) { const int s0 = sidx[(size_t)k * i]; p0 = map[s0 < 0 ? 0 : s0]; }
    }
    if (densities) {
        const double r = sqrt(rmax2);
        densities[i] = (float)((double)real / ((4.0 / 3.0) * 3.14159265358979323846 * (r * r * r)));
    }
    if (mean_dist) {
        const double x = p0.x - mean[0], y = p0.y - mean[1], z = p0.z - mean[2];
        mean_dist[i] = (float)sqrt(x * x + y * y + z * z);
    }
    // cyclic Jacobi on the symmetric 3x3
    double A[3][3] = {{c00, c01, c02}, {c01, c11, c12}, {c02, c12, c22}};
    double Q[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 30; ++sweep) {
        const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
        const double dg = A[0][0] * A[0][0] + A[1][1] * A[1][1] + A[2][2] * A[2][2];
        if (off <= 1e-32 * dg || off < 1e-300) break;
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int q = p + 1; q < 3; ++q) {
                const double apq = A[p][q];
                if (apq == 0.0) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
                for (int kk = 0; kk < 3; ++kk) {
                    const double akp = A[kk][p], akq = A[kk][q];
                    A[kk][p] = c * akp - s * akq; A[kk][q] = s * akp + c * akq;
                }
#pragma unroll
                for (int kk = 0; kk < 3; ++kk) {
                    const double apk = A[p][kk], aqk = A[q][kk];
                    A[p][kk] = c * apk - s * aqk; A[q][kk] = s * apk + c * aqk;
                }
#pragma unroll
                for (int kk = 0; kk < 3; ++kk) {
                    const double qkp = Q[kk][p], qkq = Q[kk][q];
                    Q[kk][p] = c * qkp - s * qkq; Q[kk][q] = s * qkp + c * qkq;
                }
            }
    }
    const double w0 = A[0][0], w1 = A[1][1], w2 = A[2][2];
    const double wmax = fmax(fabs(w0), fmax(fabs(w1), fabs(w2)));
    const double thr = 3.0 * 1.1920928955078125e-07 * wmax;
    const int rank = (wmax > 0) ? ((fabs(w0) > thr) + (fabs(w1) > thr) + (fabs(w2) > thr)) : 0;
    if ((eig_values || eig_vectors) && !dim2) {
        // keepEigenValues / keepEigenVectors with sortEigen: 1 (r5): eigenvalues of the scatter matrix ascending (the exchange sort (0,1) (0,2)
        // (1,2), strict), serializeEigVec of the eigenvector matrix in that column order -- entry 3 k + j = component k of eigenvector j;
        // rank < 2: upstream's degenerate answer (zeros, identity).  Static indices only (no scratch).
        double e0 = w0, e1 = w1, e2 = w2;
        int o0 = 0, o1 = 1, o2 = 2;
        if (rank >= 2) {
            if (e1 < e0) { const double t = e0; e0 = e1; e1 = t; const int u = o0; o0 = o1; o1 = u; }
            if (e2 < e0) { const double t = e0; e0 = e2; e2 = t; const int u = o0; o0 = o2; o2 = u; }
            if (e2 < e1) { const double t = e1; e1 = e2; e2 = t; const int u = o1; o1 = o2; o2 = u; }
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int oj = j == 0 ? o0 : (j == 1 ? o1 : o2);
            const double wj = j == 0 ? e0 : (j == 1 ? e1 : e2);
            if (eig_values) eig_values[3 * i + j] = rank >= 2 ? (float)wj : 0.f;
            if (eig_vectors) {
#pragma unroll
                for (int kk = 0; kk < 3; ++kk) {
                    const double q = oj == 0 ? Q[kk][0] : (oj == 1 ? Q[kk][1] : Q[kk][2]);
                    eig_vectors[9 * i + 3 * kk + j] = rank >= 2 ? (float)q : (kk == j ? 1.f : 0.f);
                }
            }
        }
    }
    float nx = 1.f, ny = 0.f, nz = 0.f; // upstream's degenerate answer: eigenvectors = identity
    if (dim2) {
        // planar cloud (z == 0: the rotations with the z axis saw zero off-diagonals): the smaller eigenvector of the plane's pair;
        // upstream needs rank + 1 >= featDim - 1 = 2 there, i.e. rank >= 1
        const double wm2 = fmax(fabs(w0), fabs(w1));
        const double thr2 = 2.0 * 1.1920928955078125e-07 * wm2;
        const int rank2 = (wm2 > 0) ? ((fabs(w0) > thr2) + (fabs(w1) > thr2)) : 0;
        if (rank2 >= 1) {
            const int e2 = w1 < w0 ? 1 : 0;
            nx = (float)(e2 == 0 ? Q[0][0] : Q[0][1]);
            ny = (float)(e2 == 0 ? Q[1][0] : Q[1][1]);
            nz = 0.f;
        }
    } else if (rank >= 2) {
        int e = 0;
        double wm = w0;
        if (w1 < wm) { wm = w1; e = 1; }
        if (w2 < wm) { wm = w2; e = 2; }
        nx = (float)(e == 0 ? Q[0][0] : (e == 1 ? Q[0][1] : Q[0][2]));
        ny = (float)(e == 0 ? Q[1][0] : (e == 1 ? Q[1][1] : Q[1][2]));
        nz = (float)(e == 0 ? Q[2][0] : (e == 1 ? Q[2][1] : Q[2][2]));
    }
    normals3[3 * i] = nx; normals3[3 * i + 1] = ny; normals3[3 * i + 2] = nz;
}

// k <= 10 (the shipped SurfaceNormalDataPointsFilter{knn: 10}) and k <= 16 keep the neighbourhood in registers; ICPMI_NORMALS_REG=0: the generic walk
static void launch_normals(hipStream_t stream, const float4* map, const int* sidx, int64_t m, int k, float* normals3, float* densities, int dim2,
                           float* mean_dist = nullptr, float* eig_values = nullptr, float* eig_vectors = nullptr, const unsigned* list = nullptr)
{
    constexpr int reg = 1;
    const dim3 grid((int)((m + 127) / 128)), block(128);
    if (reg && k <= 10) hipLaunchKernelGGL(normals_kernel<10>, grid, block, 0, stream, map, sidx, m, k, normals3, densities, dim2, mean_dist, eig_values, eig_vectors, list);
    else if (reg && k <= 16) hipLaunchKernelGGL(normals_kernel<16>, grid, block, 0, stream, map, sidx, m, k, normals3, densities, dim2, mean_dist, eig_values, eig_vectors, list);
    else hipLaunchKernelGGL(normals_kernel<0>, grid, block, 0, stream, map, sidx, m, k, normals3, densities, dim2, mean_dist, eig_values, eig_vectors, list);
}

// ---- fused input filters (Mapper::applyInputFilters): one predicate pass for a run of DistanceLimit / BoundingBox filters ----
struct FilterPack { icpmi_point_filter f[ICPMI_MAX_POINT_FILTERS]; int n; };

__global__ __launch_bounds__(256) void filter_points_kernel(const float4* __restrict__ in, int64_t n, FilterPack fp, uint8_t* __restrict__ keep)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 p = in[i];
    bool ok = true;
    for (int k = 0; k < fp.n; ++k) {
        const icpmi_point_filter& f = fp.f[k];
        if (f.type == ICPMI_FILT_DISTANCE_LIMIT) {
            // sqrtf of the sum in the oracle's order (no contraction: -ffp-contract=off on both sides)
            const float v = f.i < 0 ? sqrtf(p.x * p.x + p.y * p.y + p.z * p.z) : fabsf(f.i == 0 ? p.x : (f.i == 1 ? p.y : p.z));
            const float ad = fabsf(f.f[0]);
            ok &= f.f[1] != 0.f ? v > ad : v < ad;
        } else {
            const bool inside = p.x > f.f[0] && p.x < f.f[3] && p.y > f.f[1] && p.y < f.f[4] && p.z > f.f[2] && p.z < f.f[5];
            ok &= f.i ? !inside : inside;
        }
    }
    keep[i] = ok ? 1 : 0;
}

// ---- voxel sub-sample (OctreeMapperModule / OctreeGridDataPointsFilter stand-in, samplingMethod 0) ----
// lattice anchored at the bounding-box minimum; voxel index floor((p - lo) / edge) per axis, 21 bits
// each; the representative of a voxel is its point of smallest original index (order independent:
// atomicMin), so the keep mask is a function of the input alone.
__device__ __forceinline__ unsigned fkey(float f) { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float fkey_inv(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

__global__ __launch_bounds__(256) void bbox_min_kernel(const float4* __restrict__ in, int64_t n, unsigned* __restrict__ lo_keys)
{
    __shared__ unsigned sh[3][4];
    unsigned kx = 0xffffffffu, ky = 0xffffffffu, kz = 0xffffffffu;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float4 p = in[i];
        kx = min(kx, fkey(p.x)); ky = min(ky, fkey(p.y)); kz = min(kz, fkey(p.z));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        kx = min(kx, (unsigned)__shfl_xor((int)kx, off, 64));
        ky = min(ky, (unsigned)__shfl_xor((int)ky, off, 64));
        kz = min(kz, (unsigned)__shfl_xor((int)kz, off, 64));
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { sh[0][w] = kx; sh[1][w] = ky; sh[2][w] = kz; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const unsigned v = min(min(sh[threadIdx.x][0], sh[threadIdx.x][1]), min(sh[threadIdx.x][2], sh[threadIdx.x][3]));
        atomicMin(&lo_keys[threadIdx.x], v);
    }
}

__device__ __forceinline__ unsigned long long voxel_key(const float4 p, const unsigned* __restrict__ lo_keys, float edge)
{
    const float lo[3] = {fkey_inv(lo_keys[0]), fkey_inv(lo_keys[1]), fkey_inv(lo_keys[2])};
    const float c[3] = {p.x, p.y, p.z};
    unsigned long long key = 0;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const float v = fminf(floorf((c[r] - lo[r]) / edge), 2097151.0f);
        key = key * 2097152ull + (unsigned long long)v;
    }
    return key;
}

__device__ __forceinline__ unsigned long long mix64(unsigned long long x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}

// MurmurHash3's 32-bit finaliser: a bijection of the 32-bit integers, so "smallest hash" names exactly one point
__device__ __forceinline__ unsigned fmix32(unsigned h)
{
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}

__global__ __launch_bounds__(256) void voxel_insert_kernel(const float4* __restrict__ in, int64_t n, const unsigned* __restrict__ lo_keys,
                                                           float edge, int method, unsigned long long* __restrict__ tkeys,
                                                           unsigned* __restrict__ tvals, unsigned long long mask,
                                                           unsigned* __restrict__ slot_of)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned long long key = voxel_key(in[i], lo_keys, edge);
    unsigned long long slot = mix64(key) & mask;
    for (;;) {
        const unsigned long long prev = atomicCAS(&tkeys[slot], ~0ull, key);
        if (prev == ~0ull || prev == key) break;
        slot = (slot + 1) & mask;
    }
    atomicMin(&tvals[slot], method ? fmix32((unsigned)i) : (unsigned)i);
    slot_of[i] = (unsigned)slot;
}

// flag[i] = 1 iff i represents its voxel (T = uint8_t for the host mask, unsigned for device compaction)
template <typename T>
__global__ __launch_bounds__(256) void voxel_keep_kernel(int64_t n, const unsigned* __restrict__ tvals, const unsigned* __restrict__ slot_of,
                                                         int method, T* __restrict__ keep)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    keep[i] = tvals[slot_of[i]] == (method ? fmix32((unsigned)i) : (unsigned)i) ? 1 : 0;
}

// ---- DynamicPointsMapperModule::inPlaceUpdateMap (DynamicPointsMapperModule.cpp:34-172) -------------------
// Beams = input points in the sensor frame as (elevation, azimuth); every in-range map point looks up
// its angularly nearest beam within 2 * beamHalfAngle (the reference builds a 2-D kd-tree per call,
// :75-78; here: a bucket grid of that cell size built by counting sort, 3 x 3 cells per query, ties to
// the smallest beam index) and updates its probability of being dynamic (:97-148).
// asin / atan2 go through double and are rounded once (shared with the oracle: libm and the device
// library then agree bit for bit); everything else is the reference's float arithmetic, with the
// sub-expressions it writes with a double literal (`1.`) evaluated in double.
struct DynGrid { float cell; int ne, na; float r2; };   // r2 = (2 beamHalfAngle)^2: the search radius, DYN_RINGS cells wide

__device__ __forceinline__ void to_spherical(float x, float y, float z, float& radius, float& elev, float& azim)
{
    radius = sqrtf(x * x + y * y + z * z);
    elev = (float)asin((double)(z / radius));
    azim = (float)atan2((double)y, (double)x);
}

__device__ __forceinline__ int dyn_ecell(const DynGrid& g, float e)
{
    const int v = (int)floorf((e + 1.5707963267949f) / g.cell);
    return v < 0 ? 0 : (v > g.ne - 1 ? g.ne - 1 : v);
}
__device__ __forceinline__ int dyn_acell(const DynGrid& g, float a)
{
    const int v = (int)floorf((a + 3.14159265358979f) / g.cell);
    return v < 0 ? 0 : (v > g.na - 1 ? g.na - 1 : v);
}

// pass 1: beams to the sensor frame + angles + cell counts
__global__ __launch_bounds__(256) void dyn_beams_kernel(const float4* __restrict__ in, int64_t n, Mat16 M, DynGrid g,
                                                        float4* __restrict__ beam_xyzn, float2* __restrict__ beam_ang,
                                                        unsigned* __restrict__ keys, unsigned* __restrict__ count)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool valid = i < n;
    const float4 p = in[valid ? i : 0];
    const float3 o = xf_point(M.v, p.x, p.y, p.z, p.w);
    float radius, elev, azim;
    to_spherical(o.x, o.y, o.z, radius, elev, azim);
    const unsigned key = (unsigned)(dyn_ecell(g, elev) * g.na + dyn_acell(g, azim));
    // a lidar scan is ordered along its beams: neighbours in the array fall into the same bucket, and same-address device atomics
    // serialise -- one atomic per run of equal keys in the wave (r5; map_build.hip's counting sorts do the same)
    const WaveRun r = wave_run(key, valid);
    if (r.head) atomicAdd(&count[key], (unsigned)r.len);
    if (!valid) return;
    beam_xyzn[i] = make_float4(o.x, o.y, o.z, radius);
    beam_ang[i] = make_float2(elev, azim);
    keys[i] = key;
}

// pass 2: counting-sort scatter; a bucket entry is ONE 16-byte record {elevation, azimuth, beam index} (r5: the index used to sit in a second
// array -- a dependent load per accepted candidate); the buckets of one elevation row are contiguous, so a row of the 3 x 3 block is one run
__global__ __launch_bounds__(256) void dyn_scatter_kernel(int64_t n, const unsigned* __restrict__ keys, unsigned* __restrict__ cursor /* = starts + 1 */,
                                                          const float2* __restrict__ beam_ang, float4* __restrict__ sorted_rec)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool valid = i < n;
    const unsigned key = keys[valid ? i : 0];
    const float2 a = beam_ang[valid ? i : 0];
    const WaveRun r = wave_run(key, valid);
    unsigned base = 0;
    if (r.head) base = atomicAdd(&cursor[key], (unsigned)r.len);
    base = (unsigned)__shfl((int)base, r.head_lane, 64);
    if (!valid) return;
    sorted_rec[base + (unsigned)r.rank] = make_float4(a.x, a.y, __uint_as_float((unsigned)i), 0.f);
}

#ifndef DYN_INFLIGHT
#define DYN_INFLIGHT 4
#endif
#ifndef DYN_RINGS
#define DYN_RINGS 2   // buckets per search radius (1: 3 x 3 block of one-radius buckets, the layout until r4)
#endif
struct DynPrm { float threshold_dynamic, alpha, beta, beam_half_angle, epsilon_a, epsilon_d, sensor_max_range; };

__global__ __launch_bounds__(256) void dyn_update_kernel(const float4* __restrict__ map, const float* __restrict__ normals3, int64_t m,
                                                         Mat16 M, DynGrid g, DynPrm prm,
                                                         const float4* __restrict__ beam_xyzn, const float4* __restrict__ sorted_rec,
                                                         const unsigned* __restrict__ start, float* __restrict__ prob)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const float* T = M.v;
    const float eps = 0.0001f;
    const float4 mpt = map[i];
    const float3 mp = xf_point(T, mpt.x, mpt.y, mpt.z, mpt.w);
    const float mapNorm = sqrtf(mp.x * mp.x + mp.y * mp.y + mp.z * mp.z);
    if (!(mapNorm < prm.sensor_max_range)) return; // range cull (:60-69)
    float radius, qe, qa;
    to_spherical(mp.x, mp.y, mp.z, radius, qe, qa);
    const int ce = dyn_ecell(g, qe), ca = dyn_acell(g, qa);
    const float r2 = g.r2;
    float bd = INFINITY;
    int best = -1;
    // The nearest beam within 2 * beamHalfAngle = DYN_RINGS bucket edges: it lies in the (2 R + 1)^2 block of buckets around the point's
    // own.  The buckets (e, a - R .. a + R) of one elevation row are consecutive keys: the bounds of a row's buckets are 2 R + 2 consecutive
    // words, and all of them are requested together before anything depends on them (r4 fetched the two bounds of a bucket when it got
    // there: nine dependent round trips before the ninth bucket's records).  Own bucket first, then ring by ring: where the beams are
    // dense the nearest one is a fraction of a bucket away, and a bucket whose nearest edge is farther than the best so far cannot hold a
    // closer beam (nor an equally close one: the test is strict and leaves a margin for the rounding of the cell assignment).  The winner
    // is the minimum of (angular distance, beam index): independent of the visiting order.
    // r5: buckets of HALF the radius (R = 2, 5 x 5 block).  With one-radius buckets a point paid for every record of its own bucket before
    // the pruning could start -- 150 records where a surface is seen at a grazing angle (the synthetic scenes; a spinning lidar's own
    // returns are uniform in angle); a quarter bucket first, and the ring behind it mostly pruned: search 136 -> see DESIGN 13.2b.
    // (Scanning whole rows without the per-bucket test -- fewer branches -- looked at 3 - 5 x the records and was slower: 253 vs 225 us.)
    constexpr int R = DYN_RINGS, W = 2 * R + 1;
    const float elo = (float)ce * g.cell - 1.5707963267949f, alo = (float)ca * g.cell - 3.14159265358979f;
    float gapE[W], gapA[W];
#pragma unroll
    for (int d = 0; d < W; ++d) {
        // distance from the query to the nearest edge of the bucket d - R cells away (0 for its own)
        gapE[d] = d < R ? (qe - elo) + (float)(R - 1 - d) * g.cell : (d == R ? 0.f : (elo + g.cell - qe) + (float)(d - R - 1) * g.cell);
        gapA[d] = d < R ? (qa - alo) + (float)(R - 1 - d) * g.cell : (d == R ? 0.f : (alo + g.cell - qa) + (float)(d - R - 1) * g.cell);
    }
    unsigned sb[W][W + 1];
#pragma unroll
    for (int de = 0; de < W; ++de) {
        const int e = ce + de - R;
        const bool row = e >= 0 && e < g.ne;
#pragma unroll
        for (int x = 0; x <= W; ++x) {
            int a = ca - R + x;                       // bound x = start of bucket (e, ca - R + x)
            a = a < 0 ? 0 : (a > g.na ? g.na : a);    // (a == na: the start of the next row's first bucket = the end of this row's last)
            sb[de][x] = start[row ? (unsigned)(e * g.na + a) : 0u];
        }
    }
#pragma unroll
    for (int ring = 0; ring <= R; ++ring) {
#pragma unroll
        for (int de = 0; de < W; ++de) {
#pragma unroll
            for (int da = 0; da < W; ++da) {
                const int re = de > R ? de - R : R - de, ra = da > R ? da - R : R - da;
                if ((re > ra ? re : ra) != ring) continue;   // (compile time)
                const int e = ce + de - R, a = ca + da - R;
                if (e < 0 || e >= g.ne || a < 0 || a >= g.na) continue;
                const float ge = fmaxf(gapE[de] - 1e-5f, 0.f), ga = fmaxf(gapA[da] - 1e-5f, 0.f);
                const float dmin = ge * ge + ga * ga;
                if (dmin > r2 || dmin > bd) continue;
                // DYN_INFLIGHT records requested together (r5: one per trip made every record a full memory round trip of the wave's slowest lane)
                const unsigned jend = sb[de][da + 1];
                for (unsigned j = sb[de][da]; j < jend; j += DYN_INFLIGHT) {
                    float4 rec[DYN_INFLIGHT];
#pragma unroll
                    for (int u = 0; u < DYN_INFLIGHT; ++u) rec[u] = sorted_rec[j + u < jend ? j + u : jend - 1];
#pragma unroll
                    for (int u = 0; u < DYN_INFLIGHT; ++u) {
                        const float d0 = qe - rec[u].x, d1 = qa - rec[u].y;
                        const float d = d0 * d0 + d1 * d1;
                        if (d <= r2 && d <= bd) { // ties on the angular distance go to the smallest beam index (the bucket order is arbitrary; a clamped repeat changes nothing)
                            const int b = (int)__float_as_uint(rec[u].z);
                            if (d < bd || b < best) { bd = d; best = b; }
                        }
                    }
                }
            }
        }
    }
    if (best < 0) return; // no beam within 2 * beamHalfAngle

    const float4 ip = beam_xyzn[best];
    const float inputNorm = ip.w;
    const float dx = ip.x - mp.x, dy = ip.y - mp.y, dz = ip.z - mp.z;
    const float delta = sqrtf(dx * dx + dy * dy + dz * dz);
    const float d_max = prm.epsilon_a * inputNorm;
    const float n0 = normals3[3 * i], n1 = normals3[3 * i + 1], n2 = normals3[3 * i + 2];
    const float nx = fmaf(T[8], n2, fmaf(T[4], n1, T[0] * n0));
    const float ny = fmaf(T[9], n2, fmaf(T[5], n1, T[1] * n0));
    const float nz = fmaf(T[10], n2, fmaf(T[6], n1, T[2] * n0));
    const float ndot = (nx * mp.x + ny * mp.y + nz * mp.z) / mapNorm;

    const float w_v = (float)(eps + (1. - eps) * fabs((double)ndot));
    const float w_d1 = (float)(eps + (1. - eps) * (1. - sqrtf(bd) / (2 * prm.beam_half_angle)));
    const float offset = delta - prm.epsilon_d;
    float w_d2 = 1.f;
    if (delta < prm.epsilon_d || mapNorm > inputNorm) w_d2 = eps;
    else if (offset < d_max) w_d2 = eps + (1 - eps) * offset / d_max;
    float w_p2 = eps;
    if (delta < prm.epsilon_d) w_p2 = 1.f;
    else if (offset < d_max) w_p2 = (float)(eps + (1. - eps) * (1. - offset / d_max));

    if ((inputNorm + prm.epsilon_d + d_max) >= mapNorm) {
        const float lastDyn = prob[i];
        const float c1 = 1 - (w_v * w_d1);
        const float c2 = w_v * w_d1;
        float probDynamic, probStatic;
        if (lastDyn < prm.threshold_dynamic) {
            probDynamic = c1 * lastDyn + c2 * w_d2 * ((1 - prm.alpha) * (1 - lastDyn) + prm.beta * lastDyn);
            probStatic = c1 * (1 - lastDyn) + c2 * w_p2 * (prm.alpha * (1 - lastDyn) + (1 - prm.beta) * lastDyn);
        } else { // latched: once dynamic, always dynamic
            probDynamic = 1 - eps;
            probStatic = eps;
        }
        prob[i] = probDynamic / (probDynamic + probStatic);
    }
}

__global__ __launch_bounds__(256) void keep_flag_kernel(const float* __restrict__ d2, int64_t n, double lim, unsigned* __restrict__ flag)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    flag[i] = d2[i] >= lim ? 1u : 0u;
}

// d^2 of the k-th neighbour of every searched point (the row's last entry; +inf: fewer than k points in the cloud)
__global__ __launch_bounds__(256) void kth_d2_kernel(const float* __restrict__ d2, int64_t m, int k, const unsigned* __restrict__ list, float* __restrict__ dk)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= m) return;
    const int64_t i = list ? (int64_t)list[t] : t;
    dk[i] = d2[(size_t)k * i + (k - 1)];
}

__global__ __launch_bounds__(256) void flag_to_keep_kernel(const unsigned* __restrict__ flag, int64_t n, uint8_t* __restrict__ keep)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) keep[i] = flag[i] ? 1 : 0;
}

// stable compaction by flags: kept input i goes to slot base + pos[i] (pos = exclusive scan of the flags)
__global__ __launch_bounds__(256) void append_flagged_kernel(const float4* __restrict__ in, const float* __restrict__ in_n3, int64_t n,
                                                             const unsigned* __restrict__ flag, const unsigned* __restrict__ pos, int64_t base,
                                                             float4* __restrict__ raw, float* __restrict__ raw_n3)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || !flag[i]) return;
    const int64_t o = base + pos[i];
    raw[o] = in[i];
    if (raw_n3) {
        raw_n3[3 * o] = in_n3 ? in_n3[3 * i] : 0.f;
        raw_n3[3 * o + 1] = in_n3 ? in_n3[3 * i + 1] : 0.f;
        raw_n3[3 * o + 2] = in_n3 ? in_n3[3 * i + 2] : 0.f;
    }
}

// stable compaction: kept input i goes to slot base + pos[i] (pos = exclusive scan of the flags)
__global__ __launch_bounds__(256) void append_kept_kernel(const float4* __restrict__ in, const float* __restrict__ in_n3, int64_t n,
                                                          const float* __restrict__ d2, double lim, const unsigned* __restrict__ pos,
                                                          int64_t base, float4* __restrict__ raw, float* __restrict__ raw_n3)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || !((double)d2[i] >= lim)) return;
    const int64_t o = base + pos[i];
    raw[o] = in[i];
    if (raw_n3) {
        raw_n3[3 * o] = in_n3 ? in_n3[3 * i] : 0.f;
        raw_n3[3 * o + 1] = in_n3 ? in_n3[3 * i + 1] : 0.f;
        raw_n3[3 * o + 2] = in_n3 ? in_n3[3 * i + 2] : 0.f;
    }
}

} // namespace

// helper: a private handle on the same device/stream used to index an arbitrary cloud without
// disturbing the ICP map of the caller's handle
// The private handle lives as long as its owner (created on first use, destroyed by icpmi_destroy): its buffers are
// reused from call to call instead of ~20 hipMalloc / hipFree pairs per operator call.
struct TempCtx {
    icpmi_handle h = nullptr;
};

// private handles enqueue on the owner's stream: what they read was produced there and what they produce is consumed there,
// so stream order replaces the hipStreamSynchronize pairs an own stream needs (r2: two per surface-normal step of a map update)
static void share_stream(icpmi_ctx* c, icpmi_ctx* t)
{
    if (t->stream == c->stream) { (void)zero_state_if_pending(t); return; } // (a handle that keeps its own stream still starts from a cleared state: ADVICE r5)
    if (t->stream) (void)hipStreamSynchronize(t->stream);
    if (t->own_stream && t->stream) stream_release(t->stream);
    t->stream = c->stream; t->own_stream = false;
    drop_loop_graphs(t);
    (void)zero_state_if_pending(t); // (on the owner's stream; an error surfaces at the handle's next call)
}

static icpmi_status make_temp(icpmi_ctx* c, TempCtx& t)
{
    if (!c->temp) {
        icpmi_config cfg = c->cfg;
        icpmi_status s = create_handle(&cfg, &c->temp);
        if (s != ICPMI_OK) { c->last_error = icpmi_last_error(nullptr); c->temp = nullptr; return s; }
    }
    t.h = c->temp;
    share_stream(c, t.h);
    t.h->cfg = c->cfg;
    t.h->keep_raw = false;
    t.h->no_centre = true; // PointDistanceMapperModule.cpp:33 / SurfaceNormalDataPointsFilter build their kd-tree on the raw cloud
    t.h->single_level = false;
    return ICPMI_OK;
}

namespace { icpmi_status chain_point_distance_flags(icpmi_ctx* c, icpmi_ctx* ic, const float4* d_scan, int64_t n, float min_dist, unsigned* d_flag); }

// the index PointDistanceMapperModule searches: the resident map in its own frame (common.h: temp_raw)
// `reach`: the radius the caller will search with (minDistNewPoint): the pyramid of this index only needs the levels whose
// blocks reach that far -- level 0 alone for the usual 0.1 m against ~0.5 m cells, instead of the three levels the
// registration radius asks for
static icpmi_status raw_index(icpmi_ctx* c, icpmi_ctx** out, float reach)
{
    if (!c->temp_raw) {
        icpmi_config cfg = c->cfg;
        icpmi_status s = create_handle(&cfg, &c->temp_raw);
        if (s != ICPMI_OK) { c->last_error = icpmi_last_error(nullptr); c->temp_raw = nullptr; return s; }
        c->temp_raw_version = 0;
    }
    icpmi_ctx* t = c->temp_raw;
    share_stream(c, t);
    const float want = reach > 1e-3f ? reach : 1e-3f;
    // r4: no second index at all where the registration index carries the raw twin of its level 0 (map_build.hip: d_raw0 -- the resident
    // points, raw coordinates, in level-0 sorted order) and one 3 x 3 x 3 block of level 0 holds the search ball: the private handle
    // becomes a VIEW -- the owner's cell starts, the twin as the point array, the grid laid over the raw bounding box, centroid 0.  The
    // search is exact whatever grid it walks, so ids and d^2 are those of an index built on the raw cloud (tests: every map-update and
    // merge test compares the keep masks with the oracle's search of the raw map).  A point's cell was assigned from its centred
    // coordinates; a query's cell comes from raw ones: the two agree up to rounding far below the grid's slack.
    {
        const GridParams& g0 = c->levels.g[0];
        if (c->ins_ready && c->d_raw0 && !c->no_centre && c->m > 0 && c->m == c->m_raw && (g0.cell - 2.f * g0.slack) > want * 1.001f) {
            if (c->raw_view_version != c->map_version || t->m != c->m) {
                float maxabs = 0.f;
                for (int r = 0; r < 3; ++r) maxabs = fmaxf(maxabs, fmaxf(fabsf(c->lo_raw[r]), fabsf(c->hi_raw[r])));
                GridParams g = g0;
                g.ox = c->lo_raw[0]; g.oy = c->lo_raw[1]; g.oz = c->lo_raw[2];
                g.slack = g.cell * 1e-3f + maxabs * 2e-6f;
                if (!((g.cell - 2.f * g.slack) > want * 1.001f)) goto no_view; // (a map far from the origin: its raw coordinates round coarser)
                t->cfg = c->cfg; t->cfg.max_dist = want; t->keep_raw = false; t->no_centre = true; t->single_level = true; t->is_raw_index = false;
                t->levels = GridLevels{};
                t->levels.nlev = 1; t->levels.g[0] = g; t->levels.pts[0] = c->d_raw0; t->levels.cs[0] = c->d_cell_start; t->levels.pos0[0] = nullptr;
                t->grid = g; t->m = c->m; t->mean[0] = t->mean[1] = t->mean[2] = 0.f; t->has_normals = false; t->ins_ready = false;
                t->qsorted_n = -1; t->qsorted_src = nullptr;
                drop_loop_graphs(t);
                { const icpmi_status us = upload_level_table(t); if (us != ICPMI_OK) { c->last_error = t->last_error; return us; } }
                c->raw_view_version = c->map_version;
                c->temp_raw_version = 0; // (the handle's own arrays no longer describe anything)
            }
            ++c->raw_view_count;
            *out = t;
            return ICPMI_OK;
        }
    }
no_view:
    c->raw_view_version = 0;
    const float built_reach = t->cfg.max_dist;
    t->cfg = c->cfg; t->keep_raw = false; t->no_centre = true; t->single_level = false; t->is_raw_index = true;
    const bool current = c->temp_raw_version == c->map_version && t->m == c->m_raw && t->m > 0;
    t->cfg.max_dist = current && built_reach >= want ? built_reach : want;
    if (!current || built_reach < want) {
        if (t->stream != c->stream) HIP_TRY(c, hipStreamSynchronize(c->stream)); // the resident copy was produced on the caller's stream
        if (c->m_raw <= 0) { c->last_error = "raw_index: no resident map"; return ICPMI_ERR_INVALID_ARG; }
        // r4: the resident map grew by an append since this index was built (same raw_epoch: nobody rewrote d_raw) -- the delta is
        // merged into the cell-sorted arrays (map_build.hip: map_insert; raw coordinates, centroid 0: nothing is even recentred)
        const bool grown = t->m > 0 && t->m < c->m_raw && c->temp_raw_epoch == c->raw_epoch && c->temp_raw_m == t->m && built_reach >= want;
        icpmi_status s = map_build(t, c->d_raw, c->m_raw, nullptr, grown ? t->m : 0);
        if (s != ICPMI_OK) { c->last_error = t->last_error; return s; }
        c->temp_raw_version = c->map_version; c->temp_raw_epoch = c->raw_epoch; c->temp_raw_m = c->m_raw;
    }
    *out = t;
    return ICPMI_OK;
}

// Transformation::checkParameters: rotation part must be (close to) orthonormal with det +1
static icpmi_status check_rigid(icpmi_ctx* c, const float T[16])
{
    const double det = (double)T[0] * ((double)T[5] * T[10] - (double)T[9] * T[6]) - (double)T[4] * ((double)T[1] * T[10] - (double)T[9] * T[2]) +
                       (double)T[8] * ((double)T[1] * T[6] - (double)T[5] * T[2]);
    if (fabs(1.0 - det) > 1e-3) {
        c->last_error = "TransformationError: RigidTransformation: rotation part is not orthonormal (|1 - det| > 1e-3)";
        return ICPMI_ERR_INVALID_ARG;
    }
    return ICPMI_OK;
}

// RigidTransformation::compute on device buffers, on the handle's stream (the 4x4 goes through a small resident buffer)
icpmi_status ops_transform_dev(icpmi_ctx* c, const float T[16], const float4* d_in, int64_t n, float4* d_out)
{
    icpmi_status s = check_rigid(c, T);
    if (s != ICPMI_OK || n == 0) return s;
    hipLaunchKernelGGL(transform_kernel, dim3((int)((n + 255) / 256)), dim3(256), 0, c->stream, d_in, n, mat16(T), d_out);
    HIP_TRY(c, hipGetLastError());
    return ICPMI_OK;
}

icpmi_status ops_transform(icpmi_ctx* c, const float T[16], const float* in4, int64_t n, float* out4, const float* in_n3,
                           float* out_n3)
{
    icpmi_status cs = check_rigid(c, T);
    if (cs != ICPMI_OK) return cs;
    if (n == 0) return ICPMI_OK;
    DevBuf<float> d_n, d_no;
    DevBuf<float4> d_in, d_out;
    HIP_TRY(c, d_in.alloc((size_t)n));
    HIP_TRY(c, d_out.alloc((size_t)n));
    hipError_t e = hipMemcpyAsync(d_in, in4, (size_t)n * sizeof(float4), hipMemcpyHostToDevice, c->stream);
    const int blocks = (int)((n + 255) / 256);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(transform_kernel, dim3(blocks), dim3(256), 0, c->stream, d_in, n, mat16(T), d_out);
        e = hipMemcpyAsync(out4, d_out, (size_t)n * sizeof(float4), hipMemcpyDeviceToHost, c->stream);
    }
    if (e == hipSuccess && in_n3 && out_n3) {
        e = d_n.alloc((size_t)n * 3);
        if (e == hipSuccess) e = d_no.alloc((size_t)n * 3);
        if (e == hipSuccess) e = hipMemcpyAsync(d_n, in_n3, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(rotate3_kernel, dim3(blocks), dim3(256), 0, c->stream, d_n, n, mat16(T), d_no);
            e = hipMemcpyAsync(out_n3, d_no, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream);
        }
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    HIP_TRY(c, e);
    return ICPMI_OK;
}

icpmi_status ops_bin_cells(icpmi_ctx* c, const float* pts4, int64_t n, float cell_size, int32_t* ijk3)
{
    if (n == 0) return ICPMI_OK;
    DevBuf<float4> d_in; DevBuf<int> d_o;
    HIP_TRY(c, d_in.alloc((size_t)n));
    HIP_TRY(c, d_o.alloc((size_t)n * 3));
    hipError_t e = hipMemcpyAsync(d_in, pts4, (size_t)n * sizeof(float4), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(bin_kernel, dim3((int)((n + 255) / 256)), dim3(256), 0, c->stream, d_in, n, cell_size, d_o);
        e = hipMemcpyAsync(ijk3, d_o, (size_t)n * 3 * sizeof(int), hipMemcpyDeviceToHost, c->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    HIP_TRY(c, e);
    return ICPMI_OK;
}

// shared by surface normals and point-distance: index `cloud4` in a temp handle, kNN `q4` against it
static icpmi_status temp_knn(icpmi_ctx* c, TempCtx& t, const float* cloud4, int64_t m, const float* q4, int64_t n, int k,
                             int allow_self, bool queries_are_cloud)
{
    icpmi_ctx* tc = t.h;
    const bool self = queries_are_cloud && allow_self; // SurfaceNormalDataPointsFilter: the cloud against itself -- the sparse block grid (selfgrid.hip)
    if (self) {
        if (m <= 0 || !cloud4) { c->last_error = "set_map: empty cloud"; return ICPMI_ERR_INVALID_ARG; }
        const size_t cnt = (size_t)m * k + 1;
        if (ensure_cap(tc, &tc->d_stage_in, &tc->cap_stage_in, (size_t)m + 1) != ICPMI_OK || ensure_cap(tc, &tc->d_sidx, &tc->cap_sidx, cnt) != ICPMI_OK ||
            ensure_cap(tc, &tc->d_d2, &tc->cap_d2, cnt) != ICPMI_OK) { c->last_error = tc->last_error; return ICPMI_ERR_HIP; }
        HIP_TRY(c, hipMemcpyAsync(tc->d_stage_in, cloud4, (size_t)m * sizeof(float4), hipMemcpyHostToDevice, tc->stream));
        const icpmi_status gs = selfgrid_knn(tc, tc->d_stage_in, m, k, tc->d_sidx, tc->d_d2);
        if (gs != ICPMI_OK) c->last_error = tc->last_error;
        return gs;
    }
    tc->single_level = false;
    int32_t acc = 0;
    icpmi_status s = icpmi_set_map(t.h, cloud4, m, nullptr, &acc);
    if (s != ICPMI_OK) { c->last_error = tc->last_error; return s; }
    // queries: stage + centre on the temp map's mean (the index lives in the centred frame)
    // (the cloud itself is already there: icpmi_set_map staged it in d_stage_in -- no second upload, and no re-sizing
    // of that buffer, which would drop it)
    if (!queries_are_cloud) {
        if (ensure_cap(tc, &tc->d_stage_in, &tc->cap_stage_in, (size_t)n + 1) != ICPMI_OK) { c->last_error = tc->last_error; return ICPMI_ERR_HIP; }
        HIP_TRY(c, hipMemcpyAsync(tc->d_stage_in, q4, (size_t)n * sizeof(float4), hipMemcpyHostToDevice, tc->stream));
    }
    s = loop_prepare_reading(tc, tc->d_stage_in, n, nullptr);
    if (s != ICPMI_OK) { c->last_error = tc->last_error; return s; }
    LoopCfg lc = make_loop_cfg(tc, 1);
    lc.k = k; lc.max_dist = INFINITY; lc.maxr2 = INFINITY; lc.ring_max = 6; lc.inv1e = 1.f; lc.err2 = 1.f; // (a filter's search: exact whatever the matcher's epsilon)
    const size_t cnt = (size_t)n * k + 1;
    if (ensure_cap(tc, &tc->d_sidx, &tc->cap_sidx, cnt) != ICPMI_OK || ensure_cap(tc, &tc->d_d2, &tc->cap_d2, cnt) != ICPMI_OK ||
        ensure_cap(tc, &tc->d_hard, &tc->cap_hard, (size_t)n + 1) != ICPMI_OK) { c->last_error = tc->last_error; return ICPMI_ERR_HIP; }
    HIP_TRY(c, hipMemsetAsync(tc->d_state, 0, sizeof(IcpState), tc->stream));
    tc->nn_hist0 = nullptr; tc->nn_iter_hint = 0; tc->nn_match_pt = nullptr;
    s = nn_launch_k(tc, tc->d_reading, n, nullptr, lc, allow_self, tc->d_sidx, tc->d_d2, tc->d_state);
    if (s != ICPMI_OK) { c->last_error = tc->last_error; return s; }
    return ICPMI_OK;
}

icpmi_status ops_surface_normals(icpmi_ctx* c, const float* pts4, int64_t m, int knn, float* normals3, float* densities, int32_t* matched_ids,
                                 float* mean_dist, float* eig_values, float* eig_vectors)
{
    if (m == 0) return ICPMI_OK;
    if (knn < 1 || knn > ICPMI_MAX_K) { c->last_error = "surface_normals: knn must be in [1, 32]"; return ICPMI_ERR_INVALID_ARG; }
    TempCtx t;
    icpmi_status s = make_temp(c, t);
    if (s != ICPMI_OK) return s;
    s = temp_knn(c, t, pts4, m, nullptr, m, knn, 1, true);
    if (s != ICPMI_OK) return s;
    icpmi_ctx* tc = t.h;
    if ((eig_values || eig_vectors) && c->cfg.is_2d) { c->last_error = "surface_normals: keepEigenValues / keepEigenVectors are not served for planar clouds"; return ICPMI_ERR_UNSUPPORTED; }
    DevBuf<float> d_n, d_dens, d_md, d_ev, d_evec;
    DevBuf<int> d_ids;
    HIP_TRY(c, d_n.alloc((size_t)m * 3));
    if (densities) HIP_TRY(c, d_dens.alloc((size_t)m));
    if (mean_dist) HIP_TRY(c, d_md.alloc((size_t)m));
    if (eig_values) HIP_TRY(c, d_ev.alloc((size_t)m * 3));
    if (eig_vectors) HIP_TRY(c, d_evec.alloc((size_t)m * 9));
    launch_normals(tc->stream, tc->d_map_sorted, tc->d_sidx, m, knn, d_n, densities ? d_dens.p : (float*)nullptr, c->cfg.is_2d,
                   mean_dist ? d_md.p : (float*)nullptr, eig_values ? d_ev.p : (float*)nullptr, eig_vectors ? d_evec.p : (float*)nullptr);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && eig_values) e = hipMemcpyAsync(eig_values, d_ev, (size_t)m * 3 * sizeof(float), hipMemcpyDeviceToHost, tc->stream);
    if (e == hipSuccess && eig_vectors) e = hipMemcpyAsync(eig_vectors, d_evec, (size_t)m * 9 * sizeof(float), hipMemcpyDeviceToHost, tc->stream);
    if (e == hipSuccess && matched_ids) { // keepMatchedIds: sorted positions -> the caller's indices
        e = d_ids.alloc((size_t)m * knn);
        if (e == hipSuccess && nn_ids_to_original(tc, tc->d_sidx, m * knn, d_ids) != ICPMI_OK) { c->last_error = tc->last_error; return ICPMI_ERR_HIP; }
        if (e == hipSuccess) e = hipMemcpyAsync(matched_ids, d_ids, (size_t)m * knn * sizeof(int), hipMemcpyDeviceToHost, tc->stream);
    }
    if (e == hipSuccess && mean_dist) e = hipMemcpyAsync(mean_dist, d_md, (size_t)m * sizeof(float), hipMemcpyDeviceToHost, tc->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(normals3, d_n, (size_t)m * 3 * sizeof(float), hipMemcpyDeviceToHost, tc->stream);
    if (e == hipSuccess && densities) e = hipMemcpyAsync(densities, d_dens, (size_t)m * sizeof(float), hipMemcpyDeviceToHost, tc->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(tc->stream);
    HIP_TRY(c, e);
    return ICPMI_OK;
}

icpmi_status ops_point_distance_keep(icpmi_ctx* c, const float* map4, int64_t m, const float* in4, int64_t n, float min_dist,
                                     uint8_t* keep)
{
    if (n == 0) return ICPMI_OK;
    if (m == 0) { memset(keep, 1, (size_t)n); return ICPMI_OK; } // no neighbour: d2 = +inf >= lim
    TempCtx t;
    icpmi_status s = make_temp(c, t);
    if (s != ICPMI_OK) return s;
    s = temp_knn(c, t, map4, m, in4, n, 1, 0, false);
    if (s != ICPMI_OK) return s;
    icpmi_ctx* tc = t.h;
    DevBuf<uint8_t> d_keep;
    HIP_TRY(c, d_keep.alloc((size_t)n));
    const double lim = pd_limit(min_dist);
    hipLaunchKernelGGL(keep_kernel, dim3((int)((n + 255) / 256)), dim3(256), 0, tc->stream, tc->d_d2, n, lim, d_keep);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(keep, d_keep, (size_t)n, hipMemcpyDeviceToHost, tc->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(tc->stream);
    HIP_TRY(c, e);
    return ICPMI_OK;
}

// decimation flags of a DEVICE cloud (T = uint8_t or unsigned), stream-ordered on c->stream
template <typename T>
static icpmi_status voxel_flags_dev(icpmi_ctx* c, const float4* d_in, int64_t n, float edge, int method, T* d_keep)
{
    if (n > 0xfffffff0ll) { c->last_error = "voxel_keep: too many points"; return ICPMI_ERR_UNSUPPORTED; }
    unsigned long long cap = 1024;
    while (cap < (unsigned long long)n * 2ull) cap <<= 1;
    unsigned long long* d_keys = scratch_get<unsigned long long>(c, 0, (size_t)cap);
    unsigned* d_vals = scratch_get<unsigned>(c, 1, (size_t)cap);
    unsigned* d_slot = scratch_get<unsigned>(c, 2, (size_t)n);
    unsigned* d_lo = scratch_get<unsigned>(c, 3, 4);
    if (!d_keys || !d_vals || !d_slot || !d_lo) return ICPMI_ERR_HIP;
    HIP_TRY(c, hipMemsetAsync(d_keys, 0xff, (size_t)cap * sizeof(unsigned long long), c->stream));
    HIP_TRY(c, hipMemsetAsync(d_vals, 0xff, (size_t)cap * sizeof(unsigned), c->stream));
    HIP_TRY(c, hipMemsetAsync(d_lo, 0xff, 4 * sizeof(unsigned), c->stream));
    const int blocks = (int)((n + 255) / 256);
    const int rb = blocks < 1024 ? blocks : 1024;
    hipLaunchKernelGGL(bbox_min_kernel, dim3(rb), dim3(256), 0, c->stream, d_in, n, d_lo);
    hipLaunchKernelGGL(voxel_insert_kernel, dim3(blocks), dim3(256), 0, c->stream, d_in, n, d_lo, edge, method, d_keys, d_vals, cap - 1, d_slot);
    hipLaunchKernelGGL(voxel_keep_kernel<T>, dim3(blocks), dim3(256), 0, c->stream, n, d_vals, d_slot, method, d_keep);
    HIP_TRY(c, hipGetLastError());
    return ICPMI_OK;
}

icpmi_status ops_filter_points(icpmi_ctx* c, const float* in4, int64_t n, const icpmi_point_filter* filters, int n_filters, uint8_t* keep)
{
    if (n == 0) return ICPMI_OK;
    FilterPack fp;
    fp.n = n_filters;
    for (int k = 0; k < n_filters; ++k) fp.f[k] = filters[k];
    if (ensure_cap(c, &c->d_stage_in, &c->cap_stage_in, (size_t)n + 1) != ICPMI_OK) return ICPMI_ERR_HIP;
    uint8_t* d_keep = scratch_get<uint8_t>(c, 9, (size_t)n);
    if (!d_keep) return ICPMI_ERR_HIP;
    HIP_TRY(c, hipMemcpyAsync(c->d_stage_in, in4, (size_t)n * sizeof(float4), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(filter_points_kernel, dim3((int)((n + 255) / 256)), dim3(256), 0, c->stream, c->d_stage_in, n, fp, d_keep);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(keep, d_keep, (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return ICPMI_OK;
}

icpmi_status ops_voxel_keep_first(icpmi_ctx* c, const float* in4, int64_t n, float edge, int method, uint8_t* keep)
{
    if (n == 0) return ICPMI_OK;
    DevBuf<float4> d_in; DevBuf<uint8_t> d_keep;
    HIP_TRY(c, d_in.alloc((size_t)n));
    HIP_TRY(c, d_keep.alloc((size_t)n));
    HIP_TRY(c, hipMemcpyAsync(d_in, in4, (size_t)n * sizeof(float4), hipMemcpyHostToDevice, c->stream));
    icpmi_status s = voxel_flags_dev<uint8_t>(c, d_in, n, edge, method, d_keep);
    if (s != ICPMI_OK) return s;
    HIP_TRY(c, hipMemcpyAsync(keep, d_keep, (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return ICPMI_OK;
}

// DynamicPointsMapperModule::inPlaceUpdateMap on DEVICE arrays (T = pose^-1 as a kernel argument); d_prob updated in place.  Its scratch is
// its own (slots 10..19) and `stream` may be the handle's side stream: the module only touches the probabilities of the OLD map points, so
// the map-update chain runs it next to the decimation that follows (ops_map_update_chain).
static icpmi_status dynpts_dev(icpmi_ctx* c, const icpmi_dynpts_params* prm, const float T[16], const float4* d_in, int64_t n,
                               const float4* d_map, const float* d_nrm, int64_t m, float* d_prob, hipStream_t stream)
{
    if (n == 0 || m == 0) return ICPMI_OK; // "if (beams.empty()) return"
    DynGrid g;
    const float reach = 2 * prm->beam_half_angle;
    g.r2 = reach * reach;
    g.cell = reach / (float)DYN_RINGS;
    g.ne = (int)floorf(3.14159265358979f / g.cell) + 2;
    g.na = (int)floorf(6.28318530717959f / g.cell) + 2;
    const int64_t ncells = (int64_t)g.ne * g.na;
    if (ncells > (1ll << 28)) { c->last_error = "dynamic_points_update: beamHalfAngle too small for the angular grid"; return ICPMI_ERR_UNSUPPORTED; }
    DynPrm dp = {prm->threshold_dynamic, prm->alpha, prm->beta, prm->beam_half_angle, prm->epsilon_a, prm->epsilon_d, prm->sensor_max_range};
    float4* d_bx = scratch_get<float4>(c, 10, (size_t)n);
    float2* d_ba = scratch_get<float2>(c, 11, (size_t)n);
    unsigned* d_keys = scratch_get<unsigned>(c, 12, (size_t)n);
    unsigned* d_start = scratch_get<unsigned>(c, 13, (size_t)ncells + 2);
    unsigned* d_cnt = scratch_get<unsigned>(c, 14, (size_t)ncells + 2);
    float4* d_rec = scratch_get<float4>(c, 15, (size_t)n);
    const bool side_scan = device_scan_side_ok((int)ncells);
    unsigned* d_sums = side_scan ? scratch_get<unsigned>(c, 16, device_scan_side_words((int)ncells)) : nullptr;
    if (!d_bx || !d_ba || !d_keys || !d_start || !d_cnt || !d_rec || (side_scan && !d_sums)) return ICPMI_ERR_HIP;
    if (!side_scan && stream != c->stream) { c->last_error = "dynamic_points_update: internal -- side stream with a table the two-kernel scan cannot take"; return ICPMI_ERR_UNSUPPORTED; }
    const Mat16 M = mat16(T);
    // counts -> starts in cursor layout (map_build.hip): one table to clear, the starts are written in full by the scan
    HIP_TRY(c, hipMemsetAsync(d_cnt, 0, ((size_t)ncells + 2) * sizeof(unsigned), stream));
    const int nb = (int)((n + 255) / 256), mb = (int)((m + 255) / 256);
    hipLaunchKernelGGL(dyn_beams_kernel, dim3(nb), dim3(256), 0, stream, d_in, n, M, g, d_bx, d_ba, d_keys, d_cnt);
    icpmi_status st = side_scan ? device_exclusive_scan_cursor_side(c, stream, d_sums, d_cnt, d_start, (int)ncells, (unsigned)n)
                                : device_exclusive_scan_cursor(c, d_cnt, d_start, (int)ncells, (unsigned)n, false);
    if (st != ICPMI_OK) return st;
    hipLaunchKernelGGL(dyn_scatter_kernel, dim3(nb), dim3(256), 0, stream, n, d_keys, d_start + 1, d_ba, d_rec);
    hipLaunchKernelGGL(dyn_update_kernel, dim3(mb), dim3(256), 0, stream, d_map, d_nrm, m, M, g, dp, (const float4*)d_bx, (const float4*)d_rec, (const unsigned*)d_start, d_prob);
    HIP_TRY(c, hipGetLastError());
    return ICPMI_OK;
}

// whether the side scan of dynpts_dev is available for these parameters (the chain asks before it forks)
static bool dynpts_side_ok(const icpmi_dynpts_params* prm)
{
    const float cell = 2 * prm->beam_half_angle / (float)DYN_RINGS;
    const int64_t ncells = ((int64_t)floorf(3.14159265358979f / cell) + 2) * ((int64_t)floorf(6.28318530717959f / cell) + 2);
    return ncells <= (1ll << 28) && device_scan_side_ok((int)ncells);
}

icpmi_status ops_dynamic_points_update(icpmi_ctx* c, const icpmi_dynpts_params* prm, const float to_sensor[16], const float* in4, int64_t n,
                                       const float* map4, const float* map_normals3, int64_t m, float* prob)
{
    if (n == 0 || m == 0) return ICPMI_OK;
    DevBuf<float> d_nrm, d_prob; DevBuf<float4> d_in, d_map;
    HIP_TRY(c, d_in.alloc((size_t)n));
    HIP_TRY(c, d_map.alloc((size_t)m));
    HIP_TRY(c, d_nrm.alloc((size_t)m * 3));
    HIP_TRY(c, d_prob.alloc((size_t)m));
    HIP_TRY(c, hipMemcpyAsync(d_in, in4, (size_t)n * sizeof(float4), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(d_map, map4, (size_t)m * sizeof(float4), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(d_nrm, map_normals3, (size_t)m * 3 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(d_prob, prob, (size_t)m * sizeof(float), hipMemcpyHostToDevice, c->stream));
    icpmi_status st = dynpts_dev(c, prm, to_sensor, d_in, n, d_map, d_nrm, m, d_prob, c->stream);
    if (st != ICPMI_OK) return st;
    HIP_TRY(c, hipMemcpyAsync(prob, d_prob, (size_t)m * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return ICPMI_OK;
}

// SurfaceNormalDataPointsFilter{knn} over a DEVICE cloud into a DEVICE 3 x m array (same kernels as ops_surface_normals).
// m_old > 0 (r6): d_pts[0 .. m_old) is the cloud this function last ran on (the resident map before an append), d_normals3 still holds its
// normals and c->d_raw_dk the d^2 of every point's k-th neighbour: only the appended points and the old points an appended point can have
// entered the neighbourhood of are searched and solved again -- the other normals are what a pass over the whole cloud would write, bit for
// bit (same neighbours, same coordinates, same sums).  Reference semantics unchanged: Map.cpp:524 applies the filter to the whole map.
static icpmi_status surface_normals_dev(icpmi_ctx* c, const float4* d_pts, int64_t m, int knn, float* d_normals3, int64_t m_old = 0, bool remember = false,
                                        const unsigned** changed_list = nullptr, int64_t* changed_n = nullptr)
{
    if (changed_list) *changed_list = nullptr;
    if (changed_n) *changed_n = -1; // (-1: the whole field)
    TempCtx t;
    icpmi_status s = make_temp(c, t);
    if (s != ICPMI_OK) return s;
    icpmi_ctx* tc = t.h;
    if (tc->stream != c->stream) HIP_TRY(c, hipStreamSynchronize(c->stream)); // d_pts was produced on the caller's stream
    // r6: the sparse block grid (selfgrid.hip) -- index and search in one call
    const size_t cnt = (size_t)m * knn + 1;
    if (ensure_cap(tc, &tc->d_sidx, &tc->cap_sidx, cnt) != ICPMI_OK || ensure_cap(tc, &tc->d_d2, &tc->cap_d2, cnt) != ICPMI_OK) { c->last_error = tc->last_error; return ICPMI_ERR_HIP; }
    const bool track = remember && d_pts == c->d_raw; // (the resident map of an append-only update: remember the k-th distances for the next append;
                                                      //  a chain program may compact or reorder the points behind its normals pass: not tracked)
    if (!track) c->dk_m = 0;
    static const bool inc_on = [] { const char* e = getenv("ICPMI_NORMALS_INCREMENTAL"); return !e || atoi(e) != 0; }(); // 0: every pass over the whole cloud (diagnostic / A-B)
    const bool incremental = inc_on && track && m_old > 0 && m_old < m && c->d_raw_dk && c->dk_m == m_old && c->dk_knn == knn && c->dk_epoch == c->raw_epoch;
    if (track && ensure_cap_keep(c, &c->d_raw_dk, &c->cap_raw_dk, (size_t)m + 1, incremental ? (size_t)m_old : 0) != ICPMI_OK) return ICPMI_ERR_HIP;
    SelfGridSubset sub;
    sub.m_old = incremental ? m_old : 0; sub.d_dk = c->d_raw_dk;
    s = selfgrid_knn(tc, d_pts, m, knn, tc->d_sidx, tc->d_d2, track ? &sub : nullptr); // (tracked: the grid keeps its sorted copy for the next append)
    if (s != ICPMI_OK) { c->last_error = tc->last_error; return s; }
    const unsigned* list = incremental ? sub.d_list : nullptr;
    const int64_t todo = incremental ? sub.n_sel : m;
    if (incremental && changed_list && changed_n) { *changed_list = list; *changed_n = todo; } // (valid until the grid's next search)
    // rows of d_sidx follow the query order = the caller's order, so the normals land in place
    if (todo > 0) launch_normals(tc->stream, tc->d_map_sorted, tc->d_sidx, todo, knn, d_normals3, (float*)nullptr, c->cfg.is_2d, nullptr, nullptr, nullptr, list);
    if (track) {
        if (todo > 0) hipLaunchKernelGGL(kth_d2_kernel, dim3((int)((todo + 255) / 256)), dim3(256), 0, tc->stream, (const float*)tc->d_d2, todo, knn, list, c->d_raw_dk);
        c->dk_m = m; c->dk_knn = knn; c->dk_epoch = c->raw_epoch;
        if (incremental) { ++c->normals_incremental; c->normals_last_searched = sub.n_sel; } else { ++c->normals_full; c->normals_last_searched = m; }
    }
    HIP_TRY(c, hipGetLastError());
    if (tc->stream != c->stream) HIP_TRY(c, hipStreamSynchronize(tc->stream));
    return ICPMI_OK;
}

icpmi_status ops_map_update_point_distance(icpmi_ctx* c, const float* scan4, int64_t n, const float* scan_normals3, float min_dist,
                                           int normals_knn, uint8_t* keep_out, int64_t* appended, int64_t* new_m)
{
    if (appended) *appended = 0;
    if (new_m) *new_m = c->m_raw;
    if (n == 0) return ICPMI_OK;
    // stage the scan (and its normals)
    if (ensure_cap(c, &c->d_stage_in, &c->cap_stage_in, (size_t)n + 1) != ICPMI_OK) return ICPMI_ERR_HIP;
    HIP_TRY(c, hipMemcpyAsync(c->d_stage_in, scan4, (size_t)n * sizeof(float4), hipMemcpyHostToDevice, c->stream));
    if (scan_normals3) {
        if (ensure_cap(c, &c->d_stage_n3, &c->cap_stage_n3, (size_t)n * 3) != ICPMI_OK) return ICPMI_ERR_HIP;
        HIP_TRY(c, hipMemcpyAsync(c->d_stage_n3, scan_normals3, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    }
    return ops_map_update_dev(c, c->d_stage_in, n, scan_normals3 ? c->d_stage_n3 : nullptr, min_dist, normals_knn, keep_out, appended, new_m);
}

// the update proper, on a scan that is already in HBM (map frame)
icpmi_status ops_map_update_dev(icpmi_ctx* c, const float4* d_scan, int64_t n, const float* d_scan_n3, float min_dist, int normals_knn,
                                uint8_t* keep_out, int64_t* appended, int64_t* new_m)
{
    if (appended) *appended = 0;
    if (new_m) *new_m = c->m_raw;
    if (n == 0) return ICPMI_OK;
    if (normals_knn < 0 || normals_knn > ICPMI_MAX_K) { c->last_error = "map_update: normals_knn must be in [0, 32]"; return ICPMI_ERR_INVALID_ARG; }
    const bool scan_normals3 = d_scan_n3 != nullptr;
    const int64_t m0 = c->m > 0 ? c->m_raw : 0;
    const int blocks = (int)((n + 255) / 256);
    unsigned* d_flag = scratch_get<unsigned>(c, 6, (size_t)n + 2); // kept between calls
    unsigned* d_pos = scratch_get<unsigned>(c, 7, (size_t)n + 2);
    if (!d_flag || !d_pos) return ICPMI_ERR_HIP;
    icpmi_status s = ICPMI_OK;
    hipError_t e = hipSuccess;
    unsigned count = 0;
    const bool keep_all = !(min_dist > 0.f); // d2 >= 0 holds for every match and for none: nothing to search (the merge epoch's append, r4)
    if (m0 > 0 && !keep_all) {
        // PointDistanceMapperModule.cpp:33-42: exact NN of every input point in the map AS IT IS (raw_index: not the centred
        // registration index), self match excluded, keep iff d2 >= minDist^2
        icpmi_ctx* ri = nullptr;
        s = raw_index(c, &ri, min_dist);
        if (s == ICPMI_OK) s = chain_point_distance_flags(c, ri, d_scan, n, min_dist, d_flag);
        if (s == ICPMI_OK && keep_out) {
            uint8_t* d_keep = scratch_get<uint8_t>(c, 9, (size_t)n);
            if (!d_keep) e = hipErrorOutOfMemory;
            if (e == hipSuccess) {
                hipLaunchKernelGGL(flag_to_keep_kernel, dim3(blocks), dim3(256), 0, c->stream, (const unsigned*)d_flag, n, d_keep);
                e = hipMemcpyAsync(keep_out, d_keep, (size_t)n, hipMemcpyDeviceToHost, c->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            }
        }
        if (s == ICPMI_OK && e == hipSuccess) {
            s = device_exclusive_scan_io(c, d_flag, d_pos, (int)n, 0u);
        }
        if (s == ICPMI_OK && e == hipSuccess) {
            unsigned lp = 0, lf = 0;
            if (read_back2(c, &lp, d_pos + (n - 1), sizeof(unsigned), &lf, d_flag + (n - 1), sizeof(unsigned)) != ICPMI_OK) e = hipErrorUnknown;
            count = lp + lf;
        }
    } else {
        // PointDistanceMapperModule::createMap: the first scan is the map
        if (keep_out) memset(keep_out, 1, (size_t)n);
        count = (unsigned)n;
    }
    if (s == ICPMI_OK && e == hipSuccess && count > 0) {
        const int64_t m1 = m0 + count;
        const bool want_n = normals_knn > 0 || c->raw_has_normals || scan_normals3;
        if (m1 >= (1ll << 28)) { c->last_error = "map_update: the map would exceed 2^28-1 points"; s = ICPMI_ERR_UNSUPPORTED; }
        if (s == ICPMI_OK) s = ensure_cap_keep(c, &c->d_raw, &c->cap_raw, (size_t)m1, (size_t)m0);
        if (s == ICPMI_OK && want_n) {
            const bool had = c->raw_has_normals && m0 > 0;
            s = ensure_cap_keep(c, &c->d_raw_n3, &c->cap_raw_n3, (size_t)m1 * 3, had ? (size_t)m0 * 3 : 0);
            if (s == ICPMI_OK && !had && m0 > 0) e = hipMemsetAsync(c->d_raw_n3, 0, (size_t)m0 * 3 * sizeof(float), c->stream);
        }
        if (s == ICPMI_OK && e == hipSuccess) {
            if (m0 > 0 && !keep_all)
                hipLaunchKernelGGL(append_flagged_kernel, dim3(blocks), dim3(256), 0, c->stream, d_scan, d_scan_n3, n, (const unsigned*)d_flag,
                                   (const unsigned*)d_pos, m0, c->d_raw, want_n ? c->d_raw_n3 : (float*)nullptr);
            else { // the first scan IS the map, or every point is kept: a plain append behind the m0 points there are
                e = hipMemcpyAsync(c->d_raw + m0, d_scan, (size_t)n * sizeof(float4), hipMemcpyDeviceToDevice, c->stream);
                if (e == hipSuccess && want_n) {
                    if (scan_normals3) e = hipMemcpyAsync(c->d_raw_n3 + 3 * m0, d_scan_n3, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToDevice, c->stream);
                    else e = hipMemsetAsync(c->d_raw_n3 + 3 * m0, 0, (size_t)n * 3 * sizeof(float), c->stream);
                }
            }
            if (e == hipSuccess) e = hipGetLastError();
        }
        // SurfaceNormalDataPointsFilter over the grown map (Map.cpp:524 with examples/config.yaml:26-27)
        const unsigned* changed = nullptr; int64_t n_changed = -1;
        if (s == ICPMI_OK && e == hipSuccess && normals_knn > 0) s = surface_normals_dev(c, c->d_raw, m1, normals_knn, c->d_raw_n3, m0, true, &changed, &n_changed); // (m0: an append -- only what it changed)
        // icp.setMap(localPointCloud) (Map.cpp:528): rebuild the index from the resident copy
        // (an append: the first m0 resident points are the cloud the index was built from -- map_insert where it applies)
        // (the whole field was recomputed above / had no sorted copy: the insert gathers every normal again; r6: a FEW were recomputed -- the sorted
        //  normals move along with their points and the changed ones are patched in afterwards, map_build.hip: map_patch_normals)
        c->ins_normals_changed = (normals_knn > 0 && n_changed < 0) || !c->has_normals;
        const bool patch = normals_knn > 0 && n_changed >= 0 && !c->ins_normals_changed;
        if (s == ICPMI_OK && e == hipSuccess) s = map_build(c, c->d_raw, m1, want_n ? c->d_raw_n3 : nullptr, (m0 > 0 && c->m == m0) ? m0 : 0);
        c->ins_normals_changed = false;
        if (s == ICPMI_OK && e == hipSuccess && patch) s = map_patch_normals(c, changed, n_changed, c->d_raw, c->d_raw_n3);
        if (s == ICPMI_OK) { if (appended) *appended = count; if (new_m) *new_m = m1; }
    }
    if (s != ICPMI_OK) return s;
    HIP_TRY(c, e);
    return ICPMI_OK;
}

icpmi_status ops_get_map(icpmi_ctx* c, float* out4, float* normals3, int64_t capacity, int64_t* m)
{
    if (m) *m = c->m > 0 ? c->m_raw : 0;
    if (c->m <= 0 || !out4) return ICPMI_OK;
    if (capacity < c->m_raw) { c->last_error = "get_map: capacity too small"; return ICPMI_ERR_INVALID_ARG; }
    HIP_TRY(c, hipMemcpyAsync(out4, c->d_raw, (size_t)c->m_raw * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    if (normals3) {
        if (!c->raw_has_normals) { c->last_error = "InvalidField: the map has no normals"; return ICPMI_ERR_MISSING_NORMALS; }
        HIP_TRY(c, hipMemcpyAsync(normals3, c->d_raw_n3, (size_t)c->m_raw * 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return ICPMI_OK;
}

// =================================================================================================================
// Map::updateLocalPointCloud (Map.cpp:502-534) as one program over the resident map: mapper modules, then post
// filters.  Working set = the resident arrays themselves (features, normals, one scalar descriptor, provenance);
// compactions write into the ping-pong set and swap.  Every step is order preserving, so the result is a function
// of the inputs alone and equals the host chain built from the single operators above.
// =================================================================================================================
namespace {

__global__ __launch_bounds__(256) void chain_iota_kernel(int* __restrict__ src, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) src[i] = (int)i;
}

// scan point i (kept iff flag == nullptr or flag[i]) -> slot base + (pos ? pos[i] : i); provenance src_base + i
__global__ __launch_bounds__(256) void chain_append_kernel(const float4* __restrict__ in, const float* __restrict__ in_n3,
                                                           const float* __restrict__ in_s, int64_t n, const unsigned* __restrict__ flag,
                                                           const unsigned* __restrict__ pos, int64_t base, int src_base,
                                                           float4* __restrict__ raw, float* __restrict__ raw_n3, float* __restrict__ raw_s,
                                                           int* __restrict__ src)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || (flag && !flag[i])) return;
    const int64_t o = base + (pos ? (int64_t)pos[i] : i);
    raw[o] = in[i];
    raw_n3[3 * o] = in_n3 ? in_n3[3 * i] : 0.f;
    raw_n3[3 * o + 1] = in_n3 ? in_n3[3 * i + 1] : 0.f;
    raw_n3[3 * o + 2] = in_n3 ? in_n3[3 * i + 2] : 0.f;
    raw_s[o] = in_s ? in_s[i] : 0.f;
    src[o] = src_base + (int)i;
}

__global__ __launch_bounds__(256) void chain_compact_kernel(int64_t m, const unsigned* __restrict__ flag, const unsigned* __restrict__ pos,
                                                            const float4* __restrict__ raw, const float* __restrict__ n3,
                                                            const float* __restrict__ sc, const int* __restrict__ src,
                                                            float4* __restrict__ o_raw, float* __restrict__ o_n3, float* __restrict__ o_sc,
                                                            int* __restrict__ o_src)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= m || !flag[i]) return;
    const int64_t o = pos[i];
    o_raw[o] = raw[i];
    o_n3[3 * o] = n3[3 * i]; o_n3[3 * o + 1] = n3[3 * i + 1]; o_n3[3 * o + 2] = n3[3 * i + 2];
    o_sc[o] = sc[i];
    o_src[o] = src[i];
}

// first j with src[j] != j (m if none): the untouched head of the map, which the host need not be told about
__global__ __launch_bounds__(256) void chain_prefix_kernel(const int* __restrict__ src, int64_t m, unsigned* __restrict__ first_moved)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool moved = i < m && src[i] != (int)i;
    const unsigned long long b = __ballot(moved);
    if (b && (threadIdx.x & 63) == 0) {
        const unsigned cand = (unsigned)(i + __ffsll((long long)b) - 1);
        if (cand < __hip_atomic_load(first_moved, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(first_moved, cand);
    }
}

// CutAtDescriptorThresholdDataPointsFilter: useLargerThan drops v > threshold, else drops v < threshold
__global__ __launch_bounds__(256) void chain_cut_flag_kernel(const float* __restrict__ sc, int64_t m, float threshold, int larger,
                                                             unsigned* __restrict__ flag)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const float v = sc[i];
    flag[i] = (larger ? !(v > threshold) : !(v < threshold)) ? 1u : 0u;
}

struct Chain {
    icpmi_ctx* c;
    int64_t m = 0;         // points in the working map
    bool has_n = false;    // its normals mean something
    bool indexed = false;  // the handle's own search index still describes exactly the working map
};

// room for `need` points in the main set, the first w.m survive
icpmi_status chain_reserve(Chain& w, int64_t need)
{
    icpmi_ctx* c = w.c;
    if (need >= (1ll << 28)) { c->last_error = "map_update_chain: the map would exceed 2^28-1 points"; return ICPMI_ERR_UNSUPPORTED; }
    icpmi_status s = ensure_cap_keep(c, &c->d_raw, &c->cap_raw, (size_t)need, (size_t)w.m);
    if (s == ICPMI_OK) s = ensure_cap_keep(c, &c->d_raw_n3, &c->cap_raw_n3, (size_t)need * 3, (size_t)w.m * 3);
    if (s == ICPMI_OK) s = ensure_cap_keep(c, &c->d_raw_s, &c->cap_raw_s, (size_t)need, (size_t)w.m);
    if (s == ICPMI_OK) s = ensure_cap_keep(c, &c->d_src, &c->cap_src, (size_t)need, (size_t)w.m);
    return s;
}

icpmi_status chain_append(Chain& w, const float4* d_scan, const float* d_n3, const float* d_s, int64_t n, const unsigned* flag,
                          const unsigned* pos, int64_t count, int src_base)
{
    icpmi_ctx* c = w.c;
    if (count == 0) return ICPMI_OK;
    icpmi_status s = chain_reserve(w, w.m + count);
    if (s != ICPMI_OK) return s;
    hipLaunchKernelGGL(chain_append_kernel, dim3((int)((n + 255) / 256)), dim3(256), 0, c->stream, d_scan, d_n3, d_s, n, flag, pos, w.m,
                       src_base, c->d_raw, c->d_raw_n3, c->d_raw_s, c->d_src);
    HIP_TRY(c, hipGetLastError());
    w.m += count;
    w.indexed = false;
    return ICPMI_OK;
}

// flags (0/1) -> stable compaction of the working map; d_pos is scratch of m + 2 words
icpmi_status chain_compact(Chain& w, unsigned* d_flag, unsigned* d_pos)
{
    icpmi_ctx* c = w.c;
    const int64_t m = w.m;
    if (m == 0) return ICPMI_OK;
    int64_t count = 0;
    icpmi_status s = device_scan_flags_count(c, d_flag, d_pos, (int)m, &count);
    if (s != ICPMI_OK) return s;
    if (count == m) return ICPMI_OK; // nothing dropped
    if (ensure_cap(c, &c->d_alt_raw, &c->cap_alt_raw, (size_t)count + 1) != ICPMI_OK || ensure_cap(c, &c->d_alt_n3, &c->cap_alt_n3, (size_t)count * 3 + 1) != ICPMI_OK ||
        ensure_cap(c, &c->d_alt_s, &c->cap_alt_s, (size_t)count + 1) != ICPMI_OK || ensure_cap(c, &c->d_alt_src, &c->cap_alt_src, (size_t)count + 1) != ICPMI_OK)
        return ICPMI_ERR_HIP;
    hipLaunchKernelGGL(chain_compact_kernel, dim3((int)((m + 255) / 256)), dim3(256), 0, c->stream, m, d_flag, d_pos, c->d_raw, c->d_raw_n3,
                       c->d_raw_s, c->d_src, c->d_alt_raw, c->d_alt_n3, c->d_alt_s, c->d_alt_src);
    HIP_TRY(c, hipGetLastError());
    std::swap(c->d_raw, c->d_alt_raw); std::swap(c->cap_raw, c->cap_alt_raw);
    std::swap(c->d_raw_n3, c->d_alt_n3); std::swap(c->cap_raw_n3, c->cap_alt_n3);
    std::swap(c->d_raw_s, c->d_alt_s); std::swap(c->cap_raw_s, c->cap_alt_s);
    std::swap(c->d_src, c->d_alt_src); std::swap(c->cap_src, c->cap_alt_src);
    w.m = count;
    w.indexed = false;
    return ICPMI_OK;
}

// the working map re-ordered / decimated by an index list: new point j = old point order[j] (OctreeGridDataPointsFilter
// leaves the cloud in leaf-visiting order)
__global__ __launch_bounds__(256) void chain_gather_kernel(int64_t count, const int* __restrict__ order, const float4* __restrict__ raw,
                                                           const float* __restrict__ n3, const float* __restrict__ sc, const int* __restrict__ src,
                                                           float4* __restrict__ o_raw, float* __restrict__ o_n3, float* __restrict__ o_sc,
                                                           int* __restrict__ o_src)
{
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= count) return;
    const int64_t i = order[o];
    o_raw[o] = raw[i];
    o_n3[3 * o] = n3[3 * i]; o_n3[3 * o + 1] = n3[3 * i + 1]; o_n3[3 * o + 2] = n3[3 * i + 2];
    o_sc[o] = sc[i];
    o_src[o] = src[i];
}

icpmi_status chain_gather(Chain& w, const int* d_order, int64_t count)
{
    icpmi_ctx* c = w.c;
    if (ensure_cap(c, &c->d_alt_raw, &c->cap_alt_raw, (size_t)count + 1) != ICPMI_OK || ensure_cap(c, &c->d_alt_n3, &c->cap_alt_n3, (size_t)count * 3 + 1) != ICPMI_OK ||
        ensure_cap(c, &c->d_alt_s, &c->cap_alt_s, (size_t)count + 1) != ICPMI_OK || ensure_cap(c, &c->d_alt_src, &c->cap_alt_src, (size_t)count + 1) != ICPMI_OK)
        return ICPMI_ERR_HIP;
    if (count) hipLaunchKernelGGL(chain_gather_kernel, dim3((int)((count + 255) / 256)), dim3(256), 0, c->stream, count, d_order, c->d_raw, c->d_raw_n3,
                                  c->d_raw_s, c->d_src, c->d_alt_raw, c->d_alt_n3, c->d_alt_s, c->d_alt_src);
    HIP_TRY(c, hipGetLastError());
    std::swap(c->d_raw, c->d_alt_raw); std::swap(c->cap_raw, c->cap_alt_raw);
    std::swap(c->d_raw_n3, c->d_alt_n3); std::swap(c->cap_raw_n3, c->cap_alt_n3);
    std::swap(c->d_raw_s, c->d_alt_s); std::swap(c->cap_raw_s, c->cap_alt_s);
    std::swap(c->d_src, c->d_alt_src); std::swap(c->cap_src, c->cap_alt_src);
    w.m = count;
    w.indexed = false;
    return ICPMI_OK;
}

// PointDistanceMapperModule.cpp:33-42 on an indexing handle `ic` (the caller's own handle while its index still
// describes the working map, the private one otherwise): flag[i] = exact NN of scan i at d2 >= minDist^2
icpmi_status chain_point_distance_flags(icpmi_ctx* c, icpmi_ctx* ic, const float4* d_scan, int64_t n, float min_dist, unsigned* d_flag)
{
    const double lim = pd_limit(min_dist);
    icpmi_status s = loop_prepare_reading(ic, d_scan, n, nullptr);
    if (s != ICPMI_OK) { c->last_error = ic->last_error; return s; }
    LoopCfg lc = make_loop_cfg(ic, 1);
    lc.inv1e = 1.f; lc.err2 = 1.f; // (the module's predicate is exact whatever the matcher's epsilon)
    lc.k = 1; lc.n_out = 0; lc.max_dist = min_dist; lc.maxr2 = pd_radius2(lim); // a radius search with maxDist = minDist decides the same predicate
    const size_t cnt = (size_t)n + 1;
    if (ensure_cap(ic, &ic->d_sidx, &ic->cap_sidx, cnt) != ICPMI_OK || ensure_cap(ic, &ic->d_d2, &ic->cap_d2, cnt) != ICPMI_OK ||
        ensure_cap(ic, &ic->d_hard, &ic->cap_hard, cnt) != ICPMI_OK) { c->last_error = ic->last_error; return ICPMI_ERR_HIP; }
    HIP_TRY(c, hipMemsetAsync(ic->d_state, 0, sizeof(IcpState), ic->stream));
    ic->nn_hist0 = nullptr; ic->nn_iter_hint = 0; ic->nn_match_pt = nullptr;
    s = nn_launch_k(ic, ic->d_reading, n, nullptr, lc, 0, ic->d_sidx, ic->d_d2, ic->d_state);
    if (s != ICPMI_OK) { c->last_error = ic->last_error; return s; }
    hipLaunchKernelGGL(keep_flag_kernel, dim3((int)((n + 255) / 256)), dim3(256), 0, ic->stream, ic->d_d2, n, lim, d_flag);
    HIP_TRY(c, hipGetLastError());
    if (ic->stream != c->stream) HIP_TRY(c, hipStreamSynchronize(ic->stream));
    return ICPMI_OK;
}

} // namespace

// the side stream of the map-update chain and its two events, created on first use
static bool chain_side_ready(icpmi_ctx* c)
{
    if (c->side && c->side_fork && c->side_join) return true;
    if (!c->side && stream_acquire(&c->side) != hipSuccess) { c->side = nullptr; return false; }
    if (!c->side_fork && hipEventCreateWithFlags(&c->side_fork, hipEventDisableTiming) != hipSuccess) { c->side_fork = nullptr; return false; }
    if (!c->side_join && hipEventCreateWithFlags(&c->side_join, hipEventDisableTiming) != hipSuccess) { c->side_join = nullptr; return false; }
    return true;
}

// the working map moved by T in place: features by T, normals by its rotation (RigidTransformation::compute on the whole map,
// Map.cpp:523 / :525)
static icpmi_status chain_move(icpmi_ctx* c, const float T[16], int64_t m, bool has_n)
{
    if (m == 0) return ICPMI_OK;
    const icpmi_status s = check_rigid(c, T);
    if (s != ICPMI_OK) return s;
    hipLaunchKernelGGL(move_kernel, dim3((int)((m + 255) / 256)), dim3(256), 0, c->stream, c->d_raw, has_n ? c->d_raw_n3 : (float*)nullptr, m, mat16(T));
    HIP_TRY(c, hipGetLastError());
    return ICPMI_OK;
}

icpmi_status ops_map_update_chain(icpmi_ctx* c, const float4* d_scan, int64_t n, const float* d_scan_n3, const float* d_scan_s,
                                  const float to_sensor[16], const float from_sensor[16], const icpmi_map_op* ops, int n_ops, int n_modules,
                                  int32_t* src_out, int64_t src_capacity, int64_t* identity_prefix, int64_t* new_m)
{
    if (identity_prefix) *identity_prefix = 0;
    const int64_t m0 = c->m > 0 ? c->m_raw : 0;
    if (new_m) *new_m = m0;
    ++c->raw_epoch; // the program may move, compact or reorder the resident points: the private raw-frame index is rebuilt, not appended to
    if (n_ops < 0 || n_modules < 0 || n_modules > n_ops || (n_ops > 0 && !ops)) { c->last_error = "map_update_chain: bad program"; return ICPMI_ERR_INVALID_ARG; }
    if (src_out && src_capacity < m0 + (int64_t)n_modules * n) { c->last_error = "map_update_chain: src_capacity must be >= m_old + n_modules * n"; return ICPMI_ERR_INVALID_ARG; }
    bool uses_scalar = false;
    for (int i = 0; i < n_ops; ++i) {
        const icpmi_map_op& op = ops[i];
        switch (op.type) {
        case ICPMI_MOP_POINT_DISTANCE: if (!(op.f[0] >= 0.f)) { c->last_error = "InvalidParameter: minDistNewPoint must be >= 0"; return ICPMI_ERR_INVALID_ARG; } break;
        case ICPMI_MOP_DYNAMIC_POINTS: uses_scalar = true; if (!(op.f[3] > 0.f)) { c->last_error = "InvalidParameter: beamHalfAngle must be > 0"; return ICPMI_ERR_INVALID_ARG; } break;
        case ICPMI_MOP_VOXEL: if (!(op.f[0] > 0.f) || (op.i != 0 && op.i != 1)) { c->last_error = "map_update_chain: voxel edge must be > 0 and samplingMethod 0 or 1"; return ICPMI_ERR_INVALID_ARG; } break;
        case ICPMI_MOP_SURFACE_NORMALS: if (op.i < 1 || op.i > ICPMI_MAX_K) { c->last_error = "surface_normals: knn must be in [1, 32]"; return ICPMI_ERR_INVALID_ARG; } break;
        case ICPMI_MOP_CUT_SCALAR: uses_scalar = true; break;
        case ICPMI_MOP_OCTREE:
            if (!(op.f[0] >= 0.f) || (op.i != 0 && op.i != 1) || !(op.f[1] >= 0.f) || op.f[1] > 64.f) {
                c->last_error = "map_update_chain: octree needs maxSizeByNode >= 0, samplingMethod 0 or 1, maxPointByNode <= 64"; return ICPMI_ERR_INVALID_ARG;
            }
            break;
        default: c->last_error = "map_update_chain: unknown operator"; return ICPMI_ERR_INVALID_ARG;
        }
        if (i >= n_modules && op.type != ICPMI_MOP_SURFACE_NORMALS && op.type != ICPMI_MOP_CUT_SCALAR) { c->last_error = "map_update_chain: a mapper module after the post filters"; return ICPMI_ERR_INVALID_ARG; }
        if (i < n_modules && (op.type == ICPMI_MOP_SURFACE_NORMALS || op.type == ICPMI_MOP_CUT_SCALAR)) { c->last_error = "map_update_chain: a post filter among the mapper modules"; return ICPMI_ERR_INVALID_ARG; }
    }
    if (n_modules == 0) { c->last_error = "InvalidParameter: no mapper module configured"; return ICPMI_ERR_INVALID_ARG; }
    if (uses_scalar && n > 0 && !d_scan_s) { c->last_error = "InvalidField: the chain needs the tracked scalar descriptor on the input (AddDescriptorDataPointsFilter)"; return ICPMI_ERR_INVALID_ARG; }
    if (uses_scalar && m0 > 0 && !c->raw_has_scalar) { c->last_error = "InvalidField: the chain needs the tracked scalar descriptor on the map (icpmi_set_map_scalar)"; return ICPMI_ERR_INVALID_ARG; }
    if (n == 0 && m0 == 0) return ICPMI_OK;
    for (int i = 0; i < n_modules; ++i)
        if (ops[i].type == ICPMI_MOP_DYNAMIC_POINTS && m0 > 0 && n > 0 && !c->raw_has_normals) { // before anything is touched
            c->last_error = "InvalidField: Missing field 'normals' in map point cloud. You can add it with the SurfaceNormalDataPointsFilter in your post filters.";
            return ICPMI_ERR_MISSING_NORMALS;
        }

    Chain w;
    w.c = c; w.m = m0; w.has_n = m0 > 0 && c->raw_has_normals; w.indexed = m0 > 0;
    // the main set must exist before the first kernel touches it; normals / scalar a map never had read as zeros
    icpmi_status s = ICPMI_OK;
    {
        const bool had_n = w.has_n, had_s = m0 > 0 && c->raw_has_scalar;
        s = ensure_cap_keep(c, &c->d_raw, &c->cap_raw, (size_t)(m0 + n + 1), (size_t)m0);
        if (s == ICPMI_OK) s = ensure_cap_keep(c, &c->d_raw_n3, &c->cap_raw_n3, (size_t)(m0 + n + 1) * 3, had_n ? (size_t)m0 * 3 : 0);
        if (s == ICPMI_OK) s = ensure_cap_keep(c, &c->d_raw_s, &c->cap_raw_s, (size_t)(m0 + n + 1), had_s ? (size_t)m0 : 0);
        if (s == ICPMI_OK) s = ensure_cap(c, &c->d_src, &c->cap_src, (size_t)(m0 + n + 1));
        if (s != ICPMI_OK) return s;
        if (m0 > 0 && !had_n) HIP_TRY(c, hipMemsetAsync(c->d_raw_n3, 0, (size_t)m0 * 3 * sizeof(float), c->stream));
        if (m0 > 0 && !had_s) HIP_TRY(c, hipMemsetAsync(c->d_raw_s, 0, (size_t)m0 * sizeof(float), c->stream));
        if (m0 > 0) hipLaunchKernelGGL(chain_iota_kernel, dim3((int)((m0 + 255) / 256)), dim3(256), 0, c->stream, c->d_src, m0);
    }
    // every module appends at most the whole scan
    unsigned* d_flag = scratch_get<unsigned>(c, 6, (size_t)(m0 + (int64_t)n_modules * n + 2));
    unsigned* d_pos = scratch_get<unsigned>(c, 7, (size_t)(m0 + (int64_t)n_modules * n + 2));
    if (!d_flag || !d_pos) return ICPMI_ERR_HIP;
    const int src_base = (int)m0;
    bool created = m0 > 0; // false until the first module has created the map from the scan

    // Map.cpp:523-525: the post filters see the map in the SENSOR frame -- the whole map is moved by pose^-1, filtered, and
    // moved back by pose, on every update; the coordinates of every map point pick up that rounding each time, and so must
    // the resident copy (a replay of the bundled trajectory leaves the reference's by up to 0.4 mm otherwise: which point
    // survives a decimation or a threshold can hinge on the last bit).  from_sensor == NULL: filters in the map frame.
    const bool round_trip = from_sensor != nullptr && to_sensor != nullptr;
    bool in_sensor = false;
    // ICPMI_CHAIN_TIMING=1: wall time of every step with a stream sync behind it, on stderr (diagnostic; perturbs the overlap)
    static const bool timing = [] { const char* e = getenv("ICPMI_CHAIN_TIMING"); return e && atoi(e) != 0; }();
    auto tick = [&](const char* what, int type) {
        static thread_local std::chrono::steady_clock::time_point t0;
        if (!timing) return;
        (void)hipStreamSynchronize(c->stream);
        const auto t1 = std::chrono::steady_clock::now();
        if (what) fprintf(stderr, "[icpmi chain] %-14s type %d  m %lld  %8.1f us\n", what, type, (long long)w.m, std::chrono::duration<double, std::micro>(t1 - t0).count());
        t0 = std::chrono::steady_clock::now();
    };
    tick(nullptr, 0);
    constexpr bool overlap = true;
    bool forked = false;
    // whatever path leaves this function between a fork and its join (an error return inside the operator loop): nothing may still be
    // running on the side stream when the caller drops or reuses the arrays it works on
    struct SideGuard { icpmi_ctx* c; bool* forked; ~SideGuard() { if (*forked && c->side) (void)hipStreamSynchronize(c->side); } } side_guard{c, &forked};
    auto join = [&]() -> icpmi_status { // the handle's stream waits for the module on the side stream
        if (!forked) return ICPMI_OK;
        forked = false;
        HIP_TRY(c, hipStreamWaitEvent(c->stream, c->side_join, 0));
        return ICPMI_OK;
    };
    for (int i = 0; i < n_ops && s == ICPMI_OK; ++i) {
        const icpmi_map_op& op = ops[i];
        if (forked && op.type != ICPMI_MOP_OCTREE && op.type != ICPMI_MOP_VOXEL) { s = join(); if (s != ICPMI_OK) break; } // (cannot happen: the fork looks at the next module)
        if (i == n_modules && round_trip) { s = chain_move(c, to_sensor, w.m, w.has_n); in_sensor = true; if (s != ICPMI_OK) break; tick("to_sensor", -1); }
        switch (op.type) {
        case ICPMI_MOP_POINT_DISTANCE: {
            if (n == 0) break;
            if (!created) { s = chain_append(w, d_scan, d_scan_n3, d_scan_s, n, nullptr, nullptr, n, src_base); w.has_n = d_scan_n3 != nullptr; break; } // createMap: the scan is the map
            if (w.m == 0) { s = chain_append(w, d_scan, d_scan_n3, d_scan_s, n, nullptr, nullptr, n, src_base); break; } // no neighbour anywhere: d2 = inf
            icpmi_ctx* ic = nullptr;
            if (w.indexed) { s = raw_index(c, &ic, op.f[0]); if (s != ICPMI_OK) break; } // the resident map, in its own frame
            else {
                TempCtx t;
                s = make_temp(c, t);
                if (s != ICPMI_OK) break;
                ic = t.h;
                if (ic->stream != c->stream) HIP_TRY(c, hipStreamSynchronize(c->stream));
                int32_t acc = 0;
                ic->cfg.max_dist = op.f[0] > 1e-3f ? op.f[0] : 1e-3f; // the levels this radius needs (see raw_index)
                s = icpmi_set_map_dev(ic, (const float*)c->d_raw, w.m, nullptr, &acc);
                if (s != ICPMI_OK) { c->last_error = ic->last_error; break; }
            }
            s = chain_point_distance_flags(c, ic, d_scan, n, op.f[0], d_flag);
            if (s != ICPMI_OK) break;
            int64_t accepted = 0;
            s = device_scan_flags_count(c, d_flag, d_pos, (int)n, &accepted);
            if (s != ICPMI_OK) break;
            s = chain_append(w, d_scan, d_scan_n3, d_scan_s, n, d_flag, d_pos, accepted, src_base);
            break;
        }
        case ICPMI_MOP_DYNAMIC_POINTS: {
            if (!created) { s = chain_append(w, d_scan, d_scan_n3, d_scan_s, n, nullptr, nullptr, n, src_base); w.has_n = d_scan_n3 != nullptr; break; } // createMap: no-op on the scan
            if (n == 0 || w.m == 0) break;
            if (!w.has_n) { c->last_error = "InvalidField: Missing field 'normals' in map point cloud. You can add it with the SurfaceNormalDataPointsFilter in your post filters."; s = ICPMI_ERR_MISSING_NORMALS; break; }
            s = check_rigid(c, to_sensor);
            if (s != ICPMI_OK) break;
            icpmi_dynpts_params prm = {op.f[0], op.f[1], op.f[2], op.f[3], op.f[4], op.f[5], op.f[6]};
            // r5: the module reads the old map's points and normals and rewrites the probabilities of THOSE points; the decimation that
            // follows in the shipped chain (OctreeMapperModule: append the scan, sort / hash POSITIONS) does not look at a probability
            // until it moves the survivors.  Fork: the module on the side stream, the decimation on the handle's, joined in front of the
            // kernel that moves the descriptors (chain_gather / chain_compact).  The append may not move the arrays under the side
            // stream: their room is reserved first.
            const bool decimates_next = i + 1 < n_modules && (ops[i + 1].type == ICPMI_MOP_OCTREE || ops[i + 1].type == ICPMI_MOP_VOXEL);
            if (overlap && !timing && decimates_next && dynpts_side_ok(&prm) && chain_side_ready(c)) {
                s = chain_reserve(w, w.m + n);
                if (s != ICPMI_OK) break;
                HIP_TRY(c, hipEventRecord(c->side_fork, c->stream));
                HIP_TRY(c, hipStreamWaitEvent(c->side, c->side_fork, 0));
                s = dynpts_dev(c, &prm, to_sensor, d_scan, n, c->d_raw, c->d_raw_n3, w.m, c->d_raw_s, c->side);
                forked = true; // (also after an error: the side stream may hold half of the module)
                if (s == ICPMI_OK) HIP_TRY(c, hipEventRecord(c->side_join, c->side));
                break;
            }
            s = dynpts_dev(c, &prm, to_sensor, d_scan, n, c->d_raw, c->d_raw_n3, w.m, c->d_raw_s, c->stream);
            break;
        }
        case ICPMI_MOP_VOXEL: {
            // OctreeMapperModule.cpp:35-39: concatenate, then decimate (createMap: an empty map updated by the scan)
            s = chain_append(w, d_scan, d_scan_n3, d_scan_s, n, nullptr, nullptr, n, src_base);
            if (!created) w.has_n = d_scan_n3 != nullptr;
            if (s != ICPMI_OK || w.m == 0) break;
            s = voxel_flags_dev<unsigned>(c, c->d_raw, w.m, op.f[0], op.i, d_flag);
            if (s == ICPMI_OK) s = join();
            if (s == ICPMI_OK) s = chain_compact(w, d_flag, d_pos);
            break;
        }
        case ICPMI_MOP_OCTREE: {
            // OctreeMapperModule.cpp:35-39: concatenate, then the octree filter; the cloud comes out in leaf-visiting order
            s = chain_append(w, d_scan, d_scan_n3, d_scan_s, n, nullptr, nullptr, n, src_base);
            if (!created) w.has_n = d_scan_n3 != nullptr;
            if (s != ICPMI_OK || w.m == 0) break;
            int64_t kept = 0;
            s = octree_sample_dev(c, c->d_raw, w.m, op.f[0], (int)op.f[1], op.i, (int*)d_pos, nullptr, &kept);
            if (s == ICPMI_OK) s = join();
            if (s == ICPMI_OK) s = chain_gather(w, (const int*)d_pos, kept);
            break;
        }
        case ICPMI_MOP_SURFACE_NORMALS: {
            if (w.m == 0) break;
            s = surface_normals_dev(c, c->d_raw, w.m, op.i, c->d_raw_n3);
            w.has_n = true;
            break;
        }
        case ICPMI_MOP_CUT_SCALAR: {
            if (w.m == 0) break;
            hipLaunchKernelGGL(chain_cut_flag_kernel, dim3((int)((w.m + 255) / 256)), dim3(256), 0, c->stream, c->d_raw_s, w.m, op.f[0], op.i, d_flag);
            HIP_TRY(c, hipGetLastError());
            s = chain_compact(w, d_flag, d_pos);
            break;
        }
        }
        created = true;
        tick("op", op.type);
    }
    if (forked) { // an error between fork and join: nothing may be left running on the side stream when the arrays are dropped
        (void)hipStreamSynchronize(c->side);
        forked = false;
    }
    if (s == ICPMI_OK && round_trip) {
        if (!in_sensor) s = chain_move(c, to_sensor, w.m, w.has_n); // no post filter at all: the reference still makes the trip
        if (s == ICPMI_OK) s = chain_move(c, from_sensor, w.m, w.has_n);
        tick("from_sensor", -2);
    }
    if (s != ICPMI_OK) {
        // the resident arrays may be half way through the program: drop them, the index of the old map is intact but its
        // resident copy is not -- force the caller back to icpmi_set_map
        c->m_raw = 0; c->raw_has_normals = false; c->raw_has_scalar = false; c->m = 0;
        return s;
    }
    if (w.m == 0) {
        // the chain removed every point (e.g. CutAtDescriptorThreshold cut the whole map): the reference goes on with an empty local
        // cloud -- `icp.setMap` ignores an empty cloud and keeps its previous map (Map.cpp:528), the next scan
        // creates the map anew (Map.cpp:505-515).  Same here: the resident copy is empty, the registration index stays.
        c->m_raw = 0; c->raw_has_normals = false; c->raw_has_scalar = false;
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (new_m) *new_m = 0;
        return ICPMI_OK;
    }
    if (src_out) {
        if (src_capacity < w.m) { c->last_error = "map_update_chain: src_capacity too small"; return ICPMI_ERR_INVALID_ARG; }
        int64_t head = 0;
        if (identity_prefix) {
            unsigned first_moved = (unsigned)w.m;
            { const icpmi_status us = upload_small(c, d_pos, &first_moved, sizeof(unsigned)); if (us != ICPMI_OK) return us; }
            hipLaunchKernelGGL(chain_prefix_kernel, dim3((int)((w.m + 255) / 256)), dim3(256), 0, c->stream, c->d_src, w.m, d_pos);
            if (read_back(c, &first_moved, d_pos, sizeof(unsigned)) != ICPMI_OK) return ICPMI_ERR_HIP;
            head = first_moved;
            *identity_prefix = head;
        }
        if (w.m > head) HIP_TRY(c, hipMemcpyAsync(src_out + head, c->d_src + head, (size_t)(w.m - head) * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    }
    tick("src", -3);
    // icp.setMap(localPointCloud) (Map.cpp:528): rebuild the index from the resident copy
    s = map_build(c, c->d_raw, w.m, w.has_n ? c->d_raw_n3 : nullptr);
    if (s != ICPMI_OK) return s;
    c->raw_has_scalar = (m0 == 0 || c->raw_has_scalar) && (n == 0 || d_scan_s != nullptr); // both parts carried real values
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    tick("map_build", -4);
    if (new_m) *new_m = w.m;
    return ICPMI_OK;
}

// stable compaction of the flagged points of `in` behind position `base` of `out`
__global__ __launch_bounds__(256) void merge_compact_kernel(const float4* __restrict__ in, int64_t n, const unsigned* __restrict__ flag,
                                                            const unsigned* __restrict__ pos, float4* __restrict__ out, int64_t base)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || !flag[i]) return;
    out[base + pos[i]] = in[i];
}

// flags -> kept points appended to d_out[base ...): returns how many (one small read-back)
static icpmi_status merge_append_flagged(icpmi_ctx* c, const float4* d_in, int64_t n, unsigned* d_flag, unsigned* d_pos, float4* d_out, int64_t base,
                                         int64_t* kept)
{
    *kept = 0;
    if (n == 0) return ICPMI_OK;
    icpmi_status s = device_exclusive_scan_io(c, d_flag, d_pos, (int)n, 0u);
    if (s != ICPMI_OK) return s;
    hipLaunchKernelGGL(merge_compact_kernel, dim3((int)((n + 255) / 256)), dim3(256), 0, c->stream, d_in, n, (const unsigned*)d_flag, (const unsigned*)d_pos,
                       d_out, base);
    HIP_TRY(c, hipGetLastError());
    unsigned lp = 0, lf = 0;
    if (read_back2(c, &lp, d_pos + (n - 1), sizeof(unsigned), &lf, d_flag + (n - 1), sizeof(unsigned)) != ICPMI_OK) return ICPMI_ERR_HIP;
    *kept = (int64_t)lp + lf;
    return ICPMI_OK;
}

// ---- rank-ordered greedy merge of the gathered blocks (r4) -------------------------------------------------------------------------
// The rule (what one mapper would have appended had it taken the scans in rank order, PointDistanceMapperModule.cpp:33-42 block by block):
// a point of block r is dropped iff some KEPT point q of a block < r has FLT_EPSILON < d2(p, q) and (double)d2 < minDist^2, d2 the float
// fmaf chain of the NN kernels on the raw coordinates (sqdist3) -- exactly the predicate the per-block temporary index of r2 / r3 decided
// through its nearest neighbour.  All gathered points go into one spatial hash (cell edge a little above minDist: every such q lies in
// the 3 x 3 x 3 cells around p), then blocks 1 .. R - 1 are flagged one after the other (block r needs the flags of the blocks below it).
// Order inside a cell comes from atomics and is not reproducible; the predicate is an "exists", so the flags are.
struct MergeBlocks { long long cnt[256]; };  // (never more than 256 ranks: comm.hip)
#define MERGE_FILTER_WORDS (1u << 19)        // 2^24 bits: one per 24-bit hash of an occupied cell (a few 10^5 cells: < 2 % false positives)

__device__ __forceinline__ unsigned long long merge_cell_key(int ix, int iy, int iz)
{
    // 21 bits per axis; cells whose indices alias under the mask share a bucket -- more candidates, never fewer (distances decide)
    return ((unsigned long long)((unsigned)ix & 0x1fffffu)) | ((unsigned long long)((unsigned)iy & 0x1fffffu) << 21) |
           ((unsigned long long)((unsigned)iz & 0x1fffffu) << 42);
}
__device__ __forceinline__ int merge_cell_of(float v, float inv_h) { return (int)floorf(v * inv_h); }

__global__ __launch_bounds__(256) void merge_hash_insert_kernel(const float4* __restrict__ recv, long long maxc, int R, MergeBlocks mb, float inv_h,
                                                                unsigned long long* __restrict__ tkeys, unsigned* __restrict__ tcnt,
                                                                unsigned long long mask, unsigned* __restrict__ slot_of, unsigned* __restrict__ rank_of,
                                                                unsigned* __restrict__ bits)
{
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    if (g >= maxc * R) return;
    const int r = (int)(g / maxc);
    if (g - (long long)r * maxc >= mb.cnt[r]) return;
    const float4 p = recv[g];
    const unsigned long long key = merge_cell_key(merge_cell_of(p.x, inv_h), merge_cell_of(p.y, inv_h), merge_cell_of(p.z, inv_h));
    const unsigned long long hk = mix64(key);
    unsigned long long slot = hk & mask;
    for (;;) {
        const unsigned long long prev = atomicCAS(&tkeys[slot], ~0ull, key);
        if (prev == ~0ull || prev == key) break;
        slot = (slot + 1) & mask;
    }
    slot_of[g] = (unsigned)slot;
    rank_of[g] = atomicAdd(&tcnt[slot], 1u);
    const unsigned hb = (unsigned)(hk >> 40); // 24 bits of the hash: the occupied-cell filter (MERGE_FILTER_BITS)
    atomicOr(&bits[hb >> 5], 1u << (hb & 31u));
}

__global__ __launch_bounds__(256) void merge_hash_scatter_kernel(long long maxc, int R, MergeBlocks mb, const unsigned* __restrict__ tstart,
                                                                 const unsigned* __restrict__ slot_of, const unsigned* __restrict__ rank_of,
                                                                 const float4* __restrict__ recv, float4* __restrict__ cell_pts)
{
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    if (g >= maxc * R) return;
    const int r = (int)(g / maxc);
    if (g - (long long)r * maxc >= mb.cnt[r]) return;
    // the point itself rides in the cell-ordered list (w = its global index): a candidate is ONE contiguous 16-byte load, not an
    // index -> flag -> point chain of three dependent ones
    const float4 p = recv[g];
    cell_pts[tstart[slot_of[g]] + rank_of[g]] = make_float4(p.x, p.y, p.z, __uint_as_float((unsigned)g));
}

// flags of block r (first = the lowest non-empty block: everything kept)
__global__ __launch_bounds__(256) void merge_flag_kernel(const float4* __restrict__ recv, long long maxc, int r, long long cnt_r, int first, float inv_h,
                                                         double lim, const unsigned long long* __restrict__ tkeys, const unsigned* __restrict__ tstart,
                                                         unsigned long long mask, const float4* __restrict__ cell_pts, unsigned* __restrict__ kept,
                                                         const unsigned* __restrict__ bits)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= cnt_r) return;
    const long long g = (long long)r * maxc + i;
    if (first) { kept[g] = 1u; return; }
    const float4 p = recv[g];
    const int cx = merge_cell_of(p.x, inv_h), cy = merge_cell_of(p.y, inv_h), cz = merge_cell_of(p.z, inv_h);
    const long long lower_end = (long long)r * maxc; // global indices below this belong to lower blocks
    // Most of the 27 cells around a point are empty, and an empty cell costs a hash table its longest search (probe until a free slot,
    // one dependent round trip per step: 27 of those chains in a row were 50 - 70 us per launch).  A 2^24-bit filter of the occupied
    // cells' hashes answers "empty" for them with ONE load, all 27 in flight together; only the few cells the filter lets through
    // (occupied, or one of its < 2 % false positives) go to the table.
    unsigned cand = 0u;
    {
        unsigned w[27];
#pragma unroll
        for (int q = 0; q < 27; ++q) {
            const unsigned hb = (unsigned)(mix64(merge_cell_key(cx + (q % 3) - 1, cy + ((q / 3) % 3) - 1, cz + (q / 9) - 1)) >> 40);
            w[q] = bits[hb >> 5] >> (hb & 31u);
        }
#pragma unroll
        for (int q = 0; q < 27; ++q) cand |= (w[q] & 1u) << q;
    }
    bool drop = false;
    while (cand && !drop) {
        const int q = __ffs((int)cand) - 1;
        cand &= cand - 1u;
        const unsigned long long key = merge_cell_key(cx + (q % 3) - 1, cy + ((q / 3) % 3) - 1, cz + (q / 9) - 1);
        unsigned long long sl = mix64(key) & mask, k = tkeys[sl];
        while (k != key && k != ~0ull) { sl = (sl + 1) & mask; k = tkeys[sl]; }
        if (k != key) continue; // a false positive of the filter
        const unsigned rbq = tstart[sl], req = tstart[sl + 1];
        for (unsigned j = rbq; j < req && !drop; j += 4u) {
            float4 o[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) o[u] = cell_pts[j + u < req ? j + u : j]; // four candidates in flight
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (j + u >= req) continue;
                const unsigned o_i = __float_as_uint(o[u].w);
                if ((long long)o_i >= lower_end) continue;
                const float d2 = sqdist3(p.x, p.y, p.z, o[u].x, o[u].y, o[u].z);
                if (d2 > 1.1920929e-07f && (double)d2 < lim && kept[o_i]) drop = true; // (the flag only for a candidate that is near: rare)
            }
        }
    }
    kept[g] = drop ? 0u : 1u;
}

__global__ __launch_bounds__(256) void merge_flag_all_kernel(const float4* __restrict__ recv, long long maxc, int R, MergeBlocks mb, int first_block, float inv_h,
                                                         double lim, const unsigned long long* __restrict__ tkeys, const unsigned* __restrict__ tstart,
                                                         unsigned long long mask, const float4* __restrict__ cell_pts, unsigned* __restrict__ kept,
                                                         const unsigned* __restrict__ bits, unsigned* __restrict__ status)
{
    // r5: ALL blocks in one launch.  The greedy rule is a recursion over the rank order -- a point is dropped iff a KEPT point of a lower
    // block is near -- and r4 ran it as R - 1 launches (block r after block r - 1: ~17 us each, the R-dependent part of the epoch).  Here a
    // point that finds a near candidate of a lower block WAITS for that candidate's verdict (status: 0 undecided, 1 dropped, 2 kept;
    // agent-scope atomics) and goes on.  The dependencies point strictly downwards in the global index, workgroups are dispatched in
    // index order, and nothing a workgroup waits for lives in a later workgroup: the recursion always makes progress.  The verdicts are a
    // function of the lower blocks' verdicts alone, so they are the R - 1 launches' verdicts, whatever the timing.
    // (a WAVE never holds points of two blocks -- thread t serves point t % tpb of block t / tpb, tpb = maxc rounded up to 64: a lane that
    //  waits for a lane of its own wave would wait for ever, the other lane's verdict is stored behind the loop both of them are in)
    // EIGHT lanes per point, lane `sub` looks at cells sub, sub + 8, sub + 16, sub + 24 of the 27: a cell costs three to four dependent round
    // trips (probe, bucket bounds, candidates), one lane walking its five to eight occupied cells one after the other was a chain of ~25
    // (111 us for 113 k points at R = 8); the group's verdict is the OR of its lanes' (ballot).
    const long long t = ((long long)blockIdx.x * 256 + threadIdx.x) >> 3;
    const int sub = threadIdx.x & 7;
    const long long tpb = (maxc + 63) & ~63ll;
    const int r = (int)(t / tpb);
    const long long i = t - (long long)r * tpb;
    const bool live = r < R && i < mb.cnt[r < R ? r : 0];
    const long long g = live ? (long long)r * maxc + i : 0;
    bool drop = false;
    if (live && r > first_block) {
        const float4 p = recv[g];
        const int cx = merge_cell_of(p.x, inv_h), cy = merge_cell_of(p.y, inv_h), cz = merge_cell_of(p.z, inv_h);
        const long long lower_end = (long long)r * maxc; // global indices below this belong to lower blocks
        // Most of the 27 cells around a point are empty, and an empty cell costs a hash table its longest search (probe until a free slot).
        // A 2^24-bit filter of the occupied cells' hashes answers "empty" for them with ONE load, all of a lane's in flight together; only
        // the few cells the filter lets through (occupied, or one of its < 2 % false positives) go to the table.
        unsigned cand = 0u;
        {
            unsigned w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int q = sub + 8 * j;
                const int qq = q < 27 ? q : 0;
                const unsigned hb = (unsigned)(mix64(merge_cell_key(cx + (qq % 3) - 1, cy + ((qq / 3) % 3) - 1, cz + (qq / 9) - 1)) >> 40);
                w[j] = q < 27 ? bits[hb >> 5] >> (hb & 31u) : 0u;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) cand |= (w[j] & 1u) << j;
        }
        while (cand && !drop) {
            const int j4 = __ffs((int)cand) - 1;
            cand &= cand - 1u;
            const int q = sub + 8 * j4;
            const unsigned long long key = merge_cell_key(cx + (q % 3) - 1, cy + ((q / 3) % 3) - 1, cz + (q / 9) - 1);
            unsigned long long sl = mix64(key) & mask, k = tkeys[sl];
            while (k != key && k != ~0ull) { sl = (sl + 1) & mask; k = tkeys[sl]; }
            if (k != key) continue; // a false positive of the filter
            const unsigned rbq = tstart[sl], req = tstart[sl + 1];
            for (unsigned j = rbq; j < req && !drop; j += 4u) {
                float4 o[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) o[u] = cell_pts[j + u < req ? j + u : j]; // four candidates in flight
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (j + u >= req) continue;
                    const unsigned o_i = __float_as_uint(o[u].w);
                    if ((long long)o_i >= lower_end) continue;
                    const float d2 = sqdist3(p.x, p.y, p.z, o[u].x, o[u].y, o[u].z);
                    if (d2 > 1.1920929e-07f && (double)d2 < lim) { // a near candidate of a lower block: its verdict decides (rare)
                        unsigned v;
                        while ((v = __hip_atomic_load(&status[o_i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u) __builtin_amdgcn_s_sleep(2);
                        if (v == 2u) drop = true;
                    }
                }
            }
        }
    }
    const unsigned long long dropped = __ballot(drop);
    if (!live || sub != 0) return;
    const bool any = ((dropped >> (threadIdx.x & 56)) & 0xffull) != 0ull;
    kept[g] = any ? 0u : 1u;
    __hip_atomic_store(&status[g], any ? 1u : 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}


// recv: block r = recv[r * maxc .. r * maxc + counts[r]) (maxc = the stride of the padded layout)
static icpmi_status merge_greedy(icpmi_ctx* c, const float4* recv, const std::vector<long long>& counts, long long maxc, float min_dist, int64_t* merged_n)
{
    const int R = (int)counts.size();
    *merged_n = 0;
    const long long span = maxc * R; // padded layout of d_merge_recv: global index g = r * maxc + i
    if (span <= 0) return ICPMI_OK;
    if (span >= (1ll << 31)) { c->last_error = "staged_merge_allgather: gathered set too large"; return ICPMI_ERR_UNSUPPORTED; }
    int first_block = -1, nonempty = 0;
    for (int r = 0; r < R; ++r) if (counts[(size_t)r] > 0) { if (first_block < 0) first_block = r; ++nonempty; }
    if (first_block < 0) return ICPMI_OK;
    // kept flags over the padded layout (padding stays 0), their scan, and the compaction into d_merged
    unsigned* kept = scratch_get<unsigned>(c, 6, (size_t)span + 2);
    unsigned* pos = scratch_get<unsigned>(c, 7, (size_t)span + 2);
    if (!kept || !pos) return ICPMI_ERR_HIP;
    HIP_TRY(c, hipMemsetAsync(kept, 0, ((size_t)span + 2) * sizeof(unsigned), c->stream));
    MergeBlocks mb; memset(&mb, 0, sizeof mb);
    if (R > 256) { c->last_error = "staged_merge_allgather: more than 256 ranks"; return ICPMI_ERR_UNSUPPORTED; }
    for (int r = 0; r < R; ++r) mb.cnt[r] = counts[(size_t)r];
    const bool greedy = nonempty > 1 && min_dist > 0.f;
    const int gblocks = (int)((span + 255) / 256);
    if (!greedy) { // one block, or no distance rule: everything handed in is kept
        for (int r = 0; r < R; ++r)
            if (counts[(size_t)r] > 0)
                hipLaunchKernelGGL(merge_flag_kernel, dim3((int)((counts[(size_t)r] + 255) / 256)), dim3(256), 0, c->stream, recv, maxc, r,
                                   counts[(size_t)r], 1, 0.f, 0.0, (const unsigned long long*)nullptr, (const unsigned*)nullptr, 0ull, (const float4*)nullptr, kept, (const unsigned*)nullptr);
    } else {
        long long total = 0;
        for (long long v : counts) total += v;
        // the table has 2 .. 4 x total slots and its scan length is an int: 2^29 gathered points would make it 2^31
        if (total >= (1ll << 29)) { c->last_error = "staged_merge_allgather: gathered set too large (2^29 points or more)"; return ICPMI_ERR_UNSUPPORTED; }
        unsigned long long cap = 1024;
        while (cap < 2ull * (unsigned long long)total) cap <<= 1;
        // table: keys (u64) | counts -> starts (u32, cap + 1) ; per point: slot, rank ; cell-ordered point list
        unsigned long long* tkeys = scratch_get<unsigned long long>(c, 0, (size_t)cap);
        unsigned* tcnt = scratch_get<unsigned>(c, 1, (size_t)cap + 2);
        unsigned* slot_of = scratch_get<unsigned>(c, 2, (size_t)span + 1);
        unsigned* rank_of = scratch_get<unsigned>(c, 3, (size_t)span + 1);
        float4* cell_pts = scratch_get<float4>(c, 4, (size_t)total + 1);
        if (!tkeys || !tcnt || !slot_of || !rank_of || !cell_pts) return ICPMI_ERR_HIP;
        unsigned* bits = scratch_get<unsigned>(c, 5, (size_t)MERGE_FILTER_WORDS);
        if (!bits) return ICPMI_ERR_HIP;
        HIP_TRY(c, hipMemsetAsync(bits, 0, (size_t)MERGE_FILTER_WORDS * sizeof(unsigned), c->stream));
        HIP_TRY(c, hipMemsetAsync(tkeys, 0xff, (size_t)cap * sizeof(unsigned long long), c->stream));
        HIP_TRY(c, hipMemsetAsync(tcnt, 0, ((size_t)cap + 2) * sizeof(unsigned), c->stream));
        // cell edge: a little above minDist (a relative 1e-3 dwarfs the rounding of v * inv_h for coordinates up to ~10^4 cells from the origin;
        // beyond that the margin grows with the coordinate's ulp)
        const float h = min_dist * 1.001f + 1e-6f;
        const float inv_h = 1.0f / h;
        hipLaunchKernelGGL(merge_hash_insert_kernel, dim3(gblocks), dim3(256), 0, c->stream, recv, maxc, R, mb, inv_h, tkeys, tcnt,
                           cap - 1, slot_of, rank_of, bits);
        HIP_TRY(c, hipGetLastError());
        icpmi_status s = device_exclusive_scan(c, tcnt, (int)cap, (unsigned)total);
        if (s != ICPMI_OK) return s;
        hipLaunchKernelGGL(merge_hash_scatter_kernel, dim3(gblocks), dim3(256), 0, c->stream, maxc, R, mb, (const unsigned*)tcnt, (const unsigned*)slot_of,
                           (const unsigned*)rank_of, recv, cell_pts);
        const double lim = pd_limit(min_dist);
        // (sets too large to trust co-dispatch take the R - 1 launches of r4)
        if (span <= (1ll << 22)) {
            unsigned* status = scratch_get<unsigned>(c, 8, (size_t)span + 2);
            if (!status) return ICPMI_ERR_HIP;
            HIP_TRY(c, hipMemsetAsync(status, 0, ((size_t)span + 2) * sizeof(unsigned), c->stream));
            const long long tpb = (maxc + 63) & ~63ll;
            hipLaunchKernelGGL(merge_flag_all_kernel, dim3((unsigned)((tpb * R * 8 + 255) / 256)), dim3(256), 0, c->stream, recv, maxc, R, mb, first_block, inv_h, lim, (const unsigned long long*)tkeys,
                               (const unsigned*)tcnt, cap - 1, (const float4*)cell_pts, kept, (const unsigned*)bits, status);
        } else
        for (int r = first_block; r < R; ++r) {
            if (counts[(size_t)r] == 0) continue;
            hipLaunchKernelGGL(merge_flag_kernel, dim3((int)((counts[(size_t)r] + 255) / 256)), dim3(256), 0, c->stream, recv, maxc, r,
                               counts[(size_t)r], r == first_block ? 1 : 0, inv_h, lim, (const unsigned long long*)tkeys, (const unsigned*)tcnt, cap - 1,
                               (const float4*)cell_pts, kept, (const unsigned*)bits);
        }
        HIP_TRY(c, hipGetLastError());
    }
    // stable compaction in global order = rank order, then input order inside a block (what the per-block appends of r3 produced)
    int64_t n_kept = 0;
    icpmi_status s = merge_append_flagged(c, recv, span, kept, pos, c->d_merged, 0, &n_kept);
    if (s != ICPMI_OK) return s;
    *merged_n = n_kept;
    return ICPMI_OK;
}

// ---- r5: the one-collective epoch ------------------------------------------------------------------------------------------------------
// like merge_compact_kernel, but never past `cap` points (a block has a fixed size; the header carries the true count, an overflow sends
// every rank to the three-collective epoch)
__global__ __launch_bounds__(256) void merge_compact_cap_kernel(const float4* __restrict__ in, int64_t n, const unsigned* __restrict__ flag,
                                                                const unsigned* __restrict__ pos, float4* __restrict__ out, unsigned cap)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || !flag[i]) return;
    const unsigned p = pos[i];
    if (p < cap) out[p] = in[i];
}
// header of this rank's block: {bits(count) | bits(-1): this rank failed before the exchange, magic}
__global__ void merge_header_kernel(float4* __restrict__ hdr, const unsigned* __restrict__ count_word, int ok, unsigned fixed_count, int use_fixed)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const unsigned cnt = ok ? (use_fixed ? fixed_count : *count_word) : 0xffffffffu;
        *hdr = make_float4(__uint_as_float(cnt), __uint_as_float(ICPMI_MERGE_MAGIC), 0.f, 0.f);
    }
}
// the R headers of the gathered blocks -> host-mapped words (count, or 0xfffffffe for a header without the magic)
__global__ void merge_headers_out_kernel(const float4* __restrict__ recv, size_t block4, int R, unsigned* __restrict__ out)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    const float4 h = recv[(size_t)r * block4];
    out[r] = __float_as_uint(h.y) == ICPMI_MERGE_MAGIC ? __float_as_uint(h.x) : 0xfffffffeu;
}

// One collective per epoch: every rank hands in ONE fixed-size block {header, <= merge_block points}.  No host read-back in front of the
// collective (the accept count goes from the scan straight into the header), none between collectives (there is only one), one stream wait
// behind it (the R counts arrive in host-mapped memory).  Everything a rank can fail at before the exchange travels as count -1; buffers
// were sized with the communicator.  *fall_back: some rank accepted more points than a block holds -- every rank sees the same headers
// and takes the three-collective epoch for this scan (nothing has been appended yet).
static icpmi_status merge_epoch_one_collective(icpmi_ctx* c, const float correction[16], float min_dist, int64_t n, std::vector<long long>& counts,
                                               const float4** gathered, long long* stride, bool* fall_back, int64_t* mine_out)
{
    *fall_back = false; *mine_out = 0;
    const int R = c->comm_ranks;
    const int64_t cap = c->merge_block;
    const size_t block4 = (size_t)cap + 1;
    float4* send = c->d_merge_send;
    unsigned* d_count = reinterpret_cast<unsigned*>(c->d_comm_cnt); // (one word: the scan's sum)
    icpmi_status local = ICPMI_OK;
    bool fixed = true; unsigned fixed_count = 0;
    if (n > 0) {
        unsigned* d_flag = scratch_get<unsigned>(c, 6, (size_t)n + 2);
        unsigned* d_pos = scratch_get<unsigned>(c, 7, (size_t)n + 2);
        if (!d_flag || !d_pos) local = ICPMI_ERR_HIP;
        if (local == ICPMI_OK) local = ensure_cap(c, &c->d_stage_in, &c->cap_stage_in, (size_t)n + 1);
        if (local == ICPMI_OK) local = ops_transform_dev(c, correction, c->d_scan_map, n, c->d_stage_in);
        if (local == ICPMI_OK) {
            if (c->m > 0) {
                icpmi_ctx* ri = nullptr;
                local = raw_index(c, &ri, min_dist);
                if (local == ICPMI_OK) local = chain_point_distance_flags(c, ri, c->d_stage_in, n, min_dist, d_flag);
                if (local == ICPMI_OK) local = device_exclusive_scan_sum(c, d_flag, d_pos, (int)n, d_count);
                if (local == ICPMI_OK) {
                    hipLaunchKernelGGL(merge_compact_cap_kernel, dim3((int)((n + 255) / 256)), dim3(256), 0, c->stream, (const float4*)c->d_stage_in, n,
                                       (const unsigned*)d_flag, (const unsigned*)d_pos, send + 1, (unsigned)cap);
                    fixed = false;
                }
            } else { // no map yet: every point is new
                const int64_t cp = n < cap ? n : cap;
                if (hipMemcpyAsync(send + 1, c->d_stage_in, (size_t)cp * sizeof(float4), hipMemcpyDeviceToDevice, c->stream) != hipSuccess) local = ICPMI_ERR_HIP;
                fixed_count = (unsigned)n;
            }
        }
    }
    const std::string local_error = c->last_error;
    hipLaunchKernelGGL(merge_header_kernel, dim3(1), dim3(64), 0, c->stream, send, (const unsigned*)d_count, local == ICPMI_OK ? 1 : 0, fixed_count, fixed ? 1 : 0);
    // (collective discipline: nothing returns between here and the exchange -- a rank that left now would strand its peers inside the all-gather;
    //  a launch that failed is reported after the collective, whose own error every rank sees)
    const hipError_t header_launch = hipGetLastError();
    // ---- THE collective (a single rank without a communicator: its block is the gathered set)
    const float4* recv = send;
    if (c->comm || R > 1) {
        const icpmi_status s = comm_allgather_blocks(c, send, c->d_merge_recv, block4);
        if (s != ICPMI_OK) return s;
        recv = c->d_merge_recv;
    }
    HIP_TRY(c, header_launch);
    // ---- the R counts: host-mapped words, one wait
    unsigned* d_hdr = c->d_progress + ICPMI_PROGRESS_HDR_WORD;
    volatile unsigned* h_hdr = c->h_progress + ICPMI_PROGRESS_HDR_WORD;
    hipLaunchKernelGGL(merge_headers_out_kernel, dim3((R + 255) / 256), dim3(256), 0, c->stream, recv, block4, R, d_hdr);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    counts.assign((size_t)R, 0);
    bool over = false;
    for (int r = 0; r < R; ++r) {
        const unsigned v = h_hdr[r];
        if (v == 0xffffffffu || v == 0xfffffffeu) { // every rank leaves the epoch together
            if (local != ICPMI_OK) { c->last_error = local_error; return local; }
            c->last_error = "staged_merge_allgather: rank " + std::to_string(r) + (v == 0xffffffffu ? " failed before the exchange" : " sent a block without the header");
            return ICPMI_ERR_HIP;
        }
        counts[(size_t)r] = (long long)v;
        over |= (int64_t)v > cap;
    }
    *mine_out = counts[(size_t)c->comm_rank];
    if (over) { *fall_back = true; return ICPMI_OK; }
    *gathered = recv + 1; *stride = (long long)block4;
    return ICPMI_OK;
}

// One map-growth epoch of the scan-sharded mapper (include/icpmi.h: icpmi_staged_merge_allgather), device-resident.
// Collective discipline: a rank must never leave between two collectives its peers are going to enter.
//   * every failure before the count exchange travels as count -1 (all ranks then leave together, nobody appends);
//   * what can fail between the count exchange and the point exchange (growing the buffers to the gathered sizes) is reported in a
//     second one-word exchange (`ready`), so a rank that cannot allocate does not leave its peers in the point all-gather;
//   * after the point exchange every rank holds the same blocks and runs the same deterministic merge + append; the merged set is
//     ALWAYS appended -- a caller's copy-out buffer that is too small gets what fits and learns the full size from *merged_n
//     (fetch the rest with icpmi_staged_merged_points): the replicas cannot diverge over a host buffer.
icpmi_status ops_staged_merge_allgather(icpmi_ctx* c, const float correction[16], float min_dist, int normals_knn, int64_t* accepted_local,
                                        int64_t* appended_total, int64_t* new_m, float* merged_out4, int64_t merged_capacity, int64_t* merged_n)
{
    if (accepted_local) *accepted_local = 0;
    if (appended_total) *appended_total = 0;
    if (merged_n) *merged_n = 0;
    if (new_m) *new_m = c->m > 0 ? c->m_raw : 0;
    c->merged_last_n = 0;
    c->merged_binned = false;
    c->cells_enq_n = 0;
    // ICPMI_EPOCH_TIMING=1: wall time of the epoch's stages with a stream wait behind each, on stderr (diagnostic; perturbs the overlap)
    static const bool ep_timing = [] { const char* e = getenv("ICPMI_EPOCH_TIMING"); return e && atoi(e) != 0; }();
    auto ep_tick = [&](const char* what) {
        static thread_local std::chrono::steady_clock::time_point t0;
        if (!ep_timing) return;
        (void)hipStreamSynchronize(c->stream);
        const auto t1 = std::chrono::steady_clock::now();
        if (what) fprintf(stderr, "[icpmi epoch] %-22s %8.1f us\n", what, std::chrono::duration<double, std::micro>(t1 - t0).count());
        t0 = std::chrono::steady_clock::now();
    };
    ep_tick(nullptr);
    const int R = c->comm_ranks; // 1 without a communicator (or the loopback communicator's simulated ranks, comm.hip)
    const int64_t n = correction ? c->scan_map_n : 0; // no correction = nothing to contribute (empty scan / failed registration)
    // the exchange words: allocated by comm_init; a handle that is its own single rank gets them here (no peer can be left waiting)
    if (!c->d_comm_cnt || c->cap_comm_cnt < (size_t)2 * R + 16) {
        if (c->comm) { c->last_error = "staged_merge_allgather: communicator without exchange words"; return ICPMI_ERR_HIP; }
        if (ensure_cap(c, &c->d_comm_cnt, &c->cap_comm_cnt, (size_t)2 * R + 16) != ICPMI_OK) return ICPMI_ERR_HIP;
    }
    // ---- r5: ONE collective (fixed-size blocks with a count header); the three-collective epoch below stays as the fall-back
    if (c->merge_block == 0 && !c->comm && R == 1) { const icpmi_status rs = merge_blocks_reserve(c, 1); if (rs != ICPMI_OK) return rs; } // (its own single rank: no peer to strand)
    int64_t acc = 0;
    bool served = false;
    if (c->merge_block > 0) {
        std::vector<long long> counts;
        const float4* gathered = nullptr; long long stride = 0; bool fall_back = false; int64_t mine1 = 0;
        icpmi_status s1 = merge_epoch_one_collective(c, correction, min_dist, n, counts, &gathered, &stride, &fall_back, &mine1);
        if (s1 != ICPMI_OK) return s1;
        ep_tick("accept + collective");
        if (!fall_back) {
            if (accepted_local) *accepted_local = mine1;
            long long total = 0;
            for (long long v : counts) total += v;
            ++c->merge_fast_epochs;
            if (total == 0) { if (new_m) *new_m = c->m > 0 ? c->m_raw : 0; return ICPMI_OK; }
            s1 = merge_greedy(c, gathered, counts, stride, min_dist, &acc);
            if (s1 != ICPMI_OK) return s1;
            ep_tick("rank-ordered merge");
            served = true;
        }
    }
    if (!served) {
    ++c->merge_slow_epochs;
    // ---- this rank's accepted points: the staged scan moved by the correction (Mapper.cpp:221), PointDistance against the resident map
    icpmi_status local = ICPMI_OK;
    int64_t mine = 0;
    unsigned* d_flag = nullptr; unsigned* d_pos = nullptr;
    if (n > 0) {
        d_flag = scratch_get<unsigned>(c, 6, (size_t)n + 2);
        d_pos = scratch_get<unsigned>(c, 7, (size_t)n + 2);
        if (!d_flag || !d_pos) local = ICPMI_ERR_HIP;
        if (local == ICPMI_OK) local = ensure_cap(c, &c->d_stage_in, &c->cap_stage_in, (size_t)n + 1);
        if (local == ICPMI_OK) local = ensure_cap(c, &c->d_merge_send, &c->cap_merge_send, (size_t)n + 1);
        if (local == ICPMI_OK) local = ops_transform_dev(c, correction, c->d_scan_map, n, c->d_stage_in);
        if (local == ICPMI_OK) {
            if (c->m > 0) {
                icpmi_ctx* ri = nullptr;
                local = raw_index(c, &ri, min_dist);
                if (local == ICPMI_OK) local = chain_point_distance_flags(c, ri, c->d_stage_in, n, min_dist, d_flag);
                if (local == ICPMI_OK) local = merge_append_flagged(c, c->d_stage_in, n, d_flag, d_pos, c->d_merge_send, 0, &mine);
            } else { // no map yet: every point is new
                if (hipMemcpyAsync(c->d_merge_send, c->d_stage_in, (size_t)n * sizeof(float4), hipMemcpyDeviceToDevice, c->stream) != hipSuccess) local = ICPMI_ERR_HIP;
                mine = n;
            }
        }
    }
    const std::string local_error = c->last_error;
    if (accepted_local) *accepted_local = local == ICPMI_OK ? mine : 0;
    // ---- counts of all ranks (collective 1)
    long long* d_cnt = c->d_comm_cnt;
    long long hmine = local == ICPMI_OK ? mine : -1;
    { const icpmi_status us = upload_small(c, d_cnt + R, &hmine, sizeof hmine); if (us != ICPMI_OK) return us; }
    icpmi_status s = comm_allgather(c, d_cnt + R, d_cnt, 1, false);
    if (s != ICPMI_OK) return s;
    std::vector<long long> counts((size_t)R);
    if (read_back(c, counts.data(), d_cnt, sizeof(long long) * (size_t)R) != ICPMI_OK) return ICPMI_ERR_HIP;
    long long maxc = 0, total = 0;
    for (int r = 0; r < R; ++r)
        if (counts[(size_t)r] < 0) { // every rank leaves the epoch together
            if (local != ICPMI_OK) { c->last_error = local_error; return local; }
            c->last_error = "staged_merge_allgather: rank " + std::to_string(r) + " failed before the exchange";
            return ICPMI_ERR_HIP;
        }
    for (long long v : counts) { maxc = v > maxc ? v : maxc; total += v; }
    if (total == 0) { if (new_m) *new_m = c->m > 0 ? c->m_raw : 0; return ICPMI_OK; }
    // ---- buffers at the gathered sizes; `ready` exchange (collective 2, only when there are peers)
    icpmi_status grow = ensure_cap_keep(c, &c->d_merge_send, &c->cap_merge_send, (size_t)maxc + 1, (size_t)mine);
    if (grow == ICPMI_OK) grow = ensure_cap(c, &c->d_merge_recv, &c->cap_merge_recv, (size_t)maxc * R + 1);
    if (grow == ICPMI_OK) grow = ensure_cap(c, &c->d_merged, &c->cap_merged, (size_t)total + 1);
    const std::string grow_error = c->last_error;
    if (c->comm) {
        long long hready = grow == ICPMI_OK ? 1 : -1;
        { const icpmi_status us = upload_small(c, d_cnt + R, &hready, sizeof hready); if (us != ICPMI_OK) return us; }
        s = comm_allgather(c, d_cnt + R, d_cnt, 1, false);
        if (s != ICPMI_OK) return s;
        std::vector<long long> ready((size_t)R);
        if (read_back(c, ready.data(), d_cnt, sizeof(long long) * (size_t)R) != ICPMI_OK) return ICPMI_ERR_HIP;
        for (int r = 0; r < R; ++r)
            if (ready[(size_t)r] < 0) {
                if (grow != ICPMI_OK) { c->last_error = grow_error; return grow; }
                c->last_error = "staged_merge_allgather: rank " + std::to_string(r) + " could not size its exchange buffers";
                return ICPMI_ERR_HIP;
            }
    } else if (grow != ICPMI_OK) return grow;
    // ---- the point blocks, padded to the largest (collective 3; the payload is a few MB at most: latency, not bandwidth)
    s = comm_allgather(c, c->d_merge_send, c->d_merge_recv, (size_t)maxc * 4, true);
    if (s != ICPMI_OK) return s;
    // ---- merge in rank order; block r keeps what is at least min_dist from the points accepted from ranks < r
    // r4: ONE spatial hash over all gathered blocks and R - 1 flag passes in rank order instead of one temporary index (a dozen launches and
    // two read-backs) per block -- the same greedy rule, the same candidates' float distances, the same kept set (merge_greedy).
    s = merge_greedy(c, c->d_merge_recv, counts, maxc, min_dist, &acc);
    if (s != ICPMI_OK) return s;
    }
    // ---- r6: the merged set into the mapper's cells (cells.hip), when the handle was told to (icpmi_cell_log_configure): enqueued here, in
    //      front of the append, the table collected by icpmi_staged_bin_cells behind the epoch's last wait
    c->merged_last_n = acc;
    c->cells_enq_n = 0;
    { const icpmi_status cs = ops_cells_enqueue_in_epoch(c); if (cs != ICPMI_OK) { c->merged_last_n = 0; return cs; } }
    c->merged_last_n = 0;
    // ---- every replica appends the same set (all points kept: the distance tests are done), normals, index
    int64_t app = 0, m1 = 0;
    icpmi_status s = ops_map_update_dev(c, c->d_merged, acc, nullptr, 0.f, normals_knn, nullptr, &app, &m1);
    if (s != ICPMI_OK) return s;
    ep_tick("append + index insert");
    c->merged_last_n = acc;
    if (merged_n) *merged_n = acc;
    if (merged_out4 && merged_capacity > 0) { // what fits; *merged_n says how much there is
        const int64_t cp = acc < merged_capacity ? acc : merged_capacity;
        if (cp > 0) {
            HIP_TRY(c, hipMemcpyAsync(merged_out4, c->d_merged, (size_t)cp * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream)); // (only the copy-out needs the wait: the index update above has synchronised already)
        }
    }
    if (appended_total) *appended_total = app;
    if (new_m) *new_m = m1;
    return ICPMI_OK;
}

// the merged set of the last epoch (still in d_merged): out4 may be NULL to query *n only
icpmi_status ops_staged_merged_points(icpmi_ctx* c, float* out4, int64_t capacity, int64_t* n)
{
    const int64_t have = c->merged_last_n;
    if (n) *n = have;
    if (!out4 || have == 0) return ICPMI_OK;
    if (capacity < have) { c->last_error = "staged_merged_points: capacity too small"; return ICPMI_ERR_INVALID_ARG; }
    HIP_TRY(c, hipMemcpyAsync(out4, c->d_merged, (size_t)have * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return ICPMI_OK;
}

icpmi_status ops_staged_keep(icpmi_ctx* c, const float correction[16], float min_dist, uint8_t* keep_out, float* placed_out4)
{
    const int64_t n = c->scan_map_n;
    if (ensure_cap(c, &c->d_stage_in, &c->cap_stage_in, (size_t)n + 1) != ICPMI_OK) return ICPMI_ERR_HIP;
    icpmi_status s = ops_transform_dev(c, correction, c->d_scan_map, n, c->d_stage_in); // Mapper.cpp:221
    if (s != ICPMI_OK) return s;
    if (placed_out4) HIP_TRY(c, hipMemcpyAsync(placed_out4, c->d_stage_in, (size_t)n * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    if (c->m <= 0) { // no map yet: every point is new
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        memset(keep_out, 1, (size_t)n);
        return ICPMI_OK;
    }
    unsigned* d_flag = scratch_get<unsigned>(c, 6, (size_t)n + 2);
    uint8_t* d_keep = scratch_get<uint8_t>(c, 9, (size_t)n);
    if (!d_flag || !d_keep) return ICPMI_ERR_HIP;
    icpmi_ctx* ri = nullptr;
    s = raw_index(c, &ri, min_dist);
    if (s == ICPMI_OK) s = chain_point_distance_flags(c, ri, c->d_stage_in, n, min_dist, d_flag);
    if (s != ICPMI_OK) return s;
    hipLaunchKernelGGL(flag_to_keep_kernel, dim3((int)((n + 255) / 256)), dim3(256), 0, c->stream, (const unsigned*)d_flag, n, d_keep);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(keep_out, d_keep, (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return ICPMI_OK;
}

icpmi_status ops_map_scalar(icpmi_ctx* c, const float* set, float* get, int64_t m)
{
    if (c->m <= 0 || m != c->m_raw) { c->last_error = "map_scalar: size differs from the resident map"; return ICPMI_ERR_INVALID_ARG; }
    if (set) {
        if (ensure_cap_keep(c, &c->d_raw_s, &c->cap_raw_s, (size_t)m, 0) != ICPMI_OK) return ICPMI_ERR_HIP;
        HIP_TRY(c, hipMemcpyAsync(c->d_raw_s, set, (size_t)m * sizeof(float), hipMemcpyHostToDevice, c->stream));
        c->raw_has_scalar = true;
    }
    if (get) {
        if (!c->raw_has_scalar) { c->last_error = "InvalidField: the map has no tracked scalar descriptor"; return ICPMI_ERR_INVALID_ARG; }
        HIP_TRY(c, hipMemcpyAsync(get, c->d_raw_s, (size_t)m * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return ICPMI_OK;
}
