// solve.h -- the single-lane part of an ICP iteration (ErrorMinimizer::compute's algebra, T_iter = T_step T_iter, TransformationCheckers;
// SURVEY.md B.5 - B.8) and the reader of the fixed-point pair-sum accumulators.  Included by loop.hip (solve_kernel).  (A header since r4,
// when nn.hip's nn1_wg_kernel also ran it in every workgroup's prologue; that variant was measured again and removed in r5)
#pragma once
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------------
// small dense algebra for the single-lane solve
// ---------------------------------------------------------------------------------------------
__device__ void mat4_mul_dev(const float* A, const float* B, float* C)
{
    float R[16];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float s = A[i] * B[4 * j];
#pragma unroll
            for (int kk = 1; kk < 4; ++kk) s = fmaf(A[4 * kk + i], B[4 * j + kk], s);
            R[4 * j + i] = s;
        }
#pragma unroll
    for (int i = 0; i < 16; ++i) C[i] = R[i];
}

// symmetric Jacobi eigen-decomposition (double), n <= 6, col-major
__device__ void jacobi_eig(int n, const double* Ain, double* w, double* Q)
{
    __shared__ double A[36]; // (one lane per workgroup runs the solver: its work arrays live in LDS, not in scratch memory -- see solve_serial)
    for (int i = 0; i < n * n; ++i) A[i] = Ain[i];
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) Q[n * j + i] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0;
        for (int p = 0; p < n; ++p) for (int q = p + 1; q < n; ++q) off += A[n * q + p] * A[n * q + p];
        double dg = 0;
        for (int p = 0; p < n; ++p) dg += A[n * p + p] * A[n * p + p];
        if (off <= 1e-32 * dg || off < 1e-300) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = A[n * q + p];
                if (apq == 0.0) continue;
                const double theta = (A[n * q + q] - A[n * p + p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int kk = 0; kk < n; ++kk) {
                    const double akp = A[n * p + kk], akq = A[n * q + kk];
                    A[n * p + kk] = c * akp - s * akq; A[n * q + kk] = s * akp + c * akq;
                }
                for (int kk = 0; kk < n; ++kk) {
                    const double apk = A[n * kk + p], aqk = A[n * kk + q];
                    A[n * kk + p] = c * apk - s * aqk; A[n * kk + q] = s * apk + c * aqk;
                }
                for (int kk = 0; kk < n; ++kk) {
                    const double qkp = Q[n * p + kk], qkq = Q[n * q + kk];
                    Q[n * p + kk] = c * qkp - s * qkq; Q[n * q + kk] = s * qkp + c * qkq;
                }
            }
    }
    for (int i = 0; i < n; ++i) w[i] = A[n * i + i];
}

// Rotation of the point-to-point minimiser: R = U V^T of the float 3x3 H = U S V^T,
// with the last row of V^T negated when det(R) < 0.  The SVD is a one-sided (Hestenes) Jacobi in
// float -- the numeric spec shared with the oracle: cyclic (0,1),(0,2),(1,2) sweeps on the columns,
// rotation skipped below 1e-9 relative off-diagonal, stop at 1e-7, singular values sorted
// descending, columns of U belonging to (near) zero singular values completed by cross products.
__device__ void svd3f_dev(const float* H, float* U, float* s, float* V)
{
    float a[9], v[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int i = 0; i < 9; ++i) a[i] = H[i];
    for (int sweep = 0; sweep < 30; ++sweep) {
        float off = 0.f;
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int q = p + 1; q < 3; ++q) {
                float alpha = 0, beta = 0, gamma = 0;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    alpha += a[3 * p + i] * a[3 * p + i];
                    beta += a[3 * q + i] * a[3 * q + i];
                    gamma += a[3 * p + i] * a[3 * q + i];
                }
                if (gamma == 0.f) continue;
                const float lim = fabsf(gamma) / sqrtf(fmaxf(alpha * beta, 1.17549435e-38f));
                if (lim > off) off = lim;
                if (lim <= 1e-9f) continue;
                const float zeta = (beta - alpha) / (2.f * gamma);
                const float tt = (zeta >= 0.f ? 1.f : -1.f) / (fabsf(zeta) + sqrtf(1.f + zeta * zeta));
                const float c = 1.f / sqrtf(1.f + tt * tt), sn = c * tt;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const float ap = a[3 * p + i], aq = a[3 * q + i];
                    a[3 * p + i] = c * ap - sn * aq; a[3 * q + i] = sn * ap + c * aq;
                    const float vp = v[3 * p + i], vq = v[3 * q + i];
                    v[3 * p + i] = c * vp - sn * vq; v[3 * q + i] = sn * vp + c * vq;
                }
            }
        if (off <= 1e-7f) break;
    }
    float sv[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) sv[j] = sqrtf(a[3 * j] * a[3 * j] + a[3 * j + 1] * a[3 * j + 1] + a[3 * j + 2] * a[3 * j + 2]);
    // descending order by the exchange sort (0,1) (0,2) (1,2), strict comparisons: columns travel with their
    // singular value.  Static indices only -- a permutation array would push a, v and sv into scratch.
#define SVD3_CSWAP(I, J)                                                                      \
    if (sv[J] > sv[I]) {                                                                      \
        float t_ = sv[I]; sv[I] = sv[J]; sv[J] = t_;                                          \
        _Pragma("unroll") for (int i_ = 0; i_ < 3; ++i_) {                                    \
            t_ = a[3 * I + i_]; a[3 * I + i_] = a[3 * J + i_]; a[3 * J + i_] = t_;            \
            t_ = v[3 * I + i_]; v[3 * I + i_] = v[3 * J + i_]; v[3 * J + i_] = t_;            \
        }                                                                                     \
    }
    SVD3_CSWAP(0, 1)
    SVD3_CSWAP(0, 2)
    SVD3_CSWAP(1, 2)
#undef SVD3_CSWAP
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        s[j] = sv[j];
#pragma unroll
        for (int i = 0; i < 3; ++i) { V[3 * j + i] = v[3 * j + i]; U[3 * j + i] = a[3 * j + i]; }
    }
    // U = A V S^-1 made orthonormal by construction -- same operations, same order as the oracle (see there)
    const float tiny = s[0] * 1e-6f;
    if (!(s[0] > 0.f)) {
#pragma unroll
        for (int i = 0; i < 9; ++i) U[i] = (i % 4 == 0) ? 1.f : 0.f;
        return;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) U[i] /= s[0];
    {
        const float n0 = sqrtf(U[0] * U[0] + U[1] * U[1] + U[2] * U[2]);
#pragma unroll
        for (int i = 0; i < 3; ++i) U[i] /= n0;
    }
    bool have1 = false;
    if (s[1] > tiny) {
        float c1[3] = {U[3] / s[1], U[4] / s[1], U[5] / s[1]};
        const float d = c1[0] * U[0] + c1[1] * U[1] + c1[2] * U[2];
#pragma unroll
        for (int i = 0; i < 3; ++i) c1[i] = c1[i] - d * U[i];
        const float n1 = sqrtf(c1[0] * c1[0] + c1[1] * c1[1] + c1[2] * c1[2]);
        if (n1 > 0.5f) {
#pragma unroll
            for (int i = 0; i < 3; ++i) U[3 + i] = c1[i] / n1;
            have1 = true;
        }
    }
    if (!have1) {
        const int m = fabsf(U[0]) < fabsf(U[1]) ? (fabsf(U[0]) < fabsf(U[2]) ? 0 : 2) : (fabsf(U[1]) < fabsf(U[2]) ? 1 : 2);
        const float e[3] = {m == 0 ? 1.f : 0.f, m == 1 ? 1.f : 0.f, m == 2 ? 1.f : 0.f};
        const float d = m == 0 ? U[0] : (m == 1 ? U[1] : U[2]);
        const float w[3] = {e[0] - d * U[0], e[1] - d * U[1], e[2] - d * U[2]};
        const float nw = sqrtf(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
#pragma unroll
        for (int i = 0; i < 3; ++i) U[3 + i] = w[i] / nw;
    }
    {
        const float c2[3] = {U[6], U[7], U[8]};
        const float x[3] = {U[1] * U[5] - U[2] * U[4], U[2] * U[3] - U[0] * U[5], U[0] * U[4] - U[1] * U[3]};
        const float nx = sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
        const float sg = (have1 && s[2] > tiny && (x[0] * c2[0] + x[1] * c2[1] + x[2] * c2[2]) < 0.f) ? -1.f : 1.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) U[6 + i] = sg * (x[i] / nx);
    }
}

__device__ float det3f(const float* R)
{
    return R[0] * (R[4] * R[8] - R[7] * R[5]) - R[3] * (R[1] * R[8] - R[7] * R[2]) + R[6] * (R[1] * R[5] - R[4] * R[2]);
}

// the route through the SVD: singular or reflecting H only (see rotation_from_H) -- out of line, its registers and its
// thirty-sweep loop stay off the common path
__device__ __noinline__ void rotation_from_H_svd(const float* H, float* R)
{
    __shared__ float U[9], s[3], V[9];
    svd3f_dev(H, U, s, V);
    for (int pass = 0; pass < 2; ++pass) {
        for (int j = 0; j < 3; ++j)
            for (int i = 0; i < 3; ++i) {
                float acc = 0.f;
                for (int kk = 0; kk < 3; ++kk) acc += U[3 * kk + i] * V[3 * kk + j];
                R[3 * j + i] = acc;
            }
        if (pass == 0 && det3f(R) < 0.f) { for (int i = 0; i < 3; ++i) V[6 + i] = -V[6 + i]; }
        else break;
    }
}

// U V^T of H = U S V^T is the orthogonal polar factor of H whenever det H > 0, and the Newton iteration X <- (X + X^-T) / 2
// reaches it without U, S, V: ~4 iterations of ~60 instructions against ~18 Jacobi rotations of ~140 (the single-lane
// solve went from ~12 k to ~2.5 k clocks).  Same operations, same order as the oracle (polar_newton3f there).
__device__ bool polar_newton3f(const float* H, float* R)
{
    float n2 = 0.f;
#pragma unroll
    for (int i = 0; i < 9; ++i) n2 = fmaf(H[i], H[i], n2);
    if (!(n2 > 0.f) || n2 == INFINITY) return false;
    const float inv = 1.f / sqrtf(n2);
    float X[9], C[9], Y[9], Xn[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) X[i] = H[i] * inv;
    for (int it = 0; it < 20; ++it) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int a = (i + 1) % 3, b = (i + 2) % 3, c = (j + 1) % 3, d = (j + 2) % 3;
                const float t = X[3 * d + a] * X[3 * c + b];
                C[3 * j + i] = fmaf(X[3 * c + a], X[3 * d + b], -t);
            }
        float det = X[0] * C[0];
        det = fmaf(X[3], C[3], det);
        det = fmaf(X[6], C[6], det);
        if (it == 0 && !(det > 1e-6f)) return false;
        const float invdet = 1.f / det;
#pragma unroll
        for (int i = 0; i < 9; ++i) Y[i] = C[i] * invdet;
        if (it < 2) {
            float nx = 0.f, ny = 0.f;
#pragma unroll
            for (int i = 0; i < 9; ++i) { nx = fmaf(X[i], X[i], nx); ny = fmaf(Y[i], Y[i], ny); }
            const float mu = sqrtf(sqrtf(ny / nx)), imu = 1.f / mu;
#pragma unroll
            for (int i = 0; i < 9; ++i) Xn[i] = 0.5f * fmaf(mu, X[i], Y[i] * imu);
        } else {
#pragma unroll
            for (int i = 0; i < 9; ++i) Xn[i] = 0.5f * (X[i] + Y[i]);
        }
        float dmax = 0.f;
#pragma unroll
        for (int i = 0; i < 9; ++i) { const float dd = fabsf(Xn[i] - X[i]); if (dd > dmax) dmax = dd; X[i] = Xn[i]; }
        if (!(dmax == dmax)) return false;
        if (it >= 2 && dmax <= 3e-4f) break;
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = X[i];
    return true;
}

__device__ void rotation_from_H(const float* H, float* R)
{
    if (!polar_newton3f(H, R)) rotation_from_H_svd(H, R);
}

// solvePossiblyUnderdeterminedLinearSystem: float LLT when A is invertible, else
// the minimum-norm solution (double symmetric pseudo-inverse) -- same rule as the oracle.  N = 6, or 4 (force4DOF).
// minimum-norm branch (rank-deficient A): rare, kept out of line so that its scratch-resident
// arrays do not burden the common path
__device__ __noinline__ void solve_min_norm(int n, const float* A, const float* b, float* x);

template <int N, typename AGet, typename BGet>
__device__ __forceinline__ bool chol_solve(AGet A, BGet b, float* x)
{
    // the Cholesky route of solve_spd with the matrix behind an accessor (static indices everywhere): when A and b are read straight
    // from the pair sums in LDS, the 36 + 6 registers of a private copy are not needed -- the fused solve of nn1_wg_kernel has 80
    float dmax = 0.f;
#pragma unroll
    for (int j = 0; j < N; ++j) dmax = A(j, j) > dmax ? A(j, j) : dmax;
    const float pthr = (float)N * 1.1920928955078125e-07f * dmax;
    float L[N * N], iL[N];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        float d = A(j, j);
#pragma unroll
        for (int kk = 0; kk < N; ++kk) if (kk < j) d -= L[N * kk + j] * L[N * kk + j];
        ok = ok && (d > pthr);
        const float ljj = sqrtf(d);
        L[N * j + j] = ljj;
        const float ilj = 1.f / ljj;
        iL[j] = ilj;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if (i > j) {
                float s = A(i, j);
#pragma unroll
                for (int kk = 0; kk < N; ++kk) if (kk < j) s -= L[N * kk + i] * L[N * kk + j];
                L[N * j + i] = s * ilj;
            }
        }
    }
    if (!ok) return false;
    float y[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        float s = b(i);
#pragma unroll
        for (int kk = 0; kk < N; ++kk) if (kk < i) s -= L[N * kk + i] * y[kk];
        y[i] = s * iL[i];
    }
#pragma unroll
    for (int i = N - 1; i >= 0; --i) {
        float s = y[i];
#pragma unroll
        for (int kk = 0; kk < N; ++kk) if (kk > i) s -= L[N * i + kk] * x[kk];
        x[i] = s * iL[i];
    }
    return true;
}

template <int N>
__device__ void solve_spd(const float* A, const float* b, float* x)
{
    // invertibility rule shared with the oracle: every float Cholesky pivot > N eps_f max_j A_jj.
    // Every loop has compile-time bounds and is fully unrolled: L, y live in registers (a rolled
    // triangular loop would put them in scratch memory, ~10 us of dependent scratch traffic).
    float dmax = 0.f;
#pragma unroll
    for (int j = 0; j < N; ++j) dmax = A[N * j + j] > dmax ? A[N * j + j] : dmax;
    const float pthr = (float)N * 1.1920928955078125e-07f * dmax;
    float L[N * N], iL[N];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        float d = A[N * j + j];
#pragma unroll
        for (int kk = 0; kk < N; ++kk) if (kk < j) d -= L[N * kk + j] * L[N * kk + j];
        ok = ok && (d > pthr);
        const float ljj = sqrtf(d);
        L[N * j + j] = ljj;
        const float ilj = 1.f / ljj; // one reciprocal per pivot, multiplied through (column and both substitutions)
        iL[j] = ilj;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if (i > j) {
                float s = A[N * j + i];
#pragma unroll
                for (int kk = 0; kk < N; ++kk) if (kk < j) s -= L[N * kk + i] * L[N * kk + j];
                L[N * j + i] = s * ilj;
            }
        }
    }
    if (ok) {
        float y[N];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            float s = b[i];
#pragma unroll
            for (int kk = 0; kk < N; ++kk) if (kk < i) s -= L[N * kk + i] * y[kk];
            y[i] = s * iL[i];
        }
#pragma unroll
        for (int i = N - 1; i >= 0; --i) {
            float s = y[i];
#pragma unroll
            for (int kk = 0; kk < N; ++kk) if (kk > i) s -= L[N * i + kk] * x[kk];
            x[i] = s * iL[i];
        }
        return;
    }
    solve_min_norm(N, A, b, x);
}

__device__ __noinline__ void solve_min_norm(int n, const float* A, const float* b, float* x)
{
    __shared__ double Ad[36], w[6], Q[36], xd[6];
    for (int i = 0; i < n * n; ++i) Ad[i] = A[i];
    jacobi_eig(n, Ad, w, Q);
    double wmax = 0;
    for (int i = 0; i < n; ++i) if (fabs(w[i]) > wmax) wmax = fabs(w[i]);
    const double thr = (double)n * 1.1920928955078125e-07 * wmax;
    for (int i = 0; i < 6; ++i) xd[i] = 0.0;
    for (int e = 0; e < n; ++e) {
        if (!(w[e] > thr)) continue;
        double proj = 0;
        for (int i = 0; i < n; ++i) proj += Q[n * e + i] * (double)b[i];
        proj /= w[e];
        for (int i = 0; i < n; ++i) xd[i] += proj * Q[n * e + i];
    }
    for (int i = 0; i < n; ++i) x[i] = (float)xd[i];
}

__device__ void angle_axis_T(const float* x3, float* T)
{
    for (int i = 0; i < 16; ++i) T[i] = (i % 5 == 0) ? 1.f : 0.f;
    const float nrm = sqrtf(x3[0] * x3[0] + x3[1] * x3[1] + x3[2] * x3[2]);
    if (!(nrm > 0.f)) return;
    const float ax = x3[0] / nrm, ay = x3[1] / nrm, az = x3[2] / nrm;
    // the oracle's orc_sincos_f, operation by operation: Taylor / Horner with fmaf below 0.5 rad (a double sin + cos is ~2000
    // clocks of the single lane), through double above -- where host libm and device ocml round to the same float
    float s, c;
    if (nrm < 0.5f) {
        const float z = nrm * nrm;
        float ps = fmaf(z, 2.75573192e-06f, -1.98412698e-04f);
        ps = fmaf(z, ps, 8.33333333e-03f);
        ps = fmaf(z, ps, -1.66666667e-01f);
        s = fmaf(nrm * z, ps, nrm);
        float pc = fmaf(z, -2.75573192e-07f, 2.48015873e-05f);
        pc = fmaf(z, pc, -1.38888889e-03f);
        pc = fmaf(z, pc, 4.16666667e-02f);
        pc = fmaf(z, pc, -0.5f);
        c = fmaf(z, pc, 1.f);
    } else { s = (float)sin((double)nrm); c = (float)cos((double)nrm); }
    const float sx = s * ax, sy = s * ay, sz = s * az;
    const float cx = (1.f - c) * ax, cy = (1.f - c) * ay, cz = (1.f - c) * az;
    float tmp;
    tmp = cx * ay; T[4 * 1 + 0] = tmp - sz; T[4 * 0 + 1] = tmp + sz;
    tmp = cx * az; T[4 * 2 + 0] = tmp + sy; T[4 * 0 + 2] = tmp - sy;
    tmp = cy * az; T[4 * 2 + 1] = tmp - sx; T[4 * 1 + 2] = tmp + sx;
    T[0] = cx * ax + c; T[5] = cy * ay + c; T[10] = cz * az + c;
}

__device__ double quat_angdist(const double* a, const double* b)
{
    const double w = a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
    const double x = -a[0] * b[1] + a[1] * b[0] - a[2] * b[3] + a[3] * b[2];
    const double y = -a[0] * b[2] + a[1] * b[3] + a[2] * b[0] - a[3] * b[1];
    const double z = -a[0] * b[3] - a[1] * b[2] + a[2] * b[1] + a[3] * b[0];
    return 2.0 * atan2(sqrt(x * x + y * y + z * z), fabs(w));
}

// ---------------------------------------------------------------------------------------------
// single-wave kernel: ordered reduction of the block partials, solve, compose, checkers
// ---------------------------------------------------------------------------------------------
// the iteration's pair sums from its fixed-point accumulators (common.h: ICPMI_ACC_*): 8 copies x 32 values x 2 limbs, one load per
// thread and limb, integer sums (order-free), joined in double.  `clear`: the reader is the only one (the stand-alone solve kernel): it
// leaves the accumulators zeroed for the iteration after the next.
__device__ __forceinline__ unsigned pair_sum_mask(const LoopCfg& lc)
{
    // which of the ICPMI_NV values the chain's pair-sum kernel writes: 27 of A / b or 16 point-to-point sums, sum w (27), pairs (28), force2D's b (29..31)
    const unsigned nval = lc.minimizer == ICPMI_MIN_POINT_TO_PLANE ? 27u : (lc.minimizer == ICPMI_MIN_POINT_TO_POINT ? 16u : 0u);
    unsigned m = (nval ? ((1u << nval) - 1u) : 0u) | (1u << 27) | (1u << 28);
    if (lc.force_2d) m |= 7u << 29;
    return m;
}

__device__ __forceinline__ void read_pair_sums(unsigned long long* __restrict__ acc, double* __restrict__ tot /* LDS, ICPMI_NV */, bool clear, bool* nonfinite,
                                               unsigned mask = 0xffffffffu)
{
    __shared__ long long ph[ICPMI_ACC_COPIES][ICPMI_NV], pl[ICPMI_ACC_COPIES][ICPMI_NV];
    __shared__ unsigned s_flag;
    const int t = threadIdx.x;
    if (t < ICPMI_ACC_COPIES * ICPMI_NV) {
        const int i = t & (ICPMI_NV - 1), cp = t >> 5;
        unsigned long long a = 0ull, b = 0ull;
        if ((mask >> i) & 1u) { a = acc[ICPMI_ACC_IDX(cp, i, 0)]; b = acc[ICPMI_ACC_IDX(cp, i, 1)]; } // (every slot is its own 128-byte line: only the ones in use)
        ph[cp][i] = (long long)a; pl[cp][i] = (long long)b;
        if (clear) { if (a) acc[ICPMI_ACC_IDX(cp, i, 0)] = 0ull; if (b) acc[ICPMI_ACC_IDX(cp, i, 1)] = 0ull; }
    }
    if (t == ICPMI_ACC_COPIES * ICPMI_NV) {
        unsigned* fl = reinterpret_cast<unsigned*>(acc + ICPMI_ACC_FLAG);
        s_flag = *fl;
        if (clear && s_flag) *fl = 0u;
    }
    __syncthreads();
    if (t < ICPMI_NV) {
        long long H = 0, Lq = 0;
#pragma unroll
        for (int cp = 0; cp < ICPMI_ACC_COPIES; ++cp) { H += ph[cp][t]; Lq += pl[cp][t]; }
        tot[t] = (double)H * 65536.0 + (double)Lq * 0x1p-40;
    }
    __syncthreads();
    *nonfinite = s_flag != 0u;
}

// ONE lane: minimiser, compose, checkers on the pair sums `tot` (LDS or private); st may live in LDS (nn.hip: the solve every NN
// workgroup redoes in its prologue) or in global memory (solve_kernel).  Every array the solver indexes dynamically is a __shared__
// array (exactly one lane of a workgroup runs this code): with stack arrays the kernel needs scratch memory, and 1 568 workgroups
// walking 100 KB of scratch each turned a 2 us solve into +20 us per NN launch (r4, first version).
__device__ void solve_serial(IcpState* st, const double* tot, const LoopCfg& lc, float* T_step_out)
{
    const long long tsolve0 = clock64();

    const double wsum = tot[27];
    const long long P = (long long)(tot[28] + 0.5);
    st->pairs = P;
    st->wsum = wsum;
    if (P == 0) { st->error = ICPMI_ERR_NO_POINT_TO_MINIMIZE; st->done = 1; return; }

    float Ts[16];
    for (int i = 0; i < 16; ++i) Ts[i] = (i % 5 == 0) ? 1.f : 0.f;
    if (lc.minimizer == ICPMI_MIN_POINT_TO_POINT) {
        // H = sum w q p^T - (sum w q)(sum w p)^T / sum w, rounded to float like the reference's
        // float matrices, then R = U V^T; t = mean_q - R mean_p
        const double iw = 1.0 / wsum; // one reciprocal, multiplied through (a double division is ~30 instructions of a single lane)
        const double mpd[3] = {tot[1] * iw, tot[2] * iw, tot[3] * iw}, mqd[3] = {tot[4] * iw, tot[5] * iw, tot[6] * iw};
        __shared__ float H[9];
        for (int c = 0; c < 3; ++c)
            for (int r = 0; r < 3; ++r) H[3 * c + r] = (float)(tot[7 + 3 * c + r] - mqd[r] * tot[1 + c]);
        __shared__ float R[9];
        if (lc.is_2d) {
            // planar clouds: the proper in-plane rotation that maximises tr(R^T H), theta = atan2(H10 - H01, H00 + H11) (what the
            // 2 x 2 SVD with its reflection repair returns), same operations as the oracle
            const float a = H[0] + H[4], b2 = H[1] - H[3];
            const float r = sqrtf(a * a + b2 * b2);
            float cs = 1.f, sn = 0.f;
            if (r > 0.f) { cs = a / r; sn = b2 / r; }
            for (int i = 0; i < 9; ++i) R[i] = 0.f;
            R[0] = cs; R[1] = sn; R[3] = -sn; R[4] = cs; R[8] = 1.f;
        } else rotation_from_H(H, R);
        const float mp[3] = {(float)mpd[0], (float)mpd[1], (float)mpd[2]};
        const float mq[3] = {(float)mqd[0], (float)mqd[1], (float)mqd[2]};
        for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) Ts[4 * c + r] = R[3 * c + r];
        for (int r = 0; r < 3; ++r) Ts[12 + r] = mq[r] - (R[r] * mp[0] + R[3 + r] * mp[1] + R[6 + r] * mp[2]);
    } else if (lc.minimizer == ICPMI_MIN_POINT_TO_PLANE) {
        __shared__ float A[36], b[6], x[6];
        auto fill_Ab = [&]() { // the float copy of the system: only the restricted systems and the rank-deficient route read it
            int idx = 0;
            for (int a = 0; a < 6; ++a)
                for (int bb = a; bb < 6; ++bb) { const float v = (float)tot[idx++]; A[6 * a + bb] = v; A[6 * bb + a] = v; }
            for (int a = 0; a < 6; ++a) b[a] = (float)tot[21 + a];
        };
        if (lc.force_2d || lc.force_4dof) fill_Ab();
        if (lc.force_2d) {
            // force2D: F = [x ny - y nx; nx; ny] -- rows 2..4 of the 6-DOF F --, b from the 2-D residual (tot[29..31]); x = (yaw, tx, ty)
            __shared__ float A3[9], b3[3], x3[3];
            for (int c = 0; c < 3; ++c) { b3[c] = (float)tot[29 + c]; for (int r = 0; r < 3; ++r) A3[3 * c + r] = A[6 * (2 + c) + (2 + r)]; }
            solve_spd<3>(A3, b3, x3);
            x[0] = 0.f; x[1] = 0.f; x[2] = x3[0]; x[3] = x3[1]; x[4] = x3[2]; x[5] = 0.f;
        } else if (lc.force_4dof) {
            // force4DOF: F = [cross_z; n] -- the {2,3,4,5} sub-system of the 6-DOF sums; x = (yaw, t)
            __shared__ float A4[16], b4[4], x4[4];
            for (int c = 0; c < 4; ++c) { b4[c] = b[2 + c]; for (int r = 0; r < 4; ++r) A4[4 * c + r] = A[6 * (2 + c) + (2 + r)]; }
            solve_spd<4>(A4, b4, x4);
            x[0] = 0.f; x[1] = 0.f; x[2] = x4[0]; x[3] = x4[1]; x[4] = x4[2]; x[5] = x4[3];
        } else {
            // the 6-DOF system: Cholesky straight from the pair sums (same operations in the same order as solve_spd<6> on the float copy;
            // A(i, j) with i >= j is element (j, i) of the upper triangle in tot); the rank-deficient route takes the copy in LDS
            auto tri = [](int r, int c) { return r * 6 - r * (r - 1) / 2 + (c - r); }; // r <= c
            const bool ok = chol_solve<6>([&](int i, int j) { return (float)tot[tri(j < i ? j : i, j < i ? i : j)]; },
                                          [&](int i) { return (float)tot[21 + i]; }, x);
            if (!ok) { fill_Ab(); solve_min_norm(6, A, b, x); }
        }
        angle_axis_T(x, Ts);
        Ts[12] = x[3]; Ts[13] = x[4]; Ts[14] = x[5];
    }
    const long long tsolve1 = clock64();
    for (int i = 0; i < 16; ++i)
        if (Ts[i] != Ts[i]) { st->error = ICPMI_ERR_NAN; st->done = 1; return; }
    if (T_step_out) for (int i = 0; i < 16; ++i) T_step_out[i] = Ts[i];

    float Ti[16];
    mat4_mul_dev(Ts, st->T_iter, Ti);
    if (lc.sensor_noise) for (int i = 0; i < 16; ++i) st->T_prev[i] = st->T_iter[i]; // the pose this step's pairs were formed under
    for (int i = 0; i < 16; ++i) st->T_iter[i] = Ti[i];
    st->iter += 1;

    const long long tsolve2 = clock64();
    // ---- TransformationCheckers ----
    bool iterate = true;
    int reason = ICPMI_STOP_NONE;
    st->counter += 1;
    if (st->counter >= lc.max_iter) { iterate = false; reason = ICPMI_STOP_COUNTER; }
    if (lc.use_diff) {
        const int SL = lc.smooth;
        const int RING = ICPMI_MAX_SMOOTH + 1;
        const int slot = st->hist_n % RING;
        quat_from_T(Ti, st->hq + 4 * slot);
        for (int r = 0; r < 3; ++r) st->ht[3 * slot + r] = Ti[12 + r];
        st->hist_n += 1;
        const int hn = st->hist_n;
        {
            // the step between pose hn - 1 (just pushed) and pose hn - 2: computed once, read SL times (the older steps of the
            // window were stored by the iterations that pushed them -- same values, same summation order as recomputing)
            const int a = (hn - 1) % RING, bq = (hn - 2) % RING;
            st->hrot[a] = fabs(quat_angdist(st->hq + 4 * a, st->hq + 4 * bq));
            const double dx = st->ht[3 * a] - st->ht[3 * bq], dy = st->ht[3 * a + 1] - st->ht[3 * bq + 1],
                         dz = st->ht[3 * a + 2] - st->ht[3 * bq + 2];
            st->htr[a] = sqrt(dx * dx + dy * dy + dz * dz);
        }
        if (hn > SL) {
            double rot = 0, tr = 0;
            for (int i = hn - 1; i >= hn - SL; --i) { rot += st->hrot[i % RING]; tr += st->htr[i % RING]; }
            rot /= SL; tr /= SL;
            if (rot != rot || tr != tr) { st->error = ICPMI_ERR_NAN; st->done = 1; return; }
            if (rot < (double)lc.min_rot && tr < (double)lc.min_trans) {
                if (iterate) reason = ICPMI_STOP_DIFFERENTIAL;
                iterate = false;
            }
        }
    }
    if (lc.use_bound) {
        __shared__ double q[4];
        quat_from_T(Ti, q);
        const double rot = fabs(quat_angdist(q, st->init_q));
        const double nt = sqrt((double)Ti[12] * Ti[12] + (double)Ti[13] * Ti[13] + (double)Ti[14] * Ti[14]);
        if (rot > (double)lc.max_rot || nt > (double)lc.max_trans) { st->error = ICPMI_ERR_BOUND; st->done = 1; return; }
    }
    if (!iterate) { st->done = 1; st->stop_reason = reason; }
    const long long tsolve3 = clock64();
#ifndef ICPMI_NN_TIMING // (a timing build gives dbg[0..23] to the NN kernels' phase clocks: scripts/nn_phase.py must read them clean -- ADVICE r4)
    st->dbg[20] += (unsigned long long)(tsolve3 - tsolve0); // serial part of the solve (diagnostic)
    st->dbg[21] += 1;
    st->dbg[22] += (unsigned long long)(tsolve1 - tsolve0); // ... of which the minimiser,
    st->dbg[23] += (unsigned long long)(tsolve3 - tsolve2); // ... and the checkers
#else
    (void)tsolve1; (void)tsolve2; (void)tsolve3;
#endif
}

// the stand-alone solve: 256 threads read the accumulators (and clear them when asked), thread 0 solves
__device__ void solve_body(IcpState* __restrict__ st, unsigned long long* __restrict__ acc, bool clear, const LoopCfg& lc,
                           float* __restrict__ T_step_out, double* __restrict__ sums_out)
{
    __shared__ double tot[ICPMI_NV];
    const int t = threadIdx.x;
    bool nonfinite = false;
    read_pair_sums(acc, tot, clear, &nonfinite, pair_sum_mask(lc));
    if (t < ICPMI_NV && sums_out) sums_out[t] = tot[t];
    if (nonfinite) { if (t == 0) { st->error = ICPMI_ERR_NAN; st->done = 1; } return; }
    if (t == 0) solve_serial(st, tot, lc, T_step_out);
}

// progress word (icpmi_ctx::h_progress): visible to the host while the stream keeps running
__device__ __forceinline__ void publish_progress(const IcpState* st, unsigned* progress)
{
    if (!progress) return;
    const unsigned v = ((unsigned)(st->done != 0) << 31) | ((st->seq & 0x7ffffu) << 12) | ((unsigned)st->iter & 0xfffu);
    __hip_atomic_store(progress, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}


} // namespace
