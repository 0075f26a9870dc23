// nn.hip -- exact nearest-neighbour search against the cell-sorted map.
//
// Replaces KDTreeMatcher::findClosests -> Nabo::NNS::knn (behind `icp(input)`,
// norlab_icp_mapper/Mapper.cpp:213) and the direct libnabo calls of
// MapperModules/PointDistanceMapperModule.cpp:33-36.  Contract: squared L2
// distances, accept d2 <= maxRadius^2, unfilled slots id -1 / d2 +inf, results ascending; ties on
// equal float d2 resolve to the smallest original map index (the deterministic rule shared with the
// oracle, libnabo's own pick being traversal dependent).
//
// Kernels: nn1_wg_kernel (k = 1), nnk_ml_kernel / nnk_wg_kernel (2 <= k <= 16), nnk_kernel (17 <= k <= 32, one
// lane per query: surface normals), and the brute-force passes for queries the grid cannot decide
// (only reachable with an unbounded maxDist).  The designs are described above each kernel.
#include "common.h"

namespace {

constexpr int NN_BLOCK = 256;

typedef __attribute__((address_space(1))) unsigned gunsigned;
// sqrt for pruning radii and level choices: one v_sqrt_f32 (1 ulp) instead of the IEEE-exact sequence (~12 instructions, and the
// kernels take it per row), nudged UP by 2^-21 relative so that it never under-estimates -- a bound that is a hair too wide only
// ever adds candidates, it cannot change the exact minimum
__device__ __forceinline__ float sqrt_up(float x) { return __builtin_amdgcn_sqrtf(x) * 1.0000005f; }
typedef float vf4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) vf4 gfloat4; // 16 bytes in global memory (global_load, not flat_load)

struct Cand {
    unsigned long long key; // (d2 bits << 32) | original index
    int sidx;               // position in the sorted map
};

__device__ __forceinline__ void cand_min(Cand& a, unsigned long long key, int sidx)
{
    if (key < a.key) { a.key = key; a.sidx = sidx; }
}

template <int G>
__device__ __forceinline__ void group_reduce(Cand& c)
{
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) {
        const unsigned long long ok = __shfl_xor(c.key, off, 64);
        const int os = __shfl_xor(c.sidx, off, 64);
        cand_min(c, ok, os);
    }
}

__device__ __forceinline__ void scan_run(const float4* __restrict__ map, unsigned s, unsigned e, int sub, int G,
                                         float px, float py, float pz, bool allow_self, Cand& best)
{
    for (unsigned i = s + sub; i < e; i += G) {
        const float4 q = map[i];
        const float d2 = sqdist3(px, py, pz, q.x, q.y, q.z);
        if (allow_self || d2 > 1.1920929e-07f) cand_min(best, pack_key(d2, __float_as_uint(q.w)), (int)i);
    }
}

// x-run [x0, x1] of row (y, z), clipped to the grid; returns an empty run when outside
__device__ __forceinline__ void row_run(const GridParams& g, const unsigned* __restrict__ cs, int x0, int x1, int y, int z,
                                        unsigned& s, unsigned& e)
{
    s = 0; e = 0;
    if (y < 0 || y >= g.ny || z < 0 || z >= g.nz) return;
    if (x0 < 0) x0 = 0;
    if (x1 > g.nx - 1) x1 = g.nx - 1;
    if (x0 > x1) return;
    const int base = (z * g.ny + y) * g.nx;
    s = cs[base + x0];
    e = cs[base + x1 + 1];
}

// Brute-force pass for the queued queries (k = 1): one workgroup per query streams the whole map.
// match_pt != nullptr (r5): the loop state is kept in QUERY order (`reading` = the tile-sorted queries, the queue holds query slots): the
// matched point goes next to the match, and -- hist0 != nullptr -- the match joins the level-0 selection histogram the NN kernel built for
// the queries it decided itself (csrc/loop.hip, fused selection; copy 0 of the privatised tables).
__global__ __launch_bounds__(NN_BLOCK) void nn1_hard_kernel(const float4* __restrict__ reading, const float* __restrict__ Tptr,
                                                            const float4* __restrict__ map, int m, float maxr2, int allow_self,
                                                            int* __restrict__ out_sidx, float* __restrict__ out_d2,
                                                            IcpState* __restrict__ st, const unsigned* __restrict__ hard,
                                                            float4* __restrict__ match_pt = nullptr, unsigned* __restrict__ hist0 = nullptr)
{
    if (st->done) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) nn_stamp_close(const_cast<IcpState*>(st));
    const unsigned nh = st->hard_count;
    __shared__ unsigned long long shk[NN_BLOCK / 64];
    __shared__ int shs[NN_BLOCK / 64];
    for (unsigned h = blockIdx.x; h < nh; h += gridDim.x) {
        const int qi = (int)hard[h];
        const float4 r = reading[qi];
        float3 p;
        if (Tptr) p = xf_point(Tptr, r.x, r.y, r.z, r.w);
        else p = make_float3(r.x, r.y, r.z);
        Cand best; best.key = ~0ull; best.sidx = -1;
        scan_run(map, 0u, (unsigned)m, threadIdx.x, NN_BLOCK, p.x, p.y, p.z, allow_self != 0, best);
        group_reduce<64>(best);
        const int w = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { shk[w] = best.key; shs[w] = best.sidx; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int i = 1; i < NN_BLOCK / 64; ++i) cand_min(best, shk[i], shs[i]);
            float bd2 = __uint_as_float((unsigned)(best.key >> 32));
            int bs = best.sidx;
            if (bs < 0 || !(bd2 <= maxr2)) { bs = -1; bd2 = INFINITY; }
            out_sidx[qi] = bs;
            out_d2[qi] = bd2;
            if (match_pt) match_pt[qi] = bs >= 0 ? map[bs] : make_float4(0.f, 0.f, 0.f, 0.f);
            if (hist0 && bd2 != INFINITY && bd2 > 0.f) {
                const unsigned bits = __float_as_uint(bd2);
                atomicAdd(&hist0[ICPMI_S2_C0 + (bits >> 24)], 1u);
                atomicAdd(&hist0[ICPMI_S2_F0 + ICPMI_S2_FIDX(bits >> 16)], 1u);
            }
        }
        __syncthreads();
    }
}

__global__ void hard_reset_kernel(IcpState* st)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) { st->hard_total += st->hard_count; st->hard_count = 0; }
}

// ------------------------------------------------------------------------------------------------
// General k (2..32): one lane per query, bounded sorted list in registers / scratch, the same ring
// search.  Used by knn > 1 matchers (docs/MapperConfiguration.md:174-189 uses knn 6) and by the
// surface-normal operator (knn 10 on the map itself).
// ------------------------------------------------------------------------------------------------
// (KList<KMAX>: common.h -- the block-grid self search of selfgrid.hip keeps the same lists)
template <int KMAX>
__device__ __forceinline__ void scan_run_k(const float4* __restrict__ map, unsigned s, unsigned e, float px, float py, float pz,
                                           bool allow_self, KList<KMAX>& L)
{
    // four points in flight (r5): one point per trip made a run of c points a chain of c memory latencies -- the ring kernel that redoes the
    // tiled self-search's few left-over queries took 57 - 62 us for a few hundred of them.  Same insertion order, same lists.
    for (unsigned i = s; i < e; i += 4u) {
        float4 q[4];
#pragma unroll
        for (unsigned u = 0; u < 4u; ++u) q[u] = map[i + u < e ? i + u : i];
#pragma unroll
        for (unsigned u = 0; u < 4u; ++u) {
            if (i + u >= e) continue;
            const float d2 = sqdist3(px, py, pz, q[u].x, q[u].y, q[u].z);
            if (allow_self || d2 > 1.1920929e-07f) L.insert(pack_key(d2, __float_as_uint(q[u].w)), (int)(i + u));
        }
    }
}

template <int KMAX>
__global__ __launch_bounds__(NN_BLOCK) void nnk_kernel(const float4* __restrict__ reading, int n, const float* __restrict__ Tptr,
                                                       GridParams g, const float4* __restrict__ map,
                                                       const unsigned* __restrict__ cs, int k, float maxr2, int ring_max,
                                                       int allow_self, int* __restrict__ out_sidx, float* __restrict__ out_d2,
                                                       IcpState* __restrict__ st, unsigned* __restrict__ hard,
                                                       const unsigned* __restrict__ only, const unsigned* __restrict__ only_count)
{
    if (st && st->done) return;
    if (st && blockIdx.x == 0 && threadIdx.x == 0) nn_stamp_open(st);
    // `only` != nullptr: just the listed queries (the left-overs of the tiled self-search below)
    const int tid = blockIdx.x * NN_BLOCK + threadIdx.x;
    if (only ? tid >= (int)*only_count : tid >= n) return;
    const int qi = only ? (int)only[tid] : tid;
    const float4 r = reading[qi];
    float3 p;
    if (Tptr) p = xf_point(Tptr, r.x, r.y, r.z, r.w);
    else p = make_float3(r.x, r.y, r.z);
    const float fx = (p.x - g.ox) * g.inv_cell, fy = (p.y - g.oy) * g.inv_cell, fz = (p.z - g.oz) * g.inv_cell;
    const float flx = floorf(fx), fly = floorf(fy), flz = floorf(fz);
    const int cx = (int)fminf(fmaxf(flx, -1.0e6f), 1.0e6f);
    const int cy = (int)fminf(fmaxf(fly, -1.0e6f), 1.0e6f);
    const int cz = (int)fminf(fmaxf(flz, -1.0e6f), 1.0e6f);
    float mf = fminf(fx - flx, 1.0f - (fx - flx));
    mf = fminf(mf, fminf(fy - fly, 1.0f - (fy - fly)));
    mf = fminf(mf, fminf(fz - flz, 1.0f - (fz - flz)));
    if (!(mf >= 0.f)) mf = 0.f;

    KList<KMAX> L; L.init(k);
    bool decided = false;
    int ring = 0;
    // distances from the query to the faces of its cell along y and z (for pruning rows of ring 1)
    const float ylo = (fy - fly) * g.cell, yhi = (1.0f - (fy - fly)) * g.cell;
    const float zlo = (fz - flz) * g.cell, zhi = (1.0f - (fz - flz)) * g.cell;
    while (!decided && ring < ring_max) {
        ++ring;
        const int side = 2 * ring + 1;
        for (int r0 = 0; r0 < side * side; ++r0) {
            // ring 1 starts with the row through the query's own cell: once k candidates are held, rows and
            // cells the ball of the k-th distance cannot reach are skipped (every point within it is still seen)
            const int rr = ring == 1 ? (r0 == 0 ? 4 : (r0 <= 4 ? r0 - 1 : r0)) : r0;
            const int dy = rr % side - ring, dz = rr / side - ring;
            const bool full_row = ring == 1 || (dy == -ring || dy == ring || dz == -ring || dz == ring);
            unsigned s, e;
            if (full_row) {
                int xa = cx - ring, xb = cx + ring;
                if (ring == 1) {
                    unsigned long long kth = ~0ull;
#pragma unroll
                    for (int i = 0; i < KMAX; ++i) if (i == k - 1) kth = L.key[i];
                    if (kth != ~0ull) {
                        const float rub = sqrtf(__uint_as_float((unsigned)(kth >> 32))) * 1.000001f + g.slack;
                        const float ddy = dy == 0 ? 0.f : (dy < 0 ? ylo : yhi);
                        const float ddz = dz == 0 ? 0.f : (dz < 0 ? zlo : zhi);
                        const float rem2 = rub * rub - (ddy * ddy + ddz * ddz);
                        if (rem2 < 0.f) continue;
                        const float rem = sqrtf(rem2);
                        const int xl = (int)fmaxf(floorf((p.x - rem - g.ox) * g.inv_cell), -1.0e6f);
                        const int xh = (int)fminf(floorf((p.x + rem - g.ox) * g.inv_cell), 1.0e6f);
                        xa = xl > xa ? xl : xa;
                        xb = xh < xb ? xh : xb;
                    }
                }
                row_run(g, cs, xa, xb, cy + dy, cz + dz, s, e);
                scan_run_k<KMAX>(map, s, e, p.x, p.y, p.z, allow_self != 0, L);
            } else {
                if (cx - ring >= 0) {
                    row_run(g, cs, cx - ring, cx - ring, cy + dy, cz + dz, s, e);
                    scan_run_k<KMAX>(map, s, e, p.x, p.y, p.z, allow_self != 0, L);
                }
                if (cx + ring <= g.nx - 1) {
                    row_run(g, cs, cx + ring, cx + ring, cy + dy, cz + dz, s, e);
                    scan_run_k<KMAX>(map, s, e, p.x, p.y, p.z, allow_self != 0, L);
                }
            }
        }
        const float margin = fmaxf(((float)ring + mf) * g.cell - g.slack, 0.f);
        const float m2 = margin * margin;
        const unsigned long long kth = L.key[k - 1];
        const float kd2 = __uint_as_float((unsigned)(kth >> 32));
        const bool covers = cx - ring <= 0 && cx + ring >= g.nx - 1 && cy - ring <= 0 && cy + ring >= g.ny - 1 &&
                            cz - ring <= 0 && cz + ring >= g.nz - 1;
        decided = (kth != ~0ull && kd2 <= m2) || m2 > maxr2 || covers;
    }
    for (int j = 0; j < k; ++j) {
        float d2 = __uint_as_float((unsigned)(L.key[j] >> 32));
        int s = L.sidx[j];
        if (s < 0 || !(d2 <= maxr2)) { s = -1; d2 = INFINITY; }
        out_sidx[(size_t)k * qi + j] = s;
        out_d2[(size_t)k * qi + j] = d2;
    }
    if (!decided) {
        const unsigned slot = atomicAdd(&st->hard_count, 1u);
        hard[slot] = (unsigned)qi;
    }
}

// brute pass for general k: the workgroup streams the map, every lane keeps its own list, then k
// rounds of "extract the global minimum" merge the 256 lists.
template <int KMAX>
__global__ __launch_bounds__(NN_BLOCK) void nnk_hard_kernel(const float4* __restrict__ reading, const float* __restrict__ Tptr,
                                                            const float4* __restrict__ map, int m, int k, float maxr2,
                                                            int allow_self, int* __restrict__ out_sidx,
                                                            float* __restrict__ out_d2, IcpState* __restrict__ st,
                                                            const unsigned* __restrict__ hard)
{
    if (st->done) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) nn_stamp_close(const_cast<IcpState*>(st));
    const unsigned nh = st->hard_count;
    __shared__ unsigned long long shk[NN_BLOCK / 64];
    __shared__ int shs[NN_BLOCK / 64];
    __shared__ unsigned long long win_key;
    __shared__ int win_sidx;
    for (unsigned h = blockIdx.x; h < nh; h += gridDim.x) {
        const int qi = (int)hard[h];
        const float4 r = reading[qi];
        float3 p;
        if (Tptr) p = xf_point(Tptr, r.x, r.y, r.z, r.w);
        else p = make_float3(r.x, r.y, r.z);
        KList<KMAX> L; L.init(k);
        // (r5) Every kernel that queues a query here has written the k candidates it did find to the query's row first (nnk_kernel, nnk_ml_kernel,
        // nnk_wg_kernel, nnk_wave_kernel: rows and queue entries share one index): when that list is full its k-th key bounds the answer, and a
        // candidate above it never touches a lane's list.  Without the bound some lane of a wave inserts at nearly every candidate (a lane's list
        // changes ~50 times over its 390 candidates of a 0.1 M-point map) and the whole wave pays the 64-bit insertion each time: ~200 us per
        // QUERY -- and a lidar map's sparse periphery sends 400 - 500 points of every self search here (BASELINE config 4).  Same k points:
        // the k smallest keys are all <= the k-th key of any k valid candidates.
        const float kb = out_d2[(size_t)k * qi + (k - 1)];
        const unsigned long long bound = kb != INFINITY ? pack_key(kb, 0xffffffffu) : ~0ull;
        // ... and eight candidates are requested per trip (one load, one test, the next load was a memory round trip per candidate: what
        // was left of the kernel once the insertions were gone).  Same candidates in the same order.
        constexpr int HU = 8;
        for (unsigned i0 = threadIdx.x; i0 < (unsigned)m; i0 += NN_BLOCK * HU) {
            float4 q[HU];
#pragma unroll
            for (int u = 0; u < HU; ++u) { const unsigned i = i0 + (unsigned)u * NN_BLOCK; q[u] = map[i < (unsigned)m ? i : i0]; }
#pragma unroll
            for (int u = 0; u < HU; ++u) {
                const unsigned i = i0 + (unsigned)u * NN_BLOCK;
                const float d2 = sqdist3(p.x, p.y, p.z, q[u].x, q[u].y, q[u].z);
                const unsigned long long key = pack_key(d2, __float_as_uint(q[u].w));
                if (i < (unsigned)m && (allow_self || d2 > 1.1920929e-07f) && key <= bound) L.insert(key, (int)i);
            }
        }
        int head = 0;
        for (int j = 0; j < k; ++j) {
            Cand c;
            // static indexing: select the head entry with a compare chain
            c.key = ~0ull; c.sidx = -1;
#pragma unroll
            for (int i = 0; i < KMAX; ++i) if (i == head) { c.key = L.key[i]; c.sidx = L.sidx[i]; }
            const unsigned long long mine = c.key;
            group_reduce<64>(c);
            const int w = threadIdx.x >> 6;
            if ((threadIdx.x & 63) == 0) { shk[w] = c.key; shs[w] = c.sidx; }
            __syncthreads();
            if (threadIdx.x == 0) {
                for (int i = 1; i < NN_BLOCK / 64; ++i) cand_min(c, shk[i], shs[i]);
                win_key = c.key; win_sidx = c.sidx;
                float d2 = __uint_as_float((unsigned)(c.key >> 32));
                int s = c.sidx;
                if (s < 0 || !(d2 <= maxr2)) { s = -1; d2 = INFINITY; }
                out_sidx[(size_t)k * qi + j] = s;
                out_d2[(size_t)k * qi + j] = d2;
            }
            __syncthreads();
            if (mine == win_key && mine != ~0ull) ++head; // keys are unique (index in the low word)
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(256) void ids_kernel(const float4* __restrict__ map, const int* __restrict__ sidx, int64_t count,
                                                  int* __restrict__ ids)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const int s = sidx[i];
    ids[i] = s < 0 ? -1 : (int)__float_as_uint(map[s].w);
}

// ------------------------------------------------------------------------------------------------
// k = 1 over the grid pyramid.  (r1 - r2: nn1_ml_kernel, G lanes per query; r3: nn1_wq_kernel, a one-wave workgroup alternating
// between lane-per-query and lane-per-piece; both were kept selectable through r5 and were removed in r6 -- git history has them,
// DESIGN_history.md sections 5 / 11.2 their measurements.  What runs is nn1_wg_kernel below.)
// ------------------------------------------------------------------------------------------------
#ifdef ICPMI_NN_TIMING
#define NN_TICK(i) do { __builtin_amdgcn_s_waitcnt(0); const long long t_ = clock64(); tacc[i] += t_ - tlast; tlast = t_; } while (0)
#else
#define NN_TICK(i) do { } while (0)
#endif
// ------------------------------------------------------------------------------------------------
// The wave-queue scheme (r3) the kernel below inherits: a workgroup owns Q = 64 queries and alternates between two roles:
//   (1) LANE PER QUERY: set-up once per query, row ranges looked up, and every non-empty row cut into PIECES of <= 8 consecutive
//       candidates that go into an LDS work list {start, count, query};
//   (2) LANE PER PIECE: lane i takes pieces i, i + 64, ... whatever query they belong to -- eight independent 16-byte loads in flight
//       from ONE address register (immediate offsets; loads past the run's end are masked, the level arrays are padded), the piece's
//       best key goes into the query's LDS slot by a 64-bit ds_min; the lane whose key IS the slot's value afterwards records its
//       position (keys are unique per map point, and LDS operations of one wave complete in order: no race, no returned atomic);
//   (3) lane per query again: read the slot, apply the level's exactness rule; undecided queries repeat at the next level (or, for a
//       query that held no bound yet, with the bound its own row just gave it).
//   The candidates of all Q queries are thus spread evenly over the lanes however unevenly they are spread over the queries.
// ------------------------------------------------------------------------------------------------
// inclusive prefix sum over the 64 lanes of a wave in registers (DPP: shifts within rows of 16 lanes, then the row totals
// broadcast into the following rows); every lane must be active
__device__ __forceinline__ unsigned wave_incl_scan(unsigned v)
{
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false); // row_shr:1
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false); // row_shr:2
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false); // row_shr:4
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false); // row_shr:8
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false); // row_bcast:15 -> rows 1, 3
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false); // row_bcast:31 -> rows 2, 3
    return v;
}

// ------------------------------------------------------------------------------------------------
// nn1_wg_kernel: the wave-queue scheme with the roles given to DIFFERENT NUMBERS OF WAVES.
// (r4 carried an FSOLVE variant -- the previous iteration's solve in every workgroup's prologue, three launches per iteration.  r5 moved its
//  loads into the query round trip and measured again: fixed-count point-to-point +1.8 %, point-to-plane -1.8 %, checked 6-iteration
//  registrations -4 ... -6 %; the kernel needed scratch memory for the solver's cold paths.  Removed: DESIGN 13.3.)  The per-query set-up is cheapest
// with one lane per query (every instruction of it then serves 64 queries; with LPQ lanes per query it is paid LPQ times), but
// one wave that also looks up the nine rows and works off the ~330 pieces of its 64 queries lives for ten dependent steps --
// and at 100 k queries there are only 1.5 such waves per SIMD to hide that behind.  Here a workgroup of four waves owns 64
// queries:
//   (1a) wave 0, lane per query: transform, seed, level, pruning radius -> LDS record (the other waves are parked at the
//        barrier and issue nothing);
//   (1b) all four waves, lane = query, WAVE = ROW GROUP (rows w, w + 4, w + 8 of the 3 x 3 block): row ranges, pieces;
//   (2)  all four waves, lane per piece (one or two steps for 64 queries);
//   (3)  wave 0 decides and stores.
// Same keys and exactness rules as the r1 - r3 kernels it replaced: same bits.
// ------------------------------------------------------------------------------------------------
// r6: seven waves per SIMD = seven workgroups per CU = 1 792 resident workgroups: the 1 568 of a 100 k-query launch are all resident at once (at six, 32 of
// them waited for a slot: +0.8 ... 2.4 us per launch).  72 VGPRs without a spill since the query's point and its best-so-far moved into LDS (below);
// profiles/r6_ab_nn1_plb.txt has the A/B over sizes.  0 = the compiler's choice.
#ifndef ICPMI_NN1_WAVES
#define ICPMI_NN1_WAVES 7
#endif
#if ICPMI_NN1_WAVES > 0
#define NN1_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(ICPMI_NN1_WAVES, ICPMI_NN1_WAVES)))
#else
#define NN1_WAVES_ATTR
#endif
template <int NW, bool SELF>
__global__ __launch_bounds__(64 * NW) NN1_WAVES_ATTR void nn1_wg_kernel(const float4* __restrict__ queries, const int* __restrict__ qindex, BatchArgs ba,
                                                     const float* __restrict__ Tptr, GridLevels L, float maxr2, int* __restrict__ out_sidx,
                                                     float* __restrict__ out_d2, IcpState* __restrict__ st, unsigned* __restrict__ hard,
                                                     unsigned* __restrict__ hist0, float4* __restrict__ match_pt,
                                                     const uint4* __restrict__ ltab_g, int unseeded_lev, int seed_pre, float inv1e, float err2)
{
    // inv1e = 1 / (1 + epsilon), err2 = (1 + epsilon)^2 (both exactly 1 for the exact search: a multiplication by 1.0f changes no bit): libnabo's
    // `new_rd * maxError2 < heap.headValue()` -- what lies farther than (best so far) / (1 + epsilon) is not visited (KDTreeMatcher's epsilon)
    static_assert(NW == 3 || NW == 4, "waves per workgroup");
    constexpr int NT = 64 * NW, Q = 64;
    constexpr int NR = 3;                  // rows per lane in role 1b: rr = wave + NW sl
    constexpr int CAP = 16 * Q;            // pieces per pass (>= 9 Q: one piece per row always fits)
#ifndef ICPMI_NN1_PLB
#define ICPMI_NN1_PLB 4
#endif
    constexpr int PLB = ICPMI_NN1_PLB;     // log2 of the base piece length: 8 loads in flight per lane and piece (r6: 16 = two batches of eight per lane, so that a workgroup's ~300 pieces of 8 become <= 256 pieces and one round of phase 2: +1 %, A/B in profiles/r6_ab_nn1_plb.txt; 32: -3 %)
    const int n = ba.n[blockIdx.y];
    {
        const size_t qo = (size_t)blockIdx.y * (size_t)ba.qstride;
        queries += qo; out_sidx += qo; out_d2 += qo;
        if (qindex) qindex += qo;
        if (match_pt) match_pt += qo;
        if (hist0) hist0 += (size_t)blockIdx.y * ICPMI_SELHIST_WORDS;
        if (Tptr) Tptr = reinterpret_cast<const float*>(reinterpret_cast<const char*>(Tptr) + (size_t)blockIdx.y * sizeof(IcpState));
        st += blockIdx.y;
    }
#ifdef ICPMI_NN_TIMING
    long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tlast = clock64();
    const long long t_c0 = tlast, t_w0 = wall_clock64();
#endif
    __shared__ uint4 ltab[ICPMI_MAXLEV * 4];
    __shared__ unsigned lh[256];
    __shared__ uint2 pieces[CAP];            // {position of the first candidate in its level array, count << 8 | query slot} (the address comes from the level table)
    __shared__ float2 qdec[Q];               // what the decision of a pass needs from its row phase: squared margin, block covers the grid
    __shared__ float4 qrec[Q];               // transformed query, w = bits of its current level
    __shared__ float2 qaux[Q];               // squared pruning radius (+inf: none), flags: 1 = searching, 2 = own row only
    __shared__ unsigned long long qkey[Q];   // best (d^2, index) key of the query so far
    __shared__ unsigned qwin[Q];             // ... where that point sits: position | level << 28
    __shared__ float4 qpt[Q];                // ... and the point itself (what the loop keeps as the next iteration's seed)
    __shared__ unsigned s_any, s_total[2];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const bool w0 = wave == 0;
#ifndef ICPMI_NN_NO_HOIST
    // r4: every kernel argument the prologue reads (pointers, the level-0 grid) is requested from the kernarg segment in ONE batch here;
    // left to itself the compiler loads each one behind the branch that first uses it -- a dozen dependent scalar round trips in front
    // of the first barrier (ISA of r3's kernel).  The empty asm only pins the values into SGPRs at this point.
    {
        const GridParams g0 = L.g[0];
        asm volatile("" ::"s"(queries), "s"(qindex), "s"(out_sidx), "s"(out_d2), "s"(match_pt), "s"(ltab_g), "s"(Tptr), "s"(st), "s"(hist0), "s"(hard));
        asm volatile("" ::"s"(g0.ox), "s"(g0.oy), "s"(g0.oz), "s"(g0.cell), "s"(g0.inv_cell), "s"(g0.slack), "s"(L.nlev), "s"(maxr2), "s"(unseeded_lev), "s"(seed_pre), "s"(inv1e), "s"(err2));
    }
#endif
    const int wgs = (int)((((long long)n + Q - 1) / Q + 7) / 8 * 8);
    if ((int)blockIdx.x >= wgs) return;
    // the workgroup's queries, XCD-aware order as nn1_ml_kernel: workgroup b runs on XCD b % 8; each XCD gets one contiguous eighth.
    const int chunk = wgs >> 3;
    const int lb = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    const int slot = lane;
    const int qi = lb * Q + slot;
    const bool active = w0 && qi < n; // only wave 0 owns queries
    float4 r = make_float4(0.f, 0.f, 0.f, 1.f);
    int orig = 0, sp_kept = -1;
    float4 qs_kept = make_float4(0.f, 0.f, 0.f, 0.f);
    if (w0) { // (a wave-uniform branch: the other waves go straight to the barrier)
        r = ld_stream(queries + (active ? qi : 0));
        orig = match_pt ? qi : (qindex ? qindex[active ? qi : 0] : qi);
        if (match_pt) { sp_kept = ld_stream(out_sidx + (active ? qi : 0)); qs_kept = ld_stream(match_pt + (active ? qi : 0)); }
    }
    const int st_done = st->done, st_iter = st->iter;
    uint4 ltab_mine = make_uint4(0u, 0u, 0u, 0u);
    if (tid < ICPMI_MAXLEV * 4) ltab_mine = ltab_g[tid];
    float3 p = make_float3(0.f, 0.f, 0.f);
    if (w0) {
        if (Tptr) p = xf_point(Tptr, r.x, r.y, r.z, r.w);
        else p = make_float3(r.x, r.y, r.z);
    }
    if (st_done) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) nn_stamp_open(st); // (device clock of this launch: common.h)
    if (tid < ICPMI_MAXLEV * 4) ltab[tid] = ltab_mine;
    if (hist0) {
        for (int t = tid; t < 256; t += NT) lh[t] = 0;
        // the builder of level 0 clears level 1 of the previous iteration (loop.hip, fused selection)
        for (int gt = blockIdx.x * NT + tid; gt < 256 + 65536; gt += wgs * NT) hist0[ICPMI_S2_C1 + gt] = 0; // (the workgroups of THIS reading: a batch launches the grid of its largest)
    }
    const bool allow_self = SELF;

    Cand best; best.key = ~0ull; best.sidx = -1; // sidx = position in its level | level << 28
    bool decided = !active;
    int lev0 = unseeded_lev;
    if (w0) {   // seed: see nn1_ml_kernel -- the previous match bounds the answer; start at the first level whose block holds that ball
        int sp = -1;
        float4 qs = make_float4(0.f, 0.f, 0.f, 0.f);
        if (active && allow_self && st_iter > 0) {
            sp = match_pt ? sp_kept : out_sidx[orig];
            if (sp >= 0) qs = match_pt ? qs_kept : L.pts[0][sp];
        }
        bool want = sp >= 0;
        const float ub2 = sqdist3(p.x, p.y, p.z, qs.x, qs.y, qs.z);
        const float ub = sqrt_up(ub2);
        auto try_level = [&](int lev, const GridParams& gl) {
            const float fx = (p.x - gl.ox) * gl.inv_cell, fy = (p.y - gl.oy) * gl.inv_cell, fz = (p.z - gl.oz) * gl.inv_cell;
            float mfl = fminf(fx - floorf(fx), 1.0f - (fx - floorf(fx)));
            mfl = fminf(mfl, fminf(fy - floorf(fy), 1.0f - (fy - floorf(fy))));
            mfl = fminf(mfl, fminf(fz - floorf(fz), 1.0f - (fz - floorf(fz))));
            if (!(mfl >= 0.f)) mfl = 0.f;
            const float margin = (1.0f + mfl) * gl.cell - 2.0f * gl.slack;
            if (want && ub * inv1e * 1.000001f <= margin) {
                lev0 = lev;
                best.key = pack_key(ub2, __float_as_uint(qs.w));
                best.sidx = sp; // level 0 position
                qpt[slot] = qs; // (the seed is the best so far: a winner of a later pass overwrites it -- after the barrier below)
                want = false;
            }
        };
        try_level(0, L.g[0]);
        for (int lev = 1; lev < L.nlev; ++lev) {
            if (__ballot(want) == 0ull) break;
            try_level(lev, L.g[lev]);
        }
    }
    bool widepre = false;
    if (seed_pre && lev0 > 0 && best.key != ~0ull) { widepre = true; lev0 = 0; }

    int lev = lev0;
    bool did_pre = false; // this level's own-row pass has run
    // r6: the query's point and its best so far live in LDS from here on (qrec / qkey / qwin are the query's own slots: phase 2 updates the
    // key and the winner in place, wave 0 reads them back) -- registers that only wave 0 used and every wave paid for
    if (w0) { qrec[slot] = make_float4(p.x, p.y, p.z, __int_as_float(lev0)); qkey[slot] = best.key; qwin[slot] = (unsigned)best.sidx; }
    NN_TICK(0);
    __syncthreads(); // ltab / lh visible
    for (;;) {
        // ---- (1a) wave 0, lane per query: what the row lookups need goes into the query's record
        bool run = false, prescan = false;
        if (w0) {
            run = !decided && lev < L.nlev;
            const bool any = __ballot(run) != 0ull;
            if (lane == 0) { s_any = any ? 1u : 0u; s_total[0] = 0u; s_total[1] = 0u; }
            if (any) {
                const int lv = run ? lev : 0;
                const unsigned long long bk = qkey[slot];
                // a query that holds no bound yet (or a seed too wide for level 0, see seed_pre) first looks at the x-row through
                // its own cell only; the pass after that prunes with what it found
                prescan = run && !did_pre && (bk == ~0ull || (widepre && lev == 0));
                float rub2 = INFINITY; // squared pruning radius (with slack), +inf = no pruning
                if (!prescan && bk != ~0ull) {
                    const float slack = __uint_as_float(ltab[4 * lv + 1].y);
                    const float rub = sqrt_up(__uint_as_float((unsigned)(bk >> 32))) * inv1e * 1.000001f + slack;
                    rub2 = rub * rub;
                }
                reinterpret_cast<float*>(&qrec[slot])[3] = __int_as_float(lv);
                qaux[slot] = make_float2(rub2, __int_as_float((run ? 1 : 0) | (prescan ? 2 : 0)));
            }
        }
        __syncthreads();
        if (!s_any) break;
        NN_TICK(1);
        // ---- (1b) all waves: lane = query, wave = row group
        {
            float mf, g_cell, g_slack;
            bool covers;
            const float4 qr = qrec[slot];
            const float2 qa = qaux[slot];
            const int lv = __float_as_int(qr.w);
            const int fl = __float_as_int(qa.y);
            const bool qrun = (fl & 1) != 0, qpre = (fl & 2) != 0;
            const float rub2 = qa.x;
            GridParams g;
            const gunsigned* __restrict__ cs; // (global address space: rebuilt from integers, it would be read through the flat path)
            {
                const uint4 a = ltab[4 * lv], b = ltab[4 * lv + 1], c2 = ltab[4 * lv + 2], d = ltab[4 * lv + 3];
                g.ox = __uint_as_float(a.x); g.oy = __uint_as_float(a.y); g.oz = __uint_as_float(a.z); g.cell = __uint_as_float(a.w);
                g.inv_cell = __uint_as_float(b.x); g.slack = __uint_as_float(b.y); g.nx = (int)b.z; g.ny = (int)b.w;
                g.nz = (int)c2.x; g.ncells = (int)c2.y;
                cs = reinterpret_cast<const gunsigned*>(((unsigned long long)d.y << 32) | d.x);
            }
            const float fx = (qr.x - g.ox) * g.inv_cell, fy = (qr.y - g.oy) * g.inv_cell, fz = (qr.z - g.oz) * g.inv_cell;
            const float flx = floorf(fx), fly = floorf(fy), flz = floorf(fz);
            const int cx = (int)fminf(fmaxf(flx, -1.0e6f), 1.0e6f);
            const int cy = (int)fminf(fmaxf(fly, -1.0e6f), 1.0e6f);
            const int cz = (int)fminf(fmaxf(flz, -1.0e6f), 1.0e6f);
            mf = fminf(fx - flx, 1.0f - (fx - flx));
            mf = fminf(mf, fminf(fy - fly, 1.0f - (fy - fly)));
            mf = fminf(mf, fminf(fz - flz, 1.0f - (fz - flz)));
            if (!(mf >= 0.f)) mf = 0.f;
            g_cell = g.cell; g_slack = g.slack;
            covers = cx - 1 <= 0 && cx + 1 >= g.nx - 1 && cy - 1 <= 0 && cy + 1 >= g.ny - 1 && cz - 1 <= 0 && cz + 1 >= g.nz - 1;
            if (w0) { const float margin = fmaxf((1.0f + mf) * g_cell - g_slack, 0.f); qdec[slot] = make_float2(margin * margin, covers ? 1.f : 0.f); } // (not carried in registers across the piece steps)
            // row ranges: branch-free so that all 2 NR loads of a lane leave together (unreached rows read cs[0] twice)
            unsigned rs[NR], rn[NR];
            {
                const float ylo = (fy - fly) * g.cell, yhi = (1.0f - (fy - fly)) * g.cell;
                const float zlo = (fz - flz) * g.cell, zhi = (1.0f - (fz - flz)) * g.cell;
                unsigned ia[NR], ib[NR];
#pragma unroll
                for (int sl = 0; sl < NR; ++sl) {
                    const int rr = wave + sl * NW; // wave-uniform
                    const int dy = (rr % 3) - 1, dz = (rr / 3) - 1;
                    bool reach = qrun && rr < 9 && (!qpre || rr == 4);
                    const float ddy = dy == 0 ? 0.f : (dy < 0 ? ylo : yhi);
                    const float ddz = dz == 0 ? 0.f : (dz < 0 ? zlo : zhi);
                    const float rem2 = rub2 - (ddy * ddy + ddz * ddz); // +inf without a bound
                    reach = reach && rem2 >= 0.f;
                    int xa = cx - 1, xb = cx + 1;
                    if (rem2 != INFINITY) {
                        const float rem = sqrt_up(fmaxf(rem2, 0.f));
                        const int xl = (int)fmaxf(floorf((qr.x - rem - g.ox) * g.inv_cell), -1.0e6f);
                        const int xh = (int)fminf(floorf((qr.x + rem - g.ox) * g.inv_cell), 1.0e6f);
                        xa = xl > xa ? xl : xa;
                        xb = xh < xb ? xh : xb;
                    }
                    const int y = cy + dy, z = cz + dz;
                    xa = xa < 0 ? 0 : xa;
                    xb = xb > g.nx - 1 ? g.nx - 1 : xb;
                    reach = reach && y >= 0 && y < g.ny && z >= 0 && z < g.nz && xa <= xb;
                    const int rowbase = (z * g.ny + y) * g.nx;
                    ia[sl] = reach ? (unsigned)(rowbase + xa) : 0u;
                    ib[sl] = reach ? (unsigned)(rowbase + xb + 1) : 0u;
                }
#pragma unroll
                for (int sl = 0; sl < NR; ++sl) { rs[sl] = cs[ia[sl]]; rn[sl] = cs[ib[sl]]; }
#pragma unroll
                for (int sl = 0; sl < NR; ++sl) rn[sl] -= rs[sl]; // 0 for unreached rows (both loads hit cs[0])
            }
            // pieces: length 8; a lane's pieces start at the wave's exclusive prefix (DPP scan) behind the wave's share of the list
            // (one LDS atomic per wave).  If the workgroup's pieces overflow the list (dense cells, degenerate maps), everybody
            // repeats the count with pieces long enough to fit.
            int plb = PLB;
            unsigned base = 0;
            for (int round = 0; round < 2; ++round) {
                unsigned np = 0;
#pragma unroll
                for (int sl = 0; sl < NR; ++sl) np += (rn[sl] + ((1u << plb) - 1u)) >> plb;
                const unsigned incl = wave_incl_scan(np);
                const unsigned wtot = (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
                unsigned wb = 0;
                if (lane == 63 && wtot) wb = atomicAdd(&s_total[round], wtot);
                wb = (unsigned)__builtin_amdgcn_readlane((int)wb, 63);
                base = wb + incl - np;
                __syncthreads();
                const unsigned tot = s_total[round];
                if (tot <= (unsigned)CAP) break;
                // sum over rows of ceil(c / 2^(plb + k)) <= tot / 2^k + (number of rows <= 9 Q)
                int k = 1;
                while ((tot >> k) > (unsigned)(CAP - 9 * Q)) ++k;
                plb += k;
            }
#pragma unroll
            for (int sl = 0; sl < NR; ++sl) {
                unsigned s = rs[sl], c = rn[sl];
                while (c) {
                    const unsigned t = c < (1u << plb) ? c : (1u << plb);
                    pieces[base++] = make_uint2(s, (t << 8) | (unsigned)slot);
                    s += t; c -= t;
                }
            }
        }
        __syncthreads();
        const unsigned total = s_total[1] ? s_total[1] : s_total[0];
        NN_TICK(2);
#ifdef ICPMI_NN_TIMING
        tacc[6] += 1; tacc[7] += total;
#endif
        // ---- (2) all waves, lane per piece: eight independent 16-byte loads from ONE address register (immediate offsets).
        //      The atomics of a step and the look at their outcome are separated by a workgroup barrier.
        for (unsigned i0 = 0; i0 < total; i0 += (unsigned)NT) {
            const unsigned i = i0 + (unsigned)tid;
            const bool has = i < total;
            const uint2 e = pieces[has ? i : 0u];
            const unsigned qsl = e.y & 255u, cnt = has ? e.y >> 8 : 0u;
            const float4 qr = qrec[qsl];
            // (global address space: a pointer rebuilt from integers would otherwise be loaded through the flat path)
            const uint4 lt2 = ltab[4 * __float_as_int(qr.w) + 2];
            const gfloat4* mp = reinterpret_cast<const gfloat4*>((((unsigned long long)lt2.w << 32) | lt2.z) + (unsigned long long)e.x * 16ull);
            unsigned long long kb = ~0ull;
            unsigned pb = 0;
            float bx = 0.f, by = 0.f, bz = 0.f; // the best candidate's coordinates ride along (a reload would be one more trip)
            for (unsigned c0 = 0; c0 < cnt; c0 += 8u) { // one round unless a dense cell forced pieces longer than 8
                vf4 q[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) q[u] = mp[c0 + u]; // past the run's end: masked below (the level arrays are padded)
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float d2 = sqdist3(qr.x, qr.y, qr.z, q[u].x, q[u].y, q[u].z);
                    const unsigned long long key = pack_key(d2, __float_as_uint(q[u].w));
                    bool ok = c0 + (unsigned)u < cnt;
                    if (!allow_self) ok = ok && d2 > 1.1920929e-07f;
                    if (ok && key < kb) { kb = key; pb = c0 + (unsigned)u; bx = q[u].x; by = q[u].y; bz = q[u].z; }
                }
            }
            if (kb != ~0ull) atomicMin(&qkey[qsl], kb);
            __syncthreads();
            // keys are unique per map point: at most one lane of the workgroup finds its own key in the slot
            if (kb != ~0ull && __atomic_load_n(&qkey[qsl], __ATOMIC_RELAXED) == kb) {
                qwin[qsl] = (e.x + pb) | (__float_as_uint(qr.w) << 28);
                qpt[qsl] = make_float4(bx, by, bz, __uint_as_float((unsigned)(kb & 0xffffffffull)));
            }
        }
        __syncthreads();
        NN_TICK(3);
        // ---- (3) wave 0 decides
        if (w0 && run) {
            const unsigned long long k2 = qkey[slot];
            if (prescan) did_pre = true;
            else {
                const float2 dq = qdec[slot];
                const float m2 = dq.x;
                const bool covers = dq.y != 0.f;
                const float bd2 = __uint_as_float((unsigned)(k2 >> 32));
                decided = (k2 != ~0ull && bd2 <= m2 * err2) || m2 > maxr2 || covers;
                if (!decided) { ++lev; did_pre = false; }
            }
        }
        NN_TICK(4);
    }
    if (w0) { best.key = qkey[slot]; best.sidx = (int)qwin[slot]; }

    float bd2 = __uint_as_float((unsigned)(best.key >> 32));
    const bool found = best.key != ~0ull && bd2 <= maxr2;
    if (!found) bd2 = INFINITY;
    const bool writer = active;
    if (hist0) { // coarse level-0 histogram through LDS first: its barrier must not sit behind the global stores below
        if (writer && decided && bd2 != INFINITY && bd2 > 0.f) atomicAdd(&lh[__float_as_uint(bd2) >> 24], 1u); // (a queued query joins the histogram in the brute pass)
        __syncthreads();
        for (int t = tid; t < 256; t += NT)
            if (lh[t]) atomicAdd(&hist0[ICPMI_S2_C0 + (blockIdx.x % ICPMI_S2_COPIES) * 256 + t], lh[t]);
    }
    if (writer) {
        int bs = -1;
        float4 mpt = make_float4(0.f, 0.f, 0.f, 0.f);
        if (found) {
            const unsigned lvb = (unsigned)best.sidx >> 28, pos = (unsigned)best.sidx & 0x0fffffffu;
            if (lvb == 0) bs = (int)pos;
            else {
                const uint4 d = ltab[4 * lvb + 3];
                bs = (int)reinterpret_cast<const unsigned*>(((unsigned long long)d.w << 32) | d.z)[pos];
            }
            mpt = qpt[slot];
        }
        st_stream(out_sidx + orig, bs);
        st_stream(out_d2 + orig, bd2);
        if (match_pt) st_stream(match_pt + orig, make_float4(mpt.x, mpt.y, mpt.z, __uint_as_float((unsigned)(best.key & 0xffffffffull))));
        if (hist0 && decided && bd2 != INFINITY && bd2 > 0.f) {
            const unsigned bits = __float_as_uint(bd2);
            atomicAdd(&hist0[ICPMI_S2_F0 + (blockIdx.x % ICPMI_S2_FCOPIES) * 65536 + ICPMI_S2_FIDX(bits >> 16)], 1u);
        }
        if (!decided) {
            const unsigned hslot = atomicAdd(&st->hard_count, 1u);
            hard[hslot] = (unsigned)(match_pt ? qi : (qindex ? qindex[qi] : qi)); // the brute pass works on the order the state is kept in
        }
    }
#ifdef ICPMI_NN_TIMING
    NN_TICK(5);
    if (threadIdx.x == 0 && st_iter > 1 && (blockIdx.x % 7) == 0) { // the heaviest of a sample of the steady launches' workgroups: its life and its pieces
        atomicMax(&st->dbg[20], (unsigned long long)(clock64() - t_c0));
        atomicMax(&st->dbg[21], (unsigned long long)tacc[7]);
    }
    if (threadIdx.x == 0 && (blockIdx.x % 61) == 0) { // a sample: same-address atomics from every wave would dominate
        const int tb = st_iter > 1 ? 8 : 0; // steady launches in dbg[8..15], the first two in dbg[0..7]
        for (int i = 0; i < 6; ++i) atomicAdd(&st->dbg[tb + i], (unsigned long long)tacc[i]);
        atomicAdd(&st->dbg[tb + 6], (unsigned long long)tacc[6]);
        atomicAdd(&st->dbg[tb + 7], 1ull);
        atomicAdd(&st->dbg[16 + (st_iter > 1 ? 1 : 0)], (unsigned long long)tacc[7]);
        atomicAdd(&st->dbg[22], (unsigned long long)(clock64() - t_c0));       // shader clock ticks of this workgroup's life
        atomicAdd(&st->dbg[23], (unsigned long long)(wall_clock64() - t_w0));  // ... and 100 MHz constant-clock ticks of the same
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// 2 <= k <= 8 over the grid pyramid: the k = 1 scheme with a sorted list per lane.  G lanes serve one
// query; candidates of a level pass are dealt to the lanes as in nn1_ml_kernel, every lane keeps the
// KMAX best of ITS candidates; after a pass the group merges by KMAX rounds of "extract the group
// minimum" (shuffle butterfly) into a list replicated in all lanes, whose k-th key bounds the next
// pass (rows / cells beyond it are skipped, candidates beyond it dropped, candidates equal to a merged
// entry -- the same point seen again -- dropped).  Iterations > 0 are seeded with the previous
// iteration's k matches (lane j fetches match j), which makes the first bound tight.
// ------------------------------------------------------------------------------------------------
template <int G, int KMAX>
__global__ __launch_bounds__(NN_BLOCK) void nnk_ml_kernel(const float4* __restrict__ queries, const int* __restrict__ qindex, int n,
                                                          const float* __restrict__ Tptr, const uint4* __restrict__ ltab_g, int nlev,
                                                          int k, float maxr2, int allow_self_i, int seeded, int* __restrict__ out_sidx,
                                                          float* __restrict__ out_d2, IcpState* __restrict__ st,
                                                          unsigned* __restrict__ hard, int out_sorted, float inv1e, float err2)
{
    static_assert(G == 8, "lanes per query");
    constexpr int NB = 4;
    constexpr int NR = (9 + G - 1) / G;
    if (st->done) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) nn_stamp_open(st); // (device clock of this launch: common.h)
    __shared__ uint4 ltab[ICPMI_MAXLEV * 4];
    if (threadIdx.x < ICPMI_MAXLEV * 4) ltab[threadIdx.x] = ltab_g[threadIdx.x];
    const bool allow_self = allow_self_i != 0;
    const int chunk = gridDim.x >> 3;
    const int lb = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    const int tid = lb * NN_BLOCK + threadIdx.x;
    const int qi = tid / G;
    const int sub = tid % G;
    const bool active = qi < n;
    const float4 r = queries[active ? qi : 0];
    // out_sorted (the loop, r3): the k matches of a query live at its slot of the TILE-SORTED order -- seeds and results are
    // coalesced, and the pair sums then walk spatially coherent matches (their gathers share cache lines)
    const int orig = (qindex && !out_sorted) ? qindex[active ? qi : 0] : qi;
    float3 p;
    if (Tptr) p = xf_point(Tptr, r.x, r.y, r.z, r.w);
    else p = make_float3(r.x, r.y, r.z);
    const int lane = threadIdx.x & 63;
    const int gbase = lane - sub;
    __syncthreads();

    // private (per lane) and merged (replicated) lists; sidx = position in its level | level << 28
    unsigned long long pk[KMAX], mk[KMAX];
    int ps[KMAX], ms[KMAX];
#pragma unroll
    for (int i = 0; i < KMAX; ++i) { pk[i] = ~0ull; ps[i] = -1; mk[i] = ~0ull; ms[i] = -1; }
    unsigned long long bound = ~0ull; // k-th merged key once k candidates are held

    auto offer = [&](unsigned long long key, int sidx) {
        if (key > bound || key >= pk[KMAX - 1]) return;
        bool dup = false;
#pragma unroll
        for (int i = 0; i < KMAX; ++i) dup |= key == mk[i];
        if (dup) return;
#pragma unroll
        for (int i = KMAX - 1; i >= 0; --i) {
            const unsigned long long prev = i > 0 ? pk[i - 1] : 0ull;
            const int prevs = i > 0 ? ps[i - 1] : -1;
            if (i > 0 && key < prev) { pk[i] = prev; ps[i] = prevs; }
            else if (key < pk[i]) { pk[i] = key; ps[i] = sidx; }
        }
    };
    // merged <- the KMAX smallest of (merged U all private lists); private lists cleared.  A (G + 1)-way merge: every round takes
    // the smaller of the group minimum of the private heads (shuffle butterfly) and the head of the OLD merged list, which all
    // lanes hold (r1 pushed the old merged entries back through lane 0's private list first: KMAX sorted insertions that the
    // whole wave executed).  Private lists never hold a merged point (offer() drops those), so the two sources are disjoint.
    auto merge = [&]() {
        unsigned long long omk[KMAX];
        int oms[KMAX];
#pragma unroll
        for (int i = 0; i < KMAX; ++i) { omk[i] = mk[i]; oms[i] = ms[i]; }
        int head = 0, hm = 0;
#pragma unroll
        for (int j = 0; j < KMAX; ++j) {
            unsigned long long ck = ~0ull;
            int cs_ = -1;
#pragma unroll
            for (int i = 0; i < KMAX; ++i) if (i == head) { ck = pk[i]; cs_ = ps[i]; }
            const unsigned long long mine = ck;
#pragma unroll
            for (int off = G / 2; off > 0; off >>= 1) {
                const unsigned long long ok = __shfl_xor(ck, off, 64);
                const int os = __shfl_xor(cs_, off, 64);
                if (ok < ck) { ck = ok; cs_ = os; }
            }
            unsigned long long ok2 = ~0ull;
            int os2 = -1;
#pragma unroll
            for (int i = 0; i < KMAX; ++i) if (i == hm) { ok2 = omk[i]; os2 = oms[i]; }
            if (ok2 < ck) { mk[j] = ok2; ms[j] = os2; ++hm; }
            else {
                mk[j] = ck; ms[j] = cs_;
                if (mine == ck && mine != ~0ull) ++head; // keys are unique per map point: every holder of the winner advances
            }
        }
#pragma unroll
        for (int i = 0; i < KMAX; ++i) { pk[i] = ~0ull; ps[i] = -1; }
        unsigned long long kth = ~0ull;
#pragma unroll
        for (int i = 0; i < KMAX; ++i) if (i == k - 1) kth = mk[i];
        bound = kth;
    };

    bool decided = !active;
    int lev = 0;
    if ((seeded & 1) && active && allow_self && st->iter > 0) {
        // lane j re-evaluates previous matches j, j + G, ... under the current transform
        if (KMAX <= G) {
            if (sub < k) {
                const int sp = out_sidx[(size_t)k * orig + sub];
                if (sp >= 0) {
                    const float4 q = reinterpret_cast<const float4*>(((unsigned long long)ltab[2].w << 32) | ltab[2].z)[sp];
                    pk[0] = pack_key(sqdist3(p.x, p.y, p.z, q.x, q.y, q.z), __float_as_uint(q.w));
                    ps[0] = sp;
                }
            }
        } else {
            for (int j = sub; j < k; j += G) {
                const int sp = out_sidx[(size_t)k * orig + j];
                if (sp >= 0) {
                    const float4 q = reinterpret_cast<const float4*>(((unsigned long long)ltab[2].w << 32) | ltab[2].z)[sp];
                    offer(pack_key(sqdist3(p.x, p.y, p.z, q.x, q.y, q.z), __float_as_uint(q.w)), sp);
                }
            }
        }
    }
    if (seeded & 1) {
        merge();
        if (bound != ~0ull) { // first level whose block contains the ball of the k-th seed
            const float ub = sqrtf(__uint_as_float((unsigned)(bound >> 32))) * inv1e * 1.000001f;
            for (; lev < nlev - 1; ++lev) {
                const uint4 a = ltab[4 * lev], b = ltab[4 * lev + 1];
                const float inv = __uint_as_float(b.x);
                const float fx = (p.x - __uint_as_float(a.x)) * inv, fy = (p.y - __uint_as_float(a.y)) * inv, fz = (p.z - __uint_as_float(a.z)) * inv;
                float mfl = fminf(fx - floorf(fx), 1.0f - (fx - floorf(fx)));
                mfl = fminf(mfl, fminf(fy - floorf(fy), 1.0f - (fy - floorf(fy))));
                mfl = fminf(mfl, fminf(fz - floorf(fz), 1.0f - (fz - floorf(fz))));
                if (!(mfl >= 0.f)) mfl = 0.f;
                if (ub <= (1.0f + mfl) * __uint_as_float(a.w) - 2.0f * __uint_as_float(b.y)) break;
            }
        }
    }

    // A query that holds no bound yet first scans only the x-row through its own cell (about 1/9 of
    // the block); if that yields k candidates the same level is then searched with their bound.
    bool rowonly = bound == ~0ull;
    // Seeds too wide for level 0 (iteration 1: the first solve moved the reading a long way): the same own-row pass at level 0
    // first, pruned by the seed bound -- the k-th of the row is usually far tighter, and the coarse levels, whose blocks hold 8 / 64
    // times the candidates, are then rarely needed (r3, knn 6: iteration 1 172 -> 75 us; without this the seeded launch was SLOWER than
    // the unseeded one, 115 us)
    if ((seeded & 2) && lev > 0) { lev = 0; rowonly = true; }
    while (lev < nlev && !decided) {
        GridParams g;
        const float4* __restrict__ map;
        const unsigned* __restrict__ cs;
        {
            const uint4 a = ltab[4 * lev], b = ltab[4 * lev + 1], c2 = ltab[4 * lev + 2], d = ltab[4 * lev + 3];
            g.ox = __uint_as_float(a.x); g.oy = __uint_as_float(a.y); g.oz = __uint_as_float(a.z); g.cell = __uint_as_float(a.w);
            g.inv_cell = __uint_as_float(b.x); g.slack = __uint_as_float(b.y); g.nx = (int)b.z; g.ny = (int)b.w;
            g.nz = (int)c2.x; g.ncells = (int)c2.y;
            map = reinterpret_cast<const float4*>(((unsigned long long)c2.w << 32) | c2.z);
            cs = reinterpret_cast<const unsigned*>(((unsigned long long)d.y << 32) | d.x);
        }
        const float fx = (p.x - g.ox) * g.inv_cell, fy = (p.y - g.oy) * g.inv_cell, fz = (p.z - g.oz) * g.inv_cell;
        const float flx = floorf(fx), fly = floorf(fy), flz = floorf(fz);
        const int cx = (int)fminf(fmaxf(flx, -1.0e6f), 1.0e6f);
        const int cy = (int)fminf(fmaxf(fly, -1.0e6f), 1.0e6f);
        const int cz = (int)fminf(fmaxf(flz, -1.0e6f), 1.0e6f);
        float mf = fminf(fx - flx, 1.0f - (fx - flx));
        mf = fminf(mf, fminf(fy - fly, 1.0f - (fy - fly)));
        mf = fminf(mf, fminf(fz - flz, 1.0f - (fz - flz)));
        if (!(mf >= 0.f)) mf = 0.f;

        float rub2 = INFINITY;
        if (bound != ~0ull) {
            const float rub = sqrtf(__uint_as_float((unsigned)(bound >> 32))) * inv1e * 1.000001f + g.slack;
            rub2 = rub * rub;
        }
        unsigned rs[NR], rn[NR];
        {
            const float ylo = (fy - fly) * g.cell, yhi = (1.0f - (fy - fly)) * g.cell;
            const float zlo = (fz - flz) * g.cell, zhi = (1.0f - (fz - flz)) * g.cell;
#pragma unroll
            for (int sl = 0; sl < NR; ++sl) {
                const int rr = sub + sl * G;
                unsigned s = 0, cnt = 0;
                if (rr < 9 && (!rowonly || rr == 4)) {
                    const int dy = (rr % 3) - 1, dz = (rr / 3) - 1;
                    int xa = cx - 1, xb = cx + 1;
                    bool reach = true;
                    if (rub2 != INFINITY) {
                        const float ddy = dy == 0 ? 0.f : (dy < 0 ? ylo : yhi);
                        const float ddz = dz == 0 ? 0.f : (dz < 0 ? zlo : zhi);
                        const float rem2 = rub2 - (ddy * ddy + ddz * ddz);
                        reach = rem2 >= 0.f;
                        const float rem = sqrtf(fmaxf(rem2, 0.f));
                        const int xl = (int)fmaxf(floorf((p.x - rem - g.ox) * g.inv_cell), -1.0e6f);
                        const int xh = (int)fminf(floorf((p.x + rem - g.ox) * g.inv_cell), 1.0e6f);
                        xa = xl > xa ? xl : xa;
                        xb = xh < xb ? xh : xb;
                    }
                    if (reach) {
                        unsigned e;
                        row_run(g, cs, xa, xb, cy + dy, cz + dz, s, e);
                        cnt = e - s;
                    }
                }
                rs[sl] = s; rn[sl] = cnt;
            }
        }
        unsigned Pr[10], Or[9];
        Pr[0] = 0;
#pragma unroll
        for (int rr = 0; rr < 9; ++rr) {
            const int src = gbase + (rr % G);
            const unsigned s = __shfl(rs[rr / G], src, 64);
            const unsigned c = __shfl(rn[rr / G], src, 64);
            Or[rr] = s - Pr[rr];
            Pr[rr + 1] = Pr[rr] + c;
        }
        const unsigned total = Pr[9];
        for (unsigned k0 = (unsigned)sub; k0 < total; k0 += (unsigned)(G * NB)) {
            float4 q[NB];
            unsigned gi[NB];
            bool ok[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const unsigned kq = k0 + (unsigned)(u * G);
                ok[u] = kq < total;
                const unsigned kk = ok[u] ? kq : k0;
                unsigned off = Or[0];
#pragma unroll
                for (int rr = 1; rr < 9; ++rr) off = kk >= Pr[rr] ? Or[rr] : off;
                gi[u] = kk + off;
                q[u] = map[gi[u]];
            }
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const float d2 = sqdist3(p.x, p.y, p.z, q[u].x, q[u].y, q[u].z);
                if (ok[u] && (allow_self || d2 > 1.1920929e-07f)) offer(pack_key(d2, __float_as_uint(q[u].w)), (int)(gi[u] | ((unsigned)lev << 28)));
            }
        }
        merge();
        if (rowonly) { rowonly = false; continue; }
        const float margin = fmaxf((1.0f + mf) * g.cell - g.slack, 0.f);
        const float m2 = margin * margin;
        const float kd2 = __uint_as_float((unsigned)(bound >> 32));
        const bool covers = cx - 1 <= 0 && cx + 1 >= g.nx - 1 && cy - 1 <= 0 && cy + 1 >= g.ny - 1 && cz - 1 <= 0 && cz + 1 >= g.nz - 1;
        decided = (bound != ~0ull && kd2 <= m2 * err2) || m2 > maxr2 || covers;
        ++lev;
    }

    if (active) {
        // lane j writes results j, j + G, ... (static select from the replicated merged list)
        for (int jj = sub; jj < k; jj += G) {
        unsigned long long key = ~0ull;
        int sx = -1;
#pragma unroll
        for (int i = 0; i < KMAX; ++i) if (i == jj) { key = mk[i]; sx = ms[i]; }
        {
            float d2 = __uint_as_float((unsigned)(key >> 32));
            int bs = -1;
            if (key != ~0ull && d2 <= maxr2) {
                const unsigned lv = (unsigned)sx >> 28, pos = (unsigned)sx & 0x0fffffffu;
                if (lv == 0) bs = (int)pos;
                else {
                    const uint4 d = ltab[4 * lv + 3];
                    bs = (int)reinterpret_cast<const unsigned*>(((unsigned long long)d.w << 32) | d.z)[pos];
                }
            } else d2 = INFINITY;
            out_sidx[(size_t)k * orig + jj] = bs;
            out_d2[(size_t)k * orig + jj] = d2;
        }
        }
        if (sub == 0 && !decided) {
            const unsigned slot = atomicAdd(&st->hard_count, 1u);
            hard[slot] = (unsigned)(qindex ? qindex[qi] : qi); // the brute pass works on the caller's order
        }
    }
}

// ------------------------------------------------------------------------------------------------
// nnk_wg_kernel: 2 <= k <= 8 in the loop (seeded launches), with the roles of nn1_wg_kernel.  A workgroup of four waves owns 64
// queries:
//   (1a) wave 0, lane per query: keeps the query's sorted k-list in registers (seeded with the previous iteration's k matches
//        under the current transform), publishes the level, the pruning radius and the ACCEPT BOUND (k-th key held, else the
//        key of maxr2);
//   (1b) all waves, lane = query, wave = row group: row ranges -> pieces of 8 candidates;
//   (2)  all waves, lane per piece: candidates with key <= bound go to the query's list in LDS (one LDS atomic per piece);
//   (3)  wave 0 merges its query's list into the k-list (the same point seen again is dropped by its key) and decides.
// A list that overflows (CAPQ entries) is not lost work: the k-list of what DID arrive tightens the bound, and the pass is
// repeated -- of the candidates under the old bound at least CAPQ - k lie above the new one, so this terminates.  Keys, the
// exactness rule per level and the output are those of nnk_ml_kernel: the same k points in the same order.
// Measured (r3, knn 6, 100 k queries, 1 M points): steady launches 37.3 -> 25 us.  NOT for wide bounds: a launch whose seeds
// moved far (iteration 1) overflows the lists and repeats passes (172 us here; nnk_ml_kernel with its own-row pass first: 75 us),
// the unseeded launch takes 270 us against 115 us -- the launcher keeps nnk_ml_kernel for iterations 0 and 1.  Level 0 of the fused quantile selection
// from this kernel's tail (k atomics per query on the fine bins) made a launch 55 us; the stand-alone builder (12.5 us), which
// combines 4096 matches in LDS before it touches memory, stays.
// ------------------------------------------------------------------------------------------------
#ifndef NNK_WG_WAVES
#define NNK_WG_WAVES 5
#endif
template <int KMAX>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(KMAX <= 8 ? NNK_WG_WAVES : 3))) void nnk_wg_kernel(const float4* __restrict__ queries, const int* __restrict__ qindex, int n,
                                                     const float* __restrict__ Tptr, const uint4* __restrict__ ltab_g, int nlev,
                                                     int k, float maxr2, int* __restrict__ out_sidx, float* __restrict__ out_d2,
                                                     IcpState* __restrict__ st, unsigned* __restrict__ hard, int out_sorted, int seed_pre,
                                                     unsigned long long* __restrict__ win /* speculative level 0 of the fused selection (common.h: ICPMI_S2_WIN), or nullptr */,
                                                     float inv1e, float err2 /* KDTreeMatcher's epsilon, as in nn1_wg_kernel */)
{
    constexpr int NW = 4, NT = 64 * NW, Q = 64, NR = 3;
    constexpr int CAP = 14 * Q;   // pieces per pass (> 9 Q: one piece per row always fits)
#ifndef ICPMI_NNK_PLB
#define ICPMI_NNK_PLB 3
#endif
    constexpr int PLB = ICPMI_NNK_PLB;
#ifndef ICPMI_NNK_CAPQ
#define ICPMI_NNK_CAPQ 24
#endif
    constexpr int CAPQ = ICPMI_NNK_CAPQ;      // list entries per query and pass (> KMAX).  (48, at three waves per SIMD, for the wide first launches: slower)
    static_assert(CAPQ > KMAX, "a repeated pass must make progress");
    __shared__ uint4 ltab[ICPMI_MAXLEV * 4];
    __shared__ uint2 pieces[CAP];                 // {position of the first candidate in its level array, count << 8 | query slot}
    __shared__ float4 qrec[Q];                    // transformed query, w = bits of its current level
    __shared__ float2 qaux[Q];                    // squared pruning radius (+inf: none), flags: 1 = searching, 2 = own row only
    __shared__ unsigned long long qbound[Q];      // candidates with a key above this are dropped
    __shared__ unsigned qcnt[Q];                  // candidates offered to the list this pass (may exceed CAPQ)
    __shared__ unsigned long long qlk[CAPQ * Q];  // [entry][query]
    __shared__ unsigned qlp[CAPQ * Q];
    __shared__ unsigned s_any, s_total[2];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const bool w0 = wave == 0;
    const int wgs = (int)((((long long)n + Q - 1) / Q + 7) / 8 * 8);
    if ((int)blockIdx.x >= wgs) return;
    const int chunk = wgs >> 3;
    const int lb = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    const int slot = lane;
    const int qi = lb * Q + slot;
    const bool active = w0 && qi < n;
    if (st->done) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) nn_stamp_open(st); // (device clock of this launch: common.h)
    // the window sits around the prefix the PREVIOUS iteration's selection picked (sel2_scan_hist_kernel ran before this launch)
    const bool use_win = win != nullptr && st->iter > 0;
    const unsigned win_c = st->sel_prefix_l[0];
    const unsigned win_lo = win_c > (unsigned)(ICPMI_WIN_BINS / 2) ? win_c - (unsigned)(ICPMI_WIN_BINS / 2) : 0u;
    if (use_win) { // level 1 of the selection is cleared by whoever builds level 0 (here: the window; its last reader was the previous pair-sum kernel)
        unsigned* __restrict__ l1 = reinterpret_cast<unsigned*>(win) - ICPMI_S2_WIN + ICPMI_S2_C1;
        for (int i = (int)blockIdx.x * NT + tid; i < 256 + 65536; i += wgs * NT) l1[i] = 0u;
    }
#ifdef ICPMI_NN_TIMING
    long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tlast = clock64();
    const long long t_c0 = tlast;
    unsigned long long t_ovf = 0, t_off = 0;
    const int st_iter = st->iter;
#endif
    if (tid < ICPMI_MAXLEV * 4) ltab[tid] = ltab_g[tid];
    int orig = 0;
    float3 p = make_float3(0.f, 0.f, 0.f);
    unsigned long long mk[KMAX];
    int ms[KMAX];
#pragma unroll
    for (int i = 0; i < KMAX; ++i) { mk[i] = ~0ull; ms[i] = -1; }
    auto insert = [&](unsigned long long key, int sidx, bool check_dup) {
        if (key >= mk[KMAX - 1]) return;
        if (check_dup) {
            bool dup = false;
#pragma unroll
            for (int i = 0; i < KMAX; ++i) dup |= key == mk[i];
            if (dup) return;
        }
#pragma unroll
        for (int i = KMAX - 1; i >= 0; --i) {
            const unsigned long long prev = i > 0 ? mk[i - 1] : 0ull;
            const int prevs = i > 0 ? ms[i - 1] : -1;
            if (i > 0 && key < prev) { mk[i] = prev; ms[i] = prevs; }
            else if (key < mk[i]) { mk[i] = key; ms[i] = sidx; }
        }
    };
    auto kth = [&]() {
        unsigned long long v = ~0ull;
#pragma unroll
        for (int i = 0; i < KMAX; ++i) if (i == k - 1) v = mk[i];
        return v;
    };
    const unsigned long long maxkey = pack_key(maxr2, 0xffffffffu);
    // first level >= from whose 3 x 3 x 3 block contains the ball of this key's distance around the query
    auto level_for = [&](unsigned long long b, int from) {
        const float ub = sqrtf(__uint_as_float((unsigned)(b >> 32))) * inv1e * 1.000001f;
        int l = from;
        for (; l < nlev - 1; ++l) {
            const uint4 a = ltab_g[4 * l], bb = ltab_g[4 * l + 1];
            const float inv = __uint_as_float(bb.x);
            const float fx = (p.x - __uint_as_float(a.x)) * inv, fy = (p.y - __uint_as_float(a.y)) * inv, fz = (p.z - __uint_as_float(a.z)) * inv;
            float mfl = fminf(fx - floorf(fx), 1.0f - (fx - floorf(fx)));
            mfl = fminf(mfl, fminf(fy - floorf(fy), 1.0f - (fy - floorf(fy))));
            mfl = fminf(mfl, fminf(fz - floorf(fz), 1.0f - (fz - floorf(fz))));
            if (!(mfl >= 0.f)) mfl = 0.f;
            if (ub <= (1.0f + mfl) * __uint_as_float(a.w) - 2.0f * __uint_as_float(bb.y)) break;
        }
        return l;
    };
    bool widepre = false;
    int lev = 0;
    unsigned long long seedb = ~0ull; // key of the farthest previous match (all k valid), an upper bound of the k-th key
    if (w0) {
        const float4 r = queries[active ? qi : 0];
        orig = (qindex && !out_sorted) ? qindex[active ? qi : 0] : qi;
        if (Tptr) p = xf_point(Tptr, r.x, r.y, r.z, r.w);
        else p = make_float3(r.x, r.y, r.z);
        if (active && st->iter > 0) { // the previous k matches under the current transform (all loads first, then the insertions)
            const gfloat4* l0 = reinterpret_cast<const gfloat4*>(((unsigned long long)ltab_g[2].w << 32) | ltab_g[2].z);
            int sp[KMAX];
            vf4 sq[KMAX];
#pragma unroll
            for (int j = 0; j < KMAX; ++j) sp[j] = j < k ? out_sidx[(size_t)k * orig + j] : -1;
#pragma unroll
            for (int j = 0; j < KMAX; ++j) sq[j] = l0[sp[j] >= 0 ? sp[j] : 0];
            unsigned long long sb = 0ull; // the bound is the farthest of the k; the points themselves are met again in the pass
            unsigned long long sk[KMAX];
            bool allv = true;
#pragma unroll
            for (int j = 0; j < KMAX; ++j) {
                sk[j] = pack_key(sqdist3(p.x, p.y, p.z, sq[j].x, sq[j].y, sq[j].z), __float_as_uint(sq[j].w));
                if (j < k) { sb = sk[j] > sb ? sk[j] : sb; allv = allv && sp[j] >= 0; }
            }
            if (allv) seedb = sb;
            // epsilon > 0: the pass only visits what lies within (bound) / (1 + epsilon) and may not meet the seeds again -- they go into the
            // k-list themselves (libnabo prunes against a heap that HOLDS its k candidates); the passes then drop what they meet twice
            if (allv && inv1e != 1.0f) {
#pragma unroll
                for (int j = 0; j < KMAX; ++j) if (j < k) insert(sk[j], sp[j], true);
            }
        }
        if (seedb != ~0ull) {
            lev = level_for(seedb, 0);
            // seeds too wide for level 0 (the first iterations): the x-row through the query's own cell of level 0 first -- the k-th
            // of what it holds under the seed bound is usually much tighter, and the level is chosen again from that
            if (seed_pre && lev > 0) { widepre = true; lev = 0; }
        }
    }
    bool decided = !active;
    bool did_pre = false;
#ifdef ICPMI_NNK_DIAG
    int diag_pass = 0;
#endif
    NN_TICK(0);
    __syncthreads(); // ltab visible
    for (;;) {
        // ---- (1a)
        bool run = false, prescan = false;
        if (w0) {
            run = !decided && lev < nlev;
            const bool any = __ballot(run) != 0ull;
            if (lane == 0) { s_any = any ? 1u : 0u; s_total[0] = 0u; s_total[1] = 0u; }
            if (any) {
                const int lv = run ? lev : 0;
                unsigned long long b = kth();
                b = seedb < b ? seedb : b;
                prescan = run && !did_pre && (b == ~0ull || widepre);
                const unsigned long long acc = b < maxkey ? b : maxkey;
                float rub2 = INFINITY;
                if (!prescan && acc < 0x7f80000000000000ull) { // (a finite radius: the k-th held, or the matcher's maxDist)
                    const float slack = __uint_as_float(ltab[4 * lv + 1].y);
                    // (the k-th held shrinks by 1 / (1 + epsilon); the matcher's maxDist does not: libnabo tests `new_rd <= maxRadius2` as it is)
                    float rad = sqrt_up(__uint_as_float((unsigned)(acc >> 32)));
                    if (b != ~0ull) rad = fminf(rad, sqrt_up(__uint_as_float((unsigned)(b >> 32))) * inv1e);
                    const float rub = rad * 1.000001f + slack;
                    rub2 = rub * rub;
                }
                qrec[slot] = make_float4(p.x, p.y, p.z, __int_as_float(lv));
                qaux[slot] = make_float2(rub2, __int_as_float((run ? 1 : 0) | (prescan ? 2 : 0)));
                qbound[slot] = acc;
                qcnt[slot] = 0u;
            }
        }
        __syncthreads();
        if (!s_any) break;
        NN_TICK(1);
        // ---- (1b) all waves: lane = query, wave = row group (as nn1_wg_kernel)
        float mf, g_cell, g_slack;
        bool covers;
        {
            const float4 qr = qrec[slot];
            const float2 qa = qaux[slot];
            const int lv = __float_as_int(qr.w);
            const int fl = __float_as_int(qa.y);
            const bool qrun = (fl & 1) != 0, qpre = (fl & 2) != 0;
            const float rub2 = qa.x;
            GridParams g;
            const gunsigned* __restrict__ cs;
            {
                const uint4 a = ltab[4 * lv], b = ltab[4 * lv + 1], c2 = ltab[4 * lv + 2], d = ltab[4 * lv + 3];
                g.ox = __uint_as_float(a.x); g.oy = __uint_as_float(a.y); g.oz = __uint_as_float(a.z); g.cell = __uint_as_float(a.w);
                g.inv_cell = __uint_as_float(b.x); g.slack = __uint_as_float(b.y); g.nx = (int)b.z; g.ny = (int)b.w;
                g.nz = (int)c2.x; g.ncells = (int)c2.y;
                cs = reinterpret_cast<const gunsigned*>(((unsigned long long)d.y << 32) | d.x);
            }
            const float fx = (qr.x - g.ox) * g.inv_cell, fy = (qr.y - g.oy) * g.inv_cell, fz = (qr.z - g.oz) * g.inv_cell;
            const float flx = floorf(fx), fly = floorf(fy), flz = floorf(fz);
            const int cx = (int)fminf(fmaxf(flx, -1.0e6f), 1.0e6f);
            const int cy = (int)fminf(fmaxf(fly, -1.0e6f), 1.0e6f);
            const int cz = (int)fminf(fmaxf(flz, -1.0e6f), 1.0e6f);
            mf = fminf(fx - flx, 1.0f - (fx - flx));
            mf = fminf(mf, fminf(fy - fly, 1.0f - (fy - fly)));
            mf = fminf(mf, fminf(fz - flz, 1.0f - (fz - flz)));
            if (!(mf >= 0.f)) mf = 0.f;
            g_cell = g.cell; g_slack = g.slack;
            covers = cx - 1 <= 0 && cx + 1 >= g.nx - 1 && cy - 1 <= 0 && cy + 1 >= g.ny - 1 && cz - 1 <= 0 && cz + 1 >= g.nz - 1;
            unsigned rs[NR], rn[NR];
            {
                const float ylo = (fy - fly) * g.cell, yhi = (1.0f - (fy - fly)) * g.cell;
                const float zlo = (fz - flz) * g.cell, zhi = (1.0f - (fz - flz)) * g.cell;
                unsigned ia[NR], ib[NR];
#pragma unroll
                for (int sl = 0; sl < NR; ++sl) {
                    const int rr = wave + sl * NW;
                    const int dy = (rr % 3) - 1, dz = (rr / 3) - 1;
                    bool reach = qrun && rr < 9 && (!qpre || rr == 4);
                    const float ddy = dy == 0 ? 0.f : (dy < 0 ? ylo : yhi);
                    const float ddz = dz == 0 ? 0.f : (dz < 0 ? zlo : zhi);
                    const float rem2 = rub2 - (ddy * ddy + ddz * ddz);
                    reach = reach && rem2 >= 0.f;
                    int xa = cx - 1, xb = cx + 1;
                    if (rem2 != INFINITY) {
                        const float rem = sqrt_up(fmaxf(rem2, 0.f));
                        const int xl = (int)fmaxf(floorf((qr.x - rem - g.ox) * g.inv_cell), -1.0e6f);
                        const int xh = (int)fminf(floorf((qr.x + rem - g.ox) * g.inv_cell), 1.0e6f);
                        xa = xl > xa ? xl : xa;
                        xb = xh < xb ? xh : xb;
                    }
                    const int y = cy + dy, z = cz + dz;
                    xa = xa < 0 ? 0 : xa;
                    xb = xb > g.nx - 1 ? g.nx - 1 : xb;
                    reach = reach && y >= 0 && y < g.ny && z >= 0 && z < g.nz && xa <= xb;
                    const int rowbase = (z * g.ny + y) * g.nx;
                    ia[sl] = reach ? (unsigned)(rowbase + xa) : 0u;
                    ib[sl] = reach ? (unsigned)(rowbase + xb + 1) : 0u;
                }
#pragma unroll
                for (int sl = 0; sl < NR; ++sl) { rs[sl] = cs[ia[sl]]; rn[sl] = cs[ib[sl]]; }
#pragma unroll
                for (int sl = 0; sl < NR; ++sl) rn[sl] -= rs[sl];
            }
            int plb = PLB;
            unsigned base = 0;
            for (int round = 0; round < 2; ++round) {
                unsigned np = 0;
#pragma unroll
                for (int sl = 0; sl < NR; ++sl) np += (rn[sl] + ((1u << plb) - 1u)) >> plb;
                const unsigned incl = wave_incl_scan(np);
                const unsigned wtot = (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
                unsigned wb = 0;
                if (lane == 63 && wtot) wb = atomicAdd(&s_total[round], wtot);
                wb = (unsigned)__builtin_amdgcn_readlane((int)wb, 63);
                base = wb + incl - np;
                __syncthreads();
                const unsigned tot = s_total[round];
                if (tot <= (unsigned)CAP) break;
                int kk = 1;
                while ((tot >> kk) > (unsigned)(CAP - 9 * Q)) ++kk;
                plb += kk;
            }
#pragma unroll
            for (int sl = 0; sl < NR; ++sl) {
                unsigned s = rs[sl], c = rn[sl];
                while (c) {
                    const unsigned t = c < (1u << plb) ? c : (1u << plb);
                    pieces[base++] = make_uint2(s, (t << 8) | (unsigned)slot);
                    s += t; c -= t;
                }
            }
        }
        __syncthreads();
        const unsigned total = s_total[1] ? s_total[1] : s_total[0];
        NN_TICK(2);
#ifdef ICPMI_NN_TIMING
        tacc[6] += 1; tacc[7] += total;
#endif
        // ---- (2) all waves, lane per piece: survivors of the bound go to the query's list
        for (unsigned i0 = 0; i0 < total; i0 += (unsigned)NT) {
            const unsigned i = i0 + (unsigned)tid;
            const bool has = i < total;
            const uint2 e = pieces[has ? i : 0u];
            const unsigned qsl = e.y & 255u, cnt = has ? e.y >> 8 : 0u;
            const float4 qr = qrec[qsl];
            const unsigned long long bnd = qbound[qsl];
            const unsigned bh = (unsigned)(bnd >> 32), bl = (unsigned)bnd;
            const unsigned lvbits = __float_as_uint(qr.w) << 28;
            const uint4 lt2 = ltab[4 * __float_as_int(qr.w) + 2];
            const gfloat4* mp = reinterpret_cast<const gfloat4*>((((unsigned long long)lt2.w << 32) | lt2.z) + (unsigned long long)e.x * 16ull);
            for (unsigned c0 = 0; c0 < cnt; c0 += 8u) {
                vf4 q[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) q[u] = mp[c0 + u];
                unsigned d2b[8]; // (key = d2 bits << 32 | index bits: compared in halves, packed only when stored)
                unsigned okm = 0;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    d2b[u] = (unsigned)(pack_key(sqdist3(qr.x, qr.y, qr.z, q[u].x, q[u].y, q[u].z), 0u) >> 32);
                    const unsigned wb = __float_as_uint(q[u].w);
                    if (c0 + (unsigned)u < cnt && (d2b[u] < bh || (d2b[u] == bh && wb <= bl))) okm |= 1u << u;
                }
                if (okm) {
                    unsigned at = atomicAdd(&qcnt[qsl], (unsigned)__popc(okm));
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (okm & (1u << u)) {
                            if (at < (unsigned)CAPQ) {
                                qlk[at * Q + qsl] = ((unsigned long long)d2b[u] << 32) | __float_as_uint(q[u].w);
                                qlp[at * Q + qsl] = (e.x + c0 + (unsigned)u) | lvbits;
                            }
                            ++at;
                        }
                }
            }
        }
        __syncthreads();
        NN_TICK(3);
        // ---- (3) wave 0: list -> k-list, decide
        if (w0) {
            const unsigned offered = run ? qcnt[slot] : 0u;
#ifdef ICPMI_NN_TIMING
            t_off += offered; t_ovf += offered > (unsigned)CAPQ ? 1 : 0;
#endif
            const unsigned nq = offered < (unsigned)CAPQ ? offered : (unsigned)CAPQ;
            // (a pass over an empty k-list cannot meet a point twice: the usual case, one pass per launch)
            if (__ballot(mk[0] != ~0ull) == 0ull)
                for (unsigned e = 0; __ballot(e < nq) != 0ull; ++e) {
                    const unsigned long long key = qlk[e * Q + slot];
                    const unsigned pos = qlp[e * Q + slot];
                    if (e < nq) insert(key, (int)pos, false);
                }
            else
                for (unsigned e = 0; __ballot(e < nq) != 0ull; ++e) {
                    const unsigned long long key = qlk[e * Q + slot];
                    const unsigned pos = qlp[e * Q + slot];
                    if (e < nq) insert(key, (int)pos, true);
                }
#ifdef ICPMI_NNK_DIAG
            if (run && st->iter == ICPMI_NNK_DIAG) { // per pass of the launch of that iteration: queries, candidates offered, levels, overflows
                const int ps = diag_pass < 5 ? diag_pass : 5;
                atomicAdd(&st->dbg[ps], 1ull); atomicAdd(&st->dbg[6 + ps], (unsigned long long)offered);
                atomicAdd(&st->dbg[12 + ps], (unsigned long long)lev); atomicAdd(&st->dbg[18 + ps], offered > (unsigned)CAPQ ? 1ull : 0ull);
            }
            ++diag_pass;
#endif
            if (run) {
                if (prescan) { // (whether or not its list overflowed: the full pass follows, at the level its bound asks for)
                    did_pre = true; widepre = false;
                    unsigned long long b = kth();
                    b = seedb < b ? seedb : b;
                    if (b != ~0ull) lev = level_for(b, lev);
                } else if (offered > (unsigned)CAPQ) { } // tighter bound, same level again
                else {
                    const unsigned long long b = kth();
                    const float margin = fmaxf((1.0f + mf) * g_cell - g_slack, 0.f);
                    const float m2 = margin * margin;
                    const float kd2 = __uint_as_float((unsigned)(b >> 32));
                    decided = (b != ~0ull && kd2 <= m2 * err2) || m2 > maxr2 || covers;
                    if (!decided) { ++lev; did_pre = false; }
                }
            }
        }
        NN_TICK(4);
    }
    unsigned long long wcnt = 0ull;
    if (active) {
        int bs[KMAX];
        float bd[KMAX];
#pragma unroll
        for (int j = 0; j < KMAX; ++j) {
            float d2 = __uint_as_float((unsigned)(mk[j] >> 32));
            int b = -1;
            if (mk[j] != ~0ull && d2 <= maxr2) {
                const unsigned lv = (unsigned)ms[j] >> 28, pos = (unsigned)ms[j] & 0x0fffffffu;
                if (lv == 0) b = (int)pos;
                else {
                    const uint4 d = ltab[4 * lv + 3];
                    b = (int)reinterpret_cast<const unsigned*>(((unsigned long long)d.w << 32) | d.z)[pos];
                }
            } else d2 = INFINITY;
            bs[j] = b; bd[j] = d2;
        }
#pragma unroll
        for (int j = 0; j < KMAX; ++j)
            if (j < k) { out_sidx[(size_t)k * orig + j] = bs[j]; out_d2[(size_t)k * orig + j] = bd[j]; }
        if (!decided) {
            const unsigned hslot = atomicAdd(&st->hard_count, 1u);
            hard[hslot] = (unsigned)(qindex ? qindex[qi] : qi);
        }
        // the nine counts of the speculative window: field 0 = below, 1 .. BINS = the bins from win_lo on, BINS + 1 = above; seven bits per
        // field and lane (<= 16 distances).  Same predicate as sel2_hist0_kernel: finite and positive.
        if (use_win) {
#pragma unroll
            for (int j = 0; j < KMAX; ++j)
                if (j < k && bd[j] != INFINITY && bd[j] > 0.f) {
                    const unsigned bin = __float_as_uint(bd[j]) >> 16;
                    const unsigned f = bin < win_lo ? 0u : (bin - win_lo < (unsigned)ICPMI_WIN_BINS ? bin - win_lo + 1u : (unsigned)ICPMI_WIN_BINS + 1u);
                    wcnt += 1ull << (7u * f);
                }
        }
    }
    if (w0 && use_win) { // (the whole wave: lanes without a query carry zeros)
        // wave totals in registers: two 16-bit fields per word (<= 64 x 16 per field), DPP prefix sums, the totals read from lane 63
        static_assert(ICPMI_WIN_BINS + 2 == 9, "nine fields: five words of two, three packed atomics of three");
        unsigned tot[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const unsigned lo = (unsigned)(wcnt >> (14 * i)) & 127u, hi = i < 4 ? (unsigned)(wcnt >> (14 * i + 7)) & 127u : 0u;
            tot[i] = (unsigned)__builtin_amdgcn_readlane((int)wave_incl_scan(lo | (hi << 16)), 63);
        }
        if (lane < 3) {
            unsigned long long w = 0ull;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int f = 3 * lane + j;
                unsigned v = 0u;
#pragma unroll
                for (int i = 0; i < 5; ++i) if (i == (f >> 1)) v = (f & 1) ? tot[i] >> 16 : tot[i] & 0xffffu;
                w |= (unsigned long long)v << (21 * j);
            }
#ifndef ICPMI_WIN_DIAG_NOATOM
            if (w) atomicAdd(&win[(size_t)((blockIdx.x & (ICPMI_WIN_COPIES - 1)) * 3 + lane) * ICPMI_WIN_PAD], w);
#endif
        }
        if (blockIdx.x == 0 && lane == 0) win[ICPMI_WIN_HDR] = (unsigned long long)win_lo + 1ull;
    }
#ifdef ICPMI_NN_TIMING
    NN_TICK(5);
    if (w0 && st_iter > 1 && (blockIdx.x % 61) == 0) { atomicAdd(&st->dbg[18], t_ovf); atomicAdd(&st->dbg[19], t_off); }
    if (threadIdx.x == 0 && st_iter > 1 && (blockIdx.x % 7) == 0) {
        atomicMax(&st->dbg[20], (unsigned long long)(clock64() - t_c0));
        atomicMax(&st->dbg[21], (unsigned long long)tacc[7]);
    }
    if (threadIdx.x == 0 && (blockIdx.x % 61) == 0) {
        const int tb = st_iter > 1 ? 8 : 0;
        for (int i = 0; i < 6; ++i) atomicAdd(&st->dbg[tb + i], (unsigned long long)tacc[i]);
        atomicAdd(&st->dbg[tb + 6], (unsigned long long)tacc[6]);
        atomicAdd(&st->dbg[tb + 7], 1ull);
        atomicAdd(&st->dbg[16 + (st_iter > 1 ? 1 : 0)], (unsigned long long)tacc[7]);
    }
#endif
}

} // namespace

void nn_launch_hard_k1(icpmi_ctx* c, const float4* d_reading, const float* d_T, const LoopCfg& lc, int allow_self, int* d_sidx,
                       float* d_d2, IcpState* d_state)
{
    hipLaunchKernelGGL(nn1_hard_kernel, dim3(512), dim3(NN_BLOCK), 0, c->stream, d_reading, d_T, c->d_map_sorted, (int)c->m, lc.maxr2,
                       allow_self, d_sidx, d_d2, d_state, c->d_hard);
    hipLaunchKernelGGL(hard_reset_kernel, dim3(1), dim3(64), 0, c->stream, d_state);
}

icpmi_status nn_launch_k1(icpmi_ctx* c, const float4* d_reading, int64_t n, const float* d_T, const LoopCfg& lc, int allow_self,
                          int* d_sidx, float* d_d2, IcpState* d_state)
{
    c->nn_out_sorted = false;
    {
        // grid pyramid; sorted queries when the caller prepared them for exactly this cloud
        const bool sorted = c->batch_cur > 1 || (c->qsorted_n == n && c->qsorted_src == d_reading); // a batch is always tile-sorted
        const float4* q = sorted ? c->d_qsorted : d_reading;
        const int* qi = sorted ? c->d_qindex : nullptr;
        if (n == 0) return ICPMI_OK;
        // the brute pass may still overwrite d2 of queued queries: only build the histogram here when
        // every query is decided on the pyramid
        const GridParams& topg = c->levels.g[c->levels.nlev - 1];
        const bool needs_hard = !std::isfinite(lc.max_dist) || (topg.cell - topg.slack) <= lc.max_dist;
        // The NN kernel builds level 0 of the fused quantile selection itself (loop.hip): fine bins by
        // direct atomics, the hot coarse bins through LDS and privatised copies.  (r1 measurement: a
        // single 2048-bin level-0 histogram flushed by every workgroup cost +20 us in same-address
        // atomics; the two-tier layout removes that.)  ICPMI_NN_FUSE_HIST0=0 falls back to the
        // stand-alone builder kernel.
        const int unseeded_lev = 0;
        // r5: a chain that may need the brute pass (unbounded maxDist -- PM::ICPSequence::setDefault() --, or a maxDist beyond the top level's
        // block) keeps the fast path: the queue holds query slots and the brute pass writes the same state (point, histogram) for what it decides.
        const bool hard_sorted = needs_hard && allow_self && sorted && c->batch_cur <= 1;
        unsigned* h0 = (needs_hard && !hard_sorted) ? nullptr : c->nn_hist0;
        c->nn_builds_hist0 = h0 != nullptr;
        // loop mode keeps the per-query state in query order
        float4* mp = ((needs_hard && !hard_sorted) || !sorted) ? nullptr : c->nn_match_pt;
        c->nn_out_sorted = mp != nullptr;
        // a batch (c->batch_cur > 1, set by loop_run_batch): grid.y = readings, grid.x sized for the largest one
        const BatchArgs ba = c->batch_cur > 1 ? c->batch_args : batch_of_one(n);
        // the first solve moves the reading by the whole initial misalignment, so the seeds of iteration 1 bound the search no better than a
        // fresh own-row scan: that launch starts with the own-row pass at level 0 (seed_pre), the later ones go straight to their seed
        const bool seeded = c->nn_iter_hint > 0 && allow_self;
        const int seed_pre = (seeded && c->nn_iter_hint > 1) ? 0 : 1;
#define LAUNCH_WG(S_)                                                                                                           \
    hipLaunchKernelGGL((nn1_wg_kernel<4, S_>), dim3((int)(((n + 63) / 64 + 7) / 8 * 8), ba.nscan), dim3(256), 0, c->stream,    \
                       q, qi, ba, d_T, c->levels, lc.maxr2, d_sidx, d_d2, d_state, c->d_hard, h0, mp, c->d_lvl_tab, unseeded_lev, seed_pre, lc.inv1e, lc.err2)
        if (allow_self) LAUNCH_WG(true); else LAUNCH_WG(false);
#undef LAUNCH_WG
        const GridParams& top = c->levels.g[c->levels.nlev - 1];
        if (!std::isfinite(lc.max_dist) || (top.cell - top.slack) <= lc.max_dist) {
            if (hard_sorted && mp)
                hipLaunchKernelGGL(nn1_hard_kernel, dim3(512), dim3(NN_BLOCK), 0, c->stream, q, d_T, c->d_map_sorted, (int)c->m,
                                   lc.maxr2, allow_self, d_sidx, d_d2, d_state, c->d_hard, mp, c->nn_builds_hist0 ? c->nn_hist0 : (unsigned*)nullptr);
            else
            hipLaunchKernelGGL(nn1_hard_kernel, dim3(512), dim3(NN_BLOCK), 0, c->stream, d_reading, d_T, c->d_map_sorted, (int)c->m,
                               lc.maxr2, allow_self, d_sidx, d_d2, d_state, c->d_hard);
            // (the loop's solve_kernel empties the queue; a stage call -- icpmi_knn -- has no solve behind it)
            if (!(hard_sorted && mp)) hipLaunchKernelGGL(hard_reset_kernel, dim3(1), dim3(64), 0, c->stream, d_state);
        }
        HIP_TRY(c, hipGetLastError());
        return ICPMI_OK;
    }
}

template <int KMAX>
static icpmi_status nnk_launch_t(icpmi_ctx* c, const float4* d_reading, int64_t n, const float* d_T, const LoopCfg& lc,
                                 int allow_self, int* d_sidx, float* d_d2, IcpState* d_state)
{
    if constexpr (KMAX <= 16) if (n > 0) {
        constexpr int KM = KMAX; // (k <= 16: the cooperative kernels; r2 stopped at 8 and left knn 10 on the one-lane kernel, 5 x slower)
        {
            constexpr int G = 8;
            const bool sorted = c->qsorted_n == n && c->qsorted_src == d_reading;
            const float4* q = sorted ? c->d_qsorted : d_reading;
            const int* qi = sorted ? c->d_qindex : nullptr;
            const int seeded = (c->nn_iter_hint > 0 && allow_self) ? 3 : 0; // bit 1: wide seeds start with the own-row pass at level 0 (nnk_ml_kernel)
            const int grid = (int)(((n * G + NN_BLOCK - 1) / NN_BLOCK + 7) / 8 * 8);
            const GridParams& top = c->levels.g[c->levels.nlev - 1];
            const bool needs_hard = !std::isfinite(lc.max_dist) || (top.cell - top.slack) <= lc.max_dist;
            // loop mode: results in query order (the brute-force pass works in the caller's order: chains that may need it stay there)
            const int out_sorted = (c->nn_sorted_k && sorted && !needs_hard) ? 1 : 0;
            c->nn_out_sorted = out_sorted != 0;
            // the loop's seeded launches from iteration 2 on: nnk_wg_kernel; iterations 0 / 1 (no seed / seeds the first solve moved far), batches and
            // stage calls: nnk_ml_kernel.  (icpmi_config::knn_wg_from, a test seam: -1 = nnk_ml_kernel everywhere, 0 / 1 = nnk_wg_kernel earlier.)
            const int wg_from = c->cfg.knn_wg_from == 0 ? 2 : (c->cfg.knn_wg_from < 0 ? -1 : c->cfg.knn_wg_from - 1);
            const int wg_pre = 1;
            const bool use_wg = wg_from >= 0 && allow_self && c->nn_iter_hint >= wg_from && c->batch_cur <= 1;
            // (r5) the speculative window of the fused selection (common.h: ICPMI_S2_WIN): loops with one quantile filter whose every query is
            // decided on the pyramid (the brute pass rewrites d2 afterwards), below 2^21 matches (the packed counts cannot carry); icpmi_config::sel_window_off: off
            const int sel_win = c->cfg.sel_window_off ? 0 : 1;
            c->nn_builds_win = use_wg && sel_win && c->nn_hist0 != nullptr && d_d2 == c->d_d2 && !needs_hard && n * (int64_t)lc.k < ICPMI_WIN_MAX_COUNT;
            if (use_wg)
                hipLaunchKernelGGL((nnk_wg_kernel<KM>), dim3((int)(((n + 63) / 64 + 7) / 8 * 8)), dim3(256), 0, c->stream, q, qi,
                                   (int)n, d_T, c->d_lvl_tab, c->levels.nlev, lc.k, lc.maxr2, d_sidx, d_d2, d_state, c->d_hard, out_sorted, wg_pre,
                                   c->nn_builds_win ? reinterpret_cast<unsigned long long*>(c->nn_hist0 + ICPMI_S2_WIN) : (unsigned long long*)nullptr,
                                   lc.inv1e, lc.err2);
            else
            hipLaunchKernelGGL((nnk_ml_kernel<G, KM>), dim3(grid), dim3(NN_BLOCK), 0, c->stream, q, qi, (int)n, d_T, c->d_lvl_tab,
                               c->levels.nlev, lc.k, lc.maxr2, allow_self, seeded, d_sidx, d_d2, d_state, c->d_hard, out_sorted, lc.inv1e, lc.err2);
            if (!std::isfinite(lc.max_dist) || (top.cell - top.slack) <= lc.max_dist) {
                hipLaunchKernelGGL(nnk_hard_kernel<KMAX>, dim3(512), dim3(NN_BLOCK), 0, c->stream, d_reading, d_T, c->d_map_sorted,
                                   (int)c->m, lc.k, lc.maxr2, allow_self, d_sidx, d_d2, d_state, c->d_hard);
                hipLaunchKernelGGL(hard_reset_kernel, dim3(1), dim3(64), 0, c->stream, d_state);
            }
            HIP_TRY(c, hipGetLastError());
            return ICPMI_OK;
        }
    }
    // k = 17 .. 32: one lane per query, ring search (the cooperative kernels keep their lists in registers up to k = 16)
    if constexpr (KMAX > 16) {
        const int blocks = (int)((n + NN_BLOCK - 1) / NN_BLOCK);
        if (blocks == 0) return ICPMI_OK;
        hipLaunchKernelGGL(nnk_kernel<KMAX>, dim3(blocks), dim3(NN_BLOCK), 0, c->stream, d_reading, (int)n, d_T, c->grid,
                           c->d_map_sorted, c->d_cell_start, lc.k, lc.maxr2, lc.ring_max, allow_self, d_sidx, d_d2, d_state, c->d_hard,
                           (const unsigned*)nullptr, (const unsigned*)nullptr);
        if (!std::isfinite(lc.max_dist) || lc.ring_max < (int)ceilf(lc.max_dist / c->grid.cell) + 1) {
            hipLaunchKernelGGL(nnk_hard_kernel<KMAX>, dim3(512), dim3(NN_BLOCK), 0, c->stream, d_reading, d_T, c->d_map_sorted,
                               (int)c->m, lc.k, lc.maxr2, allow_self, d_sidx, d_d2, d_state, c->d_hard);
            hipLaunchKernelGGL(hard_reset_kernel, dim3(1), dim3(64), 0, c->stream, d_state);
        }
        HIP_TRY(c, hipGetLastError());
    }
    return ICPMI_OK;
}

icpmi_status nn_launch_k(icpmi_ctx* c, const float4* d_reading, int64_t n, const float* d_T, const LoopCfg& lc, int allow_self,
                         int* d_sidx, float* d_d2, IcpState* d_state)
{
    if (lc.k == 1) return nn_launch_k1(c, d_reading, n, d_T, lc, allow_self, d_sidx, d_d2, d_state);
    if (lc.k <= 4) return nnk_launch_t<4>(c, d_reading, n, d_T, lc, allow_self, d_sidx, d_d2, d_state);
    if (lc.k <= 6) return nnk_launch_t<6>(c, d_reading, n, d_T, lc, allow_self, d_sidx, d_d2, d_state); // knn 6: the documented chain
    if (lc.k <= 8) return nnk_launch_t<8>(c, d_reading, n, d_T, lc, allow_self, d_sidx, d_d2, d_state);
    if (lc.k <= 16) return nnk_launch_t<16>(c, d_reading, n, d_T, lc, allow_self, d_sidx, d_d2, d_state);
    if (lc.k <= 32) return nnk_launch_t<32>(c, d_reading, n, d_T, lc, allow_self, d_sidx, d_d2, d_state);
    c->last_error = "knn > 32 is not supported";
    return ICPMI_ERR_UNSUPPORTED;
}

icpmi_status nn_ids_to_original(icpmi_ctx* c, const int* d_sidx, int64_t count, int* d_ids)
{
    const int blocks = (int)((count + 255) / 256);
    if (blocks == 0) return ICPMI_OK;
    hipLaunchKernelGGL(ids_kernel, dim3(blocks), dim3(256), 0, c->stream, c->d_map_sorted, d_sidx, count, d_ids);
    HIP_TRY(c, hipGetLastError());
    return ICPMI_OK;
}
