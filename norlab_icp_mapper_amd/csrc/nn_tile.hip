// nn_tile.hip -- LDS-staged exact nearest neighbour (k = 1) for super-tile-sorted queries.
//
// Same contract and the same answers as nn1_kernel in nn.hip (KDTreeMatcher::findClosests ->
// Nabo::NNS::knn, reference call site norlab_icp_mapper/Mapper.cpp:213), different schedule:
//
//   * the reading is sorted once per registration by super-tile (map_build.hip:sort_queries) and cut
//     into work items of <= ICPMI_TQ queries that share a super-tile: one workgroup per item;
//   * per ring r the workgroup takes the bounding box of its still-undecided queries' CURRENT cells,
//     grows it by r cells, and stages that region of the cell-sorted map through LDS.  The map is
//     x-fastest, so the region is (height x depth) runs, each ONE contiguous, coalesced float4 stream
//     from HBM/L2; the per-cell boundaries of the runs are staged next to the points.  Regions larger
//     than the LDS budget are streamed in batches of whole runs;
//   * the G lanes of a query stride together over the candidates of each shell row held by the
//     current batch (16-byte ds reads, four in flight per lane) and fold (d^2, index) keys with wave
//     shuffles;
//   * the exactness rule is unchanged: a query is decided when its best distance is within its margin
//     to the searched block, when the margin exceeds maxDist, or when the block covers the grid.
//     Regions beyond the run / cell-table budget, and rings beyond ring_max, fall back to the
//     global-memory ring search / brute pass of nn.hip.
//
// HBM traffic: every map point of the region of interest is read about once per ring-1 pass (items
// overlap only by their halo), which is what the algorithmic byte count of DESIGN.md assumes.
#include "common.h"

namespace {

constexpr int TQ = ICPMI_TQ;  // queries per workgroup
constexpr int MAXROWS = 256;  // runs per staged region
constexpr int CMAX = 6144;    // staged cell-boundary entries
constexpr int GRAN = 32;      // LDS allocation granule (points) per run
constexpr int MAXBATCH = 16;  // streamed batches per region
constexpr int MAXGRAN = 512;  // granules per region (all batches)

struct Cand {
    unsigned long long key;
    int sidx;
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

__device__ __forceinline__ void scan_global(const float4* __restrict__ map, unsigned s, unsigned e, float px, float py, float pz,
                                            bool allow_self, Cand& best)
{
    for (unsigned i = s; i < e; ++i) {
        const float4 q = map[i];
        const float d2 = sqdist3(px, py, pz, q.x, q.y, q.z);
        if (allow_self || d2 > 1.1920929e-07f) cand_min(best, pack_key(d2, __float_as_uint(q.w)), (int)i);
    }
}

// candidates [a, e) of the staged array, lanes of the group take a + sub, a + sub + G, ...; four
// independent 16-byte LDS reads are issued before the first use; rejected / out-of-range slots carry
// the key ~0 so that the fold is branch free
template <int G>
__device__ __forceinline__ void scan_lds(const float4* __restrict__ P, unsigned a, unsigned e, int sub, float px, float py, float pz,
                                         bool allow_self, Cand& best)
{
    for (unsigned i = a + (unsigned)sub; i < e; i += 4u * G) {
        const unsigned i1 = i + G, i2 = i + 2 * G, i3 = i + 3 * G;
        const unsigned j1 = i1 < e ? i1 : i, j2 = i2 < e ? i2 : i, j3 = i3 < e ? i3 : i;
        const float4 q0 = P[i];
        const float4 q1 = P[j1];
        const float4 q2 = P[j2];
        const float4 q3 = P[j3];
        const float d0 = sqdist3(px, py, pz, q0.x, q0.y, q0.z);
        const float d1 = sqdist3(px, py, pz, q1.x, q1.y, q1.z);
        const float d2 = sqdist3(px, py, pz, q2.x, q2.y, q2.z);
        const float d3 = sqdist3(px, py, pz, q3.x, q3.y, q3.z);
        unsigned long long k0 = pack_key(d0, __float_as_uint(q0.w));
        unsigned long long k1 = pack_key(d1, __float_as_uint(q1.w));
        unsigned long long k2 = pack_key(d2, __float_as_uint(q2.w));
        unsigned long long k3 = pack_key(d3, __float_as_uint(q3.w));
        if (!allow_self) {
            k0 = d0 > 1.1920929e-07f ? k0 : ~0ull;
            k1 = d1 > 1.1920929e-07f ? k1 : ~0ull;
            k2 = d2 > 1.1920929e-07f ? k2 : ~0ull;
            k3 = d3 > 1.1920929e-07f ? k3 : ~0ull;
        }
        // clamped duplicates of slot i carry the same key as k0: harmless
        cand_min(best, k0, (int)i);
        cand_min(best, k1, (int)j1);
        cand_min(best, k2, (int)j2);
        cand_min(best, k3, (int)j3);
    }
}

__device__ __forceinline__ bool ring_decided(const GridParams& g, int ring, float mf, int cx, int cy, int cz, const Cand& best,
                                             float maxr2)
{
    const float margin = fmaxf(((float)ring + mf) * g.cell - g.slack, 0.f);
    const float m2 = margin * margin;
    const float bd2 = __uint_as_float((unsigned)(best.key >> 32));
    const bool covers = cx - ring <= 0 && cx + ring >= g.nx - 1 && cy - ring <= 0 && cy + ring >= g.ny - 1 && cz - ring <= 0 &&
                        cz + ring >= g.nz - 1;
    return (best.sidx >= 0 && bd2 <= m2) || m2 > maxr2 || covers;
}

// LDS layout (dynamic, every carve a multiple of 16 bytes)
template <int CAP>
struct Lds {
    static constexpr size_t off_P = 0;
    static constexpr size_t off_C = off_P + (size_t)CAP * 16;         // ushort[CMAX]
    static constexpr size_t off_GI = off_C + (size_t)CMAX * 2;        // uint2[MAXGRAN]
    static constexpr size_t off_RS = off_GI + (size_t)MAXGRAN * 8;    // uint[MAXROWS]
    static constexpr size_t off_RN = off_RS + MAXROWS * 4;            // uint[MAXROWS]
    static constexpr size_t off_RL = off_RN + MAXROWS * 4;            // uint[MAXROWS + 4]
    static constexpr size_t off_RC = off_RL + (MAXROWS + 4) * 4;      // int[MAXROWS]   first cell of the run
    static constexpr size_t off_BASE = off_RC + MAXROWS * 4;          // uint[MAXBATCH + 4]
    static constexpr size_t off_RB = off_BASE + (MAXBATCH + 4) * 4;   // uchar[MAXROWS]
    static constexpr size_t off_bb = off_RB + MAXROWS;                // int[16]
    static constexpr size_t bytes = off_bb + 64;
};

template <int G, int CAP>
__global__ __launch_bounds__(TQ * G) void nn1_tile_kernel(const float4* __restrict__ qsorted, const int* __restrict__ qindex,
                                                          const uint2* __restrict__ items, const unsigned* __restrict__ n_items,
                                                          const float* __restrict__ Tptr, GridParams g,
                                                          const float4* __restrict__ map, const unsigned* __restrict__ cs,
                                                          float maxr2, int ring_max, int allow_self_i,
                                                          int* __restrict__ out_sidx, float* __restrict__ out_d2,
                                                          IcpState* __restrict__ st, unsigned* __restrict__ hard, int timing)
{
    // one round trip: the work item (count 0 beyond the last item), the loop flag
    const uint2 item = items[blockIdx.x];
    if (item.y == 0 || st->done) return;
    (void)n_items;
    constexpr int NT = TQ * G;
    constexpr int NSTAGE = CAP / NT;
    constexpr unsigned HALF = CAP / 2;
    static_assert(CMAX * 4 <= CAP * 16, "raw cell boundaries are parked in the point area");
    using L = Lds<CAP>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4* P = reinterpret_cast<float4*>(smem + L::off_P);
    unsigned short* C = reinterpret_cast<unsigned short*>(smem + L::off_C);
    uint2* GI = reinterpret_cast<uint2*>(smem + L::off_GI);           // granule -> (global index of its first point, valid | batch << 8)
    unsigned* RS = reinterpret_cast<unsigned*>(smem + L::off_RS);     // global start of run j
    unsigned* RN = reinterpret_cast<unsigned*>(smem + L::off_RN);     // length of run j
    unsigned* RL = reinterpret_cast<unsigned*>(smem + L::off_RL);     // padded prefix of run j over the whole region
    int* RC = reinterpret_cast<int*>(smem + L::off_RC);               // linear cell index of (x0, y, z) of run j
    unsigned* BASE = reinterpret_cast<unsigned*>(smem + L::off_BASE); // padded prefix where batch b starts
    unsigned char* RB = smem + L::off_RB;                             // run -> batch
    int* bb = reinterpret_cast<int*>(smem + L::off_bb);

    const bool allow_self = allow_self_i != 0;
    const int t = threadIdx.x;
    long long tprev = timing ? clock64() : 0;
#define TS(k) do { if (timing && t == 0) { const long long now_ = clock64(); atomicAdd(&st->dbg[8 + (k)], (unsigned long long)(now_ - tprev)); tprev = now_; } } while (0)
    const int ql = t / G, sub = t % G;
    const bool active = ql < (int)item.y;
    const int qpos = (int)item.x + (active ? ql : 0);
    const float4 r = qsorted[qpos];
    const int orig = qindex[qpos];
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

    Cand best; best.key = ~0ull; best.sidx = -1; // sidx: global sorted-map position | 0x40000000, or an LDS position
    bool decided = !active;
    int ring = 0;        // rings 1..ring have been searched
    bool to_global = false;

    for (;;) {
        // ---- bounding box of the still-undecided queries ----
        if (t < 8) bb[t] = (t == 0 || t == 2 || t == 4) ? 0x7fffffff : ((t == 1 || t == 3 || t == 5) ? (int)0x80000000 : 0);
        __syncthreads();
        if (!decided && sub == 0) {
            atomicMin(&bb[0], cx); atomicMax(&bb[1], cx);
            atomicMin(&bb[2], cy); atomicMax(&bb[3], cy);
            atomicMin(&bb[4], cz); atomicMax(&bb[5], cz);
            atomicAdd(&bb[6], 1);
        }
        __syncthreads();
        TS(0);
        if (bb[6] == 0) break;
        if (ring >= ring_max) break; // leftovers go to the brute pass
        const int nr = ring + 1;
        int x0 = bb[0] - nr, x1 = bb[1] + nr, y0 = bb[2] - nr, y1 = bb[3] + nr, z0 = bb[4] - nr, z1 = bb[5] + nr;
        x0 = x0 < 0 ? 0 : x0; y0 = y0 < 0 ? 0 : y0; z0 = z0 < 0 ? 0 : z0;
        x1 = x1 > g.nx - 1 ? g.nx - 1 : x1; y1 = y1 > g.ny - 1 ? g.ny - 1 : y1; z1 = z1 > g.nz - 1 ? g.nz - 1 : z1;
        const bool empty = x0 > x1 || y0 > y1 || z0 > z1;
        const int w = x1 - x0 + 1, h = y1 - y0 + 1, d = z1 - z0 + 1;
        const long long rows_ll = empty ? 0 : (long long)h * d;
        if (!empty && (rows_ll > MAXROWS || rows_ll * (w + 1) > CMAX)) {
            to_global = true;
            if (t == 0) { atomicAdd(&st->dbg[2], 1ull); atomicAdd(&st->dbg[3], (unsigned long long)bb[6]); }
            break;
        }
        const int rows = (int)rows_ll;
        int nbatch = 0;
        if (!empty) {
            // ---- one round trip for every cell boundary of the region: entry e = (run j, cell xi),
            //      xi = 0..w, parked raw in the (still unused) point area; run bounds = first / last ----
            const int ce = rows * (w + 1);
            const float rcp = 1.0f / (float)(w + 1);
            unsigned* RAW = reinterpret_cast<unsigned*>(P);
#pragma unroll 4
            for (int e = t; e < ce; e += NT) {
                const int j = (int)(((float)e + 0.5f) * rcp); // exact: rows * (w + 1) <= CMAX << 2^22
                const int xi = e - j * (w + 1);
                const int y = y0 + j % h, z = z0 + j / h;
                RAW[e] = cs[(z * g.ny + y) * g.nx + x0 + xi];
            }
            __syncthreads();
            for (int j = t; j < MAXROWS; j += NT) {
                unsigned s = 0, e = 0;
                if (j < rows) { s = RAW[j * (w + 1)]; e = RAW[j * (w + 1) + w]; }
                RS[j] = s; RN[j] = e; // run end for now, turned into a length below
            }
            __syncthreads();
            // ---- exclusive scan of the granule-padded run lengths (first wave, 4 rows per lane) ----
            if (t < 64) {
                unsigned a[4], s = 0, mx = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned len = RN[4 * t + e] - RS[4 * t + e]; // end - start
                    RN[4 * t + e] = len;
                    a[e] = (len + (GRAN - 1)) & ~(unsigned)(GRAN - 1);
                    s += a[e];
                    mx = a[e] > mx ? a[e] : mx;
                }
                unsigned incl = s;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const unsigned o = __shfl_up(incl, off, 64);
                    if (t >= off) incl += o;
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) { const unsigned o = __shfl_xor(mx, off, 64); mx = o > mx ? o : mx; }
                unsigned ex = incl - s;
#pragma unroll
                for (int e = 0; e < 4; ++e) { RL[4 * t + e] = ex; ex += a[e]; }
                if (t == 63) { RL[MAXROWS] = incl; RL[MAXROWS + 1] = mx; }
                if (t < MAXBATCH) BASE[t] = 0xffffffffu; // a batch without a starting run stages nothing
            }
            __syncthreads();
            TS(1);
            const unsigned total = RL[MAXROWS], maxrow = RL[MAXROWS + 1];
            // batch of a run = its padded start / HALF: a batch holds < HALF + maxrow <= CAP points
            nbatch = (int)((total + HALF - 1) / HALF);
            if (maxrow > HALF || nbatch > MAXBATCH || total > (unsigned)MAXGRAN * GRAN) {
                to_global = true;
                if (t == 0) { atomicAdd(&st->dbg[2], 1ull); atomicAdd(&st->dbg[3], (unsigned long long)bb[6]); }
                break;
            }
            if (t == 0) { atomicAdd(&st->dbg[0], 1ull); atomicAdd(&st->dbg[1], (unsigned long long)total); }
            // ---- run -> batch, batch bases, granule table ----
            for (int j = t; j < rows; j += NT) {
                const unsigned rl = RL[j], rn = RN[j];
                const unsigned bj = rl / HALF;
                RB[j] = (unsigned char)bj;
                if (rn != 0) {
                    // first non-empty run of its batch publishes the batch base
                    bool first = true;
                    for (int jj = j - 1; jj >= 0; --jj) {
                        if (RN[jj] != 0) { first = (RL[jj] / HALF) != bj; break; }
                    }
                    if (first) BASE[bj] = rl;
                    const unsigned g0 = rl / GRAN, rs = RS[j];
                    const unsigned ng = (rn + GRAN - 1) / GRAN;
                    for (unsigned gi = 0; gi < ng; ++gi) {
                        const unsigned left = rn - gi * GRAN;
                        GI[g0 + gi] = make_uint2(rs + gi * GRAN, (left < GRAN ? left : (unsigned)GRAN) | (bj << 8));
                    }
                }
            }
            __syncthreads();
            // ---- cell boundaries as positions inside the batch's staged array ----
            for (int e = t; e < ce; e += NT) {
                const int j = (int)(((float)e + 0.5f) * rcp);
                unsigned v = 0;
                if (RN[j] != 0) v = (RL[j] - BASE[RB[j]]) + (RAW[e] - RS[j]);
                C[e] = (unsigned short)v;
            }
            __syncthreads();
            TS(2);
        }
        const int side = 2 * nr + 1;
        for (int b = 0; b < nbatch; ++b) {
            const unsigned bbase = BASE[b];
            if (bbase == 0xffffffffu) continue; // uniform: no run starts in this batch
            // ---- stage batch b: all loads of a lane in flight before the first LDS store ----
            {
                float4 v[NSTAGE];
                const unsigned gbase = bbase / GRAN;
#pragma unroll
                for (int u = 0; u < NSTAGE; ++u) {
                    const unsigned li = (unsigned)t + (unsigned)NT * u; // position in P
                    const unsigned gidx = gbase + li / GRAN;
                    v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (gidx * GRAN < RL[MAXROWS]) {
                        const uint2 gi = GI[gidx];
                        const unsigned o = li % GRAN;
                        if ((gi.y >> 8) == (unsigned)b && o < (gi.y & 0xffu)) v[u] = map[gi.x + o];
                    }
                }
#pragma unroll
                for (int u = 0; u < NSTAGE; ++u) P[t + NT * u] = v[u];
            }
            __syncthreads();
            TS(3);
            // ---- every undecided query searches the rows of its ring-nr shell held by this batch ----
            if (!decided) {
                if (nr == 1) {
#pragma unroll 1
                    for (int rr = 0; rr < 9; ++rr) {
                        const int y = cy + (rr % 3) - 1, z = cz + (rr / 3) - 1;
                        if (y < y0 || y > y1 || z < z0 || z > z1) continue;
                        const int j = (y - y0) + (z - z0) * h;
                        if (RB[j] != (unsigned char)b || RN[j] == 0) continue;
                        const int cb = j * (w + 1) - x0;
                        const int xa = cx - 1 < x0 ? x0 : cx - 1, xb = cx + 1 > x1 ? x1 : cx + 1;
                        if (xa <= xb) scan_lds<G>(P, C[cb + xa], C[cb + xb + 1], sub, p.x, p.y, p.z, allow_self, best);
                    }
                } else {
                    for (int rr = 0; rr < side * side; ++rr) {
                        const int dy = rr % side - nr, dz = rr / side - nr;
                        const int y = cy + dy, z = cz + dz;
                        if (y < y0 || y > y1 || z < z0 || z > z1) continue;
                        const int j = (y - y0) + (z - z0) * h;
                        if (RB[j] != (unsigned char)b || RN[j] == 0) continue;
                        const int cb = j * (w + 1) - x0;
                        const bool full_row = dy == -nr || dy == nr || dz == -nr || dz == nr;
                        int xa = cx - nr, xb = cx + nr;
                        if (full_row) {
                            xa = xa < x0 ? x0 : xa; xb = xb > x1 ? x1 : xb;
                            if (xa <= xb) scan_lds<G>(P, C[cb + xa], C[cb + xb + 1], sub, p.x, p.y, p.z, allow_self, best);
                        } else {
                            if (xa >= x0 && xa <= x1) scan_lds<G>(P, C[cb + xa], C[cb + xa + 1], sub, p.x, p.y, p.z, allow_self, best);
                            if (xb >= x0 && xb <= x1) scan_lds<G>(P, C[cb + xb], C[cb + xb + 1], sub, p.x, p.y, p.z, allow_self, best);
                        }
                    }
                }
                // a winner found in this batch carries an LDS position: make it a global position
                if (best.sidx >= 0 && (best.sidx & 0x40000000) == 0) {
                    const unsigned li = (unsigned)best.sidx;
                    const uint2 gi = GI[bbase / GRAN + li / GRAN];
                    best.sidx = (int)(gi.x + li % GRAN) | 0x40000000;
                }
            }
            __syncthreads(); // P is overwritten by the next batch / LDS tables by the next ring
            TS(4);
        }
        if (!decided) {
            group_reduce<G>(best);
            decided = ring_decided(g, nr, mf, cx, cy, cz, best, maxr2);
        }
        ring = nr;
    }

    // flag bit off: best.sidx is now a plain global sorted-map position (or -1)
    if (best.sidx >= 0) best.sidx &= 0x3fffffff;

    // ---- leftovers: global-memory ring search (region beyond the LDS table budget) ----
    if (to_global && !decided) {
        while (!decided && ring < ring_max) {
            ++ring;
            const int side = 2 * ring + 1;
            for (int rr = sub; rr < side * side; rr += G) {
                const int dy = rr % side - ring, dz = rr / side - ring;
                const bool full_row = ring == 1 || dy == -ring || dy == ring || dz == -ring || dz == ring;
                unsigned s, e;
                if (full_row) {
                    row_run(g, cs, cx - ring, cx + ring, cy + dy, cz + dz, s, e);
                    scan_global(map, s, e, p.x, p.y, p.z, allow_self, best);
                } else {
                    if (cx - ring >= 0) {
                        row_run(g, cs, cx - ring, cx - ring, cy + dy, cz + dz, s, e);
                        scan_global(map, s, e, p.x, p.y, p.z, allow_self, best);
                    }
                    if (cx + ring <= g.nx - 1) {
                        row_run(g, cs, cx + ring, cx + ring, cy + dy, cz + dz, s, e);
                        scan_global(map, s, e, p.x, p.y, p.z, allow_self, best);
                    }
                }
            }
            group_reduce<G>(best);
            decided = ring_decided(g, ring, mf, cx, cy, cz, best, maxr2);
        }
    }
    TS(5);
    if (active && sub == 0) {
        float bd2 = __uint_as_float((unsigned)(best.key >> 32));
        int bs = best.sidx;
        if (bs < 0 || !(bd2 <= maxr2)) { bs = -1; bd2 = INFINITY; }
        out_sidx[orig] = bs;
        out_d2[orig] = bd2;
        if (!decided) {
            const unsigned slot = atomicAdd(&st->hard_count, 1u);
            hard[slot] = (unsigned)orig;
        }
    }
    TS(6);
#undef TS
}

} // namespace

// defined in nn.hip
void nn_launch_hard_k1(icpmi_ctx* c, const float4* d_reading, const float* d_T, const LoopCfg& lc, int allow_self, int* d_sidx,
                       float* d_d2, IcpState* d_state);

icpmi_status nn_tile_launch_k1(icpmi_ctx* c, int64_t n, const float* d_T, const LoopCfg& lc, int allow_self, int* d_sidx, float* d_d2,
                               IcpState* d_state)
{
    const int blocks = c->q_max_items;
    if (blocks == 0 || n == 0) return ICPMI_OK;
    static int variant = -1, timing = 0;
    if (variant < 0) {
        const char* e = getenv("ICPMI_NN_VARIANT");
        variant = e ? atoi(e) : 0;
        const char* a = getenv("ICPMI_NN_TIMING");
        timing = a ? atoi(a) : 0;
    }
#define LAUNCH(G, CAP)                                                                                                            \
    do {                                                                                                                          \
        const size_t lds = Lds<CAP>::bytes;                                                                                       \
        static bool attr_set = false;                                                                                             \
        if (!attr_set) {                                                                                                          \
            HIP_TRY(c, hipFuncSetAttribute((const void*)nn1_tile_kernel<G, CAP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            attr_set = true;                                                                                                      \
        }                                                                                                                         \
        hipLaunchKernelGGL((nn1_tile_kernel<G, CAP>), dim3(blocks), dim3(TQ * G), lds, c->stream, c->d_qsorted, c->d_qindex,         \
                           c->d_qitems, c->d_q_n_items, d_T, c->grid, c->d_map_sorted, c->d_cell_start, lc.maxr2, lc.ring_max,       \
                           allow_self, d_sidx, d_d2, d_state, c->d_hard, timing);                                                 \
    } while (0)
    switch (variant) {
        case 1: LAUNCH(1, 2048); break;
        case 2: LAUNCH(2, 2048); break;
        case 3: LAUNCH(4, 2048); break;
        case 4: LAUNCH(2, 2048); break;
        case 5: LAUNCH(4, 4096); break;
        case 6: LAUNCH(8, 4096); break;
        case 7: LAUNCH(4, 2048); break;
        case 8: LAUNCH(8, 2048); break;
        default: LAUNCH(4, 2048); break;
    }
#undef LAUNCH
    if (!std::isfinite(lc.max_dist) || lc.ring_max < (int)ceilf(lc.max_dist / c->grid.cell) + 1)
        nn_launch_hard_k1(c, c->d_reading, d_T, lc, allow_self, d_sidx, d_d2, d_state);
    HIP_TRY(c, hipGetLastError());
    return ICPMI_OK;
}
