/*
 * oracle/icp_oracle.c -- CPU restatement of the ICP registration hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see icp_oracle.h).  PARITY UNPINNED at the
 * libpointmatcher / libnabo boundary: those libraries are absent from
 * /root/reference and from this image; every function below restates their
 * published algorithm as recorded in SURVEY.md Appendix B and cites the
 * reference call site that exercises it.
 */
#include "icp_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ------------------------------------------------------------------------------------------------
 * RigidTransformation::compute  (call sites Mapper.cpp:197,221; Map.cpp:523,525)
 * features' = T * features; Eigen accumulates the product column by column.
 * ---------------------------------------------------------------------------------------------- */
static inline void xf_point(const float* T, const float* p, float* o)
{
    const float x = p[0], y = p[1], z = p[2], w = p[3];
    for (int r = 0; r < 4; ++r)
        o[r] = fmaf(T[12 + r], w, fmaf(T[8 + r], z, fmaf(T[4 + r], y, T[r] * x)));
}

void orc_transform(const float* T, const float* in4, float* out4, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) {
        float o[4];
        xf_point(T, in4 + 4 * i, o);
        memcpy(out4 + 4 * i, o, sizeof o);
    }
}

void orc_rotate3(const float* T, const float* in3, float* out3, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) {
        const float x = in3[3 * i], y = in3[3 * i + 1], z = in3[3 * i + 2];
        float o[3];
        for (int r = 0; r < 3; ++r) o[r] = fmaf(T[8 + r], z, fmaf(T[4 + r], y, T[r] * x));
        memcpy(out3 + 3 * i, o, sizeof o);
    }
}

/* ------------------------------------------------------------------------------------------------
 * libnabo kd-tree (NNS::create KDTREE_LINEAR_HEAP; reference call sites
 * PointDistanceMapperModule.cpp:33-36, DynamicPointsMapperModule.cpp:75-78, and
 * KDTreeMatcher::init/findClosests behind Mapper.cpp:213 / Map.cpp:528).
 * Unbalanced tree, points in leaves, split on the dimension of largest extent at the median.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t dim;       /* split dimension, or -1 for a leaf                       */
    float   cut;       /* split value                                             */
    int32_t right;     /* index of right child (left child is node + 1) / bucket start for leaf */
    int32_t count;     /* leaf: number of points                                  */
} kd_node;

struct orc_kdtree {
    int      dim;
    int64_t  m;
    float*   pts;      /* dim-major copy: m x 4 (x,y,z,pad)                       */
    int32_t* index;    /* permutation: bucket entries -> original index           */
    float*   bpts;     /* points in bucket order (4 floats each)                  */
    kd_node* nodes;
    int64_t  n_nodes, cap_nodes;
    int      bucket;
};

static inline float coord(const orc_kdtree* t, int32_t idx, int d) { return t->pts[4 * (int64_t)idx + d]; }

/* quickselect on idx[lo..hi) so that idx[k] holds the element of rank k by (coord d, idx) */
static void kd_select(const orc_kdtree* t, int32_t* idx, int64_t lo, int64_t hi, int64_t k, int d)
{
    while (hi - lo > 1) {
        /* median of three pivot */
        int64_t mid = lo + (hi - lo) / 2;
        int32_t a = idx[lo], b = idx[mid], c = idx[hi - 1];
        float fa = coord(t, a, d), fb = coord(t, b, d), fc = coord(t, c, d);
        int32_t piv;
        if ((fa <= fb && fb <= fc) || (fc <= fb && fb <= fa)) piv = b;
        else if ((fb <= fa && fa <= fc) || (fc <= fa && fa <= fb)) piv = a;
        else piv = c;
        const float pv = coord(t, piv, d);
        int64_t i = lo, j = hi - 1;
        while (i <= j) {
            while (coord(t, idx[i], d) < pv || (coord(t, idx[i], d) == pv && idx[i] < piv)) ++i;
            while (coord(t, idx[j], d) > pv || (coord(t, idx[j], d) == pv && idx[j] > piv)) --j;
            if (i <= j) { int32_t tmp = idx[i]; idx[i] = idx[j]; idx[j] = tmp; ++i; --j; }
        }
        if (k <= j) hi = j + 1;
        else if (k >= i) lo = i;
        else return;
    }
}

static int32_t kd_new_node(orc_kdtree* t)
{
    if (t->n_nodes == t->cap_nodes) {
        t->cap_nodes = t->cap_nodes ? 2 * t->cap_nodes : 1024;
        t->nodes = (kd_node*)realloc(t->nodes, (size_t)t->cap_nodes * sizeof(kd_node));
    }
    return (int32_t)t->n_nodes++;
}

static void kd_build_rec(orc_kdtree* t, int32_t* idx, int64_t lo, int64_t hi)
{
    const int32_t me = kd_new_node(t);
    const int64_t cnt = hi - lo;
    if (cnt <= t->bucket) {
        t->nodes[me].dim = -1; t->nodes[me].cut = 0.f;
        t->nodes[me].right = (int32_t)lo; t->nodes[me].count = (int32_t)cnt;
        return;
    }
    /* widest dimension of this subset */
    float mn[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, mx[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    for (int64_t i = lo; i < hi; ++i)
        for (int d = 0; d < t->dim; ++d) {
            const float v = coord(t, idx[i], d);
            if (v < mn[d]) mn[d] = v;
            if (v > mx[d]) mx[d] = v;
        }
    int sd = 0;
    for (int d = 1; d < t->dim; ++d) if (mx[d] - mn[d] > mx[sd] - mn[sd]) sd = d;
    const int64_t mid = lo + cnt / 2;
    kd_select(t, idx, lo, hi, mid, sd);
    const float cut = coord(t, idx[mid], sd);
    kd_build_rec(t, idx, lo, mid);
    const int32_t right = (int32_t)t->n_nodes;
    kd_build_rec(t, idx, mid, hi);
    t->nodes[me].dim = sd; t->nodes[me].cut = cut; t->nodes[me].right = right; t->nodes[me].count = 0;
}

orc_kdtree* orc_kdtree_build(const float* pts4, int64_t m, int dim, int bucket_size)
{
    orc_kdtree* t = (orc_kdtree*)calloc(1, sizeof *t);
    t->dim = dim; t->m = m; t->bucket = bucket_size > 0 ? bucket_size : 8;
    t->pts = (float*)malloc((size_t)(m > 0 ? m : 1) * 4 * sizeof(float));
    memcpy(t->pts, pts4, (size_t)m * 4 * sizeof(float));
    t->index = (int32_t*)malloc((size_t)(m > 0 ? m : 1) * sizeof(int32_t));
    for (int64_t i = 0; i < m; ++i) t->index[i] = (int32_t)i;
    if (m > 0) kd_build_rec(t, t->index, 0, m);
    t->bpts = (float*)malloc((size_t)(m > 0 ? m : 1) * 4 * sizeof(float));
    for (int64_t i = 0; i < m; ++i) memcpy(t->bpts + 4 * i, t->pts + 4 * (int64_t)t->index[i], 4 * sizeof(float));
    return t;
}

void orc_kdtree_free(orc_kdtree* t)
{
    if (!t) return;
    free(t->pts); free(t->index); free(t->bpts); free(t->nodes); free(t);
}

/* bounded result list kept ascending by (d2, id): libnabo's linear heap */
typedef struct { int k; int filled; int32_t* id; float* d2; } knn_heap;

static inline int cand_less(float da, int32_t ia, float db, int32_t ib)
{
    return da < db || (da == db && ia < ib);
}

static inline void heap_insert(knn_heap* h, float d2, int32_t id)
{
    int pos = h->filled < h->k ? h->filled : h->k - 1;
    if (h->filled == h->k && !cand_less(d2, id, h->d2[pos], h->id[pos])) return;
    while (pos > 0 && cand_less(d2, id, h->d2[pos - 1], h->id[pos - 1])) {
        h->d2[pos] = h->d2[pos - 1]; h->id[pos] = h->id[pos - 1]; --pos;
    }
    h->d2[pos] = d2; h->id[pos] = id;
    if (h->filled < h->k) ++h->filled;
}

static inline float heap_worst(const knn_heap* h) { return h->filled < h->k ? INFINITY : h->d2[h->k - 1]; }

static inline float sqdist(const float* q, const float* p, int dim)
{
    const float dx = q[0] - p[0], dy = q[1] - p[1];
    float d = fmaf(dy, dy, dx * dx);
    if (dim > 2) { const float dz = q[2] - p[2]; d = fmaf(dz, dz, d); }
    return d;
}

/* err2 = (1 + epsilon)^2: libnabo's maxError2 (nabo/kdtree_cpu.cpp, recurseKnn: `if ((new_rd <= maxRadius2) && (new_rd * maxError2 <
 * heap.headValue()))`); 1 for the exact search */
static void kd_search(const orc_kdtree* t, int32_t node, const float* q, float* off, float rd,
                      float maxr2, int allow_self, knn_heap* h, float err2)
{
    const kd_node* nd = &t->nodes[node];
    if (nd->dim < 0) {
        const float* bp = t->bpts + 4 * (int64_t)nd->right;
        for (int i = 0; i < nd->count; ++i, bp += 4) {
            const float d2 = sqdist(q, bp, t->dim);
            if (d2 <= maxr2 && (allow_self || d2 > FLT_EPSILON)) heap_insert(h, d2, t->index[nd->right + i]);
        }
        return;
    }
    const int d = nd->dim;
    const float old_off = off[d];
    const float new_off = q[d] - nd->cut;
    int32_t near_c, far_c;
    if (new_off > 0.f) { near_c = nd->right; far_c = node + 1; }
    else { near_c = node + 1; far_c = nd->right; }
    kd_search(t, near_c, q, off, rd, maxr2, allow_self, h, err2);
    /* incremental distance to the far half-space (Arya & Mount); a tiny relative slack keeps the
     * prune conservative w.r.t. the fmaf-evaluated point distances */
    const float nrd = rd - old_off * old_off + new_off * new_off;
    const float bound = nrd * (1.0f - 4.0f * FLT_EPSILON) - FLT_MIN;
    if (bound <= maxr2 && (err2 == 1.0f ? bound <= heap_worst(h) : bound * err2 < heap_worst(h))) {
        off[d] = new_off;
        kd_search(t, far_c, q, off, nrd, maxr2, allow_self, h, err2);
        off[d] = old_off;
    }
}

static void knn_finish(knn_heap* h)
{
    for (int j = h->filled; j < h->k; ++j) { h->id[j] = -1; h->d2[j] = INFINITY; }
}

void orc_kdtree_knn(const orc_kdtree* t, const float* q4, int64_t n, int k, float max_radius,
                    int allow_self, int32_t* ids, float* d2, int nthreads)
{
    orc_kdtree_knn_eps(t, q4, n, k, max_radius, 0.f, allow_self, ids, d2, nthreads);
}

/* KDTreeMatcher{epsilon}: NNS::knn(query, ids, dists2, knn, epsilon, optionFlags, maxRadius) -- an approximate search; WHICH (1 + epsilon)-
 * answer comes back depends on the traversal order of the tree (libnabo's own, this one's): only the guarantee is common ground */
void orc_kdtree_knn_eps(const orc_kdtree* t, const float* q4, int64_t n, int k, float max_radius, float epsilon,
                        int allow_self, int32_t* ids, float* d2, int nthreads)
{
    const float err2 = (1.0f + epsilon) * (1.0f + epsilon);
    const float maxr2 = isinf(max_radius) ? INFINITY : max_radius * max_radius;
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 256) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int64_t i = 0; i < n; ++i) {
        knn_heap h = { k, 0, ids + (int64_t)k * i, d2 + (int64_t)k * i };
        if (t->m > 0) {
            float off[3] = { 0.f, 0.f, 0.f };
            kd_search(t, 0, q4 + 4 * i, off, 0.f, maxr2, allow_self, &h, err2);
        }
        knn_finish(&h);
    }
}

void orc_bruteforce_knn(const float* pts4, int64_t m, int dim, const float* q4, int64_t n, int k,
                        float max_radius, int allow_self, int32_t* ids, float* d2)
{
    const float maxr2 = isinf(max_radius) ? INFINITY : max_radius * max_radius;
    for (int64_t i = 0; i < n; ++i) {
        knn_heap h = { k, 0, ids + (int64_t)k * i, d2 + (int64_t)k * i };
        for (int64_t j = 0; j < m; ++j) {
            const float dd = sqdist(q4 + 4 * i, pts4 + 4 * j, dim);
            if (dd <= maxr2 && (allow_self || dd > FLT_EPSILON)) heap_insert(&h, dd, (int32_t)j);
        }
        knn_finish(&h);
    }
}

/* ------------------------------------------------------------------------------------------------
 * Matches::getDistsQuantile: nth_element over the finite, strictly positive entries,
 * index = size * quantile evaluated in float, quantile == 1 -> max.
 * ---------------------------------------------------------------------------------------------- */
static float select_rank(float* v, int64_t n, int64_t k)
{
    int64_t lo = 0, hi = n - 1;
    while (lo < hi) {
        const float a = v[lo], b = v[lo + (hi - lo) / 2], c = v[hi];
        float pv = (a < b) ? ((b < c) ? b : (a < c ? c : a)) : ((a < c) ? a : (b < c ? c : b));
        int64_t i = lo, j = hi;
        while (i <= j) {
            while (v[i] < pv) ++i;
            while (v[j] > pv) --j;
            if (i <= j) { float tmp = v[i]; v[i] = v[j]; v[j] = tmp; ++i; --j; }
        }
        if (k <= j) hi = j;
        else if (k >= i) lo = i;
        else break;
    }
    return v[k];
}

float orc_dists_quantile(const float* d2, int64_t count, float quantile)
{
    float* vals = (float*)malloc((size_t)(count > 0 ? count : 1) * sizeof(float));
    int64_t n = 0;
    for (int64_t i = 0; i < count; ++i)
        if (d2[i] != INFINITY && d2[i] > 0.f) vals[n++] = d2[i];
    float r;
    if (n == 0) r = -1.f;
    else if (quantile == 1.0f) {
        r = vals[0];
        for (int64_t i = 1; i < n; ++i) if (vals[i] > r) r = vals[i];
    } else {
        int64_t k = (int64_t)((float)n * quantile);
        if (k > n - 1) k = n - 1;
        r = select_rank(vals, n, k);
    }
    free(vals);
    return r;
}

/* VarTrimmedDistOutlierFilter::optimizeInlierRatio [UPSTREAM OutlierFiltersImpl.cpp; Phillips et al. 2007, FRMSD]: the valid
 * squared distances sorted ascending, their running sum, and over the ranks minEl = floor(minRatio N) .. maxEl = floor(maxRatio N)
 * (N = ALL entries, valid or not -- upstream's points_nbr) the minimiser of
 *     FRMS(i) = (sum of the i + 1 smallest) / ((i + 1) ((i + 1) / N)^(2 lambda));          optRatio = i_min / N  (float)
 * Restated deviations: the running sum and FRMS are formed in double (upstream: float partial_sum and Eigen float arrays, whose
 * rounding no other summation order reproduces); ranks at or beyond the number of valid entries -- where upstream's Eigen::Map
 * reads the unwritten tail of a reserved vector -- are not candidates; no candidate at all => optRatio = minEl / N. */
static int orc_cmp_float(const void* a, const void* b) { const float x = *(const float*)a, y = *(const float*)b; return x < y ? -1 : (x > y ? 1 : 0); }
float orc_var_trimmed_ratio(const float* d2, int64_t count, float min_ratio, float max_ratio, float lambda)
{
    float* vals = (float*)malloc((size_t)(count > 0 ? count : 1) * sizeof(float));
    int64_t V = 0;
    for (int64_t i = 0; i < count; ++i)
        if (d2[i] != INFINITY && d2[i] > 0.f) vals[V++] = d2[i];
    if (V == 0) { free(vals); return -1.f; }
    qsort(vals, (size_t)V, sizeof(float), orc_cmp_float);
    const int64_t min_el = (int64_t)floorf(min_ratio * (float)count), max_el = (int64_t)floorf(max_ratio * (float)count);
    const int64_t hi = max_el < V ? max_el : V;
    double cum = 0.0, best = 0.0;
    int64_t best_i = -1;
    for (int64_t i = 0; i < hi; ++i) {
        cum += (double)vals[i];
        if (i < min_el) continue;
        const double ids = (double)(i + 1), ratio = ids / (double)count;
        const double frms = cum / (ids * pow(ratio, 2.0 * (double)lambda));
        if (best_i < 0 || frms < best) { best = frms; best_i = i; }
    }
    free(vals);
    if (best_i < 0) best_i = min_el;
    return (float)best_i / (float)count;
}

/* ------------------------------------------------------------------------------------------------
 * OutlierFilters::compute. Empty chain => all ones; filters multiply.
 * ---------------------------------------------------------------------------------------------- */
int orc_outlier_weights(const orc_config* cfg, const float* d2, const int32_t* ids, int k, int64_t n,
                        const float* read_normals3, const float* ref_normals3, float* weights,
                        float* limit_out)
{
    float scale = 1.f;
    return orc_outlier_weights_ex(cfg, d2, ids, k, n, read_normals3, ref_normals3, NULL, NULL, NULL, 1, &scale, weights, limit_out);
}

/* rank size/2 of the finite entries (Matches::getMedianAbsDeviation's two nth_element calls) */
static float orc_median_finite(const float* v, int64_t cnt, int* empty)
{
    float* tmp = (float*)malloc((size_t)(cnt > 0 ? cnt : 1) * sizeof(float));
    int64_t m = 0;
    for (int64_t i = 0; i < cnt; ++i) if (v[i] != INFINITY && v[i] == v[i]) tmp[m++] = v[i];
    *empty = m == 0;
    const float r = m ? select_rank(tmp, m, m / 2) : 0.f;
    free(tmp);
    return r;
}

/* GenericDescriptorOutlierFilter{source: reading}: the 1-row descriptor of the READING point decides (same rule as source: reference,
 * OutlierFiltersImpl.cpp as recalled).  The row is handed over out of band (the checker is single threaded): orc_set_reading_scalar for the
 * stage call, orc_icp_set_reading_scalar for a registration. */
static const float* g_read_scalar = NULL;
void orc_set_reading_scalar(const float* scalar) { g_read_scalar = scalar; }

int orc_outlier_weights_ex(const orc_config* cfg, const float* d2, const int32_t* ids, int k, int64_t n, const float* read_normals3,
                           const float* ref_normals3, const float* ref_scalar, const float* step4, const float* ref4, int iteration,
                           float* robust_scale, float* weights, float* limit_out)
{
    const int64_t cnt = (int64_t)k * n;
    for (int64_t i = 0; i < cnt; ++i) weights[i] = 1.0f;
    if (limit_out) *limit_out = -1.f;
    for (int f = 0; f < cfg->n_outlier; ++f) {
        const int type = cfg->outlier[f].type;
        const float prm = cfg->outlier[f].param;
        if (type == ORC_OUT_MAXDIST) {
            const float lim = prm * prm;
            for (int64_t i = 0; i < cnt; ++i) weights[i] *= (d2[i] <= lim) ? 1.f : 0.f;
        } else if (type == ORC_OUT_MINDIST) {
            const float lim = prm * prm;
            for (int64_t i = 0; i < cnt; ++i) weights[i] *= (d2[i] >= lim) ? 1.f : 0.f;
        } else if (type == ORC_OUT_MEDIANDIST || type == ORC_OUT_TRIMMEDDIST) {
            float lim;
            if (type == ORC_OUT_MEDIANDIST) {
                const float med = orc_dists_quantile(d2, cnt, 0.5f);
                if (med < 0.f) return ORC_ERR_NO_OUTLIER_TO_FILTER;
                lim = prm * med;
            } else {
                lim = orc_dists_quantile(d2, cnt, prm);
                if (lim < 0.f) return ORC_ERR_NO_OUTLIER_TO_FILTER;
            }
            if (limit_out) *limit_out = lim;
            for (int64_t i = 0; i < cnt; ++i) weights[i] *= (d2[i] <= lim) ? 1.f : 0.f;
        } else if (type == ORC_OUT_VARTRIMMEDDIST) {
            /* VarTrimmedDistOutlierFilter{minRatio, maxRatio, lambda}: TrimmedDist at the ratio optimizeInlierRatio picks */
            const float ratio = orc_var_trimmed_ratio(d2, cnt, prm, cfg->outlier[f].param2, cfg->outlier[f].param3);
            if (ratio < 0.f) return ORC_ERR_NO_OUTLIER_TO_FILTER;
            const float lim = orc_dists_quantile(d2, cnt, ratio);
            if (lim < 0.f) return ORC_ERR_NO_OUTLIER_TO_FILTER;
            if (limit_out) *limit_out = lim;
            for (int64_t i = 0; i < cnt; ++i) weights[i] *= (d2[i] <= lim) ? 1.f : 0.f;
        } else if (type == ORC_OUT_SURFACENORMAL) {
            if (!read_normals3 || !ref_normals3) return ORC_ERR_ARG;
            const float cosmax = cosf(prm);
            for (int64_t i = 0; i < n; ++i)
                for (int j = 0; j < k; ++j) {
                    const int32_t id = ids[(int64_t)k * i + j];
                    if (id < 0) { weights[(int64_t)k * i + j] = 0.f; continue; }
                    const float* a = read_normals3 + 3 * i;
                    const float* b = ref_normals3 + 3 * (int64_t)id;
                    const float dot = fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0]));
                    weights[(int64_t)k * i + j] *= (dot > cosmax) ? 1.f : 0.f;
                }
        } else if (type == ORC_OUT_GENERICDESCRIPTOR) {
            /* GenericDescriptorOutlierFilter{source, descName, useSoftThreshold, useLargerThan, threshold} [UPSTREAM]: the 1-row
             * descriptor of the matched reference point (source: reference) decides: hard -> (desc > threshold) or (desc <
             * threshold); soft -> the weight IS the descriptor value */
            const int ip = cfg->outlier[f].iparam;
            const int from_reading = (ip & ORC_GEN_SOURCE_READING) != 0;
            if (from_reading ? !g_read_scalar : !ref_scalar) return ORC_ERR_ARG;
            for (int64_t e = 0; e < cnt; ++e) {
                const int32_t id = ids[e];
                if (id < 0) { weights[e] = 0.f; continue; }
                const float v = from_reading ? g_read_scalar[e / k] : ref_scalar[id];
                float w;
                if (ip & ORC_GEN_SOFT) w = v;
                else w = (ip & ORC_GEN_LARGER) ? (v > prm ? 1.f : 0.f) : (v < prm ? 1.f : 0.f);
                weights[e] *= w;
            }
        } else if (type == ORC_OUT_ROBUST) {
            /* RobustOutlierFilter{robustFct, tuning, scaleEstimator none | mad | berg | std, nbIterationForScale, distanceType,
             * approximation} [UPSTREAM: OutlierFiltersImpl.cpp RobustOutlierFilter / robustFiltering, libpointmatcher 1.4.x as recalled --
             * the library is not in this image]: while iteration <= nbIterationForScale (always when that is 0) the scale is
             * re-estimated --
             *   mad : sqrt(Matches::getMedianAbsDeviation()), the rank size/2 of |d2 - median(d2)| over the finite SQUARED MATCH
             *         distances (whatever distanceType says);
             *   std : sqrt(Matches::getStandardDeviation()) = sqrt(sqrt(sum (d - mean(d))^2 / (size - 1))) over EVERY entry of the
             *         distance matrix (an infinite entry -- maxDist, fewer than k matches -- makes it NaN upstream too; the registration
             *         then ends with "not a number" like every non-finite weight, see the minimizer);
             *   berg: iteration 1: 1.9 * sqrt(getDistsQuantile(0.5)), later scale = 0.85 (scale - target) + target, where target is the
             *         configured `tuning` and the M-estimator's tuning constant becomes Bergstrom's 4.3040 (cauchy), 7.0589 (tukey),
             *         2.0138 (huber); the other functions keep `tuning` for both;
             *   none: 1.
             * e2 = residual / scale^2 with residual = the squared match distance (point2point) or the squared point-to-plane distance;
             * w = the M-estimator's weight of e2 with tuning k; negative weights clamp to 0 (ARBITRARY_SMALL_VALUE underflows to 0 in
             * float); e2 >= approximation^2 forces the weight to 0.  The reductions of `std` run in double and round to float where
             * Eigen would hold a float (mean, sum of squares): Eigen's packet order is not knowable here, the double sum is within half
             * an ulp of any of them.  expf / powf go through double so that host libm and device ocml round to the same float. */
            const int ip = cfg->outlier[f].iparam;
            const int fct = ip & 15, sc = (ip >> 4) & 15, dt = (ip >> 8) & 15;
            const int nb_scale = (int)cfg->outlier[f].param2;
            float tuning = prm;
            if (sc == ORC_SCALE_BERG) tuning = fct == ORC_ROB_CAUCHY ? 4.3040f : fct == ORC_ROB_TUKEY ? 7.0589f : fct == ORC_ROB_HUBER ? 2.0138f : prm;
            if (nb_scale == 0 || iteration <= nb_scale) {
                if (sc == ORC_SCALE_MAD) {
                    int empty = 0;
                    const float med = orc_median_finite(d2, cnt, &empty);
                    if (empty) return ORC_ERR_NO_OUTLIER_TO_FILTER;
                    float* dev = (float*)malloc((size_t)cnt * sizeof(float));
                    for (int64_t e = 0; e < cnt; ++e) dev[e] = d2[e] == INFINITY ? INFINITY : fabsf(d2[e] - med);
                    const float mad = orc_median_finite(dev, cnt, &empty);
                    free(dev);
                    *robust_scale = sqrtf(mad);
                } else if (sc == ORC_SCALE_STD) {
                    double s = 0.0;
                    for (int64_t e = 0; e < cnt; ++e) s += (double)d2[e];
                    const float mean = (float)(s / (double)cnt);
                    double ss = 0.0;
                    for (int64_t e = 0; e < cnt; ++e) { const float dv = d2[e] - mean; ss += (double)(dv * dv); }
                    const float var = (float)ss / (float)(cnt - 1);
                    *robust_scale = sqrtf(sqrtf(var));
                } else if (sc == ORC_SCALE_BERG) {
                    if (iteration == 1) {
                        const float med = orc_dists_quantile(d2, cnt, 0.5f);
                        if (med < 0.f) return ORC_ERR_NO_OUTLIER_TO_FILTER;
                        *robust_scale = (float)(1.9 * (double)sqrtf(med));
                    } else {
                        const float rate = 0.85f;
                        const float t = rate * (*robust_scale - prm);
                        *robust_scale = t + prm;
                    }
                } else *robust_scale = 1.f;
            }
            const float s2 = *robust_scale * *robust_scale;
            const float kk = tuning, k2 = tuning * tuning;
            const float apx = cfg->outlier[f].param3;
            const int has_apx = apx > 0.f && apx != INFINITY;
            const float apx2 = apx * apx;
            if (limit_out && sc != ORC_SCALE_NONE) *limit_out = *robust_scale;
            for (int64_t i = 0; i < n; ++i)
                for (int j = 0; j < k; ++j) {
                    const int64_t e = (int64_t)k * i + j;
                    const int32_t id = ids[e];
                    if (id < 0 || d2[e] == INFINITY) { weights[e] = 0.f; continue; }
                    float res = d2[e];
                    if (dt == ORC_DIST_POINT2PLANE) {
                        if (!step4 || !ref4 || !ref_normals3) return ORC_ERR_ARG;
                        const float* p = step4 + 4 * i; const float* q = ref4 + 4 * (int64_t)id; const float* nn = ref_normals3 + 3 * (int64_t)id;
                        const float dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2];
                        const float dot = dx * nn[0] + dy * nn[1] + dz * nn[2];
                        res = dot * dot;
                    }
                    const float e2 = res / s2;
                    float w;
                    switch (fct) {
                    case ORC_ROB_CAUCHY: w = 1.f / (1.f + e2 / k2); break;
                    case ORC_ROB_WELSCH: w = (float)exp((double)(-e2 / k2)); break;
                    case ORC_ROB_SC: { const float t = kk + e2; w = e2 >= kk ? 4.f * k2 * (1.f / (t * t)) : 1.f; break; }
                    case ORC_ROB_GM: { const float t = kk + e2; w = k2 * (1.f / (t * t)); break; }
                    case ORC_ROB_TUKEY: { const float t = 1.f - e2 / k2; w = e2 >= k2 ? 0.f : t * t; break; }
                    case ORC_ROB_HUBER: w = e2 >= k2 ? kk * (1.f / sqrtf(e2)) : 1.f; break;
                    case ORC_ROB_L1: w = 1.f / sqrtf(e2); break;
                    default: { /* Student, d = 3 */
                        const float pw = (float)pow((double)(1.f + e2 / kk), (double)(-(kk + 3.f) / 2.f));
                        w = pw * (kk + 3.f) * (1.f / (kk + e2));
                        break; }
                    }
                    if (w <= 0.f) w = 0.f;
                    if (has_apx && e2 >= apx2) w = 0.f;
                    weights[e] *= w;
                }
        } else {
            return ORC_ERR_ARG;
        }
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------------
 * small dense algebra (float, as Eigen on PM::Matrix = Matrix<float>)
 * ---------------------------------------------------------------------------------------------- */
static void mat4_mul(const float* A, const float* B, float* C) /* col-major C = A B */
{
    float R[16];
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i) {
            float s = A[i] * B[4 * j];
            for (int kk = 1; kk < 4; ++kk) s = fmaf(A[4 * kk + i], B[4 * j + kk], s);
            R[4 * j + i] = s;
        }
    memcpy(C, R, sizeof R);
}

static void mat4_identity(float* T)
{
    memset(T, 0, 16 * sizeof(float));
    T[0] = T[5] = T[10] = T[15] = 1.f;
}

/* one-sided (Hestenes) Jacobi SVD of a 3x3 float matrix, singular values sorted descending like
 * Eigen::JacobiSVD. A = U diag(s) V^T. Columns of U for (near) zero singular values are completed
 * by cross products so that U is orthonormal. */
static void svd3f(const float* H, float* U, float* s, float* V)
{
    float a[9]; memcpy(a, H, sizeof a);                /* col-major working copy, columns rotate */
    float v[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
    for (int sweep = 0; sweep < 30; ++sweep) {
        float off = 0.f;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                float alpha = 0, beta = 0, gamma = 0;
                for (int i = 0; i < 3; ++i) {
                    alpha += a[3 * p + i] * a[3 * p + i];
                    beta += a[3 * q + i] * a[3 * q + i];
                    gamma += a[3 * p + i] * a[3 * q + i];
                }
                if (gamma == 0.f) continue;
                const float lim = fabsf(gamma) / sqrtf(fmaxf(alpha * beta, FLT_MIN));
                if (lim > off) off = lim;
                if (lim <= 1e-9f) continue;
                const float zeta = (beta - alpha) / (2.f * gamma);
                const float tt = (zeta >= 0.f ? 1.f : -1.f) / (fabsf(zeta) + sqrtf(1.f + zeta * zeta));
                const float c = 1.f / sqrtf(1.f + tt * tt), sn = c * tt;
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
    for (int j = 0; j < 3; ++j)
        sv[j] = sqrtf(a[3 * j] * a[3 * j] + a[3 * j + 1] * a[3 * j + 1] + a[3 * j + 2] * a[3 * j + 2]);
    int ord[3] = { 0, 1, 2 };
    for (int i = 0; i < 2; ++i)
        for (int j = i + 1; j < 3; ++j)
            if (sv[ord[j]] > sv[ord[i]]) { int t = ord[i]; ord[i] = ord[j]; ord[j] = t; }
    for (int j = 0; j < 3; ++j) {
        const int o = ord[j];
        s[j] = sv[o];
        for (int i = 0; i < 3; ++i) { V[3 * j + i] = v[3 * o + i]; U[3 * j + i] = a[3 * o + i]; }
    }
    /* U = A V S^-1, made orthonormal by construction (Eigen's two-sided JacobiSVD returns orthonormal factors even
     * for rank-deficient input; one-sided Jacobi leaves a column of norm ~1e-6 s0 with a direction that is mostly
     * rounding noise): u0 from the largest column; u1 = second column orthogonalised against u0, or -- when what is
     * left of it is noise -- any unit vector orthogonal to u0; u2 = +-(u0 x u1), the sign following the third column. */
    const float tiny = s[0] * 1e-6f;
    if (!(s[0] > 0.f)) { float I[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 }; memcpy(U, I, sizeof I); return; }
    for (int i = 0; i < 3; ++i) U[i] /= s[0];
    { const float n0 = sqrtf(U[0] * U[0] + U[1] * U[1] + U[2] * U[2]); for (int i = 0; i < 3; ++i) U[i] /= n0; }
    int have1 = 0;
    if (s[1] > tiny) {
        float c1[3] = { U[3] / s[1], U[4] / s[1], U[5] / s[1] };
        const float d = c1[0] * U[0] + c1[1] * U[1] + c1[2] * U[2];
        for (int i = 0; i < 3; ++i) c1[i] = c1[i] - d * U[i];
        const float n1 = sqrtf(c1[0] * c1[0] + c1[1] * c1[1] + c1[2] * c1[2]);
        if (n1 > 0.5f) { for (int i = 0; i < 3; ++i) U[3 + i] = c1[i] / n1; have1 = 1; }
    }
    if (!have1) {
        const int m = fabsf(U[0]) < fabsf(U[1]) ? (fabsf(U[0]) < fabsf(U[2]) ? 0 : 2) : (fabsf(U[1]) < fabsf(U[2]) ? 1 : 2);
        const float e[3] = { m == 0 ? 1.f : 0.f, m == 1 ? 1.f : 0.f, m == 2 ? 1.f : 0.f };
        const float d = m == 0 ? U[0] : (m == 1 ? U[1] : U[2]);
        const float w[3] = { e[0] - d * U[0], e[1] - d * U[1], e[2] - d * U[2] };
        const float nw = sqrtf(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
        for (int i = 0; i < 3; ++i) U[3 + i] = w[i] / nw;
    }
    {
        const float c2[3] = { U[6], U[7], U[8] };
        float x[3] = { U[1] * U[5] - U[2] * U[4], U[2] * U[3] - U[0] * U[5], U[0] * U[4] - U[1] * U[3] };
        const float nx = sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
        const float sg = (have1 && s[2] > tiny && (x[0] * c2[0] + x[1] * c2[1] + x[2] * c2[2]) < 0.f) ? -1.f : 1.f;
        for (int i = 0; i < 3; ++i) U[6 + i] = sg * (x[i] / nx);
    }
}

static float det3(const float* R)
{
    return R[0] * (R[4] * R[8] - R[7] * R[5]) - R[3] * (R[1] * R[8] - R[7] * R[2]) + R[6] * (R[1] * R[5] - R[4] * R[2]);
}

/* PointToPointErrorMinimizer::compute tail: R = U V^T, reflection -> negate the last
 * row of V^T.  This is the route through the SVD itself; orc_rotation_from_H takes it for the matrices its fast path declines. */
void orc_rotation_from_H_svd(const float* H, float* R)
{
    float U[9], s[3], V[9];
    svd3f(H, U, s, V);
    for (int pass = 0; pass < 2; ++pass) {
        for (int j = 0; j < 3; ++j)
            for (int i = 0; i < 3; ++i) {
                float acc = 0.f;
                for (int kk = 0; kk < 3; ++kk) acc += U[3 * kk + i] * V[3 * kk + j]; /* U(i,k) V(j,k) */
                R[3 * j + i] = acc;
            }
        if (pass == 0 && det3(R) < 0.f) { for (int i = 0; i < 3; ++i) V[6 + i] = -V[6 + i]; }
        else break;
    }
}

/* U V^T of H = U S V^T is the orthogonal polar factor of H whenever det H > 0 -- no reflection to repair -- and the Newton
 * iteration X <- (X + X^-T) / 2 reaches it without forming U, S or V (Higham 1986; Frobenius-scaled for the first two steps,
 * quadratic afterwards: the step that moves X by d leaves an error ~ d^2 / 2).  Shared numeric spec with the device: every
 * product-sum below is the explicit fmaf / rounding sequence written here.  Returns 0 -- R untouched -- when H is (close to)
 * singular or a reflection: det of the Frobenius-normalised H <= 1e-6. */
static int polar_newton3f(const float* H, float* R)
{
    float n2 = 0.f;
    for (int i = 0; i < 9; ++i) n2 = fmaf(H[i], H[i], n2);
    if (!(n2 > 0.f) || n2 == INFINITY) return 0;
    const float inv = 1.f / sqrtf(n2);
    float X[9], C[9], Y[9], Xn[9];
    for (int i = 0; i < 9; ++i) X[i] = H[i] * inv;
    for (int it = 0; it < 20; ++it) {
        /* cofactors, element (i, j) stored at [3 j + i]; X^-T = C / det */
        for (int j = 0; j < 3; ++j)
            for (int i = 0; i < 3; ++i) {
                const int a = (i + 1) % 3, b = (i + 2) % 3, c = (j + 1) % 3, d = (j + 2) % 3;
                const float t = X[3 * d + a] * X[3 * c + b];
                C[3 * j + i] = fmaf(X[3 * c + a], X[3 * d + b], -t);
            }
        float det = X[0] * C[0];
        det = fmaf(X[3], C[3], det);
        det = fmaf(X[6], C[6], det);
        if (it == 0 && !(det > 1e-6f)) return 0;
        const float invdet = 1.f / det;
        for (int i = 0; i < 9; ++i) Y[i] = C[i] * invdet;
        if (it < 2) {
            float nx = 0.f, ny = 0.f;
            for (int i = 0; i < 9; ++i) { nx = fmaf(X[i], X[i], nx); ny = fmaf(Y[i], Y[i], ny); }
            const float mu = sqrtf(sqrtf(ny / nx)), imu = 1.f / mu;
            for (int i = 0; i < 9; ++i) Xn[i] = 0.5f * fmaf(mu, X[i], Y[i] * imu);
        } else
            for (int i = 0; i < 9; ++i) Xn[i] = 0.5f * (X[i] + Y[i]);
        float dmax = 0.f;
        for (int i = 0; i < 9; ++i) { const float dd = fabsf(Xn[i] - X[i]); if (dd > dmax) dmax = dd; X[i] = Xn[i]; }
        if (!(dmax == dmax)) return 0;
        if (it >= 2 && dmax <= 3e-4f) break;
    }
    for (int i = 0; i < 9; ++i) R[i] = X[i];
    return 1;
}

void orc_rotation_from_H(const float* H, float* R)
{
    if (!polar_newton3f(H, R)) orc_rotation_from_H_svd(H, R);
}

/* symmetric Jacobi eigen-decomposition in double (n <= 6): A = Q diag(w) Q^T */
static void jacobi_eig_sym(int n, const double* Ain, double* w, double* Q)
{
    double A[36];
    for (int i = 0; i < n * n; ++i) A[i] = Ain[i];
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) Q[n * j + i] = (i == j);
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

/* solvePossiblyUnderdeterminedLinearSystem: LLT when A is invertible, otherwise the
 * minimum-norm solution. Deviation (documented): invertibility is judged on the float Cholesky
 * pivots (every pivot > 6 eps_f max_j A_jj) instead of the full-pivot QR rank, and the degenerate
 * branch uses a double-precision symmetric pseudo-inverse; both give upstream's minimum-norm answer
 * up to rounding. */
void orc_solve_n(int n, const float* A, const float* b, float* x)
{
    float dmax = 0.f;
    for (int j = 0; j < n; ++j) if (A[n * j + j] > dmax) dmax = A[n * j + j];
    const float pthr = (float)n * FLT_EPSILON * dmax;
    {
        /* float Cholesky A = L L^T, forward / backward substitution */
        float L[36], iL[6]; memset(L, 0, sizeof L);
        int ok = 1;
        for (int j = 0; j < n && ok; ++j) {
            float d = A[n * j + j];
            for (int kk = 0; kk < j; ++kk) d -= L[n * kk + j] * L[n * kk + j];
            if (!(d > pthr)) { ok = 0; break; }
            const float ljj = sqrtf(d);
            L[n * j + j] = ljj;
            const float ilj = 1.f / ljj; /* one reciprocal per pivot, multiplied through (column and both substitutions) */
            iL[j] = ilj;
            for (int i = j + 1; i < n; ++i) {
                float s = A[n * j + i];
                for (int kk = 0; kk < j; ++kk) s -= L[n * kk + i] * L[n * kk + j];
                L[n * j + i] = s * ilj;
            }
        }
        if (ok) {
            float y[6];
            for (int i = 0; i < n; ++i) {
                float s = b[i];
                for (int kk = 0; kk < i; ++kk) s -= L[n * kk + i] * y[kk];
                y[i] = s * iL[i];
            }
            for (int i = n - 1; i >= 0; --i) {
                float s = y[i];
                for (int kk = i + 1; kk < n; ++kk) s -= L[n * i + kk] * x[kk];
                x[i] = s * iL[i];
            }
            return;
        }
    }
    /* minimum-norm least squares through the eigen-decomposition */
    double Ad[36], w[6], Q[36];
    for (int i = 0; i < n * n; ++i) Ad[i] = A[i];
    jacobi_eig_sym(n, Ad, w, Q);
    double wmax = 0;
    for (int i = 0; i < n; ++i) if (fabs(w[i]) > wmax) wmax = fabs(w[i]);
    const double thr = (double)n * (double)FLT_EPSILON * wmax;
    double xd[6] = { 0, 0, 0, 0, 0, 0 };
    for (int e = 0; e < n; ++e) {
        if (!(w[e] > thr)) continue;
        double proj = 0;
        for (int i = 0; i < n; ++i) proj += Q[n * e + i] * (double)b[i];
        proj /= w[e];
        for (int i = 0; i < n; ++i) xd[i] += proj * Q[n * e + i];
    }
    for (int i = 0; i < n; ++i) x[i] = (float)xd[i];
}
void orc_solve6(const float* A, const float* b, float* x) { orc_solve_n(6, A, b, x); }

/* sin / cos of the step angle in float, specified operation by operation so that host and device produce the same bits
 * (libm's sinf and ocml's are each within an ulp, but not of each other): below 0.5 rad -- every ICP step but a wild first
 * one -- the Taylor polynomials in Horner form with fmaf (truncation < 2e-8 relative, i.e. below half an ulp; the result is
 * within ~1 ulp of the true value); above, through double, where both libraries round to the same float. */
void orc_sincos_f(float x, float* s, float* c)
{
    if (x < 0.5f) {
        const float z = x * x;
        float ps = fmaf(z, 2.75573192e-06f, -1.98412698e-04f);   /* 1/9!, -1/7! */
        ps = fmaf(z, ps, 8.33333333e-03f);                       /* 1/5! */
        ps = fmaf(z, ps, -1.66666667e-01f);                      /* -1/3! */
        *s = fmaf(x * z, ps, x);
        float pc = fmaf(z, -2.75573192e-07f, 2.48015873e-05f);   /* -1/10!, 1/8! */
        pc = fmaf(z, pc, -1.38888889e-03f);                      /* -1/6! */
        pc = fmaf(z, pc, 4.16666667e-02f);                       /* 1/4! */
        pc = fmaf(z, pc, -0.5f);
        *c = fmaf(z, pc, 1.f);
    } else { *s = (float)sin((double)x); *c = (float)cos((double)x); }
}

/* Eigen::AngleAxis(angle, axis).toRotationMatrix() in float */
static void angle_axis_to_R(const float* x3, float* T)
{
    const float nrm = sqrtf(x3[0] * x3[0] + x3[1] * x3[1] + x3[2] * x3[2]);
    mat4_identity(T);
    if (!(nrm > 0.f)) return; /* degenerate: upstream replaces the NaN rotation by identity */
    const float ax = x3[0] / nrm, ay = x3[1] / nrm, az = x3[2] / nrm;
    float s, c;
    orc_sincos_f(nrm, &s, &c);
    const float sx = s * ax, sy = s * ay, sz = s * az;
    const float cx = (1.f - c) * ax, cy = (1.f - c) * ay, cz = (1.f - c) * az;
    float tmp;
    tmp = cx * ay; T[4 * 1 + 0] = tmp - sz; T[4 * 0 + 1] = tmp + sz;
    tmp = cx * az; T[4 * 2 + 0] = tmp + sy; T[4 * 0 + 2] = tmp - sy;
    tmp = cy * az; T[4 * 2 + 1] = tmp - sx; T[4 * 1 + 2] = tmp + sx;
    T[0] = cx * ax + c; T[5] = cy * ay + c; T[10] = cz * az + c;
}

/* ------------------------------------------------------------------------------------------------
 * ErrorMinimizer::compute: gather pairs (valid dist && w != 0), then solve.
 * Sums over pairs are accumulated in double (documented deviation from Eigen's float
 * accumulation: mathematically identical, removes summation-order noise from the comparison).
 * ---------------------------------------------------------------------------------------------- */
int orc_minimize(int minimizer, const float* reading4, int64_t n, const float* ref4,
                 const float* ref_normals3, const int32_t* ids, const float* d2, const float* w, int k,
                 float* T_out, double* A_out, double* b_out, float* x_out, orc_stats* st)
{
    return orc_minimize_ex(minimizer, 0, reading4, n, ref4, ref_normals3, ids, d2, w, k, T_out, A_out, b_out, x_out, st);
}

/* force_4dof == 1: PointToPlaneErrorMinimizer{force4DOF: 1} [UPSTREAM PointToPlane.cpp compute_in_place]: the cross product is
 * reduced to its z component, F = [cross_z; n] (4 rows), x = (yaw, t); the step is AngleAxis(x0, unitZ) + t.  F's rows are rows
 * 2..5 of the 6-DOF F, so A and b are the {2,3,4,5} sub-system of the 6-DOF sums.
 * force_4dof == 2: {force2D: 1} on 3-D clouds: upstream drops the z row of the features and takes the top two rows of the normals
 * (not renormalised): F = [x ny - y nx; nx; ny] -- rows 2..4 of the 6-DOF F --, the residual is the 2-D dot (dx nx + dy ny),
 * x = (yaw, tx, ty), the step is Rotation2D(x0) + (tx, ty) embedded in the 4 x 4 identity. */
int orc_minimize_ex(int minimizer, int force_4dof, const float* reading4, int64_t n, const float* ref4,
                    const float* ref_normals3, const int32_t* ids, const float* d2, const float* w, int k,
                    float* T_out, double* A_out, double* b_out, float* x_out, orc_stats* st)
{
    int64_t P = 0;
    double wsum = 0;
    double sp[3] = { 0, 0, 0 }, sq[3] = { 0, 0, 0 }, Hs[9] = { 0 };
    double A[36] = { 0 }, b[6] = { 0 }, b2d[3] = { 0, 0, 0 };
    for (int64_t i = 0; i < n; ++i)
        for (int j = 0; j < k; ++j) {
            const int64_t e = (int64_t)k * i + j;
            if (d2[e] == INFINITY) continue;
            const float we = w[e];
            if (we == 0.f) continue;
            ++P; wsum += we;
            if (minimizer == ORC_MIN_IDENTITY) continue;
            const float* p = reading4 + 4 * i;
            const float* q = ref4 + 4 * (int64_t)ids[e];
            if (minimizer == ORC_MIN_POINT_TO_POINT) {
                for (int r = 0; r < 3; ++r) { sp[r] += (double)we * p[r]; sq[r] += (double)we * q[r]; }
                for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) Hs[3 * c + r] += (double)we * q[r] * p[c];
            } else {
                const float* nn = ref_normals3 + 3 * (int64_t)ids[e];
                /* cross = reading x normal, F = [cross; normal], dot = (p - q) . n  -- float per pair */
                const float F[6] = { p[1] * nn[2] - p[2] * nn[1], p[2] * nn[0] - p[0] * nn[2],
                                     p[0] * nn[1] - p[1] * nn[0], nn[0], nn[1], nn[2] };
                const float dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2];
                const float dot = dx * nn[0] + dy * nn[1] + dz * nn[2];
                for (int c = 0; c < 6; ++c) {
                    const double wf = (double)we * F[c];
                    for (int r = 0; r < 6; ++r) A[6 * c + r] += wf * F[r];
                    b[c] -= wf * dot;
                }
                if ((force_4dof & 3) == 2) {
                    const float dot2 = dx * nn[0] + dy * nn[1];
                    for (int c = 0; c < 3; ++c) b2d[c] -= ((double)we * F[2 + c]) * dot2;
                }
            }
        }
    if (st) {
        st->pairs = P;
        st->point_used_ratio = (float)P / (float)((int64_t)k * n);
        st->weighted_point_used_ratio = (float)(wsum / (double)((int64_t)k * n));
    }
    if (P == 0) return ORC_ERR_NO_POINT_TO_MINIMIZE;
    mat4_identity(T_out);
    /* A weight that is not a number the sums can hold (RobustOutlierFilter L1 / Huber on a residual of exactly zero: 1 / sqrt(0)) makes
     * the system meaningless: upstream's float LLT would return NaN or garbage depending on where the inf lands.  Both sides of this
     * repository agree on the outcome instead -- "transformation is not a number" (the HIP path's fixed-point accumulators refuse a
     * partial beyond +-2^77, csrc/loop.hip: accumulate_kernel). */
    {
        int bad = !(fabs(wsum) < 0x1p77);
        for (int r = 0; r < 3; ++r) bad |= !(fabs(sp[r]) < 0x1p77) || !(fabs(sq[r]) < 0x1p77) || !(fabs(b2d[r]) < 0x1p77);
        for (int r = 0; r < 9; ++r) bad |= !(fabs(Hs[r]) < 0x1p77);
        for (int r = 0; r < 36; ++r) bad |= !(fabs(A[r]) < 0x1p77);
        for (int r = 0; r < 6; ++r) bad |= !(fabs(b[r]) < 0x1p77);
        if (bad) return ORC_ERR_NAN;
    }
    if (minimizer == ORC_MIN_IDENTITY) return ORC_OK;
    if (minimizer == ORC_MIN_POINT_TO_POINT) {
        /* H = sum w (q - mq)(p - mp)^T = sum w q p^T - (sum w q)(sum w p)^T / sum w */
        float H[9], R[9];
        double mp[3], mq[3];
        const double iw = 1.0 / wsum; /* one reciprocal, multiplied through */
        for (int r = 0; r < 3; ++r) { mp[r] = sp[r] * iw; mq[r] = sq[r] * iw; }
        for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) H[3 * c + r] = (float)(Hs[3 * c + r] - mq[r] * sp[c]);
        if (A_out) for (int i = 0; i < 9; ++i) A_out[i] = H[i];
        if (force_4dof & 4) {
            /* planar clouds: the proper rotation that maximises tr(R^T H2) over the plane, R(theta) with
             * theta = atan2(H10 - H01, H00 + H11) -- what the 2 x 2 SVD with its reflection repair returns -- in closed form */
            const float a = H[0] + H[4], b2 = H[1] - H[3]; /* H(1,0) = H[3*0+1], H(0,1) = H[3*1+0] */
            const float r = sqrtf(a * a + b2 * b2);
            float cs = 1.f, sn = 0.f;
            if (r > 0.f) { cs = a / r; sn = b2 / r; }
            for (int i = 0; i < 9; ++i) R[i] = 0.f;
            R[0] = cs; R[1] = sn; R[3] = -sn; R[4] = cs; R[8] = 1.f;
        } else
        orc_rotation_from_H(H, R);
        const float mpf[3] = { (float)mp[0], (float)mp[1], (float)mp[2] };
        const float mqf[3] = { (float)mq[0], (float)mq[1], (float)mq[2] };
        for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) T_out[4 * c + r] = R[3 * c + r];
        for (int r = 0; r < 3; ++r)
            T_out[12 + r] = mqf[r] - (R[r] * mpf[0] + R[3 + r] * mpf[1] + R[6 + r] * mpf[2]);
        return ORC_OK;
    }
    float Af[36], bf[6], x[6];
    for (int i = 0; i < 36; ++i) Af[i] = (float)A[i];
    for (int i = 0; i < 6; ++i) bf[i] = (float)b[i];
    if (A_out) memcpy(A_out, A, sizeof A);
    if (b_out) memcpy(b_out, b, sizeof b);
    if ((force_4dof & 3) == 2) {
        float A3[9], b3[3], x3[3];
        for (int c = 0; c < 3; ++c) { b3[c] = (float)b2d[c]; for (int r = 0; r < 3; ++r) A3[3 * c + r] = Af[6 * (2 + c) + (2 + r)]; }
        orc_solve_n(3, A3, b3, x3);
        x[0] = 0.f; x[1] = 0.f; x[2] = x3[0]; x[3] = x3[1]; x[4] = x3[2]; x[5] = 0.f;
    } else if ((force_4dof & 3) == 1) {
        float A4[16], b4[4], x4[4];
        for (int c = 0; c < 4; ++c) { b4[c] = bf[2 + c]; for (int r = 0; r < 4; ++r) A4[4 * c + r] = Af[6 * (2 + c) + (2 + r)]; }
        orc_solve_n(4, A4, b4, x4);
        x[0] = 0.f; x[1] = 0.f; x[2] = x4[0]; x[3] = x4[1]; x[4] = x4[2]; x[5] = x4[3];
    } else orc_solve6(Af, bf, x);
    if (x_out) memcpy(x_out, x, sizeof x);
    angle_axis_to_R(x, T_out);
    T_out[12] = x[3]; T_out[13] = x[4]; T_out[14] = x[5];
    for (int i = 0; i < 16; ++i) if (T_out[i] != T_out[i]) return ORC_ERR_NAN;
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------------
 * TransformationCheckers
 * ---------------------------------------------------------------------------------------------- */
static void quat_from_R(const float* T, double* q /* w x y z */)
{
    /* Eigen::Quaternion(Matrix3) */
    const double m00 = T[0], m11 = T[5], m22 = T[10];
    double t = m00 + m11 + m22;
    if (t > 0) {
        t = sqrt(t + 1.0);
        q[0] = 0.5 * t; t = 0.5 / t;
        q[1] = (T[4 * 1 + 2] - T[4 * 2 + 1]) * t;
        q[2] = (T[4 * 2 + 0] - T[4 * 0 + 2]) * t;
        q[3] = (T[4 * 0 + 1] - T[4 * 1 + 0]) * t;
    } else {
        int i = 0;
        if (m11 > m00) i = 1;
        if (m22 > (i == 0 ? m00 : m11)) i = 2;
        const int j = (i + 1) % 3, kk = (j + 1) % 3;
#define M(r, c) ((double)T[4 * (c) + (r)])
        t = sqrt(M(i, i) - M(j, j) - M(kk, kk) + 1.0);
        double v[3];
        v[i] = 0.5 * t; t = 0.5 / t;
        q[0] = (M(kk, j) - M(j, kk)) * t;
        v[j] = (M(j, i) + M(i, j)) * t;
        v[kk] = (M(kk, i) + M(i, kk)) * t;
#undef M
        q[1] = v[0]; q[2] = v[1]; q[3] = v[2];
    }
}

static double quat_angular_distance(const double* a, const double* b)
{
    /* d = a * conj(b); 2 atan2(|d.vec|, |d.w|) */
    const double w = a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
    const double x = -a[0] * b[1] + a[1] * b[0] - a[2] * b[3] + a[3] * b[2];
    const double y = -a[0] * b[2] + a[1] * b[3] + a[2] * b[0] - a[3] * b[1];
    const double z = -a[0] * b[3] - a[1] * b[2] + a[2] * b[1] + a[3] * b[0];
    return 2.0 * atan2(sqrt(x * x + y * y + z * z), fabs(w));
}

/* ------------------------------------------------------------------------------------------------
 * ICPSequence (call sites Mapper.cpp:213, Map.cpp:111,178,528,581)
 * ---------------------------------------------------------------------------------------------- */
struct orc_icp {
    orc_config cfg;
    int64_t m;
    float mean[3];
    float* map4;       /* centred map                                       */
    float* normals3;   /* or NULL                                           */
    float* scalar;     /* or NULL: the descriptor GenericDescriptorOutlierFilter reads */
    orc_kdtree* tree;
    float* read_noise; int64_t read_noise_n; /* simpleSensorNoise of the next reading (one shot) */
    float* read_scalar; int64_t read_scalar_n; /* GenericDescriptorOutlierFilter{source: reading}: that descriptor of the next reading (one shot) */
};

orc_icp* orc_icp_create(const orc_config* cfg)
{
    orc_icp* s = (orc_icp*)calloc(1, sizeof *s);
    s->cfg = *cfg;
    if (s->cfg.knn < 1) s->cfg.knn = 1;
    if (s->cfg.nthreads < 1) s->cfg.nthreads = 1;
    return s;
}

void orc_icp_destroy(orc_icp* s)
{
    if (!s) return;
    free(s->map4); free(s->normals3); free(s->scalar); free(s->read_noise); free(s->read_scalar); orc_kdtree_free(s->tree); free(s);
}

void orc_icp_set_map_scalar(orc_icp* s, const float* scalar)
{
    free(s->scalar); s->scalar = NULL;
    if (!scalar || s->m <= 0) return;
    s->scalar = (float*)malloc((size_t)s->m * sizeof(float));
    memcpy(s->scalar, scalar, (size_t)s->m * sizeof(float));
}

void orc_icp_set_reading_scalar(orc_icp* s, const float* scalar, int64_t n)
{
    free(s->read_scalar); s->read_scalar = NULL; s->read_scalar_n = 0;
    if (!scalar || n <= 0) return;
    s->read_scalar = (float*)malloc((size_t)n * sizeof(float));
    memcpy(s->read_scalar, scalar, (size_t)n * sizeof(float));
    s->read_scalar_n = n;
}

void orc_icp_set_reading_noise(orc_icp* s, const float* noise, int64_t n)
{
    free(s->read_noise); s->read_noise = NULL; s->read_noise_n = 0;
    if (!noise || n <= 0) return;
    s->read_noise = (float*)malloc((size_t)n * sizeof(float));
    memcpy(s->read_noise, noise, (size_t)n * sizeof(float));
    s->read_noise_n = n;
}
int orc_icp_has_map(const orc_icp* s) { return s->m > 0; }
void orc_icp_get_mean(const orc_icp* s, float* mean3) { memcpy(mean3, s->mean, 3 * sizeof(float)); }

int orc_icp_set_map(orc_icp* s, const float* map4, int64_t m, const float* normals3)
{
    if (m <= 0) return 0; /* ICPSequence::setMap rejects an empty cloud, state unchanged */
    free(s->map4); free(s->normals3); free(s->scalar); orc_kdtree_free(s->tree);
    s->normals3 = NULL; s->scalar = NULL;
    s->m = m;
    double sum[3] = { 0, 0, 0 };
    for (int64_t i = 0; i < m; ++i) for (int r = 0; r < 3; ++r) sum[r] += map4[4 * i + r];
    for (int r = 0; r < 3; ++r) s->mean[r] = (float)(sum[r] / (double)m);
    s->map4 = (float*)malloc((size_t)m * 4 * sizeof(float));
    for (int64_t i = 0; i < m; ++i) {
        for (int r = 0; r < 3; ++r) s->map4[4 * i + r] = map4[4 * i + r] - s->mean[r];
        s->map4[4 * i + 3] = map4[4 * i + 3];
    }
    if (normals3) {
        s->normals3 = (float*)malloc((size_t)m * 3 * sizeof(float));
        memcpy(s->normals3, normals3, (size_t)m * 3 * sizeof(float));
    }
    /* KDTreeMatcher::init: bucketSize = knn if knn > 1 else libnabo's default 8 */
    s->tree = orc_kdtree_build(s->map4, m, 3, s->cfg.knn > 1 ? s->cfg.knn : 8);
    return 1;
}

int orc_icp_register(orc_icp* s, const float* scan4, int64_t n, const float* scan_normals3,
                     float* T_out, orc_stats* st)
{
    orc_stats local; if (!st) st = &local;
    memset(st, 0, sizeof *st);
    st->sensor_noise_overlap = -1.f;
    mat4_identity(T_out);
    const int64_t noise_n = s->read_noise_n; s->read_noise_n = 0; /* one shot */
    const int64_t rscalar_n = s->read_scalar_n; s->read_scalar_n = 0;
    g_read_scalar = rscalar_n == n ? s->read_scalar : NULL;
    if (!orc_icp_has_map(s)) return ORC_OK; /* "Ignoring attempt to perform ICP with an empty map" */
    const orc_config* cfg = &s->cfg;
    const int k = cfg->knn;
    if (cfg->minimizer == ORC_MIN_POINT_TO_PLANE && !s->normals3) return ORC_ERR_ARG;

    float Tmean[16], Tmean_inv[16];
    mat4_identity(Tmean); mat4_identity(Tmean_inv);
    for (int r = 0; r < 3; ++r) { Tmean[12 + r] = s->mean[r]; Tmean_inv[12 + r] = -s->mean[r]; }

    float* reading = (float*)malloc((size_t)(n > 0 ? n : 1) * 4 * sizeof(float));
    float* step = (float*)malloc((size_t)(n > 0 ? n : 1) * 4 * sizeof(float));
    float* step_normals = scan_normals3 ? (float*)malloc((size_t)(n > 0 ? n : 1) * 3 * sizeof(float)) : NULL;
    int32_t* ids = (int32_t*)malloc((size_t)(n > 0 ? n : 1) * k * sizeof(int32_t));
    float* d2 = (float*)malloc((size_t)(n > 0 ? n : 1) * k * sizeof(float));
    float* w = (float*)malloc((size_t)(n > 0 ? n : 1) * k * sizeof(float));
    orc_transform(Tmean_inv, scan4, reading, n);

    float T_iter[16]; mat4_identity(T_iter);
    /* checkers.init(T_iter) */
    int counter = 0;
    const int SL = cfg->smooth_length > 0 ? cfg->smooth_length : 3;
    int hist_n = 0, hist_cap = cfg->max_iterations + 2;
    if (hist_cap < 8) hist_cap = 8;
    double* hq = (double*)malloc((size_t)hist_cap * 4 * sizeof(double));
    double* ht = (double*)malloc((size_t)hist_cap * 3 * sizeof(double));
    quat_from_R(T_iter, hq); ht[0] = ht[1] = ht[2] = 0; hist_n = 1;
    float init_t[3] = { 0, 0, 0 }; double init_q[4]; memcpy(init_q, hq, sizeof init_q);

    int err = ORC_OK, iterate = 1;
    float robust_scale = 1.f; /* RobustOutlierFilter::scale, kept between iterations */
    const double t0 = now_s();
    while (iterate) {
        orc_transform(T_iter, reading, step, n);
        if (step_normals) orc_rotate3(T_iter, scan_normals3, step_normals, n);
        const double tk = now_s();
        orc_kdtree_knn(s->tree, step, n, k, cfg->max_dist, 1, ids, d2, cfg->nthreads);
        st->seconds_knn += now_s() - tk;
        err = orc_outlier_weights_ex(cfg, d2, ids, k, n, step_normals, s->normals3, s->scalar, step, s->map4, st->iterations + 1,
                                     &robust_scale, w, &st->trimmed_limit);
        if (err) break;
        float T_step[16];
        err = orc_minimize_ex(cfg->minimizer, cfg->is_2d ? (2 | 4) : (cfg->force_2d ? 2 : (cfg->force_4dof ? 1 : 0)), step, n, s->map4, s->normals3, ids, d2, w, k, T_step, NULL, NULL, NULL, st);
        if (err) break;
        mat4_mul(T_step, T_iter, T_iter);
        ++st->iterations;
        /* Counter */
        ++counter;
        if (counter >= cfg->max_iterations) { iterate = 0; st->stop_reason = ORC_STOP_COUNTER; }
        /* Differential */
        if (cfg->use_differential) {
            if (hist_n == hist_cap) {
                hist_cap *= 2;
                hq = (double*)realloc(hq, (size_t)hist_cap * 4 * sizeof(double));
                ht = (double*)realloc(ht, (size_t)hist_cap * 3 * sizeof(double));
            }
            quat_from_R(T_iter, hq + 4 * hist_n);
            for (int r = 0; r < 3; ++r) ht[3 * hist_n + r] = T_iter[12 + r];
            ++hist_n;
            if (hist_n > SL) {
                double rot = 0, tr = 0;
                for (int i = hist_n - 1; i >= hist_n - SL; --i) {
                    rot += fabs(quat_angular_distance(hq + 4 * i, hq + 4 * (i - 1)));
                    const double dx = ht[3 * i] - ht[3 * (i - 1)], dy = ht[3 * i + 1] - ht[3 * (i - 1) + 1],
                                 dz = ht[3 * i + 2] - ht[3 * (i - 1) + 2];
                    tr += sqrt(dx * dx + dy * dy + dz * dz);
                }
                rot /= SL; tr /= SL;
                if (rot != rot || tr != tr) { err = ORC_ERR_NAN; break; }
                if (rot < cfg->min_diff_rot && tr < cfg->min_diff_trans) {
                    if (iterate) st->stop_reason = ORC_STOP_DIFFERENTIAL;
                    iterate = 0;
                }
            }
        }
        /* Bound */
        if (cfg->use_bound) {
            double q[4]; quat_from_R(T_iter, q);
            const double rot = fabs(quat_angular_distance(q, init_q));
            const double dx = T_iter[12] - init_t[0], dy = T_iter[13] - init_t[1], dz = T_iter[14] - init_t[2];
            if (rot > cfg->max_rot_norm || sqrt(dx * dx + dy * dy + dz * dz) > cfg->max_trans_norm) { err = ORC_ERR_BOUND; break; }
        }
    }
    st->seconds_total = now_s() - t0;
    st->error = err;
    /* ErrorMinimizer::getOverlap() (read at Mapper.cpp:219) when the reading carries `simpleSensorNoise` and `normals` (SURVEY B.6,
     * upstream as recalled): over the LAST iteration's error elements (w != 0, finite match; `step`, `ids`, `w` still hold them) --
     * PointToPoint: |p - q| < mean|p - q| + noise_i; PointToPlane: |(p - q) . n_i / |n_i|| < noise_i, n_i the reading's normal */
    /* (PointToPointErrorMinimizer::getOverlap() asks for `simpleSensorNoise` alone; only the point-to-plane variant also for `normals`) */
    if (!err && noise_n == n && n > 0 && st->iterations > 0 && !cfg->is_2d &&
        (cfg->minimizer == ORC_MIN_POINT_TO_POINT || (cfg->minimizer == ORC_MIN_POINT_TO_PLANE && step_normals))) {
        double pairs = 0.0, sum = 0.0;
        for (int pass = 0; pass < 2; ++pass) {
            const float mean = pairs > 0.0 ? (float)(sum / pairs) : 0.f;
            double count = 0.0;
            for (int64_t i = 0; i < n; ++i)
                for (int j = 0; j < k; ++j) {
                    const int64_t e = (int64_t)k * i + j;
                    if (!(d2[e] < INFINITY) || w[e] == 0.f || ids[e] < 0) continue;
                    const float* p = step + 4 * i; const float* q = s->map4 + 4 * (int64_t)ids[e];
                    const float dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2];
                    if (cfg->minimizer == ORC_MIN_POINT_TO_POINT) {
                        const float dist = sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
                        if (pass == 0) { pairs += 1.0; sum += (double)dist; }
                        else if (dist < mean + s->read_noise[i]) count += 1.0;
                    } else {
                        const float* nr = step_normals + 3 * i;
                        const float nn = sqrtf(fmaf(nr[2], nr[2], fmaf(nr[1], nr[1], nr[0] * nr[0])));
                        const float proj = fmaf(dz, nr[2] / nn, fmaf(dy, nr[1] / nn, dx * (nr[0] / nn)));
                        if (pass == 0) pairs += 1.0;
                        else if (fabsf(proj) < s->read_noise[i]) count += 1.0;
                    }
                }
            if (pass == 1 && pairs > 0.0) st->sensor_noise_overlap = (float)(count / pairs);
        }
    }
    if (!err) {
        float tmp[16];
        mat4_mul(T_iter, Tmean_inv, tmp);
        mat4_mul(Tmean, tmp, T_out);
    }
    free(reading); free(step); free(step_normals); free(ids); free(d2); free(w); free(hq); free(ht);
    return err;
}

/* ------------------------------------------------------------------------------------------------
 * SurfaceNormalDataPointsFilter (applied at Map.cpp:524 from examples/config.yaml:26)
 * ---------------------------------------------------------------------------------------------- */
void orc_surface_normals(const float* pts4, int64_t m, int knn, float* normals3, int nthreads) { orc_surface_normals_ex(pts4, m, knn, normals3, NULL, nthreads); }
/* densities (may be NULL): keepDensities -- knn / (4/3 pi r^3), r = the largest distance of a neighbour from the centroid of the
 * neighbourhood (SurfaceNormalDataPointsFilter::computeDensity: NN.colwise().norm().maxCoeff() on the centred neighbours) */
static void orc_surface_normals_impl(const float* pts4, int64_t m, int knn, float* normals3, float* densities, int nthreads, int dim2);
/* keepMatchedIds / keepMeanDist of the filter (SurfaceNormal.cpp as recalled): the ids of the knn neighbours of every point (self included,
 * ascending distance), and the distance from the point to the mean of its neighbours.  Set before a call, consumed by it. */
static int32_t* g_sn_ids_out = NULL; static float* g_sn_meandist_out = NULL;
void orc_surface_normals_extras(int32_t* matched_ids, float* mean_dist) { g_sn_ids_out = matched_ids; g_sn_meandist_out = mean_dist; }
/* keepEigenValues / keepEigenVectors with sortEigen: 1 (SurfaceNormal.cpp as recalled): the eigenvalues of C = NN NN^T (the centred
 * neighbours' scatter matrix, NOT divided by the count) in ascending order, and serializeEigVec of the eigenvector matrix whose columns
 * follow that order: entry 3 k + j = component k of eigenvector j.  A neighbourhood of rank < 2 gets upstream's degenerate answer
 * (eigenvalues 0, eigenvectors identity).  The sign of an eigenvector is the Jacobi iteration's (upstream: Eigen's) -- compare up to sign.
 * Set before a call, consumed by it. */
static float* g_sn_eigval_out = NULL; static float* g_sn_eigvec_out = NULL;
void orc_surface_normals_eigen(float* eig_values3, float* eig_vectors9) { g_sn_eigval_out = eig_values3; g_sn_eigvec_out = eig_vectors9; }
static void orc_sn_store_eigen(int64_t i, const double* w, const double* Q, int degenerate)
{
    int o[3] = { 0, 1, 2 };
    if (!degenerate) { /* ascending, stable: the exchange sort (0,1) (0,2) (1,2) with strict comparisons */
        int t;
        if (w[o[1]] < w[o[0]]) { t = o[0]; o[0] = o[1]; o[1] = t; }
        if (w[o[2]] < w[o[0]]) { t = o[0]; o[0] = o[2]; o[2] = t; }
        if (w[o[2]] < w[o[1]]) { t = o[1]; o[1] = o[2]; o[2] = t; }
    }
    for (int j = 0; j < 3; ++j) {
        if (g_sn_eigval_out) g_sn_eigval_out[3 * i + j] = degenerate ? 0.f : (float)w[o[j]];
        if (g_sn_eigvec_out) for (int k = 0; k < 3; ++k) g_sn_eigvec_out[9 * i + 3 * k + j] = degenerate ? (k == j ? 1.f : 0.f) : (float)Q[3 * o[j] + k];
    }
}
void orc_surface_normals_ex(const float* pts4, int64_t m, int knn, float* normals3, float* densities, int nthreads)
{
    orc_surface_normals_impl(pts4, m, knn, normals3, densities, nthreads, 0);
}
/* 2-D clouds (features 3 x N upstream; here z == 0): SurfaceNormalDataPointsFilter diagonalises the 2 x 2 covariance; needs
 * rank + 1 >= featDim - 1 = 2, i.e. rank >= 1; the normal is the eigenvector of the smaller eigenvalue, in the plane */
void orc_surface_normals_2d(const float* pts4, int64_t m, int knn, float* normals3, int nthreads)
{
    orc_surface_normals_impl(pts4, m, knn, normals3, NULL, nthreads, 1);
}
static void orc_surface_normals_impl(const float* pts4, int64_t m, int knn, float* normals3, float* densities, int nthreads, int dim2)
{
    orc_kdtree* t = orc_kdtree_build(pts4, m, 3, knn > 1 ? knn : 8);
    int32_t* ids = (int32_t*)malloc((size_t)m * knn * sizeof(int32_t));
    float* d2 = (float*)malloc((size_t)m * knn * sizeof(float));
    orc_kdtree_knn(t, pts4, m, knn, INFINITY, 1, ids, d2, nthreads);
    for (int64_t i = 0; i < m; ++i) {
        double mean[3] = { 0, 0, 0 };
        int real = 0;
        for (int j = 0; j < knn; ++j) {
            const int32_t id = ids[(int64_t)knn * i + j];
            if (id < 0) continue;
            ++real;
            for (int r = 0; r < 3; ++r) mean[r] += pts4[4 * (int64_t)id + r];
        }
        double C[9] = { 0 };
        for (int r = 0; r < 3; ++r) mean[r] /= (real > 0 ? real : 1);
        double rmax2 = 0;
        for (int j = 0; j < knn; ++j) {
            const int32_t id = ids[(int64_t)knn * i + j];
            if (id < 0) continue;
            double v[3];
            for (int r = 0; r < 3; ++r) v[r] = (double)pts4[4 * (int64_t)id + r] - mean[r];
            for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) C[3 * c + r] += v[r] * v[c];
            const double r2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
            if (r2 > rmax2) rmax2 = r2;
        }
        if (densities) { const double rr = sqrt(rmax2); densities[i] = (float)((double)real / ((4.0 / 3.0) * 3.14159265358979323846 * (rr * rr * rr))); }
        if (g_sn_ids_out) for (int j = 0; j < knn; ++j) g_sn_ids_out[(int64_t)knn * i + j] = ids[(int64_t)knn * i + j];
        if (g_sn_meandist_out) {
            double dd = 0;
            for (int r = 0; r < 3; ++r) { const double v = (double)pts4[4 * i + r] - mean[r]; dd += v * v; }
            g_sn_meandist_out[i] = (float)sqrt(dd);
        }
        double w[3], Q[9];
        jacobi_eig_sym(3, C, w, Q);
        if (dim2) {
            /* z == 0: the (0,2) and (1,2) rotations of the Jacobi sweep see zero off-diagonals, w[0], w[1] are the plane's pair */
            const double wm2 = fmax(fabs(w[0]), fabs(w[1]));
            int rank2 = 0;
            for (int e = 0; e < 2; ++e) if (fabs(w[e]) > 2.0 * FLT_EPSILON * wm2 && wm2 > 0) ++rank2;
            if (rank2 < 1) { normals3[3 * i] = 1.f; normals3[3 * i + 1] = 0.f; normals3[3 * i + 2] = 0.f; continue; }
            const int e2 = w[1] < w[0] ? 1 : 0;
            normals3[3 * i] = (float)Q[3 * e2]; normals3[3 * i + 1] = (float)Q[3 * e2 + 1]; normals3[3 * i + 2] = 0.f;
            continue;
        }
        /* rank test of upstream: needs rank >= 2, otherwise eigenvalues 0 / eigenvectors identity */
        double wmax = fmax(fabs(w[0]), fmax(fabs(w[1]), fabs(w[2])));
        int rank = 0;
        for (int e = 0; e < 3; ++e) if (fabs(w[e]) > 3.0 * FLT_EPSILON * wmax && wmax > 0) ++rank;
        if (g_sn_eigval_out || g_sn_eigvec_out) orc_sn_store_eigen(i, w, Q, rank < 2);
        if (rank < 2) { normals3[3 * i] = 1.f; normals3[3 * i + 1] = 0.f; normals3[3 * i + 2] = 0.f; continue; }
        int e = 0;
        if (w[1] < w[e]) e = 1;
        if (w[2] < w[e]) e = 2;
        for (int r = 0; r < 3; ++r) normals3[3 * i + r] = (float)Q[3 * e + r];
    }
    free(ids); free(d2); orc_kdtree_free(t);
    g_sn_ids_out = NULL; g_sn_meandist_out = NULL; g_sn_eigval_out = NULL; g_sn_eigvec_out = NULL;
}

/* PointDistanceMapperModule::inPlaceUpdateMap keep mask (PointDistanceMapperModule.cpp:28-50):
 * kd-tree on the MAP, k = 1, epsilon 0, optionFlags 0 (self match not allowed), no radius;
 * keep iff dists(i) >= minDistNewPoint^2 (an unfilled slot is +inf => kept). */
void orc_point_distance_keep(const float* map4, int64_t m, const float* in4, int64_t n, float min_dist,
                             uint8_t* keep, int nthreads)
{
    orc_kdtree* t = orc_kdtree_build(map4, m, 3, 8);
    int32_t* ids = (int32_t*)malloc((size_t)(n > 0 ? n : 1) * sizeof(int32_t));
    float* d2 = (float*)malloc((size_t)(n > 0 ? n : 1) * sizeof(float));
    orc_kdtree_knn(t, in4, n, 1, INFINITY, 0, ids, d2, nthreads);
    const double lim = pow((double)min_dist, 2.0); /* std::pow(float, int) promotes to double; dists(i) is promoted for the comparison (PointDistanceMapperModule.cpp:42) */
    for (int64_t i = 0; i < n; ++i) keep[i] = (double)d2[i] >= lim;
    free(ids); free(d2); orc_kdtree_free(t);
}

/* Map::unloadCells binning (Map.cpp:206-209) with toGridCoordinate (Map.cpp:232-235) */
void orc_cell_ids(const float* pts4, int64_t n, float cell_size, int32_t* ijk3)
{
    for (int64_t i = 0; i < n; ++i)
        for (int r = 0; r < 3; ++r) ijk3[3 * i + r] = (int32_t)floorf(pts4[4 * i + r] / cell_size);
}

/* OctreeMapperModule decimation (OctreeMapperModule.cpp:35-39 -> OctreeGridDataPointsFilter, samplingMethod 0)
 * as a lattice: voxel floor((p - lo) / edge) per axis (21 bits each), lo = bounding-box minimum; the
 * representative of a voxel is its first point.  Sort-based on purpose (the device side hashes). */
typedef struct { uint64_t key; int64_t idx; } orc_vox_item;
static int orc_vox_cmp(const void* a, const void* b)
{
    const orc_vox_item* x = (const orc_vox_item*)a; const orc_vox_item* y = (const orc_vox_item*)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
}
static uint32_t orc_fmix32(uint32_t h) { h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h; }
/* method 0: the smallest index represents its voxel; method 1 (`samplingMethod: 1`, a random point, made
 * reproducible): the index with the smallest MurmurHash3 finaliser (a bijection of the 32-bit integers). */
void orc_voxel_keep(const float* in4, int64_t n, float edge, int method, uint8_t* keep)
{
    if (n <= 0) return;
    float lo[3] = {in4[0], in4[1], in4[2]};
    for (int64_t i = 1; i < n; ++i)
        for (int r = 0; r < 3; ++r) if (in4[4 * i + r] < lo[r]) lo[r] = in4[4 * i + r];
    orc_vox_item* it = (orc_vox_item*)malloc((size_t)n * sizeof(orc_vox_item));
    for (int64_t i = 0; i < n; ++i) {
        uint64_t key = 0;
        for (int r = 0; r < 3; ++r) {
            float v = floorf((in4[4 * i + r] - lo[r]) / edge);
            if (v > 2097151.0f) v = 2097151.0f;
            key = key * 2097152ull + (uint64_t)v;
        }
        it[i].key = key; it[i].idx = method ? (int64_t)orc_fmix32((uint32_t)i) : i;
        keep[i] = 0;
    }
    qsort(it, (size_t)n, sizeof(orc_vox_item), orc_vox_cmp);
    if (method) { /* undo the bijection: fmix32^-1 is not needed, look the index up again */
        orc_vox_item* h2 = (orc_vox_item*)malloc((size_t)n * sizeof(orc_vox_item));
        for (int64_t i = 0; i < n; ++i) { h2[i].key = (uint64_t)orc_fmix32((uint32_t)i); h2[i].idx = i; }
        qsort(h2, (size_t)n, sizeof(orc_vox_item), orc_vox_cmp);
        for (int64_t i = 0; i < n; ++i)
            if (i == 0 || it[i].key != it[i - 1].key) {
                /* binary search of the hash among the sorted hashes */
                int64_t lo2 = 0, hi2 = n - 1; const uint64_t want = (uint64_t)it[i].idx;
                while (lo2 < hi2) { const int64_t mid = (lo2 + hi2) / 2; if (h2[mid].key < want) lo2 = mid + 1; else hi2 = mid; }
                keep[h2[lo2].idx] = 1;
            }
        free(h2);
    } else
        for (int64_t i = 0; i < n; ++i)
            if (i == 0 || it[i].key != it[i - 1].key) keep[it[i].idx] = 1;
    free(it);
}
void orc_voxel_keep_first(const float* in4, int64_t n, float edge, uint8_t* keep) { orc_voxel_keep(in4, n, edge, 0, keep); }

/* ------------------------------------------------------------------------------------------------
 * OctreeGridDataPointsFilter behind OctreeMapperModule (OctreeMapperModule.cpp:8-12,35-39) -- the octree
 * itself, restated RECURSIVELY like upstream's Octree_<T,3>::build (libpointmatcher 1.4.x octree.hpp, as recalled):
 *   root    = the bounding CUBE of the cloud: centre = (min + max) / 2 per axis, radius = max over the axes of (max - centre);
 *   build   : a node is a leaf iff 2 * radius <= maxSizeByNode or it holds <= maxPointByNode points (or depth 21 is reached:
 *             the device side packs the path into 63 bits); otherwise its points go to 8 children, child index bit r set iff
 *             p_r > centre_r, child centre = centre +- radius / 2 (float adds), child radius = radius / 2; every child list
 *             keeps the order of the parent's list (so "first" is the smallest original index);
 *   sample  : leaves are visited depth first, children in index order 0..7; every non-empty leaf yields ONE point:
 *             samplingMethod 0 the first of its list, 1 a random one -- made reproducible: the smallest fmix32(original index);
 *   output  : the representatives in visiting order (upstream compacts the cloud in that order: the map comes out
 *             Morton-ordered, which decides who is "first" the next time two old points share a leaf).
 * order_out receives the original indices of the output points; returns their number.
 * ---------------------------------------------------------------------------------------------- */
typedef struct { const float* p; float max_size; int64_t max_pts; int method; int32_t* out; int64_t n_out; } orc_oct_ctx;
static void orc_oct_build(orc_oct_ctx* c, int32_t* idx, int64_t cnt, float cx, float cy, float cz, float radius, int depth)
{
    if (cnt == 0) return;
    if (radius * 2.f <= c->max_size || cnt <= c->max_pts || depth >= 21) {
        int32_t rep = idx[0];
        if (c->method == 1)
            for (int64_t i = 1; i < cnt; ++i) if (orc_fmix32((uint32_t)idx[i]) < orc_fmix32((uint32_t)rep)) rep = idx[i];
        c->out[c->n_out++] = rep;
        return;
    }
    int64_t n8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint8_t* oct = (uint8_t*)malloc((size_t)cnt);
    for (int64_t i = 0; i < cnt; ++i) {
        const float* q = c->p + 4 * (int64_t)idx[i];
        const int o = (q[0] > cx ? 1 : 0) | (q[1] > cy ? 2 : 0) | (q[2] > cz ? 4 : 0);
        oct[i] = (uint8_t)o; ++n8[o];
    }
    int32_t* buf = (int32_t*)malloc((size_t)cnt * sizeof(int32_t));
    int64_t start[8], fill[8], acc = 0;
    for (int o = 0; o < 8; ++o) { start[o] = fill[o] = acc; acc += n8[o]; }
    for (int64_t i = 0; i < cnt; ++i) buf[fill[oct[i]]++] = idx[i]; /* stable: parent order kept */
    free(oct);
    const float half = radius * 0.5f;
    for (int o = 0; o < 8; ++o)
        orc_oct_build(c, buf + start[o], n8[o], cx + ((o & 1) ? half : -half), cy + ((o & 2) ? half : -half), cz + ((o & 4) ? half : -half),
                      half, depth + 1);
    free(buf);
}
int64_t orc_octree_sample(const float* in4, int64_t n, float max_size, int64_t max_pts, int method, int32_t* order_out)
{
    if (n <= 0) return 0;
    float lo[3] = {in4[0], in4[1], in4[2]}, hi[3] = {in4[0], in4[1], in4[2]};
    for (int64_t i = 1; i < n; ++i)
        for (int r = 0; r < 3; ++r) {
            const float v = in4[4 * i + r];
            if (v < lo[r]) lo[r] = v;
            if (v > hi[r]) hi[r] = v;
        }
    float c[3], radius = 0.f;
    /* upstream's Octree_::build: radii = max - min; centre = min + radii * 0.5; maxRadius = max(radii) * 0.5 -- NOT (min + max) / 2
     * and max(max - centre), which can differ in the last bit and flip a `p > centre` decision */
    for (int r = 0; r < 3; ++r) { const float ext = hi[r] - lo[r]; c[r] = lo[r] + ext * 0.5f; const float rr = ext * 0.5f; if (rr > radius) radius = rr; }
    int32_t* idx = (int32_t*)malloc((size_t)n * sizeof(int32_t));
    for (int64_t i = 0; i < n; ++i) idx[i] = (int32_t)i;
    orc_oct_ctx ctx = {in4, max_size, max_pts < 1 ? 1 : max_pts, method, order_out, 0};
    orc_oct_build(&ctx, idx, n, c[0], c[1], c[2], radius, 0);
    free(idx);
    return ctx.n_out;
}

/* DynamicPointsMapperModule::inPlaceUpdateMap (DynamicPointsMapperModule.cpp:34-172), brute-force angular
 * nearest beam (the reference: 2-D kd-tree, k = 1, radius 2 * beamHalfAngle, :75-78).  Ties on equal angular
 * distance go to the smallest beam index; asin / atan2 through double, rounded once (shared with the device). */
static void orc_xf(const float* T, const float* p, float* o)
{
    for (int r = 0; r < 3; ++r) o[r] = fmaf(T[12 + r], p[3], fmaf(T[8 + r], p[2], fmaf(T[4 + r], p[1], T[r] * p[0])));
}
void orc_dynamic_points_update(const float prm[7], const float* to_sensor, const float* in4, int64_t n, const float* map4,
                               const float* map_normals3, int64_t m, float* prob, int nthreads)
{
    const float thresholdDynamic = prm[0], alpha = prm[1], beta = prm[2], beamHalfAngle = prm[3], epsilonA = prm[4], epsilonD = prm[5],
                sensorMaxRange = prm[6];
    const float eps = 0.0001f;
    if (n <= 0 || m <= 0) return;
    float* bx = (float*)malloc((size_t)n * 4 * sizeof(float));
    float* be = (float*)malloc((size_t)n * sizeof(float));
    float* ba = (float*)malloc((size_t)n * sizeof(float));
    for (int64_t i = 0; i < n; ++i) {
        float o[3];
        orc_xf(to_sensor, in4 + 4 * i, o);
        const float radius = sqrtf(o[0] * o[0] + o[1] * o[1] + o[2] * o[2]);
        bx[4 * i] = o[0]; bx[4 * i + 1] = o[1]; bx[4 * i + 2] = o[2]; bx[4 * i + 3] = radius;
        be[i] = (float)asin((double)(o[2] / radius));
        ba[i] = (float)atan2((double)o[1], (double)o[0]);
    }
    const float cell = 2 * beamHalfAngle;
    const float r2 = cell * cell;
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) schedule(static)
    for (int64_t i = 0; i < m; ++i) {
        float mp[3];
        orc_xf(to_sensor, map4 + 4 * i, mp);
        const float mapNorm = sqrtf(mp[0] * mp[0] + mp[1] * mp[1] + mp[2] * mp[2]);
        if (!(mapNorm < sensorMaxRange)) continue;
        const float qe = (float)asin((double)(mp[2] / mapNorm));
        const float qa = (float)atan2((double)mp[1], (double)mp[0]);
        float bd = INFINITY;
        int64_t best = -1;
        for (int64_t b = 0; b < n; ++b) {
            const float d0 = qe - be[b], d1 = qa - ba[b];
            const float d = d0 * d0 + d1 * d1;
            if (d <= r2 && d < bd) { bd = d; best = b; }
        }
        if (best < 0) continue;
        const float* ip = bx + 4 * best;
        const float inputNorm = ip[3];
        const float dx = ip[0] - mp[0], dy = ip[1] - mp[1], dz = ip[2] - mp[2];
        const float delta = sqrtf(dx * dx + dy * dy + dz * dz);
        const float d_max = epsilonA * inputNorm;
        const float* n3 = map_normals3 + 3 * i;
        const float nx = fmaf(to_sensor[8], n3[2], fmaf(to_sensor[4], n3[1], to_sensor[0] * n3[0]));
        const float ny = fmaf(to_sensor[9], n3[2], fmaf(to_sensor[5], n3[1], to_sensor[1] * n3[0]));
        const float nz = fmaf(to_sensor[10], n3[2], fmaf(to_sensor[6], n3[1], to_sensor[2] * n3[0]));
        const float ndot = (nx * mp[0] + ny * mp[1] + nz * mp[2]) / mapNorm;
        const float w_v = (float)(eps + (1. - eps) * fabs((double)ndot));
        const float w_d1 = (float)(eps + (1. - eps) * (1. - sqrtf(bd) / (2 * beamHalfAngle)));
        const float offset = delta - epsilonD;
        float w_d2 = 1.f;
        if (delta < epsilonD || mapNorm > inputNorm) w_d2 = eps;
        else if (offset < d_max) w_d2 = eps + (1 - eps) * offset / d_max;
        float w_p2 = eps;
        if (delta < epsilonD) w_p2 = 1.f;
        else if (offset < d_max) w_p2 = (float)(eps + (1. - eps) * (1. - offset / d_max));
        if ((inputNorm + epsilonD + d_max) >= mapNorm) {
            const float lastDyn = prob[i];
            const float c1 = 1 - (w_v * w_d1);
            const float c2 = w_v * w_d1;
            float probDynamic, probStatic;
            if (lastDyn < thresholdDynamic) {
                probDynamic = c1 * lastDyn + c2 * w_d2 * ((1 - alpha) * (1 - lastDyn) + beta * lastDyn);
                probStatic = c1 * (1 - lastDyn) + c2 * w_p2 * (alpha * (1 - lastDyn) + (1 - beta) * lastDyn);
            } else {
                probDynamic = 1 - eps;
                probStatic = eps;
            }
            prob[i] = probDynamic / (probDynamic + probStatic);
        }
    }
    free(bx); free(be); free(ba);
}

/* ------------------------------------------------------------------------------------------------
 * The DataPointsFilters of PM::ICPSequence::setDefault (Mapper.cpp:77) and MaxDensity [UPSTREAM 1.4.x, as
 * recalled].  Random numbers: std::minstd_rand (x <- 48271 x mod 2^31 - 1, defined by the C++ standard); "direct" = x / float
 * (max - min), "uniform" = std::uniform_real_distribution<float> as libstdc++ evaluates it.
 * ---------------------------------------------------------------------------------------------- */
typedef struct { uint32_t x; } orc_minstd;
static void orc_minstd_seed(orc_minstd* g, uint32_t seed) { g->x = seed % 2147483647u; if (g->x == 0) g->x = 1; }
/* test hook: the raw n-th value (n >= 1) of the stream -- [rand.predef] fixes the 10 000th of seed 1 at 399268537 (tests/test_pins.py) */
uint32_t orc_minstd_nth(uint32_t seed, uint32_t n)
{
    orc_minstd g; orc_minstd_seed(&g, seed);
    for (uint32_t i = 0; i < n; ++i) g.x = (uint32_t)(((uint64_t)g.x * 48271ull) % 2147483647ull);
    return g.x;
}
static float orc_minstd_unit(orc_minstd* g, int method)
{
    g->x = (uint32_t)(((uint64_t)g->x * 48271ull) % 2147483647ull);
    if (method == 1) { const float r = (float)(g->x - 1u) / 2147483646.0f; return r < 1.0f ? r : nextafterf(1.0f, 0.0f); }
    return (float)g->x / 2147483645.0f;
}
/* RandomSamplingDataPointsFilter{prob, randomSamplingMethod, seed >= 0}: a fresh generator per call, one number per point, kept
 * iff number < prob, never more than floor(n prob) + 1 points */
void orc_random_sampling_keep(int64_t n, float prob, int method, int seed, uint8_t* keep)
{
    orc_minstd g; orc_minstd_seed(&g, (uint32_t)seed);
    const int64_t n_out = (int64_t)((float)n * prob);
    int64_t j = 0;
    for (int64_t i = 0; i < n; ++i) keep[i] = 0;
    for (int64_t i = 0; i < n && j <= n_out; ++i)
        if (orc_minstd_unit(&g, method) < prob) { keep[i] = 1; ++j; }
}
/* MaxDensityDataPointsFilter{maxDensity, seed}: a point in a region denser than maxDensity survives with probability maxDensity / density */
void orc_max_density_keep(const float* densities, int64_t n, float max_density, int seed, uint8_t* keep)
{
    orc_minstd g; orc_minstd_seed(&g, (uint32_t)seed);
    for (int64_t i = 0; i < n; ++i) {
        keep[i] = 1;
        if (densities[i] > max_density) keep[i] = orc_minstd_unit(&g, 0) < max_density / densities[i];
    }
}
/* SamplingSurfaceNormalDataPointsFilter{ratio, knn, samplingMethod 0, maxBoxDim, seed}: median splits of the widest dimension
 * (ties by index) until a box holds <= knn points; one normal per box (rank >= 2, else the box is dropped), its points --
 * in index order -- kept with probability ratio.  order_out / normals_out (capacity n / 3 n) = kept indices and their
 * normals in box order; returns the number kept.  The split is a full sort per node (the host shell uses nth_element: the
 * two halves are the same sets). */
typedef struct { const float* p; int dim; } orc_ssn_cmp_ctx;
static orc_ssn_cmp_ctx g_ssn_cmp; /* qsort has no context argument; the oracle is single threaded here */
static int orc_ssn_cmp(const void* a, const void* b)
{
    const int32_t ia = *(const int32_t*)a, ib = *(const int32_t*)b;
    const float x = g_ssn_cmp.p[4 * (int64_t)ia + g_ssn_cmp.dim], y = g_ssn_cmp.p[4 * (int64_t)ib + g_ssn_cmp.dim];
    if (x != y) return x < y ? -1 : 1;
    return ia < ib ? -1 : (ia > ib ? 1 : 0);
}
static int orc_i32_cmp(const void* a, const void* b) { const int32_t x = *(const int32_t*)a, y = *(const int32_t*)b; return x < y ? -1 : (x > y ? 1 : 0); }
typedef struct { const float* p; float ratio; int knn; float max_box; orc_minstd g; int32_t* order; float* normals; int64_t n_out;
                 /* samplingMethod 1 (orc_sampling_surface_normal_ex): one output per surviving box */
                 int method; float* mean; int32_t* mstart; int32_t* mcount; int32_t* members; int64_t n_members; } orc_ssn_ctx;
static void orc_ssn_fuse(orc_ssn_ctx* c, int32_t* idx, int64_t cnt)
{
    qsort(idx, (size_t)cnt, sizeof(int32_t), orc_i32_cmp);
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    double mean[3] = {0, 0, 0};
    for (int64_t k = 0; k < cnt; ++k)
        for (int r = 0; r < 3; ++r) {
            const float v = c->p[4 * (int64_t)idx[k] + r];
            if (v < lo[r]) lo[r] = v;
            if (v > hi[r]) hi[r] = v;
            mean[r] += v;
        }
    float box = hi[0] - lo[0];
    if (hi[1] - lo[1] > box) box = hi[1] - lo[1];
    if (hi[2] - lo[2] > box) box = hi[2] - lo[2];
    if (box > c->max_box) return;
    for (int r = 0; r < 3; ++r) mean[r] /= (double)cnt;
    double C[9] = {0};
    for (int64_t k = 0; k < cnt; ++k) {
        double v[3];
        for (int r = 0; r < 3; ++r) v[r] = (double)c->p[4 * (int64_t)idx[k] + r] - mean[r];
        for (int cc = 0; cc < 3; ++cc) for (int r = 0; r < 3; ++r) C[3 * cc + r] += v[r] * v[cc];
    }
    double w[3], Q[9];
    jacobi_eig_sym(3, C, w, Q);
    const double wmax = fmax(fabs(w[0]), fmax(fabs(w[1]), fabs(w[2])));
    int rank = 0;
    for (int e = 0; e < 3; ++e) if (wmax > 0 && fabs(w[e]) > 3.0 * FLT_EPSILON * wmax) ++rank;
    if (rank < 2) return;
    int e = 0;
    if (w[1] < w[e]) e = 1;
    if (w[2] < w[e]) e = 2;
    if (c->method == 1) {
        /* samplingMethod 1 [UPSTREAM, as recalled: fuseRange's second branch]: the box is replaced by ONE point -- the first of its list
         * (here: its smallest index) moved to the mean of the box; existing descriptors are averaged over the members by the caller
         * (averageExistingDescriptors), who gets the member list for it.  No random number is drawn. */
        const int64_t o = c->n_out;
        c->order[o] = idx[0];
        for (int r = 0; r < 3; ++r) { c->normals[3 * o + r] = (float)Q[3 * e + r]; if (c->mean) c->mean[3 * o + r] = (float)mean[r]; }
        if (c->mstart) c->mstart[o] = (int32_t)c->n_members;
        if (c->mcount) c->mcount[o] = (int32_t)cnt;
        if (c->members) for (int64_t k = 0; k < cnt; ++k) c->members[c->n_members + k] = idx[k];
        c->n_members += cnt;
        ++c->n_out;
        return;
    }
    for (int64_t k = 0; k < cnt; ++k)
        if (orc_minstd_unit(&c->g, 0) < c->ratio) {
            c->order[c->n_out] = idx[k];
            for (int r = 0; r < 3; ++r) c->normals[3 * c->n_out + r] = (float)Q[3 * e + r];
            ++c->n_out;
        }
}
static void orc_ssn_build(orc_ssn_ctx* c, int32_t* idx, int64_t cnt, const float* lo, const float* hi)
{
    if (cnt == 0) return;
    if (cnt <= c->knn) { orc_ssn_fuse(c, idx, cnt); return; }
    int dim = 0;
    for (int r = 1; r < 3; ++r) if (hi[r] - lo[r] > hi[dim] - lo[dim]) dim = r;
    const int64_t right = cnt / 2, left = cnt - right;
    g_ssn_cmp.p = c->p; g_ssn_cmp.dim = dim;
    qsort(idx, (size_t)cnt, sizeof(int32_t), orc_ssn_cmp);
    const float cut = c->p[4 * (int64_t)idx[left] + dim];
    float lhi[3] = {hi[0], hi[1], hi[2]}, rlo[3] = {lo[0], lo[1], lo[2]};
    lhi[dim] = cut; rlo[dim] = cut;
    orc_ssn_build(c, idx, left, lo, lhi);
    orc_ssn_build(c, idx + left, right, rlo, hi);
}
int64_t orc_sampling_surface_normal(const float* pts4, int64_t n, float ratio, int knn, float max_box_dim, int seed, int32_t* order_out,
                                    float* normals_out)
{
    if (n <= 0) return 0;
    int32_t* idx = (int32_t*)malloc((size_t)n * sizeof(int32_t));
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = 0; i < n; ++i) {
        idx[i] = (int32_t)i;
        for (int r = 0; r < 3; ++r) { const float v = pts4[4 * i + r]; if (v < lo[r]) lo[r] = v; if (v > hi[r]) hi[r] = v; }
    }
    orc_ssn_ctx c; c.p = pts4; c.ratio = ratio; c.knn = knn; c.max_box = max_box_dim; c.order = order_out; c.normals = normals_out; c.n_out = 0;
    c.method = 0; c.mean = NULL; c.mstart = NULL; c.mcount = NULL; c.members = NULL; c.n_members = 0;
    orc_minstd_seed(&c.g, (uint32_t)seed);
    orc_ssn_build(&c, idx, n, lo, hi);
    free(idx);
    return c.n_out;
}
/* ... with `samplingMethod`: 0 as above (the extra outputs are not touched); 1 = one point per surviving box: order_out[j] = the smallest
 * index of box j, mean3_out[3 j ..] its new position (the mean of the box, accumulated in double in index order), normals_out its normal,
 * members_out[mstart_out[j] .. + mcount_out[j]) the members in index order (what a caller averages the descriptors over).  Boxes in
 * depth-first order.  Any of mean3_out / mstart_out / mcount_out / members_out (capacity 3 n / n / n / n) may be NULL. */
int64_t orc_sampling_surface_normal_ex(const float* pts4, int64_t n, float ratio, int knn, float max_box_dim, int seed, int method,
                                       int32_t* order_out, float* normals_out, float* mean3_out, int32_t* mstart_out, int32_t* mcount_out,
                                       int32_t* members_out)
{
    if (n <= 0) return 0;
    int32_t* idx = (int32_t*)malloc((size_t)n * sizeof(int32_t));
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = 0; i < n; ++i) {
        idx[i] = (int32_t)i;
        for (int r = 0; r < 3; ++r) { const float v = pts4[4 * i + r]; if (v < lo[r]) lo[r] = v; if (v > hi[r]) hi[r] = v; }
    }
    orc_ssn_ctx c; c.p = pts4; c.ratio = ratio; c.knn = knn; c.max_box = max_box_dim; c.order = order_out; c.normals = normals_out; c.n_out = 0;
    c.method = method == 1 ? 1 : 0; c.mean = mean3_out; c.mstart = mstart_out; c.mcount = mcount_out; c.members = members_out; c.n_members = 0;
    orc_minstd_seed(&c.g, (uint32_t)seed);
    orc_ssn_build(&c, idx, n, lo, hi);
    free(idx);
    return c.n_out;
}

/* Mapper::applyInputFilters for DistanceLimit / BoundingBox filters (Mapper.cpp:27-31, examples/config.yaml:2-18):
 * filters = n_filters rows of 8 floats {type, i, f0..f5} (type 0 distance limit: i = dim, f0 = dist, f1 = removeInside;
 * type 1 bounding box: f0..2 = min, f3..5 = max, i = removeInside). keep[i] = 1 iff every filter keeps point i. */
void orc_filter_points(const float* in4, int64_t n, const float* filters, int n_filters, uint8_t* keep)
{
    for (int64_t i = 0; i < n; ++i) {
        const float* p = in4 + 4 * i;
        int ok = 1;
        for (int k = 0; k < n_filters; ++k) {
            const float* f = filters + 8 * k;
            if ((int)f[0] == 0) {
                const int dim = (int)f[1];
                const float v = dim < 0 ? sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]) : fabsf(p[dim]);
                const float ad = fabsf(f[2]);
                ok &= f[3] != 0.f ? v > ad : v < ad;
            } else {
                const int inside = p[0] > f[2] && p[0] < f[5] && p[1] > f[3] && p[1] < f[6] && p[2] > f[4] && p[2] < f[7];
                ok &= (int)f[1] ? !inside : inside;
            }
        }
        keep[i] = (uint8_t)ok;
    }
}
