// api.hip -- the extern "C" surface declared in include/icpmi.h (see that header for the reference
// interface each entry point replaces).
#include "common.h"
#include <cstring>
#include <new>
#include <dlfcn.h>
#include <mutex>
#include <cstdio>

static std::string g_create_error;

// ICPMI_ROCTX=1: every entry point that takes a handle runs inside a roctx range named after it (rocprofv3 --marker-trace
// then shows registrations, map updates and merges as ranges above their kernels).  The marker library is looked up at run
// time: nothing links against it, and without the variable the cost is one predictable branch per call.
namespace {
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Roctx()
    {
        const char* e = getenv("ICPMI_ROCTX");
        if (!e || !atoi(e)) return;
        // rocprofv3 --marker-trace follows the SDK's marker library; the roctracer one is the fall-back for older tools
        for (const char* name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "/opt/rocm/lib/librocprofiler-sdk-roctx.so.1",
                                 "libroctx64.so.4", "libroctx64.so", "/opt/rocm/lib/libroctx64.so.4"}) {
            void* lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (!lib) continue;
            push = (int (*)(const char*))dlsym(lib, "roctxRangePushA");
            pop = (int (*)())dlsym(lib, "roctxRangePop");
            if (push && pop) return;
            push = nullptr; pop = nullptr;
        }
    }
};
Roctx& roctx() { static Roctx r; return r; }
struct RoctxRange {
    bool on;
    explicit RoctxRange(const char* name) : on(roctx().push != nullptr) { if (on) roctx().push(name); }
    ~RoctxRange() { if (on) roctx().pop(); }
};
} // namespace

// locks the handle for the rest of the calling function (see icpmi_ctx::mu) and makes its device current
#define CHECK_H(h)                                                           \
    RoctxRange _roctx_range(__func__);                                       \
    if (!(h)) return ICPMI_ERR_INVALID_ARG;                                  \
    std::lock_guard<std::recursive_mutex> _handle_lock((h)->mu);             \
    do {                                                                     \
        hipError_t _e = hipSetDevice((h)->device);                           \
        if (_e != hipSuccess) {                                              \
            (h)->last_error = std::string("hipSetDevice: ") + hipGetErrorString(_e); \
            return ICPMI_ERR_HIP;                                            \
        }                                                                    \
        if ((h)->zero_pending && zero_state_if_pending(h) != ICPMI_OK) return ICPMI_ERR_HIP; \
    } while (0)

extern "C" {

int32_t icpmi_version(void) { return ICPMI_VERSION; }
#ifndef ICPMI_SRC_STAMP
#define ICPMI_SRC_STAMP "unstamped"
#endif
#define ICPMI_STR2(x) #x
#define ICPMI_STR(x) ICPMI_STR2(x)
const char* icpmi_build_info(void) { return "icpmi " ICPMI_STR(ICPMI_VERSION) " src:" ICPMI_SRC_STAMP; }

void icpmi_config_default(icpmi_config* cfg)
{
    memset(cfg, 0, sizeof *cfg);
    cfg->device = 0;
    cfg->knn = 1;
    cfg->max_dist = INFINITY;
    cfg->epsilon = 0.f;
    cfg->n_outlier = 0;
    cfg->minimizer = ICPMI_MIN_POINT_TO_PLANE;
    cfg->max_iterations = 40;
    cfg->use_differential = 0;
    cfg->min_diff_rot = 0.001f;
    cfg->min_diff_trans = 0.001f;
    cfg->smooth_length = 3;
    cfg->use_bound = 0;
    cfg->max_rot_norm = 1.f;
    cfg->max_trans_norm = 1.f;
    cfg->grid_cell = 0.f;
    cfg->use_graph = 1;
    cfg->profile = 0;
}

static icpmi_status validate_config(const icpmi_config* cfg, std::string& err)
{
    if (cfg->knn < 1 || cfg->knn > ICPMI_MAX_K) { err = "InvalidParameter: knn must be in [1, 32]"; return ICPMI_ERR_INVALID_ARG; }
    if (!(cfg->max_dist > 0.f)) { err = "InvalidParameter: maxDist must be > 0"; return ICPMI_ERR_INVALID_ARG; }
    if (cfg->epsilon < 0.f) { err = "InvalidParameter: epsilon must be >= 0"; return ICPMI_ERR_INVALID_ARG; }
    if (cfg->n_outlier < 0 || cfg->n_outlier > ICPMI_MAX_OUTLIER) { err = "InvalidParameter: at most 8 outlier filters"; return ICPMI_ERR_INVALID_ARG; }
    for (int f = 0; f < cfg->n_outlier; ++f) {
        const int t = cfg->outlier[f].type;
        const float p = cfg->outlier[f].param;
        if (t < ICPMI_OUT_MAXDIST || t > ICPMI_OUT_VARTRIMMEDDIST) { err = "InvalidParameter: unknown outlier filter type"; return ICPMI_ERR_INVALID_ARG; }
        if (t == ICPMI_OUT_VARTRIMMEDDIST) {
            const float lo = cfg->outlier[f].param, hi = cfg->outlier[f].param2, lam = cfg->outlier[f].param3;
            if (!(lo > 0.f && lo < 1.f) || !(hi > 0.f && hi < 1.f) || !(lo <= hi) || !(lam >= 0.f)) { // upstream: ratios in ]0, 1[, lambda >= 0
                err = "InvalidParameter: VarTrimmedDist needs 0 < minRatio <= maxRatio < 1 and lambda >= 0"; return ICPMI_ERR_INVALID_ARG;
            }
        }
        if (t == ICPMI_OUT_GENERICDESCRIPTOR) {
            const int ip = cfg->outlier[f].iparam;
            if (ip & ~7) { err = "InvalidParameter: GenericDescriptorOutlierFilter: unknown flag"; return ICPMI_ERR_INVALID_ARG; }
        }
        if (t == ICPMI_OUT_ROBUST) {
            const int ip = cfg->outlier[f].iparam;
            if ((ip & 15) > ICPMI_ROB_STUDENT || ((ip >> 4) & 15) > ICPMI_SCALE_STD || ((ip >> 8) & 15) > ICPMI_DIST_POINT2PLANE || (ip >> 12)) {
                err = "InvalidParameter: RobustOutlierFilter: unknown robustFct / scaleEstimator / distanceType"; return ICPMI_ERR_INVALID_ARG;
            }
            if (!(cfg->outlier[f].param2 >= 0.f)) { err = "InvalidParameter: RobustOutlierFilter: nbIterationForScale must be >= 0"; return ICPMI_ERR_INVALID_ARG; }
            if (!(cfg->outlier[f].param3 >= 0.f)) { err = "InvalidParameter: RobustOutlierFilter: approximation must be >= 0 (0 or +inf: none)"; return ICPMI_ERR_INVALID_ARG; }
        }
        if (t == ICPMI_OUT_TRIMMEDDIST && !(p >= 0.f && p <= 1.f)) { err = "InvalidParameter: TrimmedDist ratio must be in [0, 1]"; return ICPMI_ERR_INVALID_ARG; }
        if ((t == ICPMI_OUT_MAXDIST || t == ICPMI_OUT_MINDIST || t == ICPMI_OUT_MEDIANDIST) && !(p >= 0.f)) { err = "InvalidParameter: negative outlier filter parameter"; return ICPMI_ERR_INVALID_ARG; }
    }
    if (cfg->minimizer < ICPMI_MIN_IDENTITY || cfg->minimizer > ICPMI_MIN_POINT_TO_PLANE) { err = "InvalidParameter: unknown error minimizer"; return ICPMI_ERR_INVALID_ARG; }
    if (cfg->force_4dof && cfg->force_2d) { err = "InvalidParameter: force2D and force4DOF exclude each other"; return ICPMI_ERR_INVALID_ARG; }
    if (cfg->max_iterations < 1) { err = "InvalidParameter: maxIterationCount must be >= 1"; return ICPMI_ERR_INVALID_ARG; }
    if (cfg->use_differential && (cfg->smooth_length < 1 || cfg->smooth_length > ICPMI_MAX_SMOOTH)) { err = "InvalidParameter: smoothLength must be in [1, 16]"; return ICPMI_ERR_INVALID_ARG; }
    if (cfg->grid_cell < 0.f) { err = "InvalidParameter: grid_cell must be >= 0"; return ICPMI_ERR_INVALID_ARG; }
    return ICPMI_OK;
}

} // extern "C"

// a handle without the allocation cache's handle count: what icpmi_create wraps, and what the private handles of the map-side operators are made of
icpmi_status create_handle(const icpmi_config* cfg, icpmi_handle* out)
{
    if (!cfg || !out) { g_create_error = "icpmi_create: null argument"; return ICPMI_ERR_INVALID_ARG; }
    *out = nullptr;
    icpmi_status vs = validate_config(cfg, g_create_error);
    if (vs != ICPMI_OK) return vs;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        g_create_error = std::string("icpmi_create: no HIP device available (") + hipGetErrorString(e) + ")";
        return ICPMI_ERR_HIP;
    }
    if (cfg->device < 0 || cfg->device >= ndev) { g_create_error = "icpmi_create: device ordinal out of range"; return ICPMI_ERR_INVALID_ARG; }
    icpmi_ctx* c = new (std::nothrow) icpmi_ctx();
    if (!c) { g_create_error = "icpmi_create: out of host memory"; return ICPMI_ERR_HIP; }
    c->cfg = *cfg;
    c->device = cfg->device;
#define CR(expr)                                                                                      \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess) {                                                                       \
            g_create_error = std::string(#expr) + ": " + hipGetErrorString(_e);                       \
            icpmi_destroy(c);                                                                         \
            return ICPMI_ERR_HIP;                                                                     \
        }                                                                                             \
    } while (0)
    CR(hipSetDevice(c->device));
    CR(stream_acquire(&c->stream));
    c->own_stream = true;
    CR(dev_malloc((void**)&c->d_state, sizeof(IcpState) * ICPMI_MAX_BATCH));
    c->zero_pending = true; // (common.h: zero_state_if_pending -- neither the legacy stream, which breaks another thread's capture, nor a stream this handle may never use)
    CR(dev_malloc((void**)&c->d_selhist, ICPMI_SELHIST_WORDS * sizeof(unsigned)));
    c->cap_selhist = ICPMI_SELHIST_WORDS;
    CR(hipHostMalloc((void**)&c->h_state, sizeof(IcpState) * ICPMI_MAX_BATCH, hipHostMallocMapped));
    memset(c->h_state, 0, sizeof(IcpState) * ICPMI_MAX_BATCH);
    if (hipHostGetDevicePointer((void**)&c->d_state_mirror, c->h_state, 0) != hipSuccess) { c->d_state_mirror = nullptr; (void)hipGetLastError(); } // (then loop_run copies)
    CR(hipHostMalloc((void**)&c->h_pin, ICPMI_PIN_BYTES + ICPMI_UP_SLOT * ICPMI_UP_SLOTS, hipHostMallocDefault));
    CR(hipHostMalloc((void**)&c->h_nocc, 64, hipHostMallocDefault));
    *c->h_nocc = 0;
    if (hipHostGetDevicePointer((void**)&c->d_nocc_host, c->h_nocc, 0) != hipSuccess) c->d_nocc_host = nullptr; // (then the copy launch stays)
    CR(hipHostMalloc((void**)&c->h_progress, ICPMI_PROGRESS_WORDS * sizeof(unsigned), hipHostMallocMapped)); // words 0..15: progress per reading of a batch; word 32: sequence number of the registration being launched
    memset(c->h_progress, 0, ICPMI_PROGRESS_WORDS * sizeof(unsigned));
    CR(hipHostGetDevicePointer((void**)&c->d_progress, c->h_progress, 0));
    CR(hipEventCreate(&c->ev0));
    CR(hipEventCreate(&c->ev1));
#undef CR
    *out = c;
    return ICPMI_OK;
}

extern "C" {

// handles the CALLER creates count towards the allocation cache's lifetime (private handles are created through create_handle directly)
icpmi_status icpmi_create(const icpmi_config* cfg, icpmi_handle* out)
{
    const icpmi_status s = create_handle(cfg, out);
    if (s == ICPMI_OK) {
        (*out)->counted = true;
        DevBlockCache& bc = dev_block_cache();
        std::lock_guard<std::mutex> lk(bc.mu);
        ++bc.handles;
    }
    return s;
}

icpmi_status icpmi_set_config(icpmi_handle h, const icpmi_config* cfg)
{
    CHECK_H(h);
    if (!cfg) { h->last_error = "set_config: null config"; return ICPMI_ERR_INVALID_ARG; }
    icpmi_status vs = validate_config(cfg, h->last_error);
    if (vs != ICPMI_OK) return vs;
    if (cfg->device != h->device) { h->last_error = "set_config: the device of a handle cannot change"; return ICPMI_ERR_INVALID_ARG; }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->cfg = *cfg;
    // the cached loop graph was captured for the previous chain
    drop_loop_graphs(h);
    return ICPMI_OK;
}

void icpmi_destroy(icpmi_handle c)
{
    if (!c) return;
    if (c->temp) { icpmi_destroy(c->temp); c->temp = nullptr; }
    if (c->temp_raw) { icpmi_destroy(c->temp_raw); c->temp_raw = nullptr; }
    hipSetDevice(c->device);
    selfgrid_destroy(c);
    comm_destroy(c);
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->graph_exec) hipGraphExecDestroy(c->graph_exec);
    if (c->bgraph_exec) hipGraphExecDestroy(c->bgraph_exec);
    for (int g = 0; g < 2; ++g) if (c->seg_exec[g]) hipGraphExecDestroy(c->seg_exec[g]);
    for (auto& hd : c->seg_heads) if (hd.exec) hipGraphExecDestroy(hd.exec);
    // ONE device-wide wait for all of the handle's blocks (one per block, each under the exclusive capture gate, stalled every
    // other thread's registration on this GPU ~80 times in a row)
    (void)dev_sync_for_free();
#define dev_free(p) dev_free((p), true)
    dev_free(c->d_map_sorted); dev_free(c->d_normals_sorted); dev_free(c->d_cell_start);
    for (int l = 0; l < ICPMI_MAXLEV; ++l) { dev_free(c->d_lvl_pts[l]); dev_free(c->d_lvl_cs[l]); dev_free(c->d_lvl_pos0[l]); }
    dev_free(c->d_inv);
    for (int l = 0; l < ICPMI_MAXLEV; ++l) { dev_free(c->d_lvl_key[l]); dev_free(c->d_alt_pts[l]); dev_free(c->d_alt_cs[l]); dev_free(c->d_alt_pos0[l]); dev_free(c->d_alt_key[l]); }
    dev_free(c->d_raw0); dev_free(c->d_alt_raw0); dev_free(c->d_ins_dstart0);
    dev_free(c->d_alt_nsorted); dev_free(c->d_alt_pn); dev_free(c->d_ins_key); dev_free(c->d_ins_rank);
    dev_free(c->d_keys); dev_free(c->d_fill); dev_free(c->d_blocksums); dev_free(c->d_red);
    dev_free(c->d_qsorted); dev_free(c->d_qindex); dev_free(c->d_qkeys); dev_free(c->d_qtile);
    dev_free(c->d_reading); dev_free(c->d_read_normals); dev_free(c->d_stage_in); dev_free(c->d_stage_n3);
    dev_free(c->d_match_pt); dev_free(c->d_lvl_tab); dev_free(c->d_raw); dev_free(c->d_raw_n3); dev_free(c->d_raw_s); dev_free(c->d_src); dev_free(c->d_alt_raw); dev_free(c->d_alt_n3);
    dev_free(c->d_alt_s); dev_free(c->d_alt_src); dev_free(c->d_stage_s); dev_free(c->d_merge_send); dev_free(c->d_merge_recv); dev_free(c->d_merged); dev_free(c->d_cell_log); if (c->h_cells) (void)hipHostFree(c->h_cells); dev_free(c->d_raw_dk); dev_free(c->d_comm_cnt); dev_free(c->d_read_noise); dev_free(c->d_read_scalar); dev_free(c->d_map_pn);
    for (int k = 0; k < ICPMI_SCRATCH_SLOTS; ++k) dev_free(c->scratch[k]); dev_free(c->d_scan_map); dev_free(c->d_T16);
    dev_free(c->d_sidx); dev_free(c->d_d2); dev_free(c->d_hard); dev_free(c->d_selhist);
    dev_free(c->d_state);
#undef dev_free
    if (c->h_state) hipHostFree(c->h_state);
    if (c->h_pin) hipHostFree(c->h_pin);
    if (c->h_nocc) hipHostFree(c->h_nocc);
    if (c->h_progress) hipHostFree(c->h_progress);
    if (c->ev0) hipEventDestroy(c->ev0);
    if (c->ev1) hipEventDestroy(c->ev1);
    for (hipEvent_t e : c->nn_events) hipEventDestroy(e);
    if (c->side_fork) hipEventDestroy(c->side_fork);
    if (c->side_join) hipEventDestroy(c->side_join);
    if (c->side) stream_release(c->side);
    if (c->own_stream && c->stream) stream_release(c->stream);
    const bool counted = c->counted;
    delete c;
    if (counted) { // (bookkeeping only: a mapper that is dropped and rebuilt -- the replay harness, one per pass -- finds its blocks again;
                   //  a host that wants the memory back calls icpmi_trim_cache(), measured r6: trimming here cost every new mapper ~300 hipMalloc)
        DevBlockCache& bc = dev_block_cache();
        std::lock_guard<std::mutex> lk(bc.mu);
        if (--bc.handles < 0) bc.handles = 0;
    }
}

// every device block the library's allocation cache holds back to the runtime (common.h: DevBlockCache); handles stay valid
icpmi_status icpmi_trim_cache(void)
{
    dev_cache_trim();
    return ICPMI_OK;
}


const char* icpmi_last_error(icpmi_handle h) { return h ? h->last_error.c_str() : g_create_error.c_str(); }

icpmi_status icpmi_set_stream(icpmi_handle h, void* hip_stream)
{
    CHECK_H(h);
    if (h->stream) hipStreamSynchronize(h->stream);
    if (h->own_stream && h->stream) stream_release(h->stream);
    h->stream = (hipStream_t)hip_stream;
    h->own_stream = false;
    drop_loop_graphs(h);
    return ICPMI_OK;
}

int32_t icpmi_has_map(icpmi_handle h) { return h && h->m > 0 ? 1 : 0; }

icpmi_status icpmi_get_map_mean(icpmi_handle h, float mean3[3])
{
    if (!h || !mean3) return ICPMI_ERR_INVALID_ARG;
    memcpy(mean3, h->mean, 3 * sizeof(float));
    return ICPMI_OK;
}

icpmi_status icpmi_debug_counters(icpmi_handle h, uint64_t out[24])
{
    if (!h || !out) return ICPMI_ERR_INVALID_ARG;
    for (int i = 0; i < 24; ++i) out[i] = h->h_state->dbg[i];
    // [18] / [19]: index builds served by the incremental insert / from scratch (low word: this handle's registration index; high
    // word: its private raw-frame index, ops.hip: raw_index)
    // (a -DICPMI_NN_TIMING build keeps all 24 slots for the NN kernels' phase clocks -- ADVICE r4)
#ifndef ICPMI_NN_TIMING
    out[14] = (uint64_t)h->merge_fast_epochs; // map-growth epochs served with ONE collective / with the count + ready + points exchanges (ops.hip)
    out[15] = (uint64_t)h->merge_slow_epochs;
    out[16] = (uint64_t)h->oct_respeculated; // octree filter calls whose speculated depth was too shallow (sorted twice; octree.hip)
    out[17] = (uint64_t)h->raw_view_count; // PointDistance searches served by the raw-frame view of the registration index (no second index)
    out[18] = (uint64_t)(uint32_t)h->ins_count | ((uint64_t)(uint32_t)(h->temp_raw ? h->temp_raw->ins_count : 0) << 32);
    out[19] = (uint64_t)(uint32_t)h->full_count | ((uint64_t)(uint32_t)(h->temp_raw ? h->temp_raw->full_count : 0) << 32);
    // (r6) SurfaceNormal passes over the resident map of an append-only update: served by the subset search / over the whole map; points the last pass searched
    out[20] = (uint64_t)h->normals_incremental;
    out[21] = (uint64_t)h->normals_full;
    out[22] = (uint64_t)h->normals_last_searched;
#endif
    return ICPMI_OK;
}

icpmi_status icpmi_debug_minstd_nth(icpmi_handle h, uint32_t seed, uint32_t n, uint32_t* out)
{
    CHECK_H(h);
    if (!out || n == 0) { h->last_error = "debug_minstd_nth: bad arguments"; return ICPMI_ERR_INVALID_ARG; }
    unsigned v = 0;
    const icpmi_status s = ssn_debug_minstd(h, seed, n, &v);
    *out = v;
    return s;
}

icpmi_status icpmi_get_grid_info(icpmi_handle h, float* cell, int32_t dims[3], int64_t* n_cells, int64_t* n_occupied)
{
    if (!h) return ICPMI_ERR_INVALID_ARG;
    if (cell) *cell = h->grid.cell;
    if (dims) { dims[0] = h->grid.nx; dims[1] = h->grid.ny; dims[2] = h->grid.nz; }
    if (n_cells) *n_cells = h->grid.ncells;
    if (n_occupied) *n_occupied = h->n_occupied;
    return ICPMI_OK;
}

icpmi_status icpmi_set_map_dev(icpmi_handle h, const float* d_map4, int64_t m, const float* d_normals3, int32_t* accepted)
{
    CHECK_H(h);
    if (accepted) *accepted = 0;
    if (m < 0 || (m > 0 && !d_map4)) { h->last_error = "set_map: bad arguments"; return ICPMI_ERR_INVALID_ARG; }
    if (m == 0) return ICPMI_OK; // upstream: "Ignoring attempt to setMap with an empty map", returns false
    if (m >= (1ll << 28)) { h->last_error = "set_map: more than 2^28-1 points (match positions carry the pyramid level in their top 4 bits)"; return ICPMI_ERR_UNSUPPORTED; }
    if (h->cfg.minimizer == ICPMI_MIN_POINT_TO_PLANE && !d_normals3) {
        // upstream fails later, inside the minimiser, with InvalidField("normals"); keep the map and
        // let icpmi_register report it
    }
    icpmi_status s = map_build(h, (const float4*)d_map4, m, d_normals3);
    if (s != ICPMI_OK) return s;
    if (accepted) *accepted = 1;
    return ICPMI_OK;
}

icpmi_status icpmi_set_map(icpmi_handle h, const float* map4, int64_t m, const float* normals3, int32_t* accepted)
{
    CHECK_H(h);
    if (accepted) *accepted = 0;
    if (m < 0 || (m > 0 && !map4)) { h->last_error = "set_map: bad arguments"; return ICPMI_ERR_INVALID_ARG; }
    if (m == 0) return ICPMI_OK;
    if (ensure_cap(h, &h->d_stage_in, &h->cap_stage_in, (size_t)m) != ICPMI_OK) return ICPMI_ERR_HIP;
    HIP_TRY(h, hipMemcpyAsync(h->d_stage_in, map4, (size_t)m * sizeof(float4), hipMemcpyHostToDevice, h->stream));
    if (normals3) {
        if (ensure_cap(h, &h->d_stage_n3, &h->cap_stage_n3, (size_t)m * 3) != ICPMI_OK) return ICPMI_ERR_HIP;
        HIP_TRY(h, hipMemcpyAsync(h->d_stage_n3, normals3, (size_t)m * 3 * sizeof(float), hipMemcpyHostToDevice, h->stream));
    }
    return icpmi_set_map_dev(h, (const float*)h->d_stage_in, m, normals3 ? h->d_stage_n3 : nullptr, accepted);
}

static void identity16(float* T) { for (int i = 0; i < 16; ++i) T[i] = (i % 5 == 0) ? 1.f : 0.f; }

// ICPMI_STATS_JSON=<path>: one JSON line per registration with the names libpointmatcher's inspector records for an ICP call
// (IterationsCount, OverlapRatio, ConvergenceDuration) plus the loop's own statistics.  Opened once, appended.
static FILE* stats_json()
{
    static FILE* f = [] { const char* p = getenv("ICPMI_STATS_JSON"); return (p && *p) ? fopen(p, "a") : (FILE*)nullptr; }();
    return f;
}

static void write_stats_json(const icpmi_stats& st, int64_t n, icpmi_status rs)
{
    static std::mutex mu;
    std::lock_guard<std::mutex> g(mu);
    fprintf(stats_json(), "{\"IterationsCount\": %d, \"OverlapRatio\": %.9g, \"ConvergenceDuration\": %.9g, \"PointUsedRatio\": %.9g, "
                          "\"PairsUsed\": %lld, \"ReadingPoints\": %lld, \"StopReason\": %d, \"TrimmedLimit\": %.9g, \"Status\": %d}\n",
            st.iterations, (double)st.weighted_point_used_ratio, (double)st.loop_ms * 1e-3, (double)st.point_used_ratio, (long long)st.pairs,
            (long long)n, st.stop_reason, (double)st.trimmed_limit, (int)rs);
    fflush(stats_json());
}

// fields the GenericDescriptor / Robust filters of the chain read on the map
static icpmi_status check_ext_filters(icpmi_ctx* h)
{
    for (int f = 0; f < h->cfg.n_outlier; ++f) {
        const icpmi_outlier& o = h->cfg.outlier[f];
        if (o.type == ICPMI_OUT_GENERICDESCRIPTOR && !(o.iparam & ICPMI_GEN_SOURCE_READING) && !(h->raw_has_scalar && h->m_raw == h->m)) {
            h->last_error = "InvalidField: GenericDescriptorOutlierFilter needs the tracked scalar descriptor on the map (icpmi_set_map_scalar)";
            return ICPMI_ERR_INVALID_ARG;
        }
        if (o.type == ICPMI_OUT_ROBUST && ((o.iparam >> 8) & 15) == ICPMI_DIST_POINT2PLANE && !h->has_normals) {
            h->last_error = "InvalidField: RobustOutlierFilter{distanceType: point2plane} needs the descriptor 'normals' on the map";
            return ICPMI_ERR_MISSING_NORMALS;
        }
    }
    return ICPMI_OK;
}

// the stage entry points carry no reading descriptor: a chain with GenericDescriptorOutlierFilter{source: reading} is served by icpmi_register only
static icpmi_status reject_reading_source(icpmi_ctx* h, const char* who)
{
    for (int f = 0; f < h->cfg.n_outlier; ++f)
        if (h->cfg.outlier[f].type == ICPMI_OUT_GENERICDESCRIPTOR && (h->cfg.outlier[f].iparam & ICPMI_GEN_SOURCE_READING)) {
            h->last_error = std::string(who) + ": GenericDescriptorOutlierFilter{source: reading} is only available inside icpmi_register";
            return ICPMI_ERR_UNSUPPORTED;
        }
    return ICPMI_OK;
}

static icpmi_status register_impl(icpmi_handle h, const float* d_scan4, int64_t n, const float* d_n3, int fixed_iters,
                                  float T_out[16], icpmi_stats* stats)
{
    // the one-shot reading rows (icpmi_set_reading_scalar / _sensor_noise) belong to THIS call whatever becomes of it: consumed here, before
    // any early return, so that a registration that bails out cannot leave a stale row armed for the next one
    const int64_t have_scalar_n = h->read_scalar_n, have_noise_n = h->read_noise_n;
    h->read_scalar_n = 0; h->read_noise_n = 0;
    if (!T_out) { h->last_error = "register: T_out is null"; return ICPMI_ERR_INVALID_ARG; }
    if (stats) { memset(stats, 0, sizeof *stats); stats->sensor_noise_overlap = -1.f; } // -1 = "not computed" on EVERY path
    identity16(T_out);
    if (n < 0 || (n > 0 && !d_scan4)) { h->last_error = "register: bad arguments"; return ICPMI_ERR_INVALID_ARG; }
    if (h->m <= 0) return ICPMI_OK; // "Ignoring attempt to perform ICP with an empty map" -> identity
    if (n > 0x7fffffff / ICPMI_MAX_K) { h->last_error = "register: too many points"; return ICPMI_ERR_UNSUPPORTED; }
    if (h->cfg.minimizer == ICPMI_MIN_POINT_TO_PLANE && !h->has_normals) {
        h->last_error = "InvalidField: PointToPlaneErrorMinimizer needs the descriptor 'normals' on the map";
        return ICPMI_ERR_MISSING_NORMALS;
    }
    bool needs_rn = false;
    for (int f = 0; f < h->cfg.n_outlier; ++f) needs_rn |= h->cfg.outlier[f].type == ICPMI_OUT_SURFACENORMAL;
    if (needs_rn && (!d_n3 || !h->has_normals)) {
        h->last_error = "InvalidField: SurfaceNormalOutlierFilter needs 'normals' on both reading and map";
        return ICPMI_ERR_MISSING_NORMALS;
    }
    { const icpmi_status es = check_ext_filters(h); if (es != ICPMI_OK) return es; }
    LoopCfg lc = make_loop_cfg(h, fixed_iters);
    {   // GenericDescriptorOutlierFilter{source: reading}: the row handed over for THIS reading (one shot)
        bool wants = false;
        for (int f = 0; f < h->cfg.n_outlier; ++f) wants |= h->cfg.outlier[f].type == ICPMI_OUT_GENERICDESCRIPTOR && (h->cfg.outlier[f].iparam & ICPMI_GEN_SOURCE_READING);
        if (wants && n > 0 && have_scalar_n != n) {
            h->last_error = "InvalidField: GenericDescriptorOutlierFilter{source: reading} needs the reading's descriptor (icpmi_set_reading_scalar, one row per point)";
            return ICPMI_ERR_INVALID_ARG;
        }
        lc.read_scalar = wants ? h->d_read_scalar : nullptr;
    }
    // sensor-noise overlap (icpmi_set_reading_sensor_noise): the noise row is one shot -- this registration consumes it
    // (PointToPointErrorMinimizer::getOverlap() needs `simpleSensorNoise` alone; only the point-to-plane variant also reads the reading's
    // `normals` -- ADVICE r3)
    const bool sn = have_noise_n == n && n > 0 && !lc.ext && !lc.is_2d &&
                    (lc.minimizer == ICPMI_MIN_POINT_TO_POINT || (lc.minimizer == ICPMI_MIN_POINT_TO_PLANE && d_n3));
    lc.sensor_noise = sn ? 1 : 0;
    lc.has_read_normals = (needs_rn || (sn && d_n3)) ? 1 : 0;
    if (n == 0) {
        // upstream: Trimmed/Median throw "no outlier to filter", otherwise "no point to minimize"
        bool quant = false;
        for (int f = 0; f < lc.n_out; ++f) quant |= lc.out_type[f] == ICPMI_OUT_TRIMMEDDIST || lc.out_type[f] == ICPMI_OUT_MEDIANDIST || lc.out_type[f] == ICPMI_OUT_VARTRIMMEDDIST;
        h->last_error = quant ? "ConvergenceError: no outlier to filter" : "ConvergenceError: ErrorMinimizer: no point to minimize";
        return quant ? ICPMI_ERR_NO_OUTLIER_TO_FILTER : ICPMI_ERR_NO_POINT_TO_MINIMIZE;
    }
    icpmi_stats local;
    if (!stats && stats_json()) stats = &local;
    const icpmi_status rs = loop_run(h, (const float4*)d_scan4, (needs_rn || sn) ? d_n3 : nullptr, n, lc, fixed_iters > 0, T_out, stats);
    if (stats_json() && stats) write_stats_json(*stats, n, rs);
    return rs;
}

icpmi_status icpmi_set_reading_sensor_noise(icpmi_handle h, const float* noise, int64_t n)
{
    CHECK_H(h);
    h->read_noise_n = 0;
    if (!noise || n <= 0) return ICPMI_OK;
    if (ensure_cap(h, &h->d_read_noise, &h->cap_read_noise, (size_t)n + 1) != ICPMI_OK) return ICPMI_ERR_HIP;
    HIP_TRY(h, hipMemcpyAsync(h->d_read_noise, noise, (size_t)n * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream)); // the caller's buffer is free on return
    h->read_noise_n = n;
    return ICPMI_OK;
}

icpmi_status icpmi_set_reading_scalar(icpmi_handle h, const float* scalar, int64_t n)
{
    CHECK_H(h);
    h->read_scalar_n = 0;
    if (!scalar || n <= 0) return ICPMI_OK;
    if (ensure_cap(h, &h->d_read_scalar, &h->cap_read_scalar, (size_t)n + 1) != ICPMI_OK) return ICPMI_ERR_HIP;
    HIP_TRY(h, hipMemcpyAsync(h->d_read_scalar, scalar, (size_t)n * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream)); // the caller's buffer is free on return
    h->read_scalar_n = n;
    return ICPMI_OK;
}

icpmi_status icpmi_register_dev(icpmi_handle h, const float* d_scan4, int64_t n, const float* d_scan_normals3, float T_out[16],
                                icpmi_stats* stats)
{
    CHECK_H(h);
    return register_impl(h, d_scan4, n, d_scan_normals3, 0, T_out, stats);
}

icpmi_status icpmi_register_fixed_dev(icpmi_handle h, const float* d_scan4, int64_t n, const float* d_scan_normals3,
                                      int32_t iterations, float T_out[16], icpmi_stats* stats)
{
    CHECK_H(h);
    if (iterations < 1) { h->last_error = "register_fixed: iterations must be >= 1"; return ICPMI_ERR_INVALID_ARG; }
    return register_impl(h, d_scan4, n, d_scan_normals3, iterations, T_out, stats);
}

icpmi_status icpmi_register_batch_dev(icpmi_handle h, int32_t batch, const float* const* d_scans4, const int64_t* n, int32_t fixed_iterations,
                                      float* T_out, icpmi_stats* stats, icpmi_status* status)
{
    CHECK_H(h);
    if (batch < 1 || batch > ICPMI_MAX_BATCH || !d_scans4 || !n || !T_out || fixed_iterations < 0) {
        h->last_error = "register_batch: bad arguments (1 <= batch <= 16)"; return ICPMI_ERR_INVALID_ARG;
    }
    for (int b = 0; b < batch; ++b)
        if (n[b] < 0 || (n[b] > 0 && !d_scans4[b])) { h->last_error = "register_batch: bad arguments"; return ICPMI_ERR_INVALID_ARG; }
    // The one-launch-per-iteration path serves the chains whose kernels take a batch: k = 1 on the grid pyramid without a
    // brute-force pass, at most one quantile filter, no filter that reads descriptors of the reading.  Everything else --
    // and empty readings, a handle without a map -- runs the readings one after the other: same results, no sharing.
    bool together = batch > 1 && h->m > 0 && h->cfg.knn == 1 && (h->cfg.minimizer != ICPMI_MIN_POINT_TO_PLANE || h->has_normals);
    int nquant = 0;
    for (int f = 0; f < h->cfg.n_outlier; ++f) {
        const int t = h->cfg.outlier[f].type;
        if (t == ICPMI_OUT_SURFACENORMAL || t == ICPMI_OUT_GENERICDESCRIPTOR || t == ICPMI_OUT_ROBUST || t == ICPMI_OUT_VARTRIMMEDDIST) together = false;
        if (t == ICPMI_OUT_TRIMMEDDIST || t == ICPMI_OUT_MEDIANDIST) ++nquant;
    }
    if (nquant > 1) together = false;
    for (int b = 0; b < batch; ++b) if (n[b] == 0 || n[b] > 0x7fffffff / ICPMI_MAX_K) together = false;
    if (together) {
        const GridParams& top = h->levels.g[h->levels.nlev - 1];
        if (!std::isfinite(h->cfg.max_dist) || (top.cell - top.slack) <= h->cfg.max_dist) together = false; // would need the brute-force pass
    }
    if (!together) {
        icpmi_status first = ICPMI_OK;
        for (int b = 0; b < batch; ++b) {
            const icpmi_status sb = register_impl(h, d_scans4[b], n[b], nullptr, fixed_iterations, T_out + 16 * b, stats ? stats + b : nullptr);
            if (status) status[b] = sb;
            if (sb != ICPMI_OK && first == ICPMI_OK) first = sb;
        }
        return status ? ICPMI_OK : first;
    }
    if (stats) { memset(stats, 0, sizeof(icpmi_stats) * batch); for (int b = 0; b < batch; ++b) stats[b].sensor_noise_overlap = -1.f; }
    LoopCfg lc = make_loop_cfg(h, fixed_iterations);
    return loop_run_batch(h, batch, d_scans4, n, lc, fixed_iterations > 0, T_out, stats, status);
}

icpmi_status icpmi_register(icpmi_handle h, const float* scan4, int64_t n, const float* scan_normals3, float T_out[16],
                            icpmi_stats* stats)
{
    CHECK_H(h);
    if (n < 0 || (n > 0 && !scan4)) { h->last_error = "register: bad arguments"; return ICPMI_ERR_INVALID_ARG; }
    const float* d_scan = nullptr;
    const float* d_n3 = nullptr;
    if (n > 0 && h->m > 0) {
        if (ensure_cap(h, &h->d_stage_in, &h->cap_stage_in, (size_t)n) != ICPMI_OK) return ICPMI_ERR_HIP;
        HIP_TRY(h, hipMemcpyAsync(h->d_stage_in, scan4, (size_t)n * sizeof(float4), hipMemcpyHostToDevice, h->stream));
        d_scan = (const float*)h->d_stage_in;
        if (scan_normals3) {
            if (ensure_cap(h, &h->d_stage_n3, &h->cap_stage_n3, (size_t)n * 3) != ICPMI_OK) return ICPMI_ERR_HIP;
            HIP_TRY(h, hipMemcpyAsync(h->d_stage_n3, scan_normals3, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice, h->stream));
            d_n3 = h->d_stage_n3;
        }
    } else if (n > 0) {
        d_scan = scan4; // no map: never dereferenced
    }
    return register_impl(h, d_scan, n, d_n3, 0, T_out, stats);
}

icpmi_status icpmi_transform(icpmi_handle h, const float T[16], const float* in4, int64_t n, float* out4, const float* in_normals3,
                             float* out_normals3)
{
    CHECK_H(h);
    if (!T || n < 0 || (n > 0 && (!in4 || !out4))) { h->last_error = "transform: bad arguments"; return ICPMI_ERR_INVALID_ARG; }
    return ops_transform(h, T, in4, n, out4, in_normals3, out_normals3);
}

icpmi_status icpmi_knn(icpmi_handle h, const float* q4, int64_t n, int32_t k, float max_dist, int32_t allow_self, int32_t* ids,
                       float* d2)
{
    CHECK_H(h);
    if (n < 0 || k < 1 || k > ICPMI_MAX_K || (n > 0 && (!q4 || !ids || !d2)) || !(max_dist > 0.f)) {
        h->last_error = "knn: bad arguments"; return ICPMI_ERR_INVALID_ARG;
    }
    if (n == 0) return ICPMI_OK;
    if (h->m <= 0) {
        for (int64_t i = 0; i < n * k; ++i) { ids[i] = -1; d2[i] = INFINITY; }
        return ICPMI_OK;
    }
    const size_t cnt = (size_t)n * k + 1;
    if (ensure_cap(h, &h->d_reading, &h->cap_reading, (size_t)n + 1) != ICPMI_OK || ensure_cap(h, &h->d_sidx, &h->cap_sidx, cnt) != ICPMI_OK ||
        ensure_cap(h, &h->d_d2, &h->cap_d2, cnt) != ICPMI_OK || ensure_cap(h, &h->d_hard, &h->cap_hard, (size_t)n + 1) != ICPMI_OK)
        return ICPMI_ERR_HIP;
    HIP_TRY(h, hipMemcpyAsync(h->d_reading, q4, (size_t)n * sizeof(float4), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemsetAsync(h->d_state, 0, sizeof(IcpState), h->stream));
    if (k == 1) {
        icpmi_status ss = sort_queries(h, h->d_reading, n);
        if (ss != ICPMI_OK) return ss;
    } else {
        h->qsorted_n = -1; h->qsorted_src = nullptr;
    }
    LoopCfg lc = make_loop_cfg(h, 1);
    lc.k = k; lc.max_dist = max_dist;
    lc.maxr2 = std::isinf(max_dist) ? INFINITY : max_dist * max_dist;
    if (std::isfinite(max_dist)) {
        const int need = (int)ceilf(max_dist / h->grid.cell) + 1;
        lc.ring_max = need < 16 ? need : 16;
    } else lc.ring_max = 6;
    h->nn_hist0 = nullptr; // stage call: no quantile selection follows
    h->nn_match_pt = nullptr;
    h->nn_iter_hint = 0;
    icpmi_status s = nn_launch_k(h, h->d_reading, n, nullptr, lc, allow_self, h->d_sidx, h->d_d2, h->d_state);
    if (s != ICPMI_OK) return s;
    DevBuf<int> d_ids;
    HIP_TRY(h, d_ids.alloc((size_t)n * k));
    s = nn_ids_to_original(h, h->d_sidx, n * k, d_ids);
    hipError_t e = hipSuccess;
    if (s == ICPMI_OK) {
        e = hipMemcpyAsync(ids, d_ids, (size_t)n * k * sizeof(int), hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d2, h->d_d2, (size_t)n * k * sizeof(float), hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    }
    if (s != ICPMI_OK) return s;
    HIP_TRY(h, e);
    return ICPMI_OK;
}

icpmi_status icpmi_outlier_weights(icpmi_handle h, const float* d2, const int32_t* ids, int32_t k, int64_t n,
                                   const float* read_normals3, float* weights, float* limit_out)
{
    CHECK_H(h);
    if (k < 1 || n < 0 || (n > 0 && (!d2 || !weights))) { h->last_error = "outlier_weights: bad arguments"; return ICPMI_ERR_INVALID_ARG; }
    if (n == 0) return ICPMI_OK;
    { const icpmi_status rs = reject_reading_source(h, "outlier_weights"); if (rs != ICPMI_OK) return rs; }
    LoopCfg lc = make_loop_cfg(h, 1);
    return loop_outlier_weights(h, lc, d2, ids, k, n, read_normals3, weights, limit_out);
}

icpmi_status icpmi_minimize_step(icpmi_handle h, const float* reading4, int64_t n, const float* T_iter, float T_step[16],
                                 double sums[32], icpmi_stats* stats)
{
    CHECK_H(h);
    if (n <= 0 || !reading4) { h->last_error = "minimize_step: bad arguments"; return ICPMI_ERR_INVALID_ARG; }
    { const icpmi_status es = check_ext_filters(h); if (es != ICPMI_OK) return es; }
    { const icpmi_status rs = reject_reading_source(h, "minimize_step"); if (rs != ICPMI_OK) return rs; }
    if (h->m <= 0) { h->last_error = "minimize_step: no map"; return ICPMI_ERR_INVALID_ARG; }
    if (h->cfg.minimizer == ICPMI_MIN_POINT_TO_PLANE && !h->has_normals) {
        h->last_error = "InvalidField: PointToPlaneErrorMinimizer needs the descriptor 'normals' on the map";
        return ICPMI_ERR_MISSING_NORMALS;
    }
    if (stats) { memset(stats, 0, sizeof *stats); stats->sensor_noise_overlap = -1.f; }
    if (ensure_cap(h, &h->d_reading, &h->cap_reading, (size_t)n + 1) != ICPMI_OK) return ICPMI_ERR_HIP;
    HIP_TRY(h, hipMemcpyAsync(h->d_reading, reading4, (size_t)n * sizeof(float4), hipMemcpyHostToDevice, h->stream));
    // d_reading has new contents: a tile-sorted copy of what was there before (an icpmi_knn with the same n) must not be searched in
    // its place (r3: tests/test_gpu_golden.py found the step computed on the previous call's queries)
    h->qsorted_n = -1; h->qsorted_src = nullptr;
    LoopCfg lc = make_loop_cfg(h, 1);
    for (int f = 0; f < lc.n_out; ++f)
        if (lc.out_type[f] == ICPMI_OUT_SURFACENORMAL) { h->last_error = "minimize_step: SurfaceNormal filter needs icpmi_register"; return ICPMI_ERR_UNSUPPORTED; }
    return loop_single_step(h, n, lc, T_iter, T_step, sums, stats);
}

icpmi_status icpmi_surface_normals(icpmi_handle h, const float* pts4, int64_t m, int32_t knn, float* normals3)
{
    CHECK_H(h);
    if (m < 0 || (m > 0 && (!pts4 || !normals3))) { h->last_error = "surface_normals: bad arguments"; return ICPMI_ERR_INVALID_ARG; }
    return ops_surface_normals(h, pts4, m, knn, normals3);
}

icpmi_status icpmi_surface_normals_ex(icpmi_handle h, const float* pts4, int64_t m, int32_t knn, float* normals3, float* densities)
{
    CHECK_H(h);
    if (m < 0 || (m > 0 && (!pts4 || !normals3))) { h->last_error = "surface_normals: bad arguments"; return ICPMI_ERR_INVALID_ARG; }
    return ops_surface_normals(h, pts4, m, knn, normals3, densities);
}

icpmi_status icpmi_surface_normals_ex2(icpmi_handle h, const float* pts4, int64_t m, int32_t knn, float* normals3, float* densities,
                                       int32_t* matched_ids, float* mean_dist)
{
    CHECK_H(h);
    if (m < 0 || (m > 0 && (!pts4 || !normals3))) { h->last_error = "surface_normals: bad arguments"; return ICPMI_ERR_INVALID_ARG; }
    return ops_surface_normals(h, pts4, m, knn, normals3, densities, matched_ids, mean_dist);
}

icpmi_status icpmi_surface_normals_ex3(icpmi_handle h, const float* pts4, int64_t m, int32_t knn, float* normals3, float* densities,
                                       int32_t* matched_ids, float* mean_dist, float* eig_values3, float* eig_vectors9)
{
    CHECK_H(h);
    if (m < 0 || (m > 0 && (!pts4 || !normals3))) { h->last_error = "surface_normals: bad arguments"; return ICPMI_ERR_INVALID_ARG; }
    return ops_surface_normals(h, pts4, m, knn, normals3, densities, matched_ids, mean_dist, eig_values3, eig_vectors9);
}

icpmi_status icpmi_point_distance_keep(icpmi_handle h, const float* map4, int64_t m, const float* in4, int64_t n, float min_dist,
                                       uint8_t* keep)
{
    CHECK_H(h);
    if (m < 0 || n < 0 || (m > 0 && !map4) || (n > 0 && (!in4 || !keep))) { h->last_error = "point_distance_keep: bad arguments"; return ICPMI_ERR_INVALID_ARG; }
    return ops_point_distance_keep(h, map4, m, in4, n, min_dist, keep);
}

icpmi_status icpmi_map_update_point_distance(icpmi_handle h, const float* scan4, int64_t n, const float* scan_normals3, float min_dist,
                                             int32_t normals_knn, uint8_t* keep_out, int64_t* appended, int64_t* new_m)
{
    CHECK_H(h);
    if (n < 0 || (n > 0 && !scan4) || !(min_dist >= 0.f)) { h->last_error = "map_update_point_distance: bad arguments"; return ICPMI_ERR_INVALID_ARG; }
    return ops_map_update_point_distance(h, scan4, n, scan_normals3, min_dist, normals_knn, keep_out, appended, new_m);
}

icpmi_status icpmi_register_prior(icpmi_handle h, const float* scan4, int64_t n, const float prior[16], float T_out[16], icpmi_stats* stats)
{
    CHECK_H(h);
    if (n < 0 || (n > 0 && !scan4) || !prior) { h->last_error = "register_prior: bad arguments"; return ICPMI_ERR_INVALID_ARG; }
    h->scan_map_n = 0;
    if (n > 0) {
        if (ensure_cap(h, &h->d_stage_in, &h->cap_stage_in, (size_t)n + 1) != ICPMI_OK) return ICPMI_ERR_HIP;
        if (ensure_cap(h, &h->d_scan_map, &h->cap_scan_map, (size_t)n + 1) != ICPMI_OK) return ICPMI_ERR_HIP;
        HIP_TRY(h, hipMemcpyAsync(h->d_stage_in, scan4, (size_t)n * sizeof(float4), hipMemcpyHostToDevice, h->stream));
        icpmi_status s = ops_transform_dev(h, prior, h->d_stage_in, n, h->d_scan_map); // Mapper.cpp:197
        if (s != ICPMI_OK) return s;
        h->scan_map_n = n;
    }
    return register_impl(h, (const float*)h->d_scan_map, n, nullptr, 0, T_out, stats);
}

// ... with the scan already in HBM (sensor frame): what a scan stream that lives on the device -- or a benchmark that must not time
// PCIe -- hands over; everything else as icpmi_register_prior
icpmi_status icpmi_register_prior_dev(icpmi_handle h, const float* d_scan4, int64_t n, const float prior[16], float T_out[16], icpmi_stats* stats)
{
    CHECK_H(h);
    if (n < 0 || (n > 0 && !d_scan4) || !prior) { h->last_error = "register_prior_dev: bad arguments"; return ICPMI_ERR_INVALID_ARG; }
    h->scan_map_n = 0;
    if (n > 0) {
        if (ensure_cap(h, &h->d_scan_map, &h->cap_scan_map, (size_t)n + 1) != ICPMI_OK) return ICPMI_ERR_HIP;
        icpmi_status s = ops_transform_dev(h, prior, (const float4*)d_scan4, n, h->d_scan_map); // Mapper.cpp:197
        if (s != ICPMI_OK) return s;
        h->scan_map_n = n;
    }
    return register_impl(h, (const float*)h->d_scan_map, n, nullptr, 0, T_out, stats);
}

icpmi_status icpmi_map_update_staged(icpmi_handle h, const float correction[16], float min_dist, int32_t normals_knn, uint8_t* keep_out,
                                     int64_t* appended, int64_t* new_m)
{
    CHECK_H(h);
    if (!correction || !(min_dist >= 0.f)) { h->last_error = "map_update_staged: bad arguments"; return ICPMI_ERR_INVALID_ARG; }
    if (h->scan_map_n <= 0) { h->last_error = "map_update_staged: no scan staged by icpmi_register_prior"; return ICPMI_ERR_INVALID_ARG; }
    const int64_t n = h->scan_map_n;
    if (ensure_cap(h, &h->d_stage_in, &h->cap_stage_in, (size_t)n + 1) != ICPMI_OK) return ICPMI_ERR_HIP;
    icpmi_status s = ops_transform_dev(h, correction, h->d_scan_map, n, h->d_stage_in); // Mapper.cpp:221
    if (s != ICPMI_OK) return s;
    return ops_map_update_dev(h, h->d_stage_in, n, nullptr, min_dist, normals_knn, keep_out, appended, new_m);
}

icpmi_status icpmi_staged_point_distance_keep(icpmi_handle h, const float correction[16], float min_dist, uint8_t* keep_out, float* placed_out4)
{
    CHECK_H(h);
    if (!correction || !keep_out || !(min_dist >= 0.f)) { h->last_error = "staged_point_distance_keep: bad arguments"; return ICPMI_ERR_INVALID_ARG; }
    if (h->scan_map_n <= 0) { h->last_error = "staged_point_distance_keep: no scan staged by icpmi_register_prior"; return ICPMI_ERR_INVALID_ARG; }
    return ops_staged_keep(h, correction, min_dist, keep_out, placed_out4);
}

icpmi_status icpmi_comm_get_unique_id(icpmi_comm_id* id)
{
    if (!id) { g_create_error = "comm_get_unique_id: null argument"; return ICPMI_ERR_INVALID_ARG; }
    return comm_unique_id(id, g_create_error);
}

icpmi_status icpmi_comm_init(icpmi_handle h, const icpmi_comm_id* id, int32_t n_ranks, int32_t rank)
{
    CHECK_H(h);
    if (!id || n_ranks < 1 || rank < 0 || rank >= n_ranks) { h->last_error = "comm_init: bad arguments"; return ICPMI_ERR_INVALID_ARG; }
    return comm_init(h, id, n_ranks, rank);
}

icpmi_status icpmi_comm_info(icpmi_handle h, int32_t* n_ranks, int32_t* rank, int32_t* kind)
{
    CHECK_H(h);
    int a = 1, b = 0, k = 0;
    const icpmi_status s = comm_info(h, &a, &b, &k);
    if (n_ranks) *n_ranks = a;
    if (rank) *rank = b;
    if (kind) *kind = k;
    return s;
}

icpmi_status icpmi_comm_destroy(icpmi_handle h)
{
    CHECK_H(h);
    return comm_destroy(h);
}

icpmi_status icpmi_staged_merge_allgather(icpmi_handle h, const float correction[16], float min_dist, int32_t normals_knn,
                                          int64_t* accepted_local, int64_t* appended_total, int64_t* new_m, float* merged_out4,
                                          int64_t merged_capacity, int64_t* merged_n)
{
    CHECK_H(h);
    // (argument errors are local by construction: every rank passes the same min_dist / normals_knn, and a rank that hands in
    // garbage here has a bug no protocol can paper over)
    if (!(min_dist >= 0.f) || normals_knn < 0 || normals_knn > ICPMI_MAX_K || merged_capacity < 0) {
        h->last_error = "staged_merge_allgather: bad arguments"; return ICPMI_ERR_INVALID_ARG;
    }
    // correction == NULL or nothing staged: this rank contributes no points and still takes part in the exchange
    return ops_staged_merge_allgather(h, correction, min_dist, normals_knn, accepted_local, appended_total, new_m, merged_out4, merged_capacity, merged_n);
}

icpmi_status icpmi_staged_merged_points(icpmi_handle h, float* out4, int64_t capacity, int64_t* n)
{
    CHECK_H(h);
    if (capacity < 0) { h->last_error = "staged_merged_points: bad arguments"; return ICPMI_ERR_INVALID_ARG; }
    return ops_staged_merged_points(h, out4, capacity, n);
}

icpmi_status icpmi_staged_bin_cells(icpmi_handle h, float cell_size, int32_t* ijk3, int64_t* offsets, int64_t* counts, int64_t capacity, int64_t* n_cells)
{
    CHECK_H(h);
    if (!(cell_size > 0.f) || capacity < 0) { h->last_error = "staged_bin_cells: bad arguments"; return ICPMI_ERR_INVALID_ARG; }
    return ops_staged_bin_cells(h, cell_size, ijk3, offsets, counts, capacity, n_cells);
}

icpmi_status icpmi_cell_log_configure(icpmi_handle h, float cell_size)
{
    CHECK_H(h);
    if (!(cell_size >= 0.f)) { h->last_error = "cell_log_configure: bad arguments"; return ICPMI_ERR_INVALID_ARG; }
    h->cell_auto_size = cell_size;
    return ICPMI_OK;
}

icpmi_status icpmi_cell_log_read(icpmi_handle h, int64_t offset, int64_t count, float* out4, int64_t* log_size)
{
    CHECK_H(h);
    return ops_cell_log_read(h, offset, count, out4, log_size);
}

icpmi_status icpmi_cell_log_clear(icpmi_handle h)
{
    CHECK_H(h);
    return ops_cell_log_clear(h);
}

icpmi_status icpmi_stage_discard(icpmi_handle h)
{
    CHECK_H(h);
    h->scan_map_n = 0;
    return ICPMI_OK;
}

icpmi_status icpmi_get_map(icpmi_handle h, float* out4, float* normals3, int64_t capacity, int64_t* m)
{
    CHECK_H(h);
    return ops_get_map(h, out4, normals3, capacity, m);
}

icpmi_status icpmi_voxel_keep(icpmi_handle h, const float* in4, int64_t n, float edge, int32_t method, uint8_t* keep)
{
    CHECK_H(h);
    if (n < 0 || (n > 0 && (!in4 || !keep)) || !(edge > 0.f) || (method != 0 && method != 1)) { h->last_error = "voxel_keep: bad arguments"; return ICPMI_ERR_INVALID_ARG; }
    return ops_voxel_keep_first(h, in4, n, edge, method, keep);
}

icpmi_status icpmi_sampling_surface_normal(icpmi_handle h, const float* in4, int64_t n, float ratio, int32_t knn, float max_box_dim, int32_t seed,
                                           int32_t* order_out, float* normals3_out, int64_t* n_out)
{
    CHECK_H(h);
    if (n < 0 || (n > 0 && !in4) || !(ratio > 0.f) || !(ratio <= 1.f) || knn < 3 || !(max_box_dim > 0.f) || seed < 0) {
        h->last_error = "sampling_surface_normal: bad arguments (0 < ratio <= 1, knn >= 3, maxBoxDim > 0, seed >= 0)"; return ICPMI_ERR_INVALID_ARG;
    }
    return ops_sampling_surface_normal(h, in4, n, ratio, knn, max_box_dim, seed, order_out, normals3_out, n_out);
}

icpmi_status icpmi_sampling_surface_normal_ex(icpmi_handle h, const float* in4, int64_t n, float ratio, int32_t knn, float max_box_dim, int32_t seed,
                                              int32_t method, int32_t* order_out, float* normals3_out, int64_t* n_out, float* mean3_out,
                                              int32_t* member_start_out, int32_t* member_count_out, int32_t* members_out)
{
    CHECK_H(h);
    if (n < 0 || (n > 0 && !in4) || !(ratio > 0.f) || !(ratio <= 1.f) || knn < 3 || !(max_box_dim > 0.f) || seed < 0 || (method != 0 && method != 1)) {
        h->last_error = "sampling_surface_normal: bad arguments (0 < ratio <= 1, knn >= 3, maxBoxDim > 0, seed >= 0, samplingMethod 0 or 1)"; return ICPMI_ERR_INVALID_ARG;
    }
    return ops_sampling_surface_normal_ex(h, in4, n, ratio, knn, max_box_dim, seed, method, order_out, normals3_out, n_out, mean3_out, member_start_out,
                                          member_count_out, members_out);
}

icpmi_status icpmi_octree_sample(icpmi_handle h, const float* in4, int64_t n, float max_size, int32_t max_points, int32_t method,
                                 int32_t* order_out, int32_t* leaf_of_out, int64_t* n_out)
{
    CHECK_H(h);
    if (n < 0 || (n > 0 && !in4) || !(max_size >= 0.f) || max_points < 0 || (method != 0 && method != 1)) { h->last_error = "octree_sample: bad arguments"; return ICPMI_ERR_INVALID_ARG; }
    return ops_octree_sample(h, in4, n, max_size, max_points, method, order_out, leaf_of_out, n_out);
}

icpmi_status icpmi_voxel_keep_first(icpmi_handle h, const float* in4, int64_t n, float edge, uint8_t* keep)
{
    return icpmi_voxel_keep(h, in4, n, edge, 0, keep);
}

static icpmi_status stage_chain_scalar(icpmi_handle h, const float* scan_scalar, int64_t n)
{
    if (!scan_scalar || n == 0) return ICPMI_OK;
    if (ensure_cap(h, &h->d_stage_s, &h->cap_stage_s, (size_t)n + 1) != ICPMI_OK) return ICPMI_ERR_HIP;
    HIP_TRY(h, hipMemcpyAsync(h->d_stage_s, scan_scalar, (size_t)n * sizeof(float), hipMemcpyHostToDevice, h->stream));
    return ICPMI_OK;
}

static bool chain_needs_pose(const icpmi_map_op* ops, int32_t n_ops)
{
    for (int32_t i = 0; ops && i < n_ops; ++i) if (ops[i].type == ICPMI_MOP_DYNAMIC_POINTS) return true;
    return false;
}

icpmi_status icpmi_map_update_chain(icpmi_handle h, const float* scan4, int64_t n, const float* scan_normals3, const float* scan_scalar,
                                    const float to_sensor[16], const float from_sensor[16], const icpmi_map_op* ops, int32_t n_ops,
                                    int32_t n_modules, int32_t* src_out, int64_t src_capacity, int64_t* identity_prefix, int64_t* new_m)
{
    CHECK_H(h);
    if (n < 0 || (n > 0 && !scan4) || (chain_needs_pose(ops, n_ops) && !to_sensor) || (from_sensor && !to_sensor)) { h->last_error = "map_update_chain: bad arguments"; return ICPMI_ERR_INVALID_ARG; }
    if (n > 0) {
        if (ensure_cap(h, &h->d_stage_in, &h->cap_stage_in, (size_t)n + 1) != ICPMI_OK) return ICPMI_ERR_HIP;
        HIP_TRY(h, hipMemcpyAsync(h->d_stage_in, scan4, (size_t)n * sizeof(float4), hipMemcpyHostToDevice, h->stream));
        if (scan_normals3) {
            if (ensure_cap(h, &h->d_stage_n3, &h->cap_stage_n3, (size_t)n * 3) != ICPMI_OK) return ICPMI_ERR_HIP;
            HIP_TRY(h, hipMemcpyAsync(h->d_stage_n3, scan_normals3, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice, h->stream));
        }
        icpmi_status s = stage_chain_scalar(h, scan_scalar, n);
        if (s != ICPMI_OK) return s;
    }
    return ops_map_update_chain(h, h->d_stage_in, n, scan_normals3 ? h->d_stage_n3 : nullptr, scan_scalar ? h->d_stage_s : nullptr, to_sensor,
                                from_sensor, ops, n_ops, n_modules, src_out, src_capacity, identity_prefix, new_m);
}

icpmi_status icpmi_map_update_chain_staged(icpmi_handle h, const float correction[16], const float* scan_scalar, const float to_sensor[16],
                                           const float from_sensor[16], const icpmi_map_op* ops, int32_t n_ops, int32_t n_modules,
                                           int32_t* src_out, int64_t src_capacity, int64_t* identity_prefix, int64_t* new_m)
{
    CHECK_H(h);
    if (!correction || (chain_needs_pose(ops, n_ops) && !to_sensor) || (from_sensor && !to_sensor)) { h->last_error = "map_update_chain_staged: bad arguments"; return ICPMI_ERR_INVALID_ARG; }
    if (h->scan_map_n <= 0) { h->last_error = "map_update_chain_staged: no scan staged by icpmi_register_prior"; return ICPMI_ERR_INVALID_ARG; }
    const int64_t n = h->scan_map_n;
    if (ensure_cap(h, &h->d_stage_in, &h->cap_stage_in, (size_t)n + 1) != ICPMI_OK) return ICPMI_ERR_HIP;
    icpmi_status s = ops_transform_dev(h, correction, h->d_scan_map, n, h->d_stage_in); // Mapper.cpp:221
    if (s == ICPMI_OK) s = stage_chain_scalar(h, scan_scalar, n);
    if (s != ICPMI_OK) return s;
    return ops_map_update_chain(h, h->d_stage_in, n, nullptr, scan_scalar ? h->d_stage_s : nullptr, to_sensor, from_sensor, ops, n_ops, n_modules,
                                src_out, src_capacity, identity_prefix, new_m);
}

icpmi_status icpmi_set_map_scalar(icpmi_handle h, const float* scalar, int64_t m)
{
    CHECK_H(h);
    if (!scalar) { h->last_error = "set_map_scalar: bad arguments"; return ICPMI_ERR_INVALID_ARG; }
    return ops_map_scalar(h, scalar, nullptr, m);
}

icpmi_status icpmi_get_map_scalar(icpmi_handle h, float* scalar_out, int64_t capacity)
{
    CHECK_H(h);
    if (h->m <= 0 || h->m_raw == 0) return ICPMI_OK; // no map, or an empty resident map (a chain removed every point): nothing to
                                                     // hand out -- like icpmi_get_map, which reports 0 points
    if (!scalar_out || capacity < h->m_raw) { h->last_error = "get_map_scalar: no map, or capacity too small"; return ICPMI_ERR_INVALID_ARG; }
    return ops_map_scalar(h, nullptr, scalar_out, h->m_raw);
}

icpmi_status icpmi_dynamic_points_update(icpmi_handle h, const icpmi_dynpts_params* prm, const float to_sensor[16], const float* in4,
                                         int64_t n, const float* map4, const float* map_normals3, int64_t m, float* prob_dynamic)
{
    CHECK_H(h);
    if (!prm || !to_sensor || n < 0 || m < 0 || (n > 0 && !in4) || (m > 0 && (!map4 || !map_normals3 || !prob_dynamic))) {
        h->last_error = "dynamic_points_update: bad arguments"; return ICPMI_ERR_INVALID_ARG;
    }
    if (!(prm->beam_half_angle > 0.f)) { h->last_error = "InvalidParameter: beamHalfAngle must be > 0"; return ICPMI_ERR_INVALID_ARG; }
    return ops_dynamic_points_update(h, prm, to_sensor, in4, n, map4, map_normals3, m, prob_dynamic);
}

icpmi_status icpmi_filter_points(icpmi_handle h, const float* in4, int64_t n, const icpmi_point_filter* filters, int32_t n_filters, uint8_t* keep)
{
    CHECK_H(h);
    if (n < 0 || n_filters < 0 || n_filters > ICPMI_MAX_POINT_FILTERS || (n_filters > 0 && !filters) || (n > 0 && (!in4 || !keep))) {
        h->last_error = "filter_points: bad arguments (at most 16 filters per call)"; return ICPMI_ERR_INVALID_ARG;
    }
    for (int32_t k = 0; k < n_filters; ++k) {
        const icpmi_point_filter& f = filters[k];
        if ((f.type != ICPMI_FILT_DISTANCE_LIMIT && f.type != ICPMI_FILT_BOUNDING_BOX) || (f.type == ICPMI_FILT_DISTANCE_LIMIT && (f.i < -1 || f.i > 2))) {
            h->last_error = "InvalidParameter: filter_points: unknown filter or dim outside [-1, 2]"; return ICPMI_ERR_INVALID_ARG;
        }
    }
    return ops_filter_points(h, in4, n, filters, n_filters, keep);
}

icpmi_status icpmi_bin_cells(icpmi_handle h, const float* pts4, int64_t n, float cell_size, int32_t* ijk3)
{
    CHECK_H(h);
    if (n < 0 || !(cell_size > 0.f) || (n > 0 && (!pts4 || !ijk3))) { h->last_error = "bin_cells: bad arguments"; return ICPMI_ERR_INVALID_ARG; }
    return ops_bin_cells(h, pts4, n, cell_size, ijk3);
}

} // extern "C"
