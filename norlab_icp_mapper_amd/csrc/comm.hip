// comm.hip -- the one real exchange of the multi-GPU path: every rank registers its own scan against its
// replica of the map; the points each rank accepted are all-gathered over RCCL (xGMI on an MI355X node) so that every
// replica appends the same set.  RCCL is loaded lazily (dlopen): a process that never creates a communicator does not
// need librccl.so at all, and the library binds to whichever copy the process already holds (PyTorch ships one).
#include "common.h"

#include <vector>

#include <cstdio>
#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    std::string error;
};

Rccl& rccl()
{
    static Rccl r;
    static bool tried = false;
    if (tried) return r;
    tried = true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (r.lib) break;
    }
    if (!r.lib) { r.error = std::string("cannot load librccl.so: ") + dlerror(); return r; }
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.lib, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.lib, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))dlsym(r.lib, "ncclAllGather");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
    r.CommCount = (decltype(r.CommCount))dlsym(r.lib, "ncclCommCount");       // optional: only icpmi_comm_info asks
    r.CommUserRank = (decltype(r.CommUserRank))dlsym(r.lib, "ncclCommUserRank");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.GetErrorString) {
        r.error = "librccl.so lacks an expected symbol";
        dlclose(r.lib); r.lib = nullptr;
    }
    return r;
}

} // namespace

#define RCCL_TRY(ctx, expr)                                                                        \
    do {                                                                                           \
        ncclResult_t r_ = (expr);                                                                  \
        if (r_ != ncclSuccess) {                                                                   \
            (ctx)->last_error = std::string(#expr) + ": " + rccl().GetErrorString(r_);             \
            return ICPMI_ERR_HIP;                                                                  \
        }                                                                                          \
    } while (0)

icpmi_status comm_unique_id(icpmi_comm_id* id, std::string& err)
{
    static_assert(sizeof(icpmi_comm_id) == sizeof(ncclUniqueId), "icpmi_comm_id must hold an ncclUniqueId");
    Rccl& r = rccl();
    if (!r.lib) { err = r.error; return ICPMI_ERR_HIP; }
    ncclUniqueId u;
    const ncclResult_t rc = r.GetUniqueId(&u);
    if (rc != ncclSuccess) { err = std::string("ncclGetUniqueId: ") + r.GetErrorString(rc); return ICPMI_ERR_HIP; }
    memcpy(id, &u, sizeof u);
    return ICPMI_OK;
}

// Loopback communicator (test hook; ICPMI_COMM_LOOPBACK=R in the environment AND icpmi_comm_init(..., n_ranks = 1, rank = 0)):
// R ranks simulated on ONE GPU -- every "rank" contributes this rank's own block, rank r's points moved by
// r * ICPMI_COMM_LOOPBACK_SHIFT metres along x; with ICPMI_COMM_LOOPBACK_RAGGED=1 simulated rank r hands in only the first
// floor(count * w[r % 5]) points of it, w = {1/2, 1, 1/4, 3/4, 0} (unequal and empty blocks; rank 0, "this" rank, is one of the
// small ones).  It exists so that the rank-ordered merge of icpmi_staged_merge_allgather (counts, padding, block offsets,
// rejection against the blocks of the lower ranks) can be checked against the oracle on a single-GPU box; RCCL is not touched.
// A real multi-rank job (n_ranks > 1) with the variable left set is refused: it would silently merge copies of its own block.
__global__ __launch_bounds__(256) void loop_shift_kernel(float4* __restrict__ p, size_t n, float dx)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i].x += dx;
}

// all simulated ranks' blocks in one launch (a real all-gather is one call too: R copies + R - 1 shift kernels would charge the
// loopback curve of bench.py for launches RCCL does not make)
__global__ __launch_bounds__(256) void loop_gather_kernel(const float4* __restrict__ send, float4* __restrict__ recv, size_t n, int R, float shift)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * (size_t)R) return;
    const size_t r = i / n;
    float4 p = send[i - r * n];
    if (r) p.x += (float)r * shift; // (rank 0: this rank's block as it is)
    recv[i] = p;
}

__global__ void loop_counts_kernel(const long long* __restrict__ mine, long long* __restrict__ all, int R, int ragged)
{
    const int r = threadIdx.x;
    if (r >= R) return;
    long long v = *mine;
    if (ragged && v > 0) {
        const int w = r % 5; // quarters: 2, 4, 1, 3, 0
        const long long q = w == 0 ? 2 : (w == 1 ? 4 : (w == 2 ? 1 : (w == 3 ? 3 : 0)));
        v = v * q / 4;
    }
    all[r] = v;
}

// the loopback communicator's ONE-collective exchange: block r = this rank's block with the header's count cut as loop_counts_kernel cuts
// it (ragged) and the points moved r * shift along x
__global__ __launch_bounds__(256) void loop_gather_blocks_kernel(const float4* __restrict__ send, float4* __restrict__ recv, size_t block4, int R, float shift, int ragged)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= block4 * (size_t)R) return;
    const size_t r = i / block4, e = i - r * block4;
    const float4 hdr = send[0];
    long long cnt = (long long)(int)__float_as_uint(hdr.x);
    if (ragged && cnt > 0) {
        const int w = (int)(r % 5); // quarters: 2, 4, 1, 3, 0 (loop_counts_kernel)
        const long long q = w == 0 ? 2 : (w == 1 ? 4 : (w == 2 ? 1 : (w == 3 ? 3 : 0)));
        cnt = cnt * q / 4;
    }
    if (e == 0) { recv[i] = make_float4(__uint_as_float((unsigned)(int)cnt), hdr.y, hdr.z, hdr.w); return; }
    if ((long long)e > cnt) return; // (padding: never read)
    float4 p = send[e];
    if (r) p.x += (float)r * shift;
    recv[i] = p;
}

// The exchange buffers of the one-collective epoch, sized for the communicator: send = 1 block, recv = n_ranks blocks, merged = n_ranks x
// merge_block points.  Called where no peer can be left waiting (comm_init; a single-rank handle: the first epoch).  ICPMI_MERGE_BLOCK:
// points per block (default 32 768; 0: the three-collective epoch).
icpmi_status merge_blocks_reserve(icpmi_ctx* c, int n_ranks)
{
    long long block_cfg = 32768; // (read per communicator, not once per process: the tests run both epochs in one process)
    { const char* e = getenv("ICPMI_MERGE_BLOCK"); if (e) block_cfg = atoll(e); if (block_cfg < 0) block_cfg = 0; }
    c->merge_block = 0;
    if (block_cfg == 0) return ICPMI_OK;
    if (n_ranks > 256) return ICPMI_OK; // (the R block headers arrive in 256 words of the host-mapped page, common.h: ICPMI_PROGRESS_HDR_WORD -- larger jobs keep the three-collective epoch)
    const size_t b4 = (size_t)block_cfg + 1;
    if (b4 * (size_t)n_ranks >= (1ull << 31)) return ICPMI_OK; // (the merge indexes the gathered span with 31 bits: such a job keeps the old epoch)
    if (ensure_cap(c, &c->d_merge_send, &c->cap_merge_send, b4 + 1) != ICPMI_OK) return ICPMI_ERR_HIP;
    if (ensure_cap(c, &c->d_merge_recv, &c->cap_merge_recv, b4 * (size_t)n_ranks + 1) != ICPMI_OK) return ICPMI_ERR_HIP;
    if (ensure_cap(c, &c->d_merged, &c->cap_merged, (size_t)block_cfg * (size_t)n_ranks + 1) != ICPMI_OK) return ICPMI_ERR_HIP;
    c->merge_block = block_cfg;
    return ICPMI_OK;
}

icpmi_status comm_init(icpmi_ctx* c, const icpmi_comm_id* id, int n_ranks, int rank)
{
    if (const char* lb = getenv("ICPMI_COMM_LOOPBACK")) {
        const int R = atoi(lb);
        if (R > 1) {
            if (n_ranks != 1) {
                c->last_error = "comm_init: ICPMI_COMM_LOOPBACK (a single-process test hook) is set in a job of more than one rank";
                return ICPMI_ERR_INVALID_ARG;
            }
            if (R > 256) { c->last_error = "comm_init: ICPMI_COMM_LOOPBACK > 256"; return ICPMI_ERR_INVALID_ARG; }
            if (c->comm) (void)comm_destroy(c);
            const char* sh = getenv("ICPMI_COMM_LOOPBACK_SHIFT");
            const char* rg = getenv("ICPMI_COMM_LOOPBACK_RAGGED");
            c->comm = nullptr; c->comm_ranks = R; c->comm_rank = 0; c->comm_loop_shift = sh ? (float)atof(sh) : 0.f;
            c->comm_loop_ragged = rg && atoi(rg) != 0;
            fprintf(stderr, "[icpmi] loopback communicator active: %d simulated ranks on one GPU, no RCCL (ICPMI_COMM_LOOPBACK)\n", R);
            if (ensure_cap(c, &c->d_comm_cnt, &c->cap_comm_cnt, (size_t)2 * R + 16) != ICPMI_OK) return ICPMI_ERR_HIP;
            return merge_blocks_reserve(c, R);
        }
    }
    Rccl& r = rccl();
    if (!r.lib) { c->last_error = r.error; return ICPMI_ERR_HIP; }
    if (c->comm) { RCCL_TRY(c, r.CommDestroy((ncclComm_t)c->comm)); c->comm = nullptr; }
    // the words of the epoch's count / ready exchanges: allocated here so that no allocation can fail between two collectives
    if (ensure_cap(c, &c->d_comm_cnt, &c->cap_comm_cnt, (size_t)2 * n_ranks + 16) != ICPMI_OK) return ICPMI_ERR_HIP;
    if (merge_blocks_reserve(c, n_ranks) != ICPMI_OK) return ICPMI_ERR_HIP; // (a failure here is reported before any collective exists)
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    ncclComm_t comm = nullptr;
    RCCL_TRY(c, r.CommInitRank(&comm, n_ranks, u, rank));
    c->comm = comm; c->comm_ranks = n_ranks; c->comm_rank = rank; c->comm_loop_shift = 0.f; c->comm_loop_ragged = false;
    // The block size of the one-collective epoch must be the same on every rank (ICPMI_MERGE_BLOCK is read per rank; an all-gather of blocks of
    // different sizes is undefined): gathered once, here, where every rank is inside comm_init anyway.  "All equal" is the same verdict on
    // every rank; otherwise every rank keeps the three-collective epoch.
    if (n_ranks > 1) {
        long long mine = (long long)c->merge_block;
        { const icpmi_status us = upload_small(c, c->d_comm_cnt, &mine, sizeof mine); if (us != ICPMI_OK) return us; }
        const icpmi_status gs = comm_allgather(c, c->d_comm_cnt, c->d_comm_cnt + 8, 1, false);
        if (gs != ICPMI_OK) return gs;
        std::vector<long long> all((size_t)n_ranks);
        if (read_back(c, all.data(), c->d_comm_cnt + 8, all.size() * sizeof(long long)) != ICPMI_OK) return ICPMI_ERR_HIP;
        for (long long v : all) if (v != mine) { c->merge_block = 0; break; }
    }
    return ICPMI_OK;
}

icpmi_status comm_destroy(icpmi_ctx* c)
{
    c->merge_block = 0; // (sized per communicator: the next one, or the single-rank handle's first epoch, reserves again)
    if (!c->comm) { c->comm_ranks = 1; c->comm_rank = 0; c->comm_loop_shift = 0.f; c->comm_loop_ragged = false; return ICPMI_OK; }
    Rccl& r = rccl();
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (r.lib) (void)r.CommDestroy((ncclComm_t)c->comm);
    c->comm = nullptr; c->comm_ranks = 1; c->comm_rank = 0;
    return ICPMI_OK;
}

// what the communicator itself reports (RCCL: ncclCommCount / ncclCommUserRank); kind: 0 none (single rank), 1 RCCL, 2 loopback
icpmi_status comm_info(icpmi_ctx* c, int* n_ranks, int* rank, int* kind)
{
    *n_ranks = c->comm_ranks; *rank = c->comm_rank; *kind = c->comm ? 1 : (c->comm_ranks > 1 ? 2 : 0);
    if (c->comm) {
        Rccl& r = rccl();
        if (r.CommCount && r.CommUserRank) {
            RCCL_TRY(c, r.CommCount((ncclComm_t)c->comm, n_ranks));
            RCCL_TRY(c, r.CommUserRank((ncclComm_t)c->comm, rank));
        }
    }
    return ICPMI_OK;
}

// all-gather of `count` elements per rank on the handle's stream (device buffers); no communicator = one rank = a copy
icpmi_status comm_allgather(icpmi_ctx* c, const void* d_send, void* d_recv, size_t count, bool is_float)
{
    if (!c->comm) {
        const size_t bytes = count * (is_float ? sizeof(float) : sizeof(long long));
        if (!is_float && count == 1 && c->comm_ranks > 1) { // the loopback communicator's count exchange (possibly ragged)
            hipLaunchKernelGGL(loop_counts_kernel, dim3(1), dim3(256), 0, c->stream, (const long long*)d_send, (long long*)d_recv, c->comm_ranks,
                               c->comm_loop_ragged ? 1 : 0);
            HIP_TRY(c, hipGetLastError());
            return ICPMI_OK;
        }
        if (is_float && c->comm_ranks > 1 && count >= 4 && (count & 3) == 0 && d_recv != d_send) { // the loopback communicator's point exchange
            const size_t np = count / 4;
            hipLaunchKernelGGL(loop_gather_kernel, dim3((unsigned)((np * c->comm_ranks + 255) / 256)), dim3(256), 0, c->stream, (const float4*)d_send,
                               (float4*)d_recv, np, c->comm_ranks, c->comm_loop_shift);
            HIP_TRY(c, hipGetLastError());
            return ICPMI_OK;
        }
        for (int r = 0; r < c->comm_ranks; ++r) { // one rank, or the loopback communicator's R copies
            char* dst = (char*)d_recv + (size_t)r * bytes;
            if (dst != (const char*)d_send) HIP_TRY(c, hipMemcpyAsync(dst, d_send, bytes, hipMemcpyDeviceToDevice, c->stream));
            if (is_float && r > 0 && c->comm_loop_shift != 0.f && count >= 4)
                hipLaunchKernelGGL(loop_shift_kernel, dim3((unsigned)((count / 4 + 255) / 256)), dim3(256), 0, c->stream, (float4*)dst, count / 4,
                                   (float)r * c->comm_loop_shift);
        }
        HIP_TRY(c, hipGetLastError());
        return ICPMI_OK;
    }
    RCCL_TRY(c, rccl().AllGather(d_send, d_recv, count, is_float ? ncclFloat : ncclInt64, (ncclComm_t)c->comm, c->stream));
    return ICPMI_OK;
}

icpmi_status comm_allgather_blocks(icpmi_ctx* c, const float4* d_send, float4* d_recv, size_t block4)
{
    if (c->comm) {
        RCCL_TRY(c, rccl().AllGather(d_send, d_recv, block4 * 4, ncclFloat, (ncclComm_t)c->comm, c->stream));
        return ICPMI_OK;
    }
    if (c->comm_ranks > 1) { // loopback
        hipLaunchKernelGGL(loop_gather_blocks_kernel, dim3((unsigned)((block4 * c->comm_ranks + 255) / 256)), dim3(256), 0, c->stream, d_send, d_recv, block4,
                           c->comm_ranks, c->comm_loop_shift, c->comm_loop_ragged ? 1 : 0);
        HIP_TRY(c, hipGetLastError());
        return ICPMI_OK;
    }
    // one rank: its own block is the gathered set (header + the points handed in: the padding is never read)
    HIP_TRY(c, hipMemcpyAsync(d_recv, d_send, block4 * sizeof(float4), hipMemcpyDeviceToDevice, c->stream));
    return ICPMI_OK;
}
