// tsq_comm.hip — the multi-GPU exchange behind the C-ABI: hash-radix redistribute of device-resident columns over
// RCCL send/recv (xGMI), plus the small all-reduces a distributed operator needs (joined-row counts, timing).
//
// Replaces (reference): the worker dispatch of HashJoinExec (fetchOuterSideChunks -> join workers, executor/join.go:160-231)
// and HashAggExec's partial -> final shuffle (shuffleIntermData, executor/aggregate.go:352-356) ACROSS GPUs: equi-join and
// GROUP BY are partitionable by any function of the key, so rank(key) = tsq_key_rank(key word, world) owns a key and the
// single-GPU operator runs unchanged on what a rank owns.  One process per GPU; every rank calls the same sequence.
//
//   tsq_redistribute(comm, cols, key, slot):
//     ctx stream : tsq_radix_split  -> the rows grouped by destination rank (run p = rows for rank p) + host counts
//     comm stream: ncclAllGather of the count vectors (world x world int64) -> host; then ONE group of
//                  ncclSend(run p -> p) / ncclRecv(<- p) per column and peer: an all-to-all(v) that keeps all seven xGMI
//                  links of a GPU busy at once (xGMI is point to point: a ring would be bound by one link)
//   tsq_redistribute_wait(comm, slot): the ctx stream waits for that exchange — called right before the consumer
//     (tsq_join_probe_push / tsq_agg_push of the received columns) is queued, so that the exchange of piece c + 1 runs
//     on the wire while piece c is probed.
// RCCL is loaded with dlopen at the first tsq_comm_* call: single-GPU users (and the CPU-only symbol check) never need it.
#include "tsq_internal.h"
#include "tsq_comm_plan.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#define TSQ_COMM_SLOTS 8
#define TSQ_SPLIT_MAX_PARTS 64  // = tsq_split.hip
#include <memory>
#define TSQ_MAGIC_COMM 0x7473714du /* 'tsqM' */

namespace {

struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;      // (the three below are optional: tsq_comm_info reports -1 without them)
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    std::string err;
};

RcclApi* rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (api.lib) break;
        }
        if (!api.lib) {
            api.err = std::string("dlopen(librccl): ") + (dlerror() ? dlerror() : "not found");
            return;
        }
        bool ok = true;
        auto sym = [&](const char* n) -> void* {
            void* p = dlsym(api.lib, n);
            if (!p) {
                ok = false;
                api.err = std::string("librccl lacks ") + n;
            }
            return p;
        };
        api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
        api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
        api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
        api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
        api.Send = (decltype(api.Send))sym("ncclSend");
        api.Recv = (decltype(api.Recv))sym("ncclRecv");
        api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
        api.AllReduce = (decltype(api.AllReduce))sym("ncclAllReduce");
        api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
        if (ok) {
            api.CommCount = (decltype(api.CommCount))dlsym(api.lib, "ncclCommCount");
            api.CommUserRank = (decltype(api.CommUserRank))dlsym(api.lib, "ncclCommUserRank");
            api.GetVersion = (decltype(api.GetVersion))dlsym(api.lib, "ncclGetVersion");
        }
        if (!ok) {
            dlclose(api.lib);
            api.lib = nullptr;
        }
    });
    return &api;
}

}  // namespace

// words one rank contributes to the count exchange at most: its row counts per destination, its null-bitmap mask, and for every
// var-len column its BYTE counts per destination
static inline size_t comm_lmax(int world) { return (size_t)world + 1 + (size_t)TSQ_MAX_COLS * world; }
// ... and the count exchange of up to TSQ_COMM_SLOTS prepared pieces at once: [own vectors | world x gathered vectors]
static inline size_t comm_words(int world) { return (size_t)(world + 1) * comm_lmax(world) * TSQ_COMM_SLOTS; }

struct tsq_comm {
    tsq_handle_hdr hdr;
    tsq_ctx* ctx = nullptr;
    int32_t rank = 0, world = 1;
    ncclComm_t nccl = nullptr;
    hipStream_t xs = nullptr;  // the exchange runs here, next to the operators on ctx->stream
    // L = world + 1 words per rank: its send counts, then the bit mask of its columns that carry a null bitmap.
    DevBuf cnt_dev;            // [L] this rank's vector | [world * L] gathered vectors | 8 words of all-reduce scratch
    uint64_t* cnt_host = nullptr;  // pinned mirror
    struct Slot {
        std::vector<DevBuf> send, recv;  // per column
        // a nullable column travels with one NOT-NULL byte per row (a run starts at an arbitrary bit of the sender's
        // bitmap and lands at an arbitrary bit of the receiver's): sendbm = the split's packed bitmap, sendnn / recvnn =
        // the byte flags on the wire, recvbm = the received column's packed bitmap
        std::vector<DevBuf> sendbm, sendnn, recvnn, recvbm;
        // a var-len column: send = the split's data bytes, sendoffs = the split's offsets[nrows + 1]; every run travels as its slice
        // of the offsets (rows + 1 entries, landing in recvtmp) and its bytes; recvoffs = the received column's offsets, rebased
        std::vector<DevBuf> sendoffs, recvtmp, recvoffs;
        hipEvent_t split_done = nullptr, xchg_done = nullptr;
        bool pending = false;
        bool broadcast = false;  // key_mode 2: the piece is all-gathered (every rank receives every rank's rows)
        // a piece between tsq_redistribute_prepare and tsq_redistribute_issue: its columns (split into `send`), its count vector
        // (tsq_comm_plan.h: L words) and, after tsq_redistribute_counts, every rank's vector
        int state = 0;  // 0: idle, 1: prepared, 2: counted
        std::vector<tsq_col> cols;
        int64_t nrows = 0;
        int n_var = 0;
        int var_of[TSQ_MAX_COLS];
        std::vector<uint64_t> vec, M;
    } slot[TSQ_COMM_SLOTS];
};

#define TSQ_NCCL(h, expr)                                                                                          \
    do {                                                                                                           \
        ncclResult_t _r = (expr);                                                                                  \
        if (_r != ncclSuccess) return tsq_fail((h), TSQ_ERR_HIP, std::string(#expr) + ": " + rccl()->GetErrorString(_r)); \
    } while (0)

TSQ_API tsq_status tsq_comm_unique_id(uint8_t* id_out) {
    if (!id_out) return tsq_fail(nullptr, TSQ_ERR_INVALID, "tsq_comm_unique_id: NULL argument");
    RcclApi* r = rccl();
    if (!r->lib) return tsq_fail(nullptr, TSQ_ERR_UNSUPPORTED, r->err);
    static_assert(sizeof(ncclUniqueId) <= TSQ_COMM_ID_BYTES, "unique id size");
    ncclUniqueId id;
    TSQ_NCCL(nullptr, r->GetUniqueId(&id));
    memset(id_out, 0, TSQ_COMM_ID_BYTES);
    memcpy(id_out, &id, sizeof id);
    return TSQ_OK;
}

TSQ_API tsq_status tsq_comm_create(tsq_ctx* ctx, int32_t rank, int32_t world, const uint8_t* id, tsq_comm** out) {
    tsq_ctx_lock _api_lock(ctx);
    if (!ctx || !out || !id) return tsq_fail(nullptr, TSQ_ERR_INVALID, "tsq_comm_create: NULL argument");
    *out = nullptr;
    tsq_handle_hdr* ch = &ctx->hdr;
    if (world < 1 || world > TSQ_SPLIT_MAX_PARTS || rank < 0 || rank >= world) return tsq_fail(ch, TSQ_ERR_INVALID, "tsq_comm_create: bad rank / world size");
    RcclApi* r = rccl();
    if (!r->lib) return tsq_fail(ch, TSQ_ERR_UNSUPPORTED, r->err);
    TSQ_HIP(ch, hipSetDevice(ctx->device));
    std::unique_ptr<tsq_comm> c(new tsq_comm());
    c->hdr.magic = TSQ_MAGIC_COMM;
    c->ctx = ctx;
    c->rank = rank;
    c->world = world;
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    TSQ_NCCL(ch, r->CommInitRank(&c->nccl, world, uid, rank));
    // from here on a failure must give back what exists already (the communicator, the stream, the events): tsq_comm_destroy does
    auto fail = [&](tsq_status st) {
        const std::string msg = ch->err;
        tsq_comm_destroy(c.release());
        ch->err = msg;
        return st;
    };
    const size_t words = comm_words(world) + 8;
    {
        tsq_status st = TSQ_OK;
        hipError_t e = hipStreamCreateWithFlags(&c->xs, hipStreamNonBlocking);
        if (e == hipSuccess) st = c->cnt_dev.reserve(ctx, ch, words * 8);
        if (e == hipSuccess && st == TSQ_OK) e = hipHostMalloc((void**)&c->cnt_host, words * 8, hipHostMallocDefault);
        for (auto& s : c->slot) {
            if (e == hipSuccess && st == TSQ_OK) e = hipEventCreateWithFlags(&s.split_done, hipEventDisableTiming);
            if (e == hipSuccess && st == TSQ_OK) e = hipEventCreateWithFlags(&s.xchg_done, hipEventDisableTiming);
        }
        if (st != TSQ_OK) return fail(st);
        if (e != hipSuccess) return fail(tsq_fail(ch, TSQ_ERR_HIP, std::string("tsq_comm_create: ") + hipGetErrorString(e)));
    }
    *out = c.release();
    return TSQ_OK;
}

TSQ_API void tsq_comm_destroy(tsq_comm* c) {
    tsq_ctx_lock _api_lock(tsq_ctx_of(c, TSQ_MAGIC_COMM));
    if (!c || c->hdr.magic != TSQ_MAGIC_COMM) return;
    (void)hipSetDevice(c->ctx->device);
    if (c->xs) (void)hipStreamSynchronize(c->xs);
    (void)hipStreamSynchronize(c->ctx->stream);
    for (auto& s : c->slot) {
        for (auto* v : {&s.send, &s.recv, &s.sendbm, &s.sendnn, &s.recvnn, &s.recvbm, &s.sendoffs, &s.recvtmp, &s.recvoffs})
            for (auto& b : *v) b.release();
        if (s.split_done) (void)hipEventDestroy(s.split_done);
        if (s.xchg_done) (void)hipEventDestroy(s.xchg_done);
    }
    c->cnt_dev.release();
    if (c->cnt_host) (void)hipHostFree(c->cnt_host);
    if (c->nccl) (void)rccl()->CommDestroy(c->nccl);
    if (c->xs) (void)hipStreamDestroy(c->xs);
    c->hdr.magic = 0;
    delete c;
}

namespace {

// packed bitmap (bit = 1: NOT NULL, util/chunk/column.go:89-92) <-> one byte per row, on the exchange stream
__global__ void __launch_bounds__(256) k_offsets_shift(const int64_t* src, int64_t* dst, int64_t n, int64_t delta) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i] + delta;
}
__global__ void __launch_bounds__(256) k_bits_to_bytes(const uint8_t* bits, uint8_t* bytes, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) bytes[i] = (bits[i >> 3] >> (i & 7)) & 1;
}
__global__ void __launch_bounds__(256) k_bytes_to_bits(const uint8_t* bytes, uint8_t* bits, int64_t n) {
    const int64_t nb = (n + 7) >> 3;
    for (int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x; b < nb; b += (int64_t)gridDim.x * 256) {
        uint32_t v = 0;
        for (int k = 0; k < 8 && b * 8 + k < n; k++) v |= bytes[b * 8 + k] ? (1u << k) : 0u;
        bits[b] = (uint8_t)v;
    }
}

// all-reduce of n (<= 8) 8-byte host words through the scratch behind the count matrix; synchronises the exchange stream
tsq_status allreduce8(tsq_comm* c, void* inout, int32_t n, ncclDataType_t dt, int32_t op) {
    tsq_handle_hdr* h = &c->hdr;
    if (!inout || n < 1 || n > 8 || op < 0 || op > 2) return tsq_fail(h, TSQ_ERR_INVALID, "all-reduce: 1..8 words, op 0 (sum) / 1 (max) / 2 (min)");
    TSQ_HIP(h, hipSetDevice(c->ctx->device));
    const size_t off = comm_words(c->world);
    uint64_t* dev = c->cnt_dev.as<uint64_t>() + off;
    uint64_t* host = c->cnt_host + off;
    TSQ_HIP(h, hipStreamSynchronize(c->ctx->stream));  // a barrier-like call: everything this rank queued is done
    memcpy(host, inout, (size_t)n * 8);
    TSQ_HIP(h, hipMemcpyAsync(dev, host, (size_t)n * 8, hipMemcpyHostToDevice, c->xs));
    const ncclRedOp_t rop = op == 0 ? ncclSum : (op == 1 ? ncclMax : ncclMin);
    TSQ_NCCL(h, rccl()->AllReduce(dev, dev, (size_t)n, dt, rop, c->nccl, c->xs));
    TSQ_HIP(h, hipMemcpyAsync(host, dev, (size_t)n * 8, hipMemcpyDeviceToHost, c->xs));
    TSQ_HIP(h, hipStreamSynchronize(c->xs));
    memcpy(inout, host, (size_t)n * 8);
    return TSQ_OK;
}

}  // namespace

TSQ_API tsq_status tsq_comm_allreduce_i64(tsq_comm* c, int64_t* inout, int32_t n, int32_t op) {
    tsq_ctx_lock _api_lock(tsq_ctx_of(c, TSQ_MAGIC_COMM));
    if (!c || c->hdr.magic != TSQ_MAGIC_COMM) return TSQ_ERR_INVALID;
    return allreduce8(c, inout, n, ncclInt64, op);
}
TSQ_API tsq_status tsq_comm_allreduce_f64(tsq_comm* c, double* inout, int32_t n, int32_t op) {
    tsq_ctx_lock _api_lock(tsq_ctx_of(c, TSQ_MAGIC_COMM));
    if (!c || c->hdr.magic != TSQ_MAGIC_COMM) return TSQ_ERR_INVALID;
    return allreduce8(c, inout, n, ncclDouble, op);
}
// ---- collective steps of an operator (declared in tsq_internal.h)
bool tsq_comm_usable(const tsq_comm* c, const tsq_ctx* ctx) { return c && c->hdr.magic == TSQ_MAGIC_COMM && c->ctx == ctx && c->nccl; }
int32_t tsq_comm_world_size(const tsq_comm* c) { return c->world; }
tsq_status tsq_comm_allreduce_host_i64(tsq_comm* c, int64_t* inout, int32_t n, int32_t op) { return allreduce8(c, inout, n, ncclInt64, op); }
tsq_status tsq_comm_allreduce_dev_sum(tsq_comm* c, void* buf, size_t count, int elem_bytes) {
    tsq_handle_hdr* h = &c->hdr;
    if (!buf || (elem_bytes != 1 && elem_bytes != 4)) return tsq_fail(h, TSQ_ERR_INVALID, "device all-reduce: 1- or 4-byte elements");
    TSQ_HIP(h, hipSetDevice(c->ctx->device));
    TSQ_HIP(h, hipStreamSynchronize(c->ctx->stream));  // the buffer's producer kernels ran on the context's stream
    if (count) TSQ_NCCL(h, rccl()->AllReduce(buf, buf, count, elem_bytes == 1 ? ncclUint8 : ncclUint32, ncclSum, c->nccl, c->xs));
    TSQ_HIP(h, hipStreamSynchronize(c->xs));
    return TSQ_OK;
}

TSQ_API tsq_status tsq_comm_barrier(tsq_comm* c) {
    tsq_ctx_lock _api_lock(tsq_ctx_of(c, TSQ_MAGIC_COMM));
    if (!c || c->hdr.magic != TSQ_MAGIC_COMM) return TSQ_ERR_INVALID;
    int64_t one = 1;
    return allreduce8(c, &one, 1, ncclInt64, 0);
}

// what the communicator itself says it is: ranks and this rank as RCCL counts them (ncclCommCount / ncclCommUserRank), the library's
// version — the bench line carries them so that a multi-GPU number can be checked against the collective library, not the launcher
TSQ_API tsq_status tsq_comm_info(tsq_comm* c, int32_t* rank_out, int32_t* nranks_out, int32_t* rccl_version_out) {
    tsq_ctx_lock _api_lock(tsq_ctx_of(c, TSQ_MAGIC_COMM));
    if (!c || c->hdr.magic != TSQ_MAGIC_COMM) return TSQ_ERR_INVALID;
    RcclApi* r = rccl();
    int v = -1;
    if (rank_out) {
        v = -1;
        if (r->CommUserRank && c->nccl) TSQ_NCCL(&c->hdr, r->CommUserRank(c->nccl, &v));
        *rank_out = v;
    }
    if (nranks_out) {
        v = -1;
        if (r->CommCount && c->nccl) TSQ_NCCL(&c->hdr, r->CommCount(c->nccl, &v));
        *nranks_out = v;
    }
    if (rccl_version_out) {
        v = -1;
        if (r->GetVersion) TSQ_NCCL(&c->hdr, r->GetVersion(&v));
        *rccl_version_out = v;
    }
    return TSQ_OK;
}

namespace {

// ---- stage 1: the piece is split on the operator stream into the slot's send buffers; its count vector is known on the host
tsq_status comm_prepare(tsq_comm* c, const tsq_col* cols, int32_t n_cols, int32_t key_col, int32_t key_mode, int64_t nrows, int32_t slot) {
    tsq_handle_hdr* h = &c->hdr;
    tsq_ctx* ctx = c->ctx;
    if (!cols || n_cols < 1 || n_cols > TSQ_MAX_COLS || key_col < 0 || key_col >= n_cols || nrows < 0 || slot < 0 || slot >= TSQ_COMM_SLOTS)
        return tsq_fail(h, TSQ_ERR_INVALID, "tsq_redistribute: bad arguments");
    tsq_comm::Slot& s = c->slot[slot];
    uint64_t my_mask = 0;
    int n_var = 0;  // column -> its index among the var-len columns
    for (int i = 0; i < n_cols; i++) {
        if (!(cols[i].flags & TSQ_COL_DEVICE)) return tsq_fail(h, TSQ_ERR_INVALID, "tsq_redistribute: columns must be device resident");
        if (cols[i].type < TSQ_I64 || cols[i].type > TSQ_BYTES) return tsq_fail(h, TSQ_ERR_INVALID, "tsq_redistribute: unknown column type");
        if (cols[i].type == TSQ_BYTES && !cols[i].offsets) return tsq_fail(h, TSQ_ERR_INVALID, "tsq_redistribute: a var-len column needs offsets");
        if (cols[i].null_bitmap) my_mask |= 1ull << i;
        s.var_of[i] = cols[i].type == TSQ_BYTES ? n_var++ : -1;
    }
    const int* var_of = s.var_of;
    TSQ_HIP(h, hipSetDevice(ctx->device));
    const int W = c->world, L = W + 1 + n_var * W;
    if (s.pending) TSQ_HIP(h, hipStreamSynchronize(c->xs));  // the previous exchange of this slot (its buffers are rewritten below)
    s.pending = false;
    s.state = 0;
    for (auto* v : {&s.send, &s.recv, &s.sendbm, &s.sendnn, &s.recvnn, &s.recvbm, &s.sendoffs, &s.recvtmp, &s.recvoffs}) v->resize(n_cols);
    // ---- the data bytes of the var-len columns (the split writes as many as it reads)
    int64_t in_bytes[TSQ_MAX_COLS] = {0};
    if (n_var && nrows > 0) {
        for (int i = 0; i < n_cols; i++)
            if (var_of[i] >= 0) TSQ_HIP(h, hipMemcpyAsync(ctx->pinned + i, cols[i].offsets + nrows, 8, hipMemcpyDeviceToHost, ctx->stream));
        TSQ_HIP(h, hipStreamSynchronize(ctx->stream));
        for (int i = 0; i < n_cols; i++)
            if (var_of[i] >= 0) in_bytes[i] = (int64_t)ctx->pinned[i];
    }
    // ---- split on the operator stream (ends with the host counts: tsq_radix_split synchronises).  Rows with a NULL key go
    // to rank 0 (they never join; GROUP BY makes them one group)
    std::vector<tsq_col> sc(n_cols);
    for (int i = 0; i < n_cols; i++) {
        const bool var = var_of[i] >= 0;
        const size_t es = var ? 0 : tsq_elem_size(cols[i].type);
        TSQ_TRY(s.send[i].reserve(ctx, h, (var ? (size_t)in_bytes[i] : (size_t)std::max<int64_t>(nrows, 1) * es) + 64));
        sc[i] = cols[i];
        sc[i].data = s.send[i].p;
        sc[i].null_bitmap = nullptr;
        if (var) {
            TSQ_TRY(s.sendoffs[i].reserve(ctx, h, ((size_t)nrows + 1) * 8 + 64));
            sc[i].offsets = s.sendoffs[i].as<int64_t>();
        }
        if (cols[i].null_bitmap) {
            TSQ_TRY(s.sendbm[i].reserve(ctx, h, tsq_bitmap_bytes(nrows) + 64));
            sc[i].null_bitmap = s.sendbm[i].as<uint8_t>();
        }
    }
    int64_t sendc[TSQ_SPLIT_MAX_PARTS] = {0};
    s.broadcast = key_mode == TSQ_KEYMODE_BROADCAST;
    if (s.broadcast) {
        // an all-gather of the columns: no split — the send buffers hold the columns once, every rank gets all rows (tsq_comm_plan.h)
        for (int i = 0; i < n_cols && nrows > 0; i++) {
            const bool var = var_of[i] >= 0;
            const size_t bytes = var ? (size_t)in_bytes[i] : (size_t)nrows * tsq_elem_size(cols[i].type);
            if (bytes) TSQ_HIP(h, hipMemcpyAsync(s.send[i].p, cols[i].data, bytes, hipMemcpyDeviceToDevice, ctx->stream));
            if (var) TSQ_HIP(h, hipMemcpyAsync(s.sendoffs[i].p, cols[i].offsets, ((size_t)nrows + 1) * 8, hipMemcpyDeviceToDevice, ctx->stream));
            if (cols[i].null_bitmap) TSQ_HIP(h, hipMemcpyAsync(s.sendbm[i].p, cols[i].null_bitmap, tsq_bitmap_bytes(nrows), hipMemcpyDeviceToDevice, ctx->stream));
        }
        for (int p = 0; p < W; p++) sendc[p] = nrows;
    } else if (nrows > 0) {
        tsq_status st = tsq_radix_split(ctx, cols, n_cols, key_col, key_mode, nrows, W, sc.data(), sendc);
        if (st != TSQ_OK) return tsq_fail(h, st, ctx->hdr.err);
    }
    // ---- counts, nullable-column mask and the var-len columns' byte counts: this rank's vector (tsq_comm_plan.h)
    s.vec.assign((size_t)L, 0);
    for (int p = 0; p < W; p++) s.vec[(size_t)p] = (uint64_t)sendc[p];
    s.vec[(size_t)W] = my_mask;
    if (n_var && nrows > 0) {  // byte boundaries of the runs: offsets[first row of run p]
        std::vector<int64_t> bounds((size_t)n_var * (W + 1));
        for (int i = 0; i < n_cols; i++) {
            if (var_of[i] < 0) continue;
            int64_t row = 0;
            for (int p = 0; p <= W; p++) {
                if (s.broadcast) {  // every run is the whole column
                    bounds[(size_t)var_of[i] * (W + 1) + p] = (int64_t)p * in_bytes[i];
                    continue;
                }
                TSQ_HIP(h, hipMemcpyAsync(&bounds[(size_t)var_of[i] * (W + 1) + p], s.sendoffs[i].as<int64_t>() + row, 8, hipMemcpyDeviceToHost, ctx->stream));
                if (p < W) row += sendc[p];
            }
        }
        TSQ_HIP(h, hipStreamSynchronize(ctx->stream));
        for (int v = 0; v < n_var; v++)
            for (int p = 0; p < W; p++) s.vec[(size_t)W + 1 + (size_t)v * W + p] = (uint64_t)(bounds[(size_t)v * (W + 1) + p + 1] - bounds[(size_t)v * (W + 1) + p]);
    }
    s.cols.assign(cols, cols + n_cols);
    s.nrows = nrows;
    s.n_var = n_var;
    s.state = 1;
    return TSQ_OK;
}

// ---- stage 2: ONE all-gather for the count vectors of every listed (prepared) piece: each rank learns every piece's world x L matrix
tsq_status comm_counts(tsq_comm* c, const int32_t* slots, int32_t n_slots) {
    tsq_handle_hdr* h = &c->hdr;
    if (!slots || n_slots < 1 || n_slots > TSQ_COMM_SLOTS) return tsq_fail(h, TSQ_ERR_INVALID, "tsq_redistribute_counts: 1..8 slots");
    const int W = c->world;
    size_t Lsum = 0;
    for (int k = 0; k < n_slots; k++) {
        if (slots[k] < 0 || slots[k] >= TSQ_COMM_SLOTS || c->slot[slots[k]].state != 1) return tsq_fail(h, TSQ_ERR_INVALID, "tsq_redistribute_counts: a slot that was not prepared");
        for (int q = 0; q < k; q++)
            if (slots[q] == slots[k]) return tsq_fail(h, TSQ_ERR_INVALID, "tsq_redistribute_counts: a slot listed twice");
        Lsum += c->slot[slots[k]].vec.size();
    }
    if ((size_t)(W + 1) * Lsum > comm_words(W)) return tsq_fail(h, TSQ_ERR_INVALID, "tsq_redistribute_counts: count vectors too long");
    TSQ_HIP(h, hipSetDevice(c->ctx->device));
    uint64_t* cd = c->cnt_dev.as<uint64_t>();
    std::vector<const std::vector<uint64_t>*> vecs;
    std::vector<size_t> Ls;
    for (int k = 0; k < n_slots; k++) {
        vecs.push_back(&c->slot[slots[k]].vec);
        Ls.push_back(c->slot[slots[k]].vec.size());
    }
    (void)tsq_comm_pack_counts(vecs, c->cnt_host);
    TSQ_HIP(h, hipMemcpyAsync(cd, c->cnt_host, Lsum * 8, hipMemcpyHostToDevice, c->xs));
    TSQ_NCCL(h, rccl()->AllGather(cd, cd + Lsum, Lsum, ncclInt64, c->nccl, c->xs));
    TSQ_HIP(h, hipMemcpyAsync(c->cnt_host + Lsum, cd + Lsum, (size_t)W * Lsum * 8, hipMemcpyDeviceToHost, c->xs));
    TSQ_HIP(h, hipStreamSynchronize(c->xs));
    const uint64_t* G = c->cnt_host + Lsum;  // rank q's vectors, piece after piece, at G + q * Lsum (tsq_comm_plan.h)
    for (int k = 0; k < n_slots; k++) {
        tsq_comm::Slot& s = c->slot[slots[k]];
        s.M = tsq_comm_unpack_counts(G, W, Ls, (size_t)k);
        s.state = 2;
    }
    return TSQ_OK;
}

// ---- stage 3: the exchange of one counted piece, queued on the communicator's stream (no host synchronisation)
tsq_status comm_issue(tsq_comm* c, int32_t slot, tsq_col* out_cols, int32_t n_out, int64_t* nrows_out) {
    tsq_handle_hdr* h = &c->hdr;
    tsq_ctx* ctx = c->ctx;
    if (slot < 0 || slot >= TSQ_COMM_SLOTS || !out_cols || !nrows_out) return tsq_fail(h, TSQ_ERR_INVALID, "tsq_redistribute_issue: bad arguments");
    tsq_comm::Slot& s = c->slot[slot];
    if (s.state != 2) return tsq_fail(h, TSQ_ERR_INVALID, "tsq_redistribute_issue: the slot's counts were not exchanged");
    const int n_cols = (int)s.cols.size();
    if (n_out != n_cols) return tsq_fail(h, TSQ_ERR_INVALID, "tsq_redistribute_issue: as many output columns as the piece has");
    const tsq_col* cols = s.cols.data();
    const int* var_of = s.var_of;
    const int64_t nrows = s.nrows;
    const int W = c->world;
    TSQ_HIP(h, hipSetDevice(ctx->device));
    const uint64_t* M = s.M.data();  // M[q * L + ...]: rank q's vector
    // ---- who sends what to whom, where it lands, how received offsets are rebased: tsq_comm_plan.h (walked on the CPU for world
    // sizes 2, 4 and 8 by tests/hostsim)
    int32_t es_of[TSQ_MAX_COLS];
    for (int i = 0; i < n_cols; i++) es_of[i] = var_of[i] >= 0 ? 0 : (int32_t)tsq_elem_size(cols[i].type);
    const tsq_comm_plan pl = tsq_comm_make_plan(c->rank, W, n_cols, es_of, M, s.broadcast);
    const int64_t total = pl.total_rows;
    const uint64_t mask = pl.mask;  // a column is nullable for everybody as soon as one rank holds NULLs in it
    for (int i = 0; i < n_cols; i++) {
        const bool var = var_of[i] >= 0;
        TSQ_TRY(s.recv[i].reserve(ctx, h, (var ? (size_t)pl.recv_bytes[(size_t)i] : (size_t)std::max<int64_t>(total, 1) * tsq_elem_size(cols[i].type)) + 64));
        if (var) {
            TSQ_TRY(s.recvtmp[i].reserve(ctx, h, ((size_t)total + W + 1) * 8 + 64));
            TSQ_TRY(s.recvoffs[i].reserve(ctx, h, ((size_t)total + 1) * 8 + 64));
        }
        if ((mask >> i) & 1) {
            TSQ_TRY(s.sendnn[i].reserve(ctx, h, (size_t)std::max<int64_t>(nrows, 1) + 64));
            TSQ_TRY(s.recvnn[i].reserve(ctx, h, (size_t)std::max<int64_t>(total, 1) + 64));
            TSQ_TRY(s.recvbm[i].reserve(ctx, h, tsq_bitmap_bytes(total) + 64));
        }
    }
    // ---- the exchange: after everything queued on the operator stream so far (the split, and the consumers of this slot's
    // previous contents), one group of sends and receives
    TSQ_HIP(h, hipEventRecord(s.split_done, ctx->stream));
    TSQ_HIP(h, hipStreamWaitEvent(c->xs, s.split_done, 0));
    for (int i = 0; i < n_cols; i++) {  // the NOT-NULL bytes of what this rank sends
        if (!((mask >> i) & 1) || nrows == 0) continue;
        if (cols[i].null_bitmap) {
            const int grid = (int)std::min<int64_t>((nrows + 255) / 256, (int64_t)ctx->num_cus * 8);
            hipLaunchKernelGGL(k_bits_to_bytes, dim3(grid), dim3(256), 0, c->xs, s.sendbm[i].as<uint8_t>(), s.sendnn[i].as<uint8_t>(), nrows);
            TSQ_HIP(h, hipGetLastError());
        } else {
            TSQ_HIP(h, hipMemsetAsync(s.sendnn[i].p, 1, (size_t)nrows, c->xs));
        }
    }
    TSQ_NCCL(h, rccl()->GroupStart());
    {
        // an error inside the group must still close it (an open group swallows every later RCCL call of this process)
        tsq_status gs = TSQ_OK;
        for (const tsq_comm_xfer& x : pl.xfers) {
            const char* sbase = (const char*)(x.kind == TSQ_XFER_DATA ? s.send[x.col].p : (x.kind == TSQ_XFER_OFFS ? s.sendoffs[x.col].p : s.sendnn[x.col].p));
            char* rbase = (char*)(x.kind == TSQ_XFER_DATA ? s.recv[x.col].p : (x.kind == TSQ_XFER_OFFS ? s.recvtmp[x.col].p : s.recvnn[x.col].p));
            if (x.peer == c->rank) {
                if (x.send_len) {
                    hipError_t e = hipMemcpyAsync(rbase + x.recv_off, sbase + x.send_off, x.send_len, hipMemcpyDeviceToDevice, c->xs);
                    if (e != hipSuccess) { gs = tsq_fail(h, TSQ_ERR_HIP, std::string("hipMemcpyAsync(own run): ") + hipGetErrorString(e)); break; }
                }
                continue;
            }
            ncclResult_t r1 = ncclSuccess;
            if (x.send_len) r1 = rccl()->Send(sbase + x.send_off, x.send_len, ncclChar, x.peer, c->nccl, c->xs);
            if (r1 == ncclSuccess && x.recv_len) r1 = rccl()->Recv(rbase + x.recv_off, x.recv_len, ncclChar, x.peer, c->nccl, c->xs);
            if (r1 != ncclSuccess) { gs = tsq_fail(h, TSQ_ERR_HIP, std::string("ncclSend / ncclRecv: ") + rccl()->GetErrorString(r1)); break; }
        }
        const ncclResult_t ge = rccl()->GroupEnd();
        if (gs != TSQ_OK) return gs;
        if (ge != ncclSuccess) return tsq_fail(h, TSQ_ERR_HIP, std::string("ncclGroupEnd: ") + rccl()->GetErrorString(ge));
    }
    for (int i = 0; i < n_cols; i++)
        if (var_of[i] >= 0) TSQ_HIP(h, hipMemsetAsync(s.recvoffs[i].p, 0, 8, c->xs));
    for (const tsq_comm_shift& sh : pl.shifts) {  // received offsets -> the column's offsets
        const int grid = (int)std::min<int64_t>(((int64_t)sh.rows + 255) / 256, (int64_t)ctx->num_cus * 8);
        hipLaunchKernelGGL(k_offsets_shift, dim3(grid), dim3(256), 0, c->xs, s.recvtmp[sh.col].as<int64_t>() + sh.src_entry, s.recvoffs[sh.col].as<int64_t>() + sh.dst_entry,
                           (int64_t)sh.rows, sh.delta);
        TSQ_HIP(h, hipGetLastError());
    }
    for (int i = 0; i < n_cols; i++) {  // received bytes -> the packed bitmap of the received column
        if (!((mask >> i) & 1) || total == 0) continue;
        const int grid = (int)std::min<int64_t>(((total + 7) / 8 + 255) / 256, (int64_t)ctx->num_cus * 8);
        hipLaunchKernelGGL(k_bytes_to_bits, dim3(grid), dim3(256), 0, c->xs, s.recvnn[i].as<uint8_t>(), s.recvbm[i].as<uint8_t>(), total);
        TSQ_HIP(h, hipGetLastError());
    }
    TSQ_HIP(h, hipEventRecord(s.xchg_done, c->xs));
    s.pending = true;
    for (int i = 0; i < n_cols; i++) {
        out_cols[i] = cols[i];
        out_cols[i].data = s.recv[i].p;
        out_cols[i].null_bitmap = ((mask >> i) & 1) ? s.recvbm[i].as<uint8_t>() : nullptr;
        out_cols[i].offsets = var_of[i] >= 0 ? s.recvoffs[i].as<int64_t>() : nullptr;
        out_cols[i].length = total;
        out_cols[i].flags = TSQ_COL_DEVICE;
    }
    *nrows_out = total;
    s.state = 0;
    return TSQ_OK;
}

}  // namespace

TSQ_API tsq_status tsq_redistribute(tsq_comm* c, const tsq_col* cols, int32_t n_cols, int32_t key_col, int32_t key_mode, int64_t nrows,
                                    int32_t slot, tsq_col* out_cols, int64_t* nrows_out) {
    tsq_ctx_lock _api_lock(tsq_ctx_of(c, TSQ_MAGIC_COMM));
    if (!c || c->hdr.magic != TSQ_MAGIC_COMM) return TSQ_ERR_INVALID;
    if (!out_cols || !nrows_out) return tsq_fail(&c->hdr, TSQ_ERR_INVALID, "tsq_redistribute: bad arguments");
    TSQ_TRY(comm_prepare(c, cols, n_cols, key_col, key_mode, nrows, slot));
    TSQ_TRY(comm_counts(c, &slot, 1));
    return comm_issue(c, slot, out_cols, n_cols, nrows_out);
}

// The same exchange in three calls, so that a plan that redistributes its input in PIECES pays ONE count exchange (one blocking
// all-gather + host synchronisation) for all of them: prepare every piece (slot c = piece c), exchange the counts of all slots,
// then issue piece after piece — nothing between two issues waits for the host.
TSQ_API tsq_status tsq_redistribute_prepare(tsq_comm* c, const tsq_col* cols, int32_t n_cols, int32_t key_col, int32_t key_mode, int64_t nrows, int32_t slot) {
    tsq_ctx_lock _api_lock(tsq_ctx_of(c, TSQ_MAGIC_COMM));
    if (!c || c->hdr.magic != TSQ_MAGIC_COMM) return TSQ_ERR_INVALID;
    return comm_prepare(c, cols, n_cols, key_col, key_mode, nrows, slot);
}
TSQ_API tsq_status tsq_redistribute_counts(tsq_comm* c, const int32_t* slots, int32_t n_slots) {
    tsq_ctx_lock _api_lock(tsq_ctx_of(c, TSQ_MAGIC_COMM));
    if (!c || c->hdr.magic != TSQ_MAGIC_COMM) return TSQ_ERR_INVALID;
    return comm_counts(c, slots, n_slots);
}
TSQ_API tsq_status tsq_redistribute_issue(tsq_comm* c, int32_t slot, tsq_col* out_cols, int32_t n_cols, int64_t* nrows_out) {
    tsq_ctx_lock _api_lock(tsq_ctx_of(c, TSQ_MAGIC_COMM));
    if (!c || c->hdr.magic != TSQ_MAGIC_COMM) return TSQ_ERR_INVALID;
    return comm_issue(c, slot, out_cols, n_cols, nrows_out);
}


TSQ_API tsq_status tsq_redistribute_wait(tsq_comm* c, int32_t slot) {
    tsq_ctx_lock _api_lock(tsq_ctx_of(c, TSQ_MAGIC_COMM));
    if (!c || c->hdr.magic != TSQ_MAGIC_COMM) return TSQ_ERR_INVALID;
    if (slot < 0 || slot >= TSQ_COMM_SLOTS) return tsq_fail(&c->hdr, TSQ_ERR_INVALID, "tsq_redistribute_wait: bad slot");
    TSQ_HIP(&c->hdr, hipSetDevice(c->ctx->device));
    if (c->slot[slot].pending) TSQ_HIP(&c->hdr, hipStreamWaitEvent(c->ctx->stream, c->slot[slot].xchg_done, 0));
    return TSQ_OK;
}
