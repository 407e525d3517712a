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
        if (!ok) {
            dlclose(api.lib);
            api.lib = nullptr;
        }
    });
    return &api;
}

}  // namespace

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
        hipEvent_t split_done = nullptr, xchg_done = nullptr;
        bool pending = false;
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
    TSQ_HIP(ch, hipStreamCreateWithFlags(&c->xs, hipStreamNonBlocking));
    const size_t words = (size_t)(world + 1) * (world + 1) + 8;
    TSQ_TRY(c->cnt_dev.reserve(ctx, ch, words * 8));
    TSQ_HIP(ch, hipHostMalloc((void**)&c->cnt_host, words * 8, hipHostMallocDefault));
    for (auto& s : c->slot) {
        TSQ_HIP(ch, hipEventCreateWithFlags(&s.split_done, hipEventDisableTiming));
        TSQ_HIP(ch, hipEventCreateWithFlags(&s.xchg_done, hipEventDisableTiming));
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
        for (auto* v : {&s.send, &s.recv, &s.sendbm, &s.sendnn, &s.recvnn, &s.recvbm})
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
    const size_t off = (size_t)(c->world + 1) * (c->world + 1);
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
TSQ_API tsq_status tsq_comm_barrier(tsq_comm* c) {
    tsq_ctx_lock _api_lock(tsq_ctx_of(c, TSQ_MAGIC_COMM));
    if (!c || c->hdr.magic != TSQ_MAGIC_COMM) return TSQ_ERR_INVALID;
    int64_t one = 1;
    return allreduce8(c, &one, 1, ncclInt64, 0);
}

TSQ_API tsq_status tsq_redistribute(tsq_comm* c, const tsq_col* cols, int32_t n_cols, int32_t key_col, int32_t key_mode, int64_t nrows,
                                    int32_t slot, tsq_col* out_cols, int64_t* nrows_out) {
    tsq_ctx_lock _api_lock(tsq_ctx_of(c, TSQ_MAGIC_COMM));
    if (!c || c->hdr.magic != TSQ_MAGIC_COMM) return TSQ_ERR_INVALID;
    tsq_handle_hdr* h = &c->hdr;
    tsq_ctx* ctx = c->ctx;
    if (!cols || !out_cols || !nrows_out || n_cols < 1 || n_cols > TSQ_MAX_COLS || key_col < 0 || key_col >= n_cols || nrows < 0 || slot < 0 || slot >= TSQ_COMM_SLOTS)
        return tsq_fail(h, TSQ_ERR_INVALID, "tsq_redistribute: bad arguments");
    uint64_t my_mask = 0;
    for (int i = 0; i < n_cols; i++) {
        if (!(cols[i].flags & TSQ_COL_DEVICE)) return tsq_fail(h, TSQ_ERR_INVALID, "tsq_redistribute: columns must be device resident");
        if (cols[i].type < TSQ_I64 || cols[i].type > TSQ_F64) return tsq_fail(h, TSQ_ERR_UNSUPPORTED, "tsq_redistribute: fixed-width columns only");
        if (cols[i].null_bitmap) my_mask |= 1ull << i;
    }
    TSQ_HIP(h, hipSetDevice(ctx->device));
    tsq_comm::Slot& s = c->slot[slot];
    const int W = c->world, L = W + 1;
    if (s.pending) TSQ_HIP(h, hipStreamSynchronize(c->xs));  // the previous exchange of this slot (its buffers are rewritten below)
    s.pending = false;
    for (auto* v : {&s.send, &s.recv, &s.sendbm, &s.sendnn, &s.recvnn, &s.recvbm}) v->resize(n_cols);
    // ---- split on the operator stream (ends with the host counts: tsq_radix_split synchronises).  Rows with a NULL key go
    // to rank 0 (they never join; GROUP BY makes them one group)
    std::vector<tsq_col> sc(n_cols);
    for (int i = 0; i < n_cols; i++) {
        const size_t es = tsq_elem_size(cols[i].type);
        TSQ_TRY(s.send[i].reserve(ctx, h, (size_t)std::max<int64_t>(nrows, 1) * es + 64));
        sc[i] = cols[i];
        sc[i].data = s.send[i].p;
        sc[i].null_bitmap = nullptr;
        if (cols[i].null_bitmap) {
            TSQ_TRY(s.sendbm[i].reserve(ctx, h, tsq_bitmap_bytes(nrows) + 64));
            sc[i].null_bitmap = s.sendbm[i].as<uint8_t>();
        }
    }
    int64_t sendc[TSQ_SPLIT_MAX_PARTS] = {0};
    if (nrows > 0) {
        tsq_status st = tsq_radix_split(ctx, cols, n_cols, key_col, key_mode, nrows, W, sc.data(), sendc);
        if (st != TSQ_OK) return tsq_fail(h, st, ctx->hdr.err);
    }
    // ---- counts and nullable-column masks: every rank learns the whole world x (world + 1) matrix
    uint64_t* cd = c->cnt_dev.as<uint64_t>();
    for (int p = 0; p < W; p++) c->cnt_host[p] = (uint64_t)sendc[p];
    c->cnt_host[W] = my_mask;
    TSQ_HIP(h, hipMemcpyAsync(cd, c->cnt_host, (size_t)L * 8, hipMemcpyHostToDevice, c->xs));
    TSQ_NCCL(h, rccl()->AllGather(cd, cd + L, (size_t)L, ncclInt64, c->nccl, c->xs));
    TSQ_HIP(h, hipMemcpyAsync(c->cnt_host + L, cd + L, (size_t)W * L * 8, hipMemcpyDeviceToHost, c->xs));
    TSQ_HIP(h, hipStreamSynchronize(c->xs));
    int64_t recvc[TSQ_SPLIT_MAX_PARTS], total = 0;
    uint64_t mask = 0;  // a column is nullable for everybody as soon as one rank holds NULLs in it
    for (int p = 0; p < W; p++) {
        recvc[p] = (int64_t)c->cnt_host[L + (size_t)p * L + c->rank];  // what rank p sends to this rank
        total += recvc[p];
        mask |= c->cnt_host[L + (size_t)p * L + W];
    }
    for (int i = 0; i < n_cols; i++) {
        TSQ_TRY(s.recv[i].reserve(ctx, h, (size_t)std::max<int64_t>(total, 1) * tsq_elem_size(cols[i].type) + 64));
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
    for (int i = 0; i < n_cols; i++) {
        const size_t es = tsq_elem_size(cols[i].type);
        const bool nn = (mask >> i) & 1;
        size_t so = 0, ro = 0;  // in rows
        for (int p = 0; p < W; p++) {
            const size_t sr = (size_t)sendc[p], rr = (size_t)recvc[p];
            if (p == c->rank) {
                if (sr) TSQ_HIP(h, hipMemcpyAsync((char*)s.recv[i].p + ro * es, (const char*)s.send[i].p + so * es, sr * es, hipMemcpyDeviceToDevice, c->xs));
                if (sr && nn) TSQ_HIP(h, hipMemcpyAsync((char*)s.recvnn[i].p + ro, (const char*)s.sendnn[i].p + so, sr, hipMemcpyDeviceToDevice, c->xs));
            } else {
                if (sr) TSQ_NCCL(h, rccl()->Send((const char*)s.send[i].p + so * es, sr * es, ncclChar, p, c->nccl, c->xs));
                if (rr) TSQ_NCCL(h, rccl()->Recv((char*)s.recv[i].p + ro * es, rr * es, ncclChar, p, c->nccl, c->xs));
                if (sr && nn) TSQ_NCCL(h, rccl()->Send((const char*)s.sendnn[i].p + so, sr, ncclChar, p, c->nccl, c->xs));
                if (rr && nn) TSQ_NCCL(h, rccl()->Recv((char*)s.recvnn[i].p + ro, rr, ncclChar, p, c->nccl, c->xs));
            }
            so += sr;
            ro += rr;
        }
    }
    TSQ_NCCL(h, rccl()->GroupEnd());
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
        out_cols[i].offsets = nullptr;
        out_cols[i].length = total;
        out_cols[i].flags = TSQ_COL_DEVICE;
    }
    *nrows_out = total;
    return TSQ_OK;
}

TSQ_API tsq_status tsq_redistribute_wait(tsq_comm* c, int32_t slot) {
    tsq_ctx_lock _api_lock(tsq_ctx_of(c, TSQ_MAGIC_COMM));
    if (!c || c->hdr.magic != TSQ_MAGIC_COMM) return TSQ_ERR_INVALID;
    if (slot < 0 || slot >= TSQ_COMM_SLOTS) return tsq_fail(&c->hdr, TSQ_ERR_INVALID, "tsq_redistribute_wait: bad slot");
    TSQ_HIP(&c->hdr, hipSetDevice(c->ctx->device));
    if (c->slot[slot].pending) TSQ_HIP(&c->hdr, hipStreamWaitEvent(c->ctx->stream, c->slot[slot].xchg_done, 0));
    return TSQ_OK;
}
