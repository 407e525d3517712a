// tsq_internal.h — host-side plumbing shared by the libtsq translation units (not part of the ABI).
#ifndef TSQ_INTERNAL_H
#define TSQ_INTERNAL_H

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <thread>
#include <string>
#include <unordered_map>
#include <vector>

#include "tsq_arena.h"
#include "tsq_device.h"

#define TSQ_API extern "C" __attribute__((visibility("default")))

// every handle starts with this header so tsq_last_error(handle) works on any of them
struct tsq_handle_hdr {
    uint32_t magic;
    std::string err;
};
#define TSQ_MAGIC_CTX 0x74737143u   /* 'tsqC' */
#define TSQ_MAGIC_JOIN 0x7473714au  /* 'tsqJ' */
#define TSQ_MAGIC_AGG 0x74737141u   /* 'tsqA' */
#define TSQ_MAGIC_EXPR 0x74737145u  /* 'tsqE' */

void tsq_set_global_error(const std::string& s);

struct tsq_ctx {
    tsq_handle_hdr hdr;
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipDeviceProp_t prop;
    int num_cus = 256;
    // small pinned scratch for scalar results (counts, error words)
    uint64_t* pinned = nullptr;      // host-mapped, 64 words
    uint64_t* dscratch = nullptr;    // device, 64 words
    // hiprtc modules of destroyed expression handles: unloaded with the context.  (Unloading a module as soon as its
    // handle is destroyed made LATER, unrelated kernels fault intermittently on ROCm 7.2 — 3 of 12 runs of
    // tools/bench_kernels.py, address 0x100000 — although the stream had been synchronised before the unload.)
    std::vector<hipModule_t> retired_modules;
    std::mutex retired_mu;
    // plan cache of specialised expression kernels: generated source -> loaded module.  A query that runs again (or a
    // second operator with the same expression list) skips the hiprtc compile; a failed compile is remembered too.
    struct JitEntry {
        hipModule_t mod = nullptr;
        hipFunction_t f_expr = nullptr, f_filter = nullptr;
        std::string log;
        double compile_ms = 0;  // hiprtc + module load of this program set (once per distinct tree and context)
        // TSQ_JIT_AUTO compiles on a helper thread (the interpreter kernels serve the handle meanwhile): the thread leaves the code object
        // here and sets `state` to 2; the first launch after that loads the module on the caller's thread (state 3 = loaded or failed for good)
        std::vector<char> code;
        std::atomic<int> state{0};  // 0 new, 1 compiling on `worker`, 2 code ready, 3 final
        std::thread worker;
    };
    std::unordered_map<std::string, JitEntry> jit_cache;
    std::mutex jit_mu;
    // One context = one stream + one block of pinned scratch words shared by all of its handles.  The Go callers run
    // operators of one plan on several goroutines (executor/join.go:207, aggregate.go:512, projection.go:312-347), so two
    // handles of the same context may be entered at the same time: every entry point that touches the stream or the
    // scratch holds this lock for its duration (the GPU work is serialised by the stream anyway).  tsq_*_cancel does not
    // take it: it only sets an atomic flag.  Recursive: entry points call each other's helpers.
    std::recursive_mutex api_mu;
    // test / measurement knobs (tsq_ctx_set_knob, include/tsq.h): TSQ_KNOB_DEFAULT = not set
    int64_t knob[TSQ_KNOB_COUNT];
    // device-memory pool: hipMalloc costs ~35 ms per GB, and one radix / pre-aggregation batch needs several GB of
    // partition buffers — per handle that was 100+ ms of allocation for a 20 ms aggregate.  Buffers released by a handle
    // are kept (up to pool_cap bytes) and handed to the next one; everything runs on ctx->stream, so stream order
    // protects a buffer that is recycled while kernels of its previous owner are still queued.
    std::mutex pool_mu;
    std::vector<std::pair<void*, size_t>> pool;
    size_t pool_bytes = 0, pool_cap = (size_t)48 << 30;
    std::unordered_map<void*, size_t> user_allocs;  // live tsq_dev_alloc blocks -> capacity (guarded by pool_mu)
    std::unordered_map<void*, size_t> host_allocs;  // live tsq_host_alloc blocks (pinned) -> capacity (guarded by pool_mu)
    // ARENA (tsq_ctx_reserve): one slab allocated up front — by the host process when it creates the context, outside any query —
    // that the buffers of every operator are carved from, so that the FIRST build / aggregate of a session does not pay hipMalloc's
    // first touch (35 ms per GB: 310 ms for the 1e8-row build side).  Free ranges by offset, merged with their neighbours on release;
    // a request takes the smallest free range that holds it (256-byte granules).  A request the arena cannot hold falls through to
    // the pool / hipMalloc.  Guarded by pool_mu.
    char* arena_base = nullptr;
    tsq_arena_ranges arena;  // tsq_arena.h
};
inline void* tsq_arena_get(tsq_ctx* ctx, size_t bytes, size_t* got) {  // (pool_mu held)
    size_t off = 0;
    if (!ctx->arena_base || !ctx->arena.get(bytes, &off, got)) return nullptr;
    return ctx->arena_base + off;
}
inline bool tsq_arena_put(tsq_ctx* ctx, void* p, size_t cap) {  // (pool_mu held) true: p was the arena's
    if (!ctx->arena_base || (char*)p < ctx->arena_base || (char*)p >= ctx->arena_base + ctx->arena.size) return false;
    ctx->arena.put((size_t)((char*)p - ctx->arena_base), cap);
    return true;
}
inline void* tsq_pool_get(tsq_ctx* ctx, size_t bytes, size_t* got) {
    std::lock_guard<std::mutex> g(ctx->pool_mu);
    if (void* a = tsq_arena_get(ctx, bytes, got)) return a;
    int best = -1;
    for (int i = 0; i < (int)ctx->pool.size(); i++) {
        const size_t c = ctx->pool[i].second;
        if (c >= bytes && c <= bytes + bytes / 2 + (1 << 20) && (best < 0 || c < ctx->pool[best].second)) best = i;
    }
    if (best < 0) return nullptr;
    void* p = ctx->pool[best].first;
    *got = ctx->pool[best].second;
    ctx->pool_bytes -= *got;
    ctx->pool.erase(ctx->pool.begin() + best);
    return p;
}
inline void tsq_pool_put(tsq_ctx* ctx, void* p, size_t cap) {
    {
        std::lock_guard<std::mutex> g(ctx->pool_mu);
        if (tsq_arena_put(ctx, p, cap)) return;
        if (cap >= (1 << 16) && ctx->pool_bytes + cap <= ctx->pool_cap && ctx->pool.size() < 256) {
            ctx->pool.emplace_back(p, cap);
            ctx->pool_bytes += cap;
            return;
        }
    }
    (void)hipFree(p);
}

inline int64_t tsq_knob(const tsq_ctx* ctx, int k, int64_t dflt) { return ctx->knob[k] == TSQ_KNOB_DEFAULT ? dflt : ctx->knob[k]; }

struct tsq_ctx_lock {
    std::unique_lock<std::recursive_mutex> g;
    explicit tsq_ctx_lock(tsq_ctx* c) { if (c) g = std::unique_lock<std::recursive_mutex>(c->api_mu); }
};
template <class H>
inline tsq_ctx* tsq_ctx_of(H* h, uint32_t magic) { return h && h->hdr.magic == magic ? h->ctx : nullptr; }

inline tsq_status tsq_fail(tsq_handle_hdr* h, tsq_status s, const std::string& msg) {
    if (h) h->err = msg;
    tsq_set_global_error(msg);
    return s;
}

#define TSQ_HIP(h, expr)                                                                               \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess) {                                                                        \
            tsq_status _s = (_e == hipErrorOutOfMemory) ? TSQ_ERR_OOM_DEVICE : TSQ_ERR_HIP;            \
            return tsq_fail((h), _s, std::string(#expr) + ": " + hipGetErrorString(_e));               \
        }                                                                                              \
    } while (0)

#define TSQ_TRY(expr)                     \
    do {                                  \
        tsq_status _s = (expr);           \
        if (_s != TSQ_OK) return _s;      \
    } while (0)

// growable device buffer (never shrinks); contents are NOT preserved on growth unless keep=true
size_t tsq_user_alloc_bytes(tsq_ctx* ctx, const void* p);  // tsq_ctx.hip
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    tsq_ctx* owner = nullptr;
    bool foreign = false;  // p is the CALLER's buffer (TSQ_COL_RETAIN): never freed here; a reserve() beyond cap copies it into an own block
    tsq_status reserve(tsq_ctx* ctx, tsq_handle_hdr* h, size_t bytes, bool keep = false, size_t used = 0) {
        if (bytes <= cap) return TSQ_OK;
        size_t ncap = bytes;
        if (keep && cap) ncap = std::max(bytes, cap + cap / 2);
        void* np = tsq_pool_get(ctx, ncap, &ncap);
        if (!np) TSQ_HIP(h, hipMalloc(&np, ncap));
        if (keep && p && used) {
            hipError_t e = hipMemcpyAsync(np, p, used, hipMemcpyDeviceToDevice, ctx->stream);
            if (e != hipSuccess) {
                tsq_pool_put(ctx, np, ncap);
                return tsq_fail(h, TSQ_ERR_HIP, std::string("hipMemcpyAsync(grow): ") + hipGetErrorString(e));
            }
            (void)hipStreamSynchronize(ctx->stream);
        } else if (p) {
            (void)hipStreamSynchronize(ctx->stream);  // queued kernels may still read the buffer that is handed back (hipFree used to wait too)
        }
        release();
        p = np;
        cap = ncap;
        owner = ctx;
        return TSQ_OK;
    }
    void release() {
        if (p && !foreign) {
            if (owner) tsq_pool_put(owner, p, cap);
            else (void)hipFree(p);
        }
        p = nullptr;
        cap = 0;
        foreign = false;
    }
    void adopt(void* ptr, size_t bytes) {  // keep the caller's buffer as this column's storage
        release();
        p = ptr;
        cap = bytes;
        foreign = true;
    }
    template <class T>
    T* as() const { return (T*)p; }
};

// Pinned host memory is expensive to get (hipHostMalloc pins page by page: ~0.3 ms per MB, 100 ms for the 320 MB of result columns of
// bench.py's pcie_inclusive_1e7) and every join result batch, every staging area asked for its own: released buffers are kept in a
// process-wide pool (up to 8 GB, best fit within 2x) and handed to the next owner.  A buffer comes back only after its owner has
// synchronised the stream that wrote it (result batches are released once they were pulled, staging areas at destroy).  The pool is
// never destroyed (static teardown would run after the HIP runtime's).
struct tsq_pinned_pool {
    std::mutex mu;
    std::vector<std::pair<void*, size_t>> free_list;
    size_t bytes = 0;
};
inline tsq_pinned_pool& tsq_pinned() {
    static tsq_pinned_pool* pool = new tsq_pinned_pool();
    return *pool;
}
struct PinnedBuf {
    void* p = nullptr;
    size_t cap = 0;
    tsq_status reserve(tsq_handle_hdr* h, size_t bytes) {
        if (bytes <= cap) return TSQ_OK;
        release();
        {
            tsq_pinned_pool& pool = tsq_pinned();
            std::lock_guard<std::mutex> g(pool.mu);
            int best = -1;
            for (int i = 0; i < (int)pool.free_list.size(); i++) {
                const size_t c = pool.free_list[i].second;
                if (c >= bytes && c <= 2 * bytes + (1 << 20) && (best < 0 || c < pool.free_list[best].second)) best = i;
            }
            if (best >= 0) {
                p = pool.free_list[best].first;
                cap = pool.free_list[best].second;
                pool.bytes -= cap;
                pool.free_list.erase(pool.free_list.begin() + best);
                return TSQ_OK;
            }
        }
        // (sizes rounded up — 64 KB granules below 1 MB, 1 MB granules above — so that the next batch of nearly the same size fits)
        const size_t gran = bytes >= ((size_t)1 << 20) ? ((size_t)1 << 20) : ((size_t)1 << 16);
        const size_t want = (bytes + gran - 1) / gran * gran;
        TSQ_HIP(h, hipHostMalloc(&p, want, hipHostMallocDefault));
        cap = want;
        return TSQ_OK;
    }
    void release() {
        if (p) {
            tsq_pinned_pool& pool = tsq_pinned();
            bool kept = false;
            {
                std::lock_guard<std::mutex> g(pool.mu);
                if (cap >= ((size_t)1 << 16) && pool.bytes + cap <= ((size_t)8 << 30) && pool.free_list.size() < 512) {
                    pool.free_list.emplace_back(p, cap);
                    pool.bytes += cap;
                    kept = true;
                }
            }
            if (!kept) (void)hipHostFree(p);
        }
        p = nullptr;
        cap = 0;
    }
};

// host-side copy of a large chunk between the caller's memory and pinned staging: one core moves ~10 GB/s, the link 50+, so a copy of
// several MB is cut into slices for a few threads (a 1 Mi-row push / pull of the cgo shim; 1024-row chunks stay on the calling thread)
inline void tsq_host_copy(void* dst, const void* src, size_t n) {
    constexpr size_t kMin = (size_t)2 << 20;
    if (n < 2 * kMin) {
        memcpy(dst, src, n);
        return;
    }
    const size_t parts = std::min<size_t>(8, n / kMin);
    const size_t per = ((n + parts - 1) / parts + 63) & ~(size_t)63;
    std::vector<std::thread> th;
    th.reserve(parts - 1);
    for (size_t p = 1; p < parts; p++) {
        const size_t lo = p * per;
        if (lo >= n) break;
        const size_t len = std::min(per, n - lo);
        th.emplace_back([=] { memcpy((char*)dst + lo, (const char*)src + lo, len); });
    }
    memcpy(dst, src, std::min(per, n));
    for (auto& t : th) t.join();
}

// caller's chunk -> pinned staging (HostStage::add, a 1024-row push = 8 KB per column): the staged bytes are next read by a DMA engine,
// never by this core, so they are written with NON-TEMPORAL stores (SSE2 movntdq through clang's builtin; 64 bytes per step) — a plain
// memcpy first reads every destination line into the cache to own it, a third of the memory traffic of the copy.  The sfence at the
// end orders the write-combining buffers before the hipMemcpyAsync that follows.  TSQ_KNOB_HOST_NT_COPY = 0: memcpy.
inline std::atomic<int> tsq_host_nt_copy_on{1};
inline void tsq_stage_copy(void* dst, const void* src, size_t n) {
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
    if (n >= 1024 && n < ((size_t)4 << 20) && tsq_host_nt_copy_on.load(std::memory_order_relaxed)) {
        typedef long long tsq_v2di __attribute__((vector_size(16), aligned(16)));
        typedef long long tsq_v2di_u __attribute__((vector_size(16), aligned(1)));
        char* d = (char*)dst;
        const char* s = (const char*)src;
        const size_t head = (16 - ((uintptr_t)d & 15)) & 15;
        if (head) {
            memcpy(d, s, head);
            d += head;
            s += head;
            n -= head;
        }
        size_t i = 0;
        for (; i + 64 <= n; i += 64) {
            const tsq_v2di a = *(const tsq_v2di_u*)(s + i), b = *(const tsq_v2di_u*)(s + i + 16), c = *(const tsq_v2di_u*)(s + i + 32), e = *(const tsq_v2di_u*)(s + i + 48);
            __builtin_nontemporal_store(a, (tsq_v2di*)(d + i));
            __builtin_nontemporal_store(b, (tsq_v2di*)(d + i + 16));
            __builtin_nontemporal_store(c, (tsq_v2di*)(d + i + 32));
            __builtin_nontemporal_store(e, (tsq_v2di*)(d + i + 48));
        }
        if (i < n) memcpy(d + i, s + i, n - i);
        asm volatile("sfence" ::: "memory");
        return;
    }
#endif
    tsq_host_copy(dst, src, n);
}

inline int tsq_elem_size(int32_t type) { return type == TSQ_F32 ? 4 : 8; }
inline size_t tsq_bitmap_bytes(int64_t rows) { return (size_t)((rows + 7) / 8); }

// grid sizing for HBM-bound grid-stride kernels: enough waves to cover latency, ≤ 8 blocks/CU
inline int tsq_grid_for(const tsq_ctx* ctx, int64_t work_items, int block, int items_per_thread = 1) {
    int64_t blocks = (work_items + (int64_t)block * items_per_thread - 1) / ((int64_t)block * items_per_thread);
    int64_t cap = (int64_t)ctx->num_cus * 8;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

// kernels defined in tsq_ctx.hip that other units launch through these wrappers
// first_row (device, optional): rows below *first_row (a multiple of 32) have their bits already — the pass starts there
tsq_status tsq_launch_pack_bitmap(tsq_ctx* ctx, tsq_handle_hdr* h, const uint8_t* notnull_bytes, uint8_t* bitmap, int64_t n, const unsigned long long* first_row = nullptr);

// tsq_comm.hip, for the COLLECTIVE steps of an operator (tsq_join_build_finish_shared): every rank of the communicator calls them
// in the same order.  The device all-reduce sums `count` elements of 1 or 4 bytes in place: it waits for the context's stream,
// runs on the communicator's exchange stream and returns when the result is in `buf`.
struct tsq_comm;
bool tsq_comm_usable(const tsq_comm* c, const tsq_ctx* ctx);
int32_t tsq_comm_world_size(const tsq_comm* c);
tsq_status tsq_comm_allreduce_dev_sum(tsq_comm* c, void* buf, size_t count, int elem_bytes);
tsq_status tsq_comm_allreduce_host_i64(tsq_comm* c, int64_t* inout, int32_t n, int32_t op);  // op: 0 sum, 1 max, 2 min; n <= 8

#endif
