// tsq_ctx.hip — context, device memory, timers, synthetic-table generator (gfx950).
#include "tsq_stage.h"

static std::mutex g_err_mu;
static std::string g_err;
static thread_local std::string g_err_ret;

void tsq_set_global_error(const std::string& s) {
    std::lock_guard<std::mutex> lk(g_err_mu);
    g_err = s;
}

TSQ_API int32_t tsq_abi_version(void) { return TSQ_ABI_VERSION; }

TSQ_API int32_t tsq_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

TSQ_API const char* tsq_last_error(const void* handle) {
    if (handle) {
        const tsq_handle_hdr* h = (const tsq_handle_hdr*)handle;
        if (h->magic == TSQ_MAGIC_CTX || h->magic == TSQ_MAGIC_JOIN || h->magic == TSQ_MAGIC_AGG ||
            h->magic == TSQ_MAGIC_EXPR)
            return h->err.c_str();
    }
    std::lock_guard<std::mutex> lk(g_err_mu);
    g_err_ret = g_err;
    return g_err_ret.c_str();
}

TSQ_API tsq_status tsq_ctx_create(int32_t device, tsq_ctx** out) {
    if (!out) return tsq_fail(nullptr, TSQ_ERR_INVALID, "tsq_ctx_create: out == NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return tsq_fail(nullptr, TSQ_ERR_NO_DEVICE,
                        "tsq_ctx_create: no HIP device visible; libtsq has no CPU fallback");
    if (device < 0 || device >= n) return tsq_fail(nullptr, TSQ_ERR_INVALID, "tsq_ctx_create: bad device index");
    tsq_ctx* c = new tsq_ctx();
    c->hdr.magic = TSQ_MAGIC_CTX;
    c->device = device;
    for (int64_t& k : c->knob) k = TSQ_KNOB_DEFAULT;
    tsq_handle_hdr* h = &c->hdr;
    auto fail = [&](hipError_t e, const char* what) {
        tsq_status s = tsq_fail(nullptr, e == hipErrorOutOfMemory ? TSQ_ERR_OOM_DEVICE : TSQ_ERR_HIP,
                                std::string(what) + ": " + hipGetErrorString(e));
        delete c;
        return s;
    };
    (void)h;
    hipError_t e;
    if ((e = hipSetDevice(device)) != hipSuccess) return fail(e, "hipSetDevice");
    if ((e = hipGetDeviceProperties(&c->prop, device)) != hipSuccess) return fail(e, "hipGetDeviceProperties");
    c->num_cus = c->prop.multiProcessorCount > 0 ? c->prop.multiProcessorCount : 256;
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) return fail(e, "hipStreamCreate");
    c->own_stream = true;
    if ((e = hipEventCreate(&c->ev0)) != hipSuccess) return fail(e, "hipEventCreate");
    if ((e = hipEventCreate(&c->ev1)) != hipSuccess) return fail(e, "hipEventCreate");
    if ((e = hipHostMalloc((void**)&c->pinned, 64 * sizeof(uint64_t), hipHostMallocDefault)) != hipSuccess)
        return fail(e, "hipHostMalloc");
    if ((e = hipMalloc((void**)&c->dscratch, 64 * sizeof(uint64_t))) != hipSuccess) return fail(e, "hipMalloc");
    *out = c;
    return TSQ_OK;
}

TSQ_API tsq_status tsq_ctx_set_stream(tsq_ctx* ctx, void* hip_stream) {
    tsq_ctx_lock _api_lock(ctx);
    if (!ctx) return TSQ_ERR_INVALID;
    TSQ_HIP(&ctx->hdr, hipSetDevice(ctx->device));
    if (ctx->own_stream && ctx->stream) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamDestroy(ctx->stream);
    }
    ctx->stream = (hipStream_t)hip_stream;
    ctx->own_stream = false;
    return TSQ_OK;
}

TSQ_API tsq_status tsq_ctx_sync(tsq_ctx* ctx) {
    tsq_ctx_lock _api_lock(ctx);
    if (!ctx) return TSQ_ERR_INVALID;
    TSQ_HIP(&ctx->hdr, hipSetDevice(ctx->device));
    TSQ_HIP(&ctx->hdr, hipStreamSynchronize(ctx->stream));
    return TSQ_OK;
}

TSQ_API void tsq_ctx_destroy(tsq_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->dscratch) (void)hipFree(ctx->dscratch);
    (void)hipDeviceSynchronize();
    for (hipModule_t m : ctx->retired_modules) (void)hipModuleUnload(m);
    for (auto& kv : ctx->jit_cache) {
        if (kv.second.worker.joinable()) kv.second.worker.join();  // (a compile still running on its helper thread)
        if (kv.second.mod) (void)hipModuleUnload(kv.second.mod);
    }
    for (auto& b : ctx->pool) (void)hipFree(b.first);
    for (auto& b : ctx->user_allocs)  // blocks the caller never handed back (the arena's go with the slab)
        if (!ctx->arena_base || (char*)b.first < ctx->arena_base || (char*)b.first >= ctx->arena_base + ctx->arena.size) (void)hipFree(b.first);
    if (ctx->arena_base) (void)hipFree(ctx->arena_base);
    for (auto& b : ctx->host_allocs) {  // pinned blocks the caller never handed back: to the process-wide pool
        PinnedBuf pb;
        pb.p = b.first;
        pb.cap = b.second;
        pb.release();
    }
    ctx->hdr.magic = 0;
    delete ctx;
}

TSQ_API tsq_status tsq_ctx_set_knob(tsq_ctx* ctx, int32_t knob, int64_t value) {
    tsq_ctx_lock _api_lock(ctx);
    if (!ctx || ctx->hdr.magic != TSQ_MAGIC_CTX) return TSQ_ERR_INVALID;
    if (knob < 0 || knob >= TSQ_KNOB_COUNT) return tsq_fail(&ctx->hdr, TSQ_ERR_INVALID, "tsq_ctx_set_knob: unknown knob");
    ctx->knob[knob] = value;
    if (knob == TSQ_KNOB_HOST_NT_COPY) tsq_host_nt_copy_on.store(value == 0 ? 0 : 1, std::memory_order_relaxed);  // (process-wide: the staging copies have no context at hand)
    return TSQ_OK;
}

// The arena (tsq_internal.h): one slab for the buffers of every operator of this context, allocated here — when the host process
// sets the context up — instead of piece by piece inside the first query.
TSQ_API tsq_status tsq_ctx_reserve(tsq_ctx* ctx, int64_t bytes) {
    tsq_ctx_lock _api_lock(ctx);
    if (!ctx || bytes < 0) return TSQ_ERR_INVALID;
    tsq_handle_hdr* h = &ctx->hdr;
    TSQ_HIP(h, hipSetDevice(ctx->device));
    TSQ_HIP(h, hipStreamSynchronize(ctx->stream));
    std::lock_guard<std::mutex> g(ctx->pool_mu);
    if (ctx->arena_base) {
        if (ctx->arena.used) return tsq_fail(h, TSQ_ERR_INVALID, "tsq_ctx_reserve: the arena holds live buffers (reserve before the first operator, or after the last one is destroyed)");
        (void)hipFree(ctx->arena_base);
        ctx->arena_base = nullptr;
        ctx->arena.reset(0);
    }
    if (bytes == 0) return TSQ_OK;
    const size_t sz = ((size_t)bytes + 255) & ~(size_t)255;
    void* p = nullptr;
    TSQ_HIP(h, hipMalloc(&p, sz));
    hipError_t e = hipMemsetAsync(p, 0, sz, ctx->stream);  // every page is mapped before the first operator runs
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        (void)hipFree(p);
        return tsq_fail(h, TSQ_ERR_HIP, std::string("tsq_ctx_reserve: ") + hipGetErrorString(e));
    }
    ctx->arena_base = (char*)p;
    ctx->arena.reset(sz);
    return TSQ_OK;
}
TSQ_API tsq_status tsq_ctx_arena_stats(tsq_ctx* ctx, int64_t* size_out, int64_t* used_out, int64_t* peak_out) {
    if (!ctx) return TSQ_ERR_INVALID;
    std::lock_guard<std::mutex> g(ctx->pool_mu);
    if (size_out) *size_out = (int64_t)ctx->arena.size;
    if (used_out) *used_out = (int64_t)ctx->arena.used;
    if (peak_out) *peak_out = (int64_t)ctx->arena.peak;
    return TSQ_OK;
}

// tsq_dev_alloc / tsq_dev_free go through the context pool as well: a device-resident operator pipeline allocates its
// output chunks (GBs) per query, and hipMalloc is ~35 ms per GB.  The size of every live allocation is remembered so that
// tsq_dev_free can hand the block back; a recycled block is safe because every libtsq kernel and copy runs on ctx->stream.
TSQ_API tsq_status tsq_dev_alloc(tsq_ctx* ctx, int64_t bytes, void** out) {
    tsq_ctx_lock _api_lock(ctx);
    if (!ctx || !out || bytes < 0) return TSQ_ERR_INVALID;
    TSQ_HIP(&ctx->hdr, hipSetDevice(ctx->device));
    *out = nullptr;
    if (bytes == 0) bytes = 8;
    size_t cap = (size_t)bytes + 64;  // (64 readable bytes behind the last one: what the operators' own column buffers have — a block a join keeps as it is, TSQ_COL_RETAIN, is read with 16-byte loads)
    void* p = tsq_pool_get(ctx, cap, &cap);
    if (!p) TSQ_HIP(&ctx->hdr, hipMalloc(&p, cap));
    {
        std::lock_guard<std::mutex> g(ctx->pool_mu);
        ctx->user_allocs[p] = cap;
    }
    *out = p;
    return TSQ_OK;
}
// bytes of the tsq_dev_alloc block that starts at p (0: not such a block)
size_t tsq_user_alloc_bytes(tsq_ctx* ctx, const void* p) {
    std::lock_guard<std::mutex> g(ctx->pool_mu);
    auto it = ctx->user_allocs.find(const_cast<void*>(p));
    return it == ctx->user_allocs.end() ? 0 : it->second;
}
TSQ_API tsq_status tsq_dev_free(tsq_ctx* ctx, void* p) {
    tsq_ctx_lock _api_lock(ctx);
    if (!ctx) return TSQ_ERR_INVALID;
    TSQ_HIP(&ctx->hdr, hipSetDevice(ctx->device));
    if (!p) return TSQ_OK;
    size_t cap = 0;
    {
        std::lock_guard<std::mutex> g(ctx->pool_mu);
        auto it = ctx->user_allocs.find(p);
        if (it == ctx->user_allocs.end()) return tsq_fail(&ctx->hdr, TSQ_ERR_INVALID, "tsq_dev_free: pointer was not returned by tsq_dev_alloc on this context");
        cap = it->second;
        ctx->user_allocs.erase(it);
    }
    tsq_pool_put(ctx, p, cap);
    return TSQ_OK;
}
// Pinned (page-locked) host memory for the chunks a host hands to / takes from the operators: the DMA engines reach it directly
// (~55 GB/s on this box; a pageable buffer is copied through a bounce buffer at a fifth of that, and its first touch page-faults).
// The blocks come from the process-wide pinned pool (PinnedBuf), so a query that runs again finds them there.
TSQ_API tsq_status tsq_host_alloc(tsq_ctx* ctx, int64_t bytes, void** out) {
    tsq_ctx_lock _api_lock(ctx);
    if (!ctx || !out || bytes < 0) return TSQ_ERR_INVALID;
    TSQ_HIP(&ctx->hdr, hipSetDevice(ctx->device));
    *out = nullptr;
    PinnedBuf b;
    TSQ_TRY(b.reserve(&ctx->hdr, (size_t)(bytes > 0 ? bytes : 8)));
    {
        std::lock_guard<std::mutex> g(ctx->pool_mu);
        ctx->host_allocs[b.p] = b.cap;
    }
    *out = b.p;  // (ownership moves to the caller: b is not released)
    return TSQ_OK;
}
TSQ_API tsq_status tsq_host_free(tsq_ctx* ctx, void* p) {
    tsq_ctx_lock _api_lock(ctx);
    if (!ctx) return TSQ_ERR_INVALID;
    if (!p) return TSQ_OK;
    PinnedBuf b;
    {
        std::lock_guard<std::mutex> g(ctx->pool_mu);
        auto it = ctx->host_allocs.find(p);
        if (it == ctx->host_allocs.end()) return tsq_fail(&ctx->hdr, TSQ_ERR_INVALID, "tsq_host_free: pointer was not returned by tsq_host_alloc on this context");
        b.p = p;
        b.cap = it->second;
        ctx->host_allocs.erase(it);
    }
    b.release();  // back to the pinned pool
    return TSQ_OK;
}
TSQ_API tsq_status tsq_dev_memset(tsq_ctx* ctx, void* p, int32_t byte, int64_t bytes) {
    tsq_ctx_lock _api_lock(ctx);
    if (!ctx) return TSQ_ERR_INVALID;
    TSQ_HIP(&ctx->hdr, hipSetDevice(ctx->device));
    TSQ_HIP(&ctx->hdr, hipMemsetAsync(p, byte, (size_t)bytes, ctx->stream));
    return TSQ_OK;
}
TSQ_API tsq_status tsq_copy_h2d(tsq_ctx* ctx, void* dst_dev, const void* src_host, int64_t bytes) {
    tsq_ctx_lock _api_lock(ctx);
    if (!ctx) return TSQ_ERR_INVALID;
    TSQ_HIP(&ctx->hdr, hipSetDevice(ctx->device));
    TSQ_HIP(&ctx->hdr, hipMemcpyAsync(dst_dev, src_host, (size_t)bytes, hipMemcpyHostToDevice, ctx->stream));
    TSQ_HIP(&ctx->hdr, hipStreamSynchronize(ctx->stream));
    return TSQ_OK;
}
TSQ_API tsq_status tsq_copy_d2h(tsq_ctx* ctx, void* dst_host, const void* src_dev, int64_t bytes) {
    tsq_ctx_lock _api_lock(ctx);
    if (!ctx) return TSQ_ERR_INVALID;
    TSQ_HIP(&ctx->hdr, hipSetDevice(ctx->device));
    TSQ_HIP(&ctx->hdr, hipMemcpyAsync(dst_host, src_dev, (size_t)bytes, hipMemcpyDeviceToHost, ctx->stream));
    TSQ_HIP(&ctx->hdr, hipStreamSynchronize(ctx->stream));
    return TSQ_OK;
}

TSQ_API tsq_status tsq_copy_d2d(tsq_ctx* ctx, void* dst_dev, const void* src_dev, int64_t bytes) {
    tsq_ctx_lock _api_lock(ctx);
    if (!ctx || bytes < 0) return TSQ_ERR_INVALID;
    if (bytes == 0) return TSQ_OK;
    TSQ_HIP(&ctx->hdr, hipSetDevice(ctx->device));
    TSQ_HIP(&ctx->hdr, hipMemcpyAsync(dst_dev, src_dev, (size_t)bytes, hipMemcpyDeviceToDevice, ctx->stream));  // stream ordered, not synchronised
    return TSQ_OK;
}

// append n rows of a device null bitmap at bit position dst_rows of another one (Column.appendNullBitmap / Decoder.decodeColumn's
// shift-and-or, util/chunk/codec.go:325-343): chunks whose row counts are not multiples of 8 are concatenated on the device
TSQ_API tsq_status tsq_bitmap_append(tsq_ctx* ctx, uint8_t* dst_bitmap, int64_t dst_rows, const uint8_t* src_bitmap, int64_t n) {
    tsq_ctx_lock _api_lock(ctx);
    if (!ctx || !dst_bitmap || dst_rows < 0 || n < 0) return TSQ_ERR_INVALID;
    if (n == 0) return TSQ_OK;
    TSQ_HIP(&ctx->hdr, hipSetDevice(ctx->device));
    return tsq_launch_append_bits(ctx, &ctx->hdr, dst_bitmap, dst_rows, src_bitmap, n);  // src == NULL: n set bits
}

TSQ_API tsq_status tsq_timer_start(tsq_ctx* ctx) {
    tsq_ctx_lock _api_lock(ctx);
    if (!ctx) return TSQ_ERR_INVALID;
    TSQ_HIP(&ctx->hdr, hipSetDevice(ctx->device));
    TSQ_HIP(&ctx->hdr, hipEventRecord(ctx->ev0, ctx->stream));
    return TSQ_OK;
}
TSQ_API tsq_status tsq_timer_stop_ms(tsq_ctx* ctx, double* ms_out) {
    tsq_ctx_lock _api_lock(ctx);
    if (!ctx || !ms_out) return TSQ_ERR_INVALID;
    TSQ_HIP(&ctx->hdr, hipSetDevice(ctx->device));
    TSQ_HIP(&ctx->hdr, hipEventRecord(ctx->ev1, ctx->stream));
    TSQ_HIP(&ctx->hdr, hipEventSynchronize(ctx->ev1));
    float ms = 0;
    TSQ_HIP(&ctx->hdr, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    *ms_out = (double)ms;
    return TSQ_OK;
}

// ------------------------------------------------------------------ K11: synthetic columns
// One row per lane, grid-stride; the null bitmap is produced with a wave ballot: one 64-bit word
// per wave-iteration, stored by lane 0 (bitmap words are 8-byte aligned because rows advance in
// multiples of 64 from a 64-aligned base).
__global__ void __launch_bounds__(256) k_gen_column(tsq_gen_spec spec, int64_t nrows, uint64_t* __restrict__ dst,
                                                    uint64_t* __restrict__ bm_words, const uint64_t* __restrict__ src) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t nround = (nrows + 63) & ~(int64_t)63;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nround; k += stride) {
        const bool in = k < nrows;
        const uint64_t i = (uint64_t)(spec.start + k);
        bool isnull = in ? tsq_gen_is_null(spec, i) : true;
        if (in) {
            uint64_t s = (spec.kind == TSQ_GEN_HASH_OF_COL) ? src[k] : 0;
            dst[k] = isnull ? 0 : tsq_gen_value(spec, i, s);
        }
        if (bm_words) {
            unsigned long long notnull = __ballot(!isnull);
            if ((threadIdx.x & 63) == 0) {
                // tail word: only write the bytes that belong to the bitmap
                int64_t word = k >> 6;
                int64_t bytes_left = (int64_t)((nrows + 7) / 8) - word * 8;
                if (bytes_left >= 8) bm_words[word] = notnull;
                else {
                    uint8_t* b = (uint8_t*)(bm_words + word);
                    for (int64_t j = 0; j < bytes_left; j++) b[j] = (uint8_t)(notnull >> (8 * j));
                }
            }
        }
    }
}

TSQ_API tsq_status tsq_gen_column(tsq_ctx* ctx, const tsq_gen_spec* spec, int64_t nrows, void* dst,
                                  uint8_t* null_bitmap, const void* src) {
    tsq_ctx_lock _api_lock(ctx);
    if (!ctx || !spec || !dst || nrows < 0) return tsq_fail(ctx ? &ctx->hdr : nullptr, TSQ_ERR_INVALID, "tsq_gen_column: bad args");
    if ((spec->kind == TSQ_GEN_AFFINE || spec->kind == TSQ_GEN_RAND_MOD) && spec->m == 0)
        return tsq_fail(&ctx->hdr, TSQ_ERR_INVALID, "tsq_gen_column: m == 0");
    if (spec->kind == TSQ_GEN_AFFINE && (spec->m >> 31))
        return tsq_fail(&ctx->hdr, TSQ_ERR_INVALID, "tsq_gen_column: AFFINE needs m < 2^31");
    if (spec->kind == TSQ_GEN_HASH_OF_COL && !src)
        return tsq_fail(&ctx->hdr, TSQ_ERR_INVALID, "tsq_gen_column: HASH_OF_COL needs src");
    if (spec->null_pct > 0 && !null_bitmap)
        return tsq_fail(&ctx->hdr, TSQ_ERR_INVALID, "tsq_gen_column: null_pct > 0 needs a null bitmap");
    if (((uintptr_t)null_bitmap & 7) != 0)
        return tsq_fail(&ctx->hdr, TSQ_ERR_INVALID, "tsq_gen_column: null bitmap must be 8-byte aligned");
    if (nrows == 0) return TSQ_OK;
    TSQ_HIP(&ctx->hdr, hipSetDevice(ctx->device));
    int grid = tsq_grid_for(ctx, nrows, 256);
    hipLaunchKernelGGL(k_gen_column, dim3(grid), dim3(256), 0, ctx->stream, *spec, nrows, (uint64_t*)dst,
                       (uint64_t*)null_bitmap, (const uint64_t*)src);
    TSQ_HIP(&ctx->hdr, hipGetLastError());
    return TSQ_OK;
}

// ------------------------------------------------------------------ byte flags -> null bitmap
// notnull_bytes[i] != 0  ->  bit i set.  One 64-row word per wave-iteration (ballot).
__global__ void __launch_bounds__(256) k_pack_bitmap(const uint8_t* __restrict__ flags, uint8_t* __restrict__ bitmap, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t nround = (n + 63) & ~(int64_t)63;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nround; k += stride) {
        bool nn = k < n ? flags[k] != 0 : false;
        unsigned long long w = __ballot(nn);
        if ((threadIdx.x & 63) == 0) {
            int64_t byte0 = (k >> 6) * 8;
            int64_t bytes_left = (n + 7) / 8 - byte0;
            if (bytes_left > 8) bytes_left = 8;
            for (int64_t j = 0; j < bytes_left; j++) bitmap[byte0 + j] = (uint8_t)(w >> (8 * j));
        }
    }
}

// the same, 32 rows per thread: two 16-byte loads of flags -> one 4-byte store (round 6: the byte-per-lane kernel above took 0.27 ms per
// 1e8 rows — 1.1 of the 4.3 ms of a nullable LEFT OUTER probe pass went into packing four bitmaps).  A flag byte becomes its bit with
// an OR-fold (any non-zero value counts) and a multiply that gathers the eight low bits of a 64-bit word into one byte.
__device__ __forceinline__ uint32_t tsq_flags8_to_bits(uint64_t w) {
    w |= w >> 4;
    w |= w >> 2;
    w |= w >> 1;
    w &= 0x0101010101010101ull;
    return (uint32_t)((w * 0x0102040810204080ull) >> 56);
}
__global__ void __launch_bounds__(256) k_pack_bitmap32(const uint4* __restrict__ flags, uint32_t* __restrict__ bitmap, int64_t nwords, const unsigned long long* first_row) {
    const int64_t k0 = first_row ? (int64_t)(*first_row >> 5) : 0;  // (words below: written by the kernel that produced the rows)
    for (int64_t k = k0 + (int64_t)blockIdx.x * 256 + threadIdx.x; k < nwords; k += (int64_t)gridDim.x * 256) {
        const uint4 a = flags[2 * k], b = flags[2 * k + 1];
        const uint32_t b0 = tsq_flags8_to_bits((uint64_t)a.x | ((uint64_t)a.y << 32)), b1 = tsq_flags8_to_bits((uint64_t)a.z | ((uint64_t)a.w << 32));
        const uint32_t b2 = tsq_flags8_to_bits((uint64_t)b.x | ((uint64_t)b.y << 32)), b3 = tsq_flags8_to_bits((uint64_t)b.z | ((uint64_t)b.w << 32));
        bitmap[k] = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
    }
}

tsq_status tsq_launch_pack_bitmap(tsq_ctx* ctx, tsq_handle_hdr* h, const uint8_t* notnull_bytes, uint8_t* bitmap, int64_t n, const unsigned long long* first_row) {
    if (n <= 0) return TSQ_OK;
    // whole 32-row words through the wide kernel (buffers of the operators start on 256-byte boundaries), the tail byte by byte
    int64_t done = 0;
    if (n >= 4096 && ((uintptr_t)notnull_bytes & 15u) == 0 && ((uintptr_t)bitmap & 3u) == 0) {
        const int64_t nwords = n / 32;
        hipLaunchKernelGGL(k_pack_bitmap32, dim3(tsq_grid_for(ctx, nwords, 256)), dim3(256), 0, ctx->stream, reinterpret_cast<const uint4*>(notnull_bytes), reinterpret_cast<uint32_t*>(bitmap), nwords, first_row);
        TSQ_HIP(h, hipGetLastError());
        done = nwords * 32;
        if (done == n) return TSQ_OK;
    }
    int grid = tsq_grid_for(ctx, n - done, 256);
    hipLaunchKernelGGL(k_pack_bitmap, dim3(grid), dim3(256), 0, ctx->stream, notnull_bytes + done, bitmap + done / 8, n - done);
    TSQ_HIP(h, hipGetLastError());
    return TSQ_OK;
}

// ------------------------------------------------------------------ bitmap append (device chunk.List growth)
// dst bits [dst_off, dst_off+n) := src bits [0,n) (or all ones when src == nullptr).  One thread per
// destination byte; edge bytes are read-modify-written by their single owner (appends are stream
// ordered, so there is no concurrent writer).
__global__ void __launch_bounds__(256) k_append_bits(uint8_t* dst, int64_t dst_off, const uint8_t* src, int64_t n) {
    const int64_t first_byte = dst_off >> 3, last_byte = (dst_off + n - 1) >> 3;
    const int64_t nbytes = last_byte - first_byte + 1;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < nbytes; j += stride) {
        const int64_t B = first_byte + j;
        uint8_t v = dst[B];
        for (int bit = 0; bit < 8; bit++) {
            const int64_t g = B * 8 + bit;
            if (g < dst_off || g >= dst_off + n) continue;
            const int64_t s = g - dst_off;
            const bool one = src ? ((src[s >> 3] >> (s & 7)) & 1) : true;
            v = one ? (uint8_t)(v | (1u << bit)) : (uint8_t)(v & ~(1u << bit));
        }
        dst[B] = v;
    }
}


tsq_status tsq_launch_append_bits(tsq_ctx* ctx, tsq_handle_hdr* h, uint8_t* dst, int64_t dst_off, const uint8_t* src_dev, int64_t n) {
    if (n <= 0) return TSQ_OK;
    int64_t nbytes = ((dst_off + n - 1) >> 3) - (dst_off >> 3) + 1;
    int grid = tsq_grid_for(ctx, nbytes, 256);
    hipLaunchKernelGGL(k_append_bits, dim3(grid), dim3(256), 0, ctx->stream, dst, dst_off, src_dev, n);
    TSQ_HIP(h, hipGetLastError());
    return TSQ_OK;
}

// ---------------------------------------------------------------- var-len columns (util/chunk/column.go:28-34: offsets[n + 1] + data)
// dst[i] = src[i] + delta: the offsets of an appended chunk, or of a pulled slice, moved to their new base
__global__ void __launch_bounds__(256) k_offsets_rebase(int64_t* dst, const int64_t* src, int64_t n, int64_t delta) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] = src[i] + delta;
}
tsq_status tsq_launch_offsets_rebase(tsq_ctx* ctx, tsq_handle_hdr* h, int64_t* dst, const int64_t* src_dev, int64_t n, int64_t delta) {
    if (n <= 0) return TSQ_OK;
    hipLaunchKernelGGL(k_offsets_rebase, dim3(tsq_grid_for(ctx, n, 256)), dim3(256), 0, ctx->stream, dst, src_dev, n, delta);
    TSQ_HIP(h, hipGetLastError());
    return TSQ_OK;
}

// exclusive prefix sum of n int64 lengths in place; v[n] receives the total.  Three launches: sums of 2048-element blocks, a
// one-workgroup scan of those, the blocks again with their bases.
#define TSQ_SCAN_BLK 2048
__global__ void __launch_bounds__(256) k_scan64_sums(const int64_t* v, int64_t n, unsigned long long* sums) {
    __shared__ unsigned long long s_w[4];
    const int64_t b0 = (int64_t)blockIdx.x * TSQ_SCAN_BLK;
    unsigned long long acc = 0;
    for (int k = 0; k < TSQ_SCAN_BLK / 256; k++) {
        const int64_t i = b0 + k * 256 + threadIdx.x;
        if (i < n) acc += (unsigned long long)v[i];
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
__global__ void __launch_bounds__(1024) k_scan64_top(unsigned long long* sums, int n) {  // exclusive, in place; sums[n] = total
    __shared__ unsigned long long s_w[16];
    const int per = (n + 1023) / 1024, lo = threadIdx.x * per;
    unsigned long long sum = 0;
    for (int i = lo; i < lo + per && i < n; i++) sum += sums[i];
    unsigned long long x = sum;
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long y = __shfl_up(x, o, 64);
        if ((int)(threadIdx.x & 63) >= o) x += y;
    }
    if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = x;
    __syncthreads();
    unsigned long long pre = 0, tot = 0;
    for (int w = 0; w < 16; w++) {
        if (w < (int)(threadIdx.x >> 6)) pre += s_w[w];
        tot += s_w[w];
    }
    unsigned long long run = pre + x - sum;
    for (int i = lo; i < lo + per && i < n; i++) {
        const unsigned long long c = sums[i];
        sums[i] = run;
        run += c;
    }
    if (threadIdx.x == 0) sums[n] = tot;
}
__global__ void __launch_bounds__(256) k_scan64_apply(int64_t* v, int64_t n, const unsigned long long* sums) {
    __shared__ unsigned long long s_w[4];
    __shared__ unsigned long long s_run;
    const int64_t b0 = (int64_t)blockIdx.x * TSQ_SCAN_BLK;
    if (threadIdx.x == 0) s_run = sums[blockIdx.x];
    __syncthreads();
    for (int k = 0; k < TSQ_SCAN_BLK / 256; k++) {
        const int64_t i = b0 + k * 256 + threadIdx.x;
        const unsigned long long c = i < n ? (unsigned long long)v[i] : 0ull;
        unsigned long long x = c;
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long y = __shfl_up(x, o, 64);
            if ((int)(threadIdx.x & 63) >= o) x += y;
        }
        if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = x;
        __syncthreads();
        unsigned long long pre = s_run;
        for (int w = 0; w < (int)(threadIdx.x >> 6); w++) pre += s_w[w];
        if (i < n) v[i] = (int64_t)(pre + x - c);
        __syncthreads();
        if (threadIdx.x == 0) s_run += s_w[0] + s_w[1] + s_w[2] + s_w[3];
        __syncthreads();
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) v[n] = (int64_t)sums[gridDim.x];
}
// v: n + 1 entries; scratch: (blocks + 1) words
tsq_status tsq_launch_scan64(tsq_ctx* ctx, tsq_handle_hdr* h, int64_t* v, int64_t n, DevBuf& scratch) {
    if (n <= 0) {
        TSQ_HIP(h, hipMemsetAsync(v, 0, 8, ctx->stream));
        return TSQ_OK;
    }
    const int64_t blocks = (n + TSQ_SCAN_BLK - 1) / TSQ_SCAN_BLK;
    if (blocks > 0x7ffffff0LL) return tsq_fail(h, TSQ_ERR_UNSUPPORTED, "scan: too many rows");
    TSQ_TRY(scratch.reserve(ctx, h, (size_t)(blocks + 1) * 8 + 64));
    hipLaunchKernelGGL(k_scan64_sums, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, v, n, scratch.as<unsigned long long>());
    hipLaunchKernelGGL(k_scan64_top, dim3(1), dim3(1024), 0, ctx->stream, scratch.as<unsigned long long>(), (int)blocks);
    hipLaunchKernelGGL(k_scan64_apply, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, v, n, scratch.as<unsigned long long>());
    TSQ_HIP(h, hipGetLastError());
    return TSQ_OK;
}
