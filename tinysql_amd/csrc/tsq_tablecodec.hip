// tsq_tablecodec.hip — the record keys of a table scan <-> handles on the GPU (SURVEY.md §8 f, rank 4: tablecodec).
//
// Replaces tablecodec.DecodeRowKey (tablecodec/tablecodec.go:235-242), called once per scanned KV pair by mocktikv's
// tableScanExec.getRowFromPoint / getRowFromRange (store/mockstore/mocktikv/executor.go:124-196) before the row's value is
// decoded — handles[] is what tsq_rowcodec_decode takes for the PK-handle column — and EncodeRowKeyWithHandle (:65-70), which
// turns handles into the point keys / range ends of a scan (executor/table_reader.go, distsql.TableHandlesToKVRanges).
//
//   record key = 't' | EncodeInt(tableID) | "_r" | EncodeInt(handle)      19 bytes, back to back in `keys`
//
// HBM-bound byte work, no MFMA.  Keys are 19 bytes apart, so a lane-per-key read straight from HBM would touch every 64-byte
// line from three or four lanes with unaligned 8-byte accesses; instead a workgroup copies the contiguous bytes of its 1024 keys
// into LDS with aligned 16-byte loads (tsq_rc_tile_plan, the stored-row decoder's tile staging) and every lane parses its keys
// out of LDS with three aligned words + a funnel shift per 8-byte field.  The encoder builds the keys of a tile in an LDS image
// at the skew of its destination and stores it with aligned 16-byte vectors (tsq_enc_copy_plan, the response encoder's copy-out).
// Algorithmic bytes per key: 19 B + 8 B handle (+ 8 B table id when asked for).
#include "tsq_internal.h"
#include "tsq_rowcodec_dp.h"
#include "tsq_encode_dp.h"
#include "tsq_tablecodec_dp.h"

#define TC_NT 256
#define TC_KPL 4                      // keys per lane and tile
#define TC_TILE (TC_NT * TC_KPL)      // 1024 keys = 19 456 bytes
#define TC_LDS (TC_TILE * 19 + 64)    // + alignment skew (<= 15) + the slack the word reader may touch past the last key

struct TcDecArgs {
    const uint8_t* keys;
    const int64_t* offsets;  // nullptr: key r = bytes [19 r, 19 r + 19)
    int64_t n, n_bytes;
    int64_t* handles;
    int64_t* table_ids;      // nullptr: not wanted
    unsigned long long* err; // min over (key << 4 | code); ~0 = no error
};
struct TcEncArgs {
    int64_t table_id;
    const int64_t* handles;
    int64_t n;
    uint8_t* keys;
};

namespace {

struct TcLds {  // a key inside the staged tile: aligned words + funnel shift (the tile has 16 bytes of slack after its last byte)
    const uint32_t* w;
    uint32_t base;
    __device__ __forceinline__ uint32_t operator()(uint32_t i) const { return ((const uint8_t*)w)[base + i]; }
    __device__ __forceinline__ uint64_t le(uint32_t p, uint32_t) const {
        const uint32_t q = base + p, i = q >> 2;
        return tsq_rc_funnel(w[i], w[i + 1], w[i + 2], q);
    }
};
struct TcGlobal {  // a key in global memory (var-len key lists whose tile does not fit the LDS budget): exactly the bytes asked for
    const uint8_t* p;
    __device__ __forceinline__ uint32_t operator()(uint32_t i) const { return p[i]; }
    __device__ __forceinline__ uint64_t le(uint32_t q, uint32_t n) const { return tsq_rc_le_bytes(*this, q, n); }
};

__global__ void __launch_bounds__(TC_NT) k_rowkeys_decode(TcDecArgs a) {
    __shared__ uint4 s_tile[TC_LDS / 16 + 1];
    const uint32_t tid = threadIdx.x;
    const int64_t n_tiles = (a.n + TC_TILE - 1) / TC_TILE;
    for (int64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const int64_t r0 = t * TC_TILE, r1 = r0 + TC_TILE < a.n ? r0 + TC_TILE : a.n;
        const int64_t tile_lo = a.offsets ? a.offsets[r0] : r0 * 19, tile_hi = a.offsets ? a.offsets[r1] : r1 * 19;
        const tsq_rc_plan plan = tsq_rc_tile_plan((uint64_t)(uintptr_t)a.keys, tile_lo, tile_hi, a.n_bytes, TC_LDS - 32);
        if (plan.staged) {
            const uint4* src = reinterpret_cast<const uint4*>(a.keys + plan.copy_from);  // 16-byte aligned by construction
            for (uint32_t i = tid; i < plan.n_vec; i += TC_NT) s_tile[i] = src[i];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < TC_KPL; k++) {
            const int64_t r = r0 + (int64_t)k * TC_NT + tid;  // consecutive lanes, consecutive keys: coalesced handle stores
            if (r >= r1) continue;
            const int64_t lo = a.offsets ? a.offsets[r] : r * 19, hi = a.offsets ? a.offsets[r + 1] : lo + 19;
            int code;
            int64_t table_id = 0, handle = 0;
            if (lo < tile_lo || hi < lo || hi > tile_hi || lo < 0 || hi > a.n_bytes) {
                code = TC_INVALID_KEY;  // offsets not non-decreasing inside the tile's span / the key bytes
            } else if (plan.staged) {
                const TcLds rd{reinterpret_cast<const uint32_t*>(s_tile), plan.skew + (uint32_t)(lo - tile_lo)};
                code = tsq_tc_decode_row_key(rd, (uint32_t)(hi - lo > 0xffff ? 0xffff : hi - lo), &table_id, &handle);
            } else {
                const TcGlobal rd{a.keys + lo};
                code = tsq_tc_decode_row_key(rd, (uint32_t)(hi - lo > 0xffff ? 0xffff : hi - lo), &table_id, &handle);
            }
            a.handles[r] = handle;
            if (a.table_ids) a.table_ids[r] = table_id;
            if (code != TC_OK) atomicMin(a.err, ((unsigned long long)r << 4) | (unsigned long long)code);
        }
        __syncthreads();  // every wave is done with the tile before the next one is written over it
    }
}

__global__ void __launch_bounds__(TC_NT) k_rowkeys_encode(TcEncArgs a) {
    __shared__ uint4 s_img[TC_LDS / 16 + 1];
    uint8_t* img = reinterpret_cast<uint8_t*>(s_img);
    const uint32_t tid = threadIdx.x;
    const int64_t n_tiles = (a.n + TC_TILE - 1) / TC_TILE;
    for (int64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const int64_t r0 = t * TC_TILE, r1 = r0 + TC_TILE < a.n ? r0 + TC_TILE : a.n;
        const int64_t base = r0 * 19;
        const uint32_t T = (uint32_t)(r1 - r0) * 19u;
        const tsq_enc_copy plan = tsq_enc_copy_plan((uint64_t)(uintptr_t)a.keys, base, T);
#pragma unroll
        for (int k = 0; k < TC_KPL; k++) {
            const int64_t r = r0 + (int64_t)k * TC_NT + tid;
            if (r >= r1) continue;
            uint64_t p0, p1;
            uint32_t p2;
            tsq_tc_encode_row_key(a.table_id, a.handles[r], &p0, &p1, &p2);
            const uint32_t pos = plan.skew + (uint32_t)(r - r0) * 19u;
#pragma unroll
            for (uint32_t i = 0; i < 8; i++) img[pos + i] = (uint8_t)(p0 >> (8 * i));
#pragma unroll
            for (uint32_t i = 0; i < 8; i++) img[pos + 8 + i] = (uint8_t)(p1 >> (8 * i));
#pragma unroll
            for (uint32_t i = 0; i < 3; i++) img[pos + 16 + i] = (uint8_t)(p2 >> (8 * i));
        }
        __syncthreads();
        uint8_t* g = a.keys + base - plan.skew;  // 16-byte aligned by construction
        if (tid < 16 && plan.skew + tid < plan.head_end) g[plan.skew + tid] = img[plan.skew + tid];
        if (tid >= 16 && tid < 32 && plan.tail_lo + (tid - 16) < plan.tail_end) g[plan.tail_lo + (tid - 16)] = img[plan.tail_lo + (tid - 16)];
        for (uint32_t i = plan.body_lo + tid; i < plan.body_hi; i += TC_NT) reinterpret_cast<uint4*>(g)[i] = s_img[i];
        __syncthreads();  // the image is reused by the next tile
    }
}

}  // namespace

// ====================================================================== host side
TSQ_API tsq_status tsq_rowkeys_decode(tsq_ctx* ctx, const uint8_t* keys, int64_t n_bytes, const int64_t* key_offsets, int64_t n_keys,
                                      uint32_t data_flags, int64_t* handles_out, int64_t* table_ids_out, int64_t* nkeys_out) {
    tsq_ctx_lock _api_lock(ctx);
    if (!ctx) return TSQ_ERR_INVALID;
    tsq_handle_hdr* h = &ctx->hdr;
    if (nkeys_out) *nkeys_out = 0;
    if (!nkeys_out || n_keys < 0 || n_bytes < 0 || (n_keys > 0 && (!keys || !handles_out)))
        return tsq_fail(h, TSQ_ERR_INVALID, "tsq_rowkeys_decode: bad arguments");
    if (!key_offsets && n_bytes != n_keys * 19) return tsq_fail(h, TSQ_ERR_INVALID, "tsq_rowkeys_decode: n_bytes != 19 * n_keys without key_offsets");
    if (n_keys >= (1LL << 40)) return tsq_fail(h, TSQ_ERR_UNSUPPORTED, "more than 2^40 keys per call");
    if (n_keys == 0) return TSQ_OK;
    TSQ_HIP(h, hipSetDevice(ctx->device));
    const bool dev = data_flags & TSQ_COL_DEVICE;
    DevBuf dkeys, doffs, dh, dt, derr;
    auto release_all = [&]() { for (DevBuf* b : {&dkeys, &doffs, &dh, &dt, &derr}) b->release(); };
    auto fail = [&](tsq_status st) { release_all(); return st; };
    TcDecArgs a;
    memset(&a, 0, sizeof a);
    a.n = n_keys;
    a.n_bytes = n_bytes;
    tsq_status s = derr.reserve(ctx, h, 64);
    hipError_t e = hipSuccess;
    if (s == TSQ_OK && !dev) {
        s = dkeys.reserve(ctx, h, (size_t)n_bytes + 64);
        if (s == TSQ_OK && key_offsets) s = doffs.reserve(ctx, h, ((size_t)n_keys + 1) * 8 + 64);
        if (s == TSQ_OK) s = dh.reserve(ctx, h, (size_t)n_keys * 8 + 64);
        if (s == TSQ_OK && table_ids_out) s = dt.reserve(ctx, h, (size_t)n_keys * 8 + 64);
        if (s == TSQ_OK && n_bytes > 0) e = hipMemcpyAsync(dkeys.p, keys, (size_t)n_bytes, hipMemcpyHostToDevice, ctx->stream);
        if (s == TSQ_OK && e == hipSuccess && key_offsets) e = hipMemcpyAsync(doffs.p, key_offsets, ((size_t)n_keys + 1) * 8, hipMemcpyHostToDevice, ctx->stream);
        a.keys = dkeys.as<uint8_t>();
        a.offsets = key_offsets ? doffs.as<int64_t>() : nullptr;
        a.handles = dh.as<int64_t>();
        a.table_ids = table_ids_out ? dt.as<int64_t>() : nullptr;
    } else {
        a.keys = keys;
        a.offsets = key_offsets;
        a.handles = handles_out;
        a.table_ids = table_ids_out;
    }
    if (s != TSQ_OK) return fail(s);
    if (e != hipSuccess) return fail(tsq_fail(h, TSQ_ERR_HIP, std::string("tsq_rowkeys_decode(H2D): ") + hipGetErrorString(e)));
    a.err = derr.as<unsigned long long>();
    e = hipMemsetAsync(a.err, 0xff, 8, ctx->stream);
    if (e == hipSuccess) {
        const int64_t n_tiles = (n_keys + TC_TILE - 1) / TC_TILE;
        const int grid = (int)std::min<int64_t>(n_tiles, (int64_t)ctx->num_cus * 8);  // 19.5 KB of LDS per workgroup: eight per CU
        hipLaunchKernelGGL(k_rowkeys_decode, dim3(grid), dim3(TC_NT), 0, ctx->stream, a);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(ctx->pinned, a.err, 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return fail(tsq_fail(h, TSQ_ERR_HIP, std::string("tsq_rowkeys_decode: ") + hipGetErrorString(e)));
    const uint64_t errw = ctx->pinned[0];
    const int64_t good = errw == ~0ull ? n_keys : (int64_t)(errw >> 4);  // the scan has handed the rows before the first bad key on
    if (!dev && good > 0) {
        e = hipMemcpyAsync(handles_out, dh.p, (size_t)good * 8, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess && table_ids_out) e = hipMemcpyAsync(table_ids_out, dt.p, (size_t)good * 8, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) return fail(tsq_fail(h, TSQ_ERR_HIP, std::string("tsq_rowkeys_decode(D2H): ") + hipGetErrorString(e)));
    }
    release_all();
    *nkeys_out = good;
    if (errw != ~0ull) return tsq_fail(h, TSQ_ERR_INVALID, "invalid key");  // errInvalidKey (tablecodec.go:237)
    return TSQ_OK;
}

TSQ_API tsq_status tsq_rowkeys_encode(tsq_ctx* ctx, int64_t table_id, const int64_t* handles, int64_t n_keys, uint32_t data_flags, uint8_t* keys_out) {
    tsq_ctx_lock _api_lock(ctx);
    if (!ctx) return TSQ_ERR_INVALID;
    tsq_handle_hdr* h = &ctx->hdr;
    if (n_keys < 0 || (n_keys > 0 && (!handles || !keys_out))) return tsq_fail(h, TSQ_ERR_INVALID, "tsq_rowkeys_encode: bad arguments");
    if (n_keys >= (1LL << 40)) return tsq_fail(h, TSQ_ERR_UNSUPPORTED, "more than 2^40 keys per call");
    if (n_keys == 0) return TSQ_OK;
    TSQ_HIP(h, hipSetDevice(ctx->device));
    const bool dev = data_flags & TSQ_COL_DEVICE;
    DevBuf dh, dk;
    auto fail = [&](tsq_status st) { dh.release(); dk.release(); return st; };
    TcEncArgs a;
    a.table_id = table_id;
    a.n = n_keys;
    a.handles = handles;
    a.keys = keys_out;
    hipError_t e = hipSuccess;
    if (!dev) {
        tsq_status s = dh.reserve(ctx, h, (size_t)n_keys * 8 + 64);
        if (s == TSQ_OK) s = dk.reserve(ctx, h, (size_t)n_keys * 19 + 64);
        if (s != TSQ_OK) return fail(s);
        e = hipMemcpyAsync(dh.p, handles, (size_t)n_keys * 8, hipMemcpyHostToDevice, ctx->stream);
        a.handles = dh.as<int64_t>();
        a.keys = dk.as<uint8_t>();
    }
    if (e == hipSuccess) {
        const int64_t n_tiles = (n_keys + TC_TILE - 1) / TC_TILE;
        const int grid = (int)std::min<int64_t>(n_tiles, (int64_t)ctx->num_cus * 8);
        hipLaunchKernelGGL(k_rowkeys_encode, dim3(grid), dim3(TC_NT), 0, ctx->stream, a);
        e = hipGetLastError();
    }
    if (e == hipSuccess && !dev) e = hipMemcpyAsync(keys_out, dk.p, (size_t)n_keys * 19, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && !dev) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return fail(tsq_fail(h, TSQ_ERR_HIP, std::string("tsq_rowkeys_encode: ") + hipGetErrorString(e)));
    dh.release();
    dk.release();
    return TSQ_OK;
}
