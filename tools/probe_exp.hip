// probe_exp.hip — experiments that explain the partitioned probe's behaviour (why the L2 locality
// does / does not materialise).  Not product code.
//   E1 workgroup -> XCD placement census
//   E2 probe kernel with table accesses folded into a tiny per-XCD slice (upper bound: all L2 hits)
//   E3 probe kernel with a streaming prefetch of the next table slice(s)
// usage: probe_exp [bits=10] [mode=-1 (all)] [J/CU=4]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../tinysql_amd/csrc/tsq_radix.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_gen(uint64_t* bk, uint64_t* pk, int64_t nb, int64_t np) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += stride) bk[i] = (2654435761ULL * (uint64_t)i + 12345ULL) % (uint64_t)nb;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < np; i += stride) pk[i] = tsq_splitmix64(42 ^ (1ULL << 56) ^ (uint64_t)i) % (uint64_t)nb;
}
__global__ void __launch_bounds__(256) k_build(const uint64_t* bk, int64_t nb, JoinTable t) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nb; r += stride) {
        const uint64_t kw = bk[r];
        uint64_t bkt = tsq_mulhi64(tsq_mix64(kw), t.nbuckets);
        bool done = false;
        while (!done) {
            unsigned long long* base = (unsigned long long*)(t.keys + bkt * TSQ_BUCKET);
            for (int s = 0; s < TSQ_BUCKET && !done; s++) {
                if (__hip_atomic_load(base + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == TSQ_EMPTY_KEY) {
                    unsigned long long old = atomicCAS(base + s, (unsigned long long)TSQ_EMPTY_KEY, (unsigned long long)kw);
                    if (old == TSQ_EMPTY_KEY) done = true;
                }
            }
            bkt = (bkt + 1 == t.nbuckets) ? 0 : bkt + 1;
        }
    }
}
// T1: `readers` workgroups, all on XCD 0, each stream through the same sequence of fresh regions
__global__ void __launch_bounds__(256) k_shared_read(const ulonglong2* buf, size_t region_bytes, int nregions, unsigned long long* out) {
    if ((blockIdx.x & 7) != 0) return;
    uint64_t acc = 0;
    const size_t n16 = region_bytes / 16;
    for (int r = 0; r < nregions; r++) {
        const ulonglong2* p = buf + (size_t)r * n16;
        for (size_t i = threadIdx.x; i < n16; i += 256) {
            const ulonglong2 v = p[i];
            acc += v.x ^ v.y;
        }
    }
    if (acc == 0x1234567) atomicAdd(out, 1ull);
}
__global__ void k_census(uint32_t* out) {
    if (threadIdx.x == 0) out[blockIdx.x] = tsq_xcc_id();
}

// MODE 0: as the product kernel.  MODE 1: bucket folded into 4096 buckets (256 KB) per virtual XCD
// (counts are wrong by construction).  MODE 2: prefetch the slice of partition pi + DEPTH.
template <int U, int MODE>
__global__ void __launch_bounds__(256) k_probe_exp(RadixProbeArgs a, int depth_arg) {
    int depth = depth_arg;
    __shared__ uint32_t s_base[TSQ_RADIX_MAXSEG], s_n[TSQ_RADIX_MAXSEG];
    const uint32_t tid = threadIdx.x;
    const uint32_t vx = blockIdx.x & 7u, j = blockIdx.x >> 3, J = gridDim.x >> 3;
    const uint32_t P = 1u << a.st.bits, NP = P >> 3, R = a.st.R, cap = a.st.cap;
    const uint32_t nseg_p = R >= J ? R / J : 1u;
    const uint32_t S = R >= J ? 1u : J / R;
    const uint32_t nsegs = NP * nseg_p;
    for (uint32_t sg = tid; sg < nsegs; sg += 256) {
        const uint32_t pi = sg / nseg_p, m = sg % nseg_p, p = pi * 8 + vx;
        const uint32_t r = R >= J ? j + m * J : j % R, s = R >= J ? 0u : j / R;
        const uint32_t region = p * R + r;
        uint32_t len = a.st.cursor[region];
        const uint32_t ve = a.st.valid_end[region];
        len = len < ve ? len : ve;
        len = len < cap ? len : cap;
        const uint32_t lo = (uint32_t)((uint64_t)len * s / S), hi = (uint32_t)((uint64_t)len * (s + 1) / S);
        s_base[sg] = region * cap + lo;
        s_n[sg] = hi - lo;
    }
    __syncthreads();
    uint32_t seg = 0, off = 0;
    uint64_t cnt = 0;
    bool alive = true;
    uint32_t pf_acc = 0, pf_prev = 0;
    uint32_t last_pi = 0xffffffffu;
    const uint32_t shift = 64 - a.st.bits;
    auto prefetch_slice = [&](uint32_t pi) {
        if (pi >= NP) return;
        const uint64_t p = (uint64_t)pi * 8 + vx;
        const uint64_t b0 = tsq_mulhi64(p << shift, a.t.nbuckets);
        const uint64_t b1 = p + 1 == P ? a.t.nbuckets : tsq_mulhi64((p + 1) << shift, a.t.nbuckets);
        const uint64_t nlines = ((b1 - b0) * 64 + 127) / 128;  // 128-byte L2 lines
        const uint64_t per = (nlines + J - 1) / J;
        for (uint64_t l = tid; l < per; l += 256) {
            const uint64_t line = j * per + l;
            if (line < nlines) {
                const uint32_t v = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(a.t.keys + b0 * TSQ_BUCKET) + line * 128);
                pf_acc ^= pf_prev;
                pf_prev = v;
            }
        }
    };
    if (MODE >= 2)
        for (int d = 0; d < depth; d++) prefetch_slice((uint32_t)d);
    if (MODE == 3) {  // wait until the whole XCD team has its share of the first slices in L2
        pf_acc ^= pf_prev;
        pf_prev = 0;
        __syncthreads();
        if (tid == 0) {
            if (pf_acc == 0x12345678u) atomicAdd(&a.counters[7], 1ull);
            __hip_atomic_fetch_add(&a.counters[8 + vx], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long target = (unsigned long long)J * (unsigned long long)(depth >> 8 ? (depth >> 8) : 1);
            for (int spin = 0; spin < 2000000; spin++) {
                if (__hip_atomic_load(&a.counters[8 + vx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) break;
                __builtin_amdgcn_s_sleep(2);
            }
        }
        __syncthreads();
        depth &= 255;
    }
    auto fetch = [&](uint64_t& k) -> bool {
        while (seg < nsegs && off >= s_n[seg]) {
            seg++;
            off = 0;
        }
        if (seg >= nsegs) {
            alive = false;
            return false;
        }
        if (MODE >= 2) {
            const uint32_t pi = seg / nseg_p;
            if (pi != last_pi) {
                last_pi = pi;
                prefetch_slice(pi + depth);
            }
        }
        const uint32_t i = off + tid;
        off += 256;
        if (i < s_n[seg]) {
            k = a.st.keys[(size_t)s_base[seg] + i];
            return true;
        }
        return false;
    };
    uint64_t kn[U];
    bool vn[U];
#pragma unroll
    for (int u = 0; u < U; u++) { kn[u] = 0; vn[u] = fetch(kn[u]); }
    bool more = alive || vn[0];
    while (more) {
        uint64_t k[U];
        bool v[U];
#pragma unroll
        for (int u = 0; u < U; u++) { k[u] = kn[u]; v[u] = vn[u]; }
        bool first_alive = false;
#pragma unroll
        for (int u = 0; u < U; u++) {
            kn[u] = 0;
            vn[u] = fetch(kn[u]);
            if (u == 0) first_alive = alive;
        }
        more = first_alive;
        uint64_t bkt[U];
        ulonglong2 L[U][4];
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (v[u]) {
                bkt[u] = tsq_mulhi64(tsq_mix64(k[u]), a.t.nbuckets);
                if (MODE == 1) bkt[u] = (bkt[u] & 4095u) + vx * 4096u;
                const ulonglong2* line = reinterpret_cast<const ulonglong2*>(a.t.keys + bkt[u] * TSQ_BUCKET);
                L[u][0] = line[0]; L[u][1] = line[1]; L[u][2] = line[2]; L[u][3] = line[3];
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (v[u]) {
                const uint64_t kw = k[u];
                const uint64_t w[8] = {L[u][0].x, L[u][0].y, L[u][1].x, L[u][1].y, L[u][2].x, L[u][2].y, L[u][3].x, L[u][3].y};
                uint32_t c = 0;
                bool has_empty = false;
#pragma unroll
                for (int s = 0; s < TSQ_BUCKET; s++) {
                    c += w[s] == kw ? 1u : 0u;
                    has_empty |= w[s] == TSQ_EMPTY_KEY;
                }
                if (MODE != 1 && !has_empty) c += radix_probe_spill(a.t, kw, bkt[u]);
                cnt += c;
            }
        }
    }
    cnt = wave_sum_u64(cnt);
    if ((tid & 63) == 0 && cnt) atomicAdd(&a.counters[0], (unsigned long long)cnt);
    pf_acc ^= pf_prev;
    if (pf_acc == 0x12345678u) atomicAdd(&a.counters[7], 1ull);
}


// E5: ordered dynamic chunk queue per XCD.  Workgroups of virtual XCD vx draw chunk tickets from
// queue[vx]; tickets enumerate the chunks (CH = 256*U keys) of partitions vx, vx+8, ... in order,
// so the keys in flight on one XCD always belong to a window of a few consecutive partitions.
template <int U, bool NT, bool PF>
__global__ void __launch_bounds__(256) k_probe_q(RadixProbeArgs a, int depth, unsigned long long* queue) {
    constexpr uint32_t CH = 256 * U;
    __shared__ uint32_t s_len[TSQ_RADIX_MAXSEG];     // [pi*8 + r]
    __shared__ uint32_t s_cstart[TSQ_RADIX_MAX_P / 8 + 1];
    __shared__ uint32_t s_wsum[4];
    __shared__ uint32_t s_ticket;
    const uint32_t tid = threadIdx.x;
    const uint32_t vx = blockIdx.x & 7u;
    const uint32_t P = 1u << a.st.bits, NP = P >> 3, cap = a.st.cap;
    const uint32_t shift = 64 - a.st.bits;
    for (uint32_t i = tid; i < NP * 8; i += 256) {
        const uint32_t region = ((i >> 3) * 8 + vx) * 8 + (i & 7);
        uint32_t len = a.st.cursor[region];
        const uint32_t ve = a.st.valid_end[region];
        len = len < ve ? len : ve;
        s_len[i] = len < cap ? len : cap;
    }
    __syncthreads();
    {   // chunks per partition -> exclusive prefix (NP <= 256: one partition per thread)
        uint32_t nch = 0;
        if (tid < NP)
            for (int r = 0; r < 8; r++) nch += (s_len[tid * 8 + r] + CH - 1) / CH;
        uint32_t total;
        const uint32_t ex = block_excl_scan<256>(nch, s_wsum, &total);
        if (tid < NP) s_cstart[tid] = ex;
        if (tid == 0) s_cstart[NP] = total;
    }
    __syncthreads();
    const uint32_t nchunks = s_cstart[NP];
    uint64_t cnt = 0;
    uint32_t pf_acc = 0, pf_prev = 0;
    for (;;) {
        if (tid == 0) s_ticket = (uint32_t)__hip_atomic_fetch_add(&queue[vx * 64], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const uint32_t t = s_ticket;
        __syncthreads();
        if (t >= nchunks) break;
        // decode: partition by binary search, then region/chunk by walking the 8 regions
        uint32_t lo = 0, hi = NP;  // invariant: s_cstart[lo] <= t < s_cstart[hi]
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (s_cstart[mid] <= t) lo = mid; else hi = mid;
        }
        const uint32_t pi = lo;
        uint32_t c = t - s_cstart[pi], r = 0;
        for (; r < 8; r++) {
            const uint32_t cr = (s_len[pi * 8 + r] + CH - 1) / CH;
            if (c < cr) break;
            c -= cr;
        }
        const uint32_t p = pi * 8 + vx;
        const uint32_t len = s_len[pi * 8 + r];
        const uint32_t n = len - c * CH < CH ? len - c * CH : CH;
        const uint64_t* src = a.st.keys + (size_t)(p * 8 + r) * cap + (size_t)c * CH;
        if (PF) {  // this ticket's share of the table slice of partition pi + depth
            const uint32_t pj = pi + (uint32_t)depth;
            if (pj < NP) {
                const uint32_t ord = t - s_cstart[pi], nchp = s_cstart[pi + 1] - s_cstart[pi];
                const uint64_t pp = (uint64_t)pj * 8 + vx;
                const uint64_t b0 = tsq_mulhi64(pp << shift, a.t.nbuckets);
                const uint64_t b1 = pp + 1 == P ? a.t.nbuckets : tsq_mulhi64((pp + 1) << shift, a.t.nbuckets);
                const uint64_t nlines = ((b1 - b0) * 64 + 127) / 128;
                const uint64_t l0 = nlines * ord / nchp, l1 = nlines * (ord + 1) / nchp;
                for (uint64_t l = l0 + tid; l < l1; l += 256) {
                    const uint32_t v = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(a.t.keys + b0 * TSQ_BUCKET) + l * 128);
                    pf_acc ^= pf_prev;
                    pf_prev = v;
                }
            }
        }
        uint64_t k[U];
        bool v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t i = u * 256 + tid;
            v[u] = i < n;
            k[u] = 0;
            if (v[u]) k[u] = NT ? __builtin_nontemporal_load(src + i) : src[i];
        }
        uint64_t bkt[U];
        ulonglong2 L[U][4];
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (v[u]) {
                bkt[u] = tsq_mulhi64(tsq_mix64(k[u]), a.t.nbuckets);
                const ulonglong2* line = reinterpret_cast<const ulonglong2*>(a.t.keys + bkt[u] * TSQ_BUCKET);
                L[u][0] = line[0]; L[u][1] = line[1]; L[u][2] = line[2]; L[u][3] = line[3];
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (v[u]) {
                const uint64_t kw = k[u];
                const uint64_t w[8] = {L[u][0].x, L[u][0].y, L[u][1].x, L[u][1].y, L[u][2].x, L[u][2].y, L[u][3].x, L[u][3].y};
                uint32_t cc = 0;
                bool has_empty = false;
#pragma unroll
                for (int s = 0; s < TSQ_BUCKET; s++) {
                    cc += w[s] == kw ? 1u : 0u;
                    has_empty |= w[s] == TSQ_EMPTY_KEY;
                }
                if (!has_empty) cc += radix_probe_spill(a.t, kw, bkt[u]);
                cnt += cc;
            }
        }
    }
    cnt = wave_sum_u64(cnt);
    if ((tid & 63) == 0 && cnt) atomicAdd(&a.counters[0], (unsigned long long)cnt);
    pf_acc ^= pf_prev;
    if (pf_acc == 0x12345678u) atomicAdd(&a.counters[7], 1ull);
}

// E7: E5 with DEFERRED spill handling: a lane whose home bucket is full parks (key, bucket) in an LDS
// list instead of walking the next bucket inside the hot loop; the list is drained by full waves.
// (E5)  Workgroups of virtual XCD vx draw chunk tickets from
// queue[vx]; tickets enumerate the chunks (CH = 256*U keys) of partitions vx, vx+8, ... in order,
// so the keys in flight on one XCD always belong to a window of a few consecutive partitions.
template <int U, bool NT, bool PF>
__global__ void __launch_bounds__(256) k_probe_q7(RadixProbeArgs a, int depth, unsigned long long* queue) {
    constexpr uint32_t CH = 256 * U;
    __shared__ uint32_t s_len[TSQ_RADIX_MAXSEG];     // [pi*8 + r]
    __shared__ uint32_t s_cstart[TSQ_RADIX_MAX_P / 8 + 1];
    __shared__ uint32_t s_wsum[4];
    __shared__ uint32_t s_ticket;
    constexpr uint32_t SPCAP = 2 * CH > 1024 ? 2 * CH : 1024;
    __shared__ uint64_t s_spk[SPCAP], s_spb[SPCAP];
    __shared__ uint32_t s_spn;
    if (threadIdx.x == 0) s_spn = 0;
    const uint32_t tid = threadIdx.x;
    const uint32_t vx = blockIdx.x & 7u;
    const uint32_t P = 1u << a.st.bits, NP = P >> 3, cap = a.st.cap;
    const uint32_t shift = 64 - a.st.bits;
    for (uint32_t i = tid; i < NP * 8; i += 256) {
        const uint32_t region = ((i >> 3) * 8 + vx) * 8 + (i & 7);
        uint32_t len = a.st.cursor[region];
        const uint32_t ve = a.st.valid_end[region];
        len = len < ve ? len : ve;
        s_len[i] = len < cap ? len : cap;
    }
    __syncthreads();
    {   // chunks per partition -> exclusive prefix (NP <= 256: one partition per thread)
        uint32_t nch = 0;
        if (tid < NP)
            for (int r = 0; r < 8; r++) nch += (s_len[tid * 8 + r] + CH - 1) / CH;
        uint32_t total;
        const uint32_t ex = block_excl_scan<256>(nch, s_wsum, &total);
        if (tid < NP) s_cstart[tid] = ex;
        if (tid == 0) s_cstart[NP] = total;
    }
    __syncthreads();
    const uint32_t nchunks = s_cstart[NP];
    uint64_t cnt = 0;
    uint32_t pf_acc = 0, pf_prev = 0;
    for (;;) {
        if (tid == 0) s_ticket = (uint32_t)__hip_atomic_fetch_add(&queue[vx * 64], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const uint32_t t = s_ticket;
        const uint32_t spn = s_spn;
        __syncthreads();
        if (t >= nchunks || spn + CH > SPCAP) {  // drain the parked keys with full waves
            for (uint32_t i = tid; i < spn; i += 256) cnt += radix_probe_spill(a.t, s_spk[i], s_spb[i]);
            __syncthreads();
            if (tid == 0) s_spn = 0;
        }
        if (t >= nchunks) break;
        // decode: partition by binary search, then region/chunk by walking the 8 regions
        uint32_t lo = 0, hi = NP;  // invariant: s_cstart[lo] <= t < s_cstart[hi]
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (s_cstart[mid] <= t) lo = mid; else hi = mid;
        }
        const uint32_t pi = lo;
        uint32_t c = t - s_cstart[pi], r = 0;
        for (; r < 8; r++) {
            const uint32_t cr = (s_len[pi * 8 + r] + CH - 1) / CH;
            if (c < cr) break;
            c -= cr;
        }
        const uint32_t p = pi * 8 + vx;
        const uint32_t len = s_len[pi * 8 + r];
        const uint32_t n = len - c * CH < CH ? len - c * CH : CH;
        const uint64_t* src = a.st.keys + (size_t)(p * 8 + r) * cap + (size_t)c * CH;
        if (PF) {  // this ticket's share of the table slice of partition pi + depth
            const uint32_t pj = pi + (uint32_t)depth;
            if (pj < NP) {
                const uint32_t ord = t - s_cstart[pi], nchp = s_cstart[pi + 1] - s_cstart[pi];
                const uint64_t pp = (uint64_t)pj * 8 + vx;
                const uint64_t b0 = tsq_mulhi64(pp << shift, a.t.nbuckets);
                const uint64_t b1 = pp + 1 == P ? a.t.nbuckets : tsq_mulhi64((pp + 1) << shift, a.t.nbuckets);
                const uint64_t nlines = ((b1 - b0) * 64 + 127) / 128;
                const uint64_t l0 = nlines * ord / nchp, l1 = nlines * (ord + 1) / nchp;
                for (uint64_t l = l0 + tid; l < l1; l += 256) {
                    const uint32_t v = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(a.t.keys + b0 * TSQ_BUCKET) + l * 128);
                    pf_acc ^= pf_prev;
                    pf_prev = v;
                }
            }
        }
        uint64_t k[U];
        bool v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t i = u * 256 + tid;
            v[u] = i < n;
            k[u] = 0;
            if (v[u]) k[u] = NT ? __builtin_nontemporal_load(src + i) : src[i];
        }
        uint64_t bkt[U];
        ulonglong2 L[U][4];
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (v[u]) {
                bkt[u] = tsq_mulhi64(tsq_mix64(k[u]), a.t.nbuckets);
                const ulonglong2* line = reinterpret_cast<const ulonglong2*>(a.t.keys + bkt[u] * TSQ_BUCKET);
                L[u][0] = line[0]; L[u][1] = line[1]; L[u][2] = line[2]; L[u][3] = line[3];
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (v[u]) {
                const uint64_t kw = k[u];
                const uint64_t w[8] = {L[u][0].x, L[u][0].y, L[u][1].x, L[u][1].y, L[u][2].x, L[u][2].y, L[u][3].x, L[u][3].y};
                uint32_t cc = 0;
                bool has_empty = false;
#pragma unroll
                for (int s = 0; s < TSQ_BUCKET; s++) {
                    cc += w[s] == kw ? 1u : 0u;
                    has_empty |= w[s] == TSQ_EMPTY_KEY;
                }
                if (!has_empty) {
                    const uint32_t sl = atomicAdd(&s_spn, 1u);
                    s_spk[sl] = kw;
                    s_spb[sl] = bkt[u];
                }
                cnt += cc;
            }
        }
    }
    cnt = wave_sum_u64(cnt);
    if ((tid & 63) == 0 && cnt) atomicAdd(&a.counters[0], (unsigned long long)cnt);
    pf_acc ^= pf_prev;
    if (pf_acc == 0x12345678u) atomicAdd(&a.counters[7], 1ull);
}

// E7t: instrumented (wall_clock64 per phase, 100 MHz) copy of E7: E5 with DEFERRED spill handling: a lane whose home bucket is full parks (key, bucket) in an LDS
// list instead of walking the next bucket inside the hot loop; the list is drained by full waves.
// (E5)  Workgroups of virtual XCD vx draw chunk tickets from
// queue[vx]; tickets enumerate the chunks (CH = 256*U keys) of partitions vx, vx+8, ... in order,
// so the keys in flight on one XCD always belong to a window of a few consecutive partitions.
template <int U, bool NT, bool PF>
__global__ void __launch_bounds__(256) k_probe_q7t(RadixProbeArgs a, int depth, unsigned long long* queue) {
    constexpr uint32_t CH = 256 * U;
    __shared__ uint32_t s_len[TSQ_RADIX_MAXSEG];     // [pi*8 + r]
    __shared__ uint32_t s_cstart[TSQ_RADIX_MAX_P / 8 + 1];
    __shared__ uint32_t s_wsum[4];
    __shared__ uint32_t s_ticket;
    constexpr uint32_t SPCAP = 2 * CH > 1024 ? 2 * CH : 1024;
    __shared__ uint64_t s_spk[SPCAP], s_spb[SPCAP];
    __shared__ uint32_t s_spn;
    if (threadIdx.x == 0) s_spn = 0;
    const uint32_t tid = threadIdx.x;
    const uint32_t vx = blockIdx.x & 7u;
    const uint32_t P = 1u << a.st.bits, NP = P >> 3, cap = a.st.cap;
    const uint32_t shift = 64 - a.st.bits;
    for (uint32_t i = tid; i < NP * 8; i += 256) {
        const uint32_t region = ((i >> 3) * 8 + vx) * 8 + (i & 7);
        uint32_t len = a.st.cursor[region];
        const uint32_t ve = a.st.valid_end[region];
        len = len < ve ? len : ve;
        s_len[i] = len < cap ? len : cap;
    }
    __syncthreads();
    {   // chunks per partition -> exclusive prefix (NP <= 256: one partition per thread)
        uint32_t nch = 0;
        if (tid < NP)
            for (int r = 0; r < 8; r++) nch += (s_len[tid * 8 + r] + CH - 1) / CH;
        uint32_t total;
        const uint32_t ex = block_excl_scan<256>(nch, s_wsum, &total);
        if (tid < NP) s_cstart[tid] = ex;
        if (tid == 0) s_cstart[NP] = total;
    }
    __syncthreads();
    const uint32_t nchunks = s_cstart[NP];
    uint64_t cnt = 0;
    uint32_t pf_acc = 0, pf_prev = 0;
    uint64_t tA = 0, tB = 0, tC = 0, tD = 0, tE = 0, nchk = 0;
    for (;;) {
        const uint64_t c0 = wall_clock64();
        if (tid == 0) s_ticket = (uint32_t)__hip_atomic_fetch_add(&queue[vx * 64], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const uint32_t t = s_ticket;
        const uint32_t spn = s_spn;
        __syncthreads();
        if (t >= nchunks || spn + CH > SPCAP) {  // drain the parked keys with full waves
            for (uint32_t i = tid; i < spn; i += 256) cnt += radix_probe_spill(a.t, s_spk[i], s_spb[i]);
            __syncthreads();
            if (tid == 0) s_spn = 0;
        }
        if (t >= nchunks) break;
        const uint64_t c1 = wall_clock64();
        // decode: partition by binary search, then region/chunk by walking the 8 regions
        uint32_t lo = 0, hi = NP;  // invariant: s_cstart[lo] <= t < s_cstart[hi]
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (s_cstart[mid] <= t) lo = mid; else hi = mid;
        }
        const uint32_t pi = lo;
        uint32_t c = t - s_cstart[pi], r = 0;
        for (; r < 8; r++) {
            const uint32_t cr = (s_len[pi * 8 + r] + CH - 1) / CH;
            if (c < cr) break;
            c -= cr;
        }
        const uint32_t p = pi * 8 + vx;
        const uint32_t len = s_len[pi * 8 + r];
        const uint32_t n = len - c * CH < CH ? len - c * CH : CH;
        const uint64_t* src = a.st.keys + (size_t)(p * 8 + r) * cap + (size_t)c * CH;
        if (PF) {  // this ticket's share of the table slice of partition pi + depth
            const uint32_t pj = pi + (uint32_t)depth;
            if (pj < NP) {
                const uint32_t ord = t - s_cstart[pi], nchp = s_cstart[pi + 1] - s_cstart[pi];
                const uint64_t pp = (uint64_t)pj * 8 + vx;
                const uint64_t b0 = tsq_mulhi64(pp << shift, a.t.nbuckets);
                const uint64_t b1 = pp + 1 == P ? a.t.nbuckets : tsq_mulhi64((pp + 1) << shift, a.t.nbuckets);
                const uint64_t nlines = ((b1 - b0) * 64 + 127) / 128;
                const uint64_t l0 = nlines * ord / nchp, l1 = nlines * (ord + 1) / nchp;
                for (uint64_t l = l0 + tid; l < l1; l += 256) {
                    const uint32_t v = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(a.t.keys + b0 * TSQ_BUCKET) + l * 128);
                    pf_acc ^= pf_prev;
                    pf_prev = v;
                }
            }
        }
        const uint64_t c2 = wall_clock64();
        uint64_t k[U];
        bool v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t i = u * 256 + tid;
            v[u] = i < n;
            k[u] = 0;
            if (v[u]) k[u] = NT ? __builtin_nontemporal_load(src + i) : src[i];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint64_t c3 = wall_clock64();
        uint64_t bkt[U];
        ulonglong2 L[U][4];
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (v[u]) {
                bkt[u] = tsq_mulhi64(tsq_mix64(k[u]), a.t.nbuckets);
                const ulonglong2* line = reinterpret_cast<const ulonglong2*>(a.t.keys + bkt[u] * TSQ_BUCKET);
                L[u][0] = line[0]; L[u][1] = line[1]; L[u][2] = line[2]; L[u][3] = line[3];
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint64_t c4 = wall_clock64();
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (v[u]) {
                const uint64_t kw = k[u];
                const uint64_t w[8] = {L[u][0].x, L[u][0].y, L[u][1].x, L[u][1].y, L[u][2].x, L[u][2].y, L[u][3].x, L[u][3].y};
                uint32_t cc = 0;
                bool has_empty = false;
#pragma unroll
                for (int s = 0; s < TSQ_BUCKET; s++) {
                    cc += w[s] == kw ? 1u : 0u;
                    has_empty |= w[s] == TSQ_EMPTY_KEY;
                }
                if (!has_empty) {
                    const uint32_t sl = atomicAdd(&s_spn, 1u);
                    s_spk[sl] = kw;
                    s_spb[sl] = bkt[u];
                }
                cnt += cc;
            }
        }
        const uint64_t c5 = wall_clock64();
        tA += c1 - c0; tB += c2 - c1; tC += c3 - c2; tD += c4 - c3; tE += c5 - c4; nchk++;
    }
    if (tid == 0) {
        atomicAdd(&a.counters[16], (unsigned long long)tA); atomicAdd(&a.counters[17], (unsigned long long)tB); atomicAdd(&a.counters[18], (unsigned long long)tC);
        atomicAdd(&a.counters[19], (unsigned long long)tD); atomicAdd(&a.counters[20], (unsigned long long)tE); atomicAdd(&a.counters[21], (unsigned long long)nchk);
    }
    cnt = wave_sum_u64(cnt);
    if ((tid & 63) == 0 && cnt) atomicAdd(&a.counters[0], (unsigned long long)cnt);
    pf_acc ^= pf_prev;
    if (pf_acc == 0x12345678u) atomicAdd(&a.counters[7], 1ull);
}

// E6: E5 + software pipeline.  Thread 0 is the scout: the ticket atomic for chunk i+2 is in flight
// while chunk i+1's descriptor is decoded and chunk i is probed; key loads of chunk i+1 are issued
// AFTER the table loads of chunk i so that waiting for the table lines does not wait for them.
template <int U, bool PF>
__global__ void __launch_bounds__(256) k_probe_q2(RadixProbeArgs a, int depth, unsigned long long* queue) {
    constexpr uint32_t CH = 256 * U;
    constexpr uint32_t END = 0xffffffffu;
    __shared__ uint32_t s_len[TSQ_RADIX_MAXSEG];
    __shared__ uint32_t s_cstart[TSQ_RADIX_MAX_P / 8 + 1];
    __shared__ uint32_t s_wsum[4];
    __shared__ uint64_t s_dsrc[2], s_dpf[2];
    __shared__ uint32_t s_dn[2], s_dl0[2], s_dl1[2];
    const uint32_t tid = threadIdx.x;
    const uint32_t vx = blockIdx.x & 7u;
    const uint32_t P = 1u << a.st.bits, NP = P >> 3, cap = a.st.cap;
    const uint32_t shift = 64 - a.st.bits;
    for (uint32_t i = tid; i < NP * 8; i += 256) {
        const uint32_t region = ((i >> 3) * 8 + vx) * 8 + (i & 7);
        uint32_t len = a.st.cursor[region];
        const uint32_t ve = a.st.valid_end[region];
        len = len < ve ? len : ve;
        s_len[i] = len < cap ? len : cap;
    }
    __syncthreads();
    {
        uint32_t nch = 0;
        if (tid < NP)
            for (int r = 0; r < 8; r++) nch += (s_len[tid * 8 + r] + CH - 1) / CH;
        uint32_t total;
        const uint32_t ex = block_excl_scan<256>(nch, s_wsum, &total);
        if (tid < NP) s_cstart[tid] = ex;
        if (tid == 0) s_cstart[NP] = total;
    }
    __syncthreads();
    const uint32_t nchunks = s_cstart[NP];
    auto take = [&]() -> uint32_t { return (uint32_t)__hip_atomic_fetch_add(&queue[vx * 64], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto decode = [&](uint32_t t, int slot) {  // thread 0 only
        if (t >= nchunks) { s_dn[slot] = END; return; }
        uint32_t lo = 0, hi = NP;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (s_cstart[mid] <= t) lo = mid; else hi = mid;
        }
        const uint32_t pi = lo;
        uint32_t c = t - s_cstart[pi], r = 0;
        for (; r < 8; r++) {
            const uint32_t cr = (s_len[pi * 8 + r] + CH - 1) / CH;
            if (c < cr) break;
            c -= cr;
        }
        const uint32_t p = pi * 8 + vx, len = s_len[pi * 8 + r];
        s_dn[slot] = len - c * CH < CH ? len - c * CH : CH;
        s_dsrc[slot] = (uint64_t)(a.st.keys + (size_t)(p * 8 + r) * cap + (size_t)c * CH);
        if (PF) {
            const uint32_t pj = pi + (uint32_t)depth;
            uint32_t l0 = 0, l1 = 0;
            uint64_t b0 = 0;
            if (pj < NP) {
                const uint32_t ord = t - s_cstart[pi], nchp = s_cstart[pi + 1] - s_cstart[pi];
                const uint64_t pp = (uint64_t)pj * 8 + vx;
                b0 = tsq_mulhi64(pp << shift, a.t.nbuckets);
                const uint64_t b1 = pp + 1 == P ? a.t.nbuckets : tsq_mulhi64((pp + 1) << shift, a.t.nbuckets);
                const uint64_t nlines = ((b1 - b0) * 64 + 127) / 128;
                l0 = (uint32_t)(nlines * ord / nchp);
                l1 = (uint32_t)(nlines * (ord + 1) / nchp);
            }
            s_dpf[slot] = (uint64_t)(a.t.keys + b0 * TSQ_BUCKET);
            s_dl0[slot] = l0;
            s_dl1[slot] = l1;
        }
    };
    uint64_t cnt = 0;
    uint32_t pf_acc = 0, pf_prev = 0;
    uint32_t tk_pending = 0;
    if (tid == 0) {
        const uint32_t tA = take(), tB = take();
        tk_pending = take();
        decode(tA, 0);
        decode(tB, 1);
    }
    __syncthreads();
    uint64_t kn[U];
    uint32_t nn = s_dn[0];
#pragma unroll
    for (int u = 0; u < U; u++) {
        kn[u] = 0;
        const uint32_t i = u * 256 + tid;
        if (nn != END && i < nn) kn[u] = __builtin_nontemporal_load((const uint64_t*)s_dsrc[0] + i);
    }
    for (uint32_t it = 0;; it++) {
        const uint32_t n = nn;
        if (n == END) break;
        uint64_t k[U], bkt[U];
        ulonglong2 L[U][4];
#pragma unroll
        for (int u = 0; u < U; u++) {
            k[u] = kn[u];
            if (u * 256 + tid < n) {
                bkt[u] = tsq_mulhi64(tsq_mix64(k[u]), a.t.nbuckets);
                const ulonglong2* line = reinterpret_cast<const ulonglong2*>(a.t.keys + bkt[u] * TSQ_BUCKET);
                L[u][0] = line[0]; L[u][1] = line[1]; L[u][2] = line[2]; L[u][3] = line[3];
            }
        }
        // next chunk's keys (descriptor slot (it+1)&1) — issued after the table loads
        const int ns = (it + 1) & 1;
        nn = s_dn[ns];
#pragma unroll
        for (int u = 0; u < U; u++) {
            kn[u] = 0;
            const uint32_t i = u * 256 + tid;
            if (nn != END && i < nn) kn[u] = __builtin_nontemporal_load((const uint64_t*)s_dsrc[ns] + i);
        }
        if (PF && nn != END) {
            const char* pb = (const char*)s_dpf[ns];
            for (uint32_t l = s_dl0[ns] + tid; l < s_dl1[ns]; l += 256) {
                const uint32_t v = *reinterpret_cast<const uint32_t*>(pb + (size_t)l * 128);
                pf_acc ^= pf_prev;
                pf_prev = v;
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (u * 256 + tid < n) {
                const uint64_t kw = k[u];
                const uint64_t w[8] = {L[u][0].x, L[u][0].y, L[u][1].x, L[u][1].y, L[u][2].x, L[u][2].y, L[u][3].x, L[u][3].y};
                uint32_t cc = 0;
                bool has_empty = false;
#pragma unroll
                for (int s = 0; s < TSQ_BUCKET; s++) {
                    cc += w[s] == kw ? 1u : 0u;
                    has_empty |= w[s] == TSQ_EMPTY_KEY;
                }
                if (!has_empty) cc += radix_probe_spill(a.t, kw, bkt[u]);
                cnt += cc;
            }
        }
        __syncthreads();  // everyone has read descriptor slot ns^1 (= it & 1) long ago and slot ns above
        if (tid == 0) {   // chunk it+2 -> slot it & 1
            decode(tk_pending, it & 1);
            tk_pending = take();
        }
        __syncthreads();
    }
    cnt = wave_sum_u64(cnt);
    if ((tid & 63) == 0 && cnt) atomicAdd(&a.counters[0], (unsigned long long)cnt);
    pf_acc ^= pf_prev;
    if (pf_acc == 0x12345678u) atomicAdd(&a.counters[7], 1ull);
}

// E8: ordered queue + software pipeline + deferred spill.  One __syncthreads per chunk.  Thread 0
// (scout) decodes the ticket of chunk it+2 while the table loads of chunk it are in flight.
template <int U, bool QUAD>
__global__ void __launch_bounds__(256) k_probe_q8(RadixProbeArgs a, unsigned long long* queue) {
    constexpr uint32_t CH = 256 * U;
    constexpr uint32_t END = 0xffffffffu;
    constexpr uint32_t SPCAP = 2 * CH > 1024 ? 2 * CH : 1024;
    __shared__ uint32_t s_len[TSQ_RADIX_MAXSEG];
    __shared__ uint32_t s_cstart[TSQ_RADIX_MAX_P / 8 + 1];
    __shared__ uint32_t s_wsum[4];
    __shared__ uint64_t s_dsrc[2];
    __shared__ uint32_t s_dn[2];
    __shared__ uint64_t s_spk[SPCAP], s_spb[SPCAP];
    __shared__ uint32_t s_spn;
    const uint32_t tid = threadIdx.x;
    const uint32_t vx = blockIdx.x & 7u;
    const uint32_t P = 1u << a.st.bits, NP = P >> 3, cap = a.st.cap;
    if (tid == 0) s_spn = 0;
    for (uint32_t i = tid; i < NP * 8; i += 256) {
        const uint32_t region = ((i >> 3) * 8 + vx) * 8 + (i & 7);
        uint32_t len = a.st.cursor[region];
        const uint32_t ve = a.st.valid_end[region];
        len = len < ve ? len : ve;
        s_len[i] = len < cap ? len : cap;
    }
    __syncthreads();
    {
        uint32_t nch = 0;
        if (tid < NP)
            for (int r = 0; r < 8; r++) nch += (s_len[tid * 8 + r] + CH - 1) / CH;
        uint32_t total;
        const uint32_t ex = block_excl_scan<256>(nch, s_wsum, &total);
        if (tid < NP) s_cstart[tid] = ex;
        if (tid == 0) s_cstart[NP] = total;
    }
    __syncthreads();
    const uint32_t nchunks = s_cstart[NP];
    auto take = [&]() -> uint32_t { return (uint32_t)__hip_atomic_fetch_add(&queue[vx * 64], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto decode = [&](uint32_t t, int slot) {  // thread 0 only
        if (t >= nchunks) { s_dn[slot] = END; return; }
        uint32_t lo = 0, hi = NP;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (s_cstart[mid] <= t) lo = mid; else hi = mid;
        }
        const uint32_t pi = lo;
        uint32_t c = t - s_cstart[pi], r = 0;
        for (; r < 8; r++) {
            const uint32_t cr = (s_len[pi * 8 + r] + CH - 1) / CH;
            if (c < cr) break;
            c -= cr;
        }
        const uint32_t p = pi * 8 + vx, len = s_len[pi * 8 + r];
        s_dn[slot] = len - c * CH < CH ? len - c * CH : CH;
        s_dsrc[slot] = (uint64_t)(a.st.keys + (size_t)(p * 8 + r) * cap + (size_t)c * CH);
    };
    uint64_t cnt = 0;
    uint32_t tk_pending = 0;
    if (tid == 0) {
        const uint32_t tA = take(), tB = take();
        tk_pending = take();
        decode(tA, 0);
        decode(tB, 1);
    }
    __syncthreads();
    uint64_t kn[U];
    uint32_t nn = s_dn[0];
#pragma unroll
    for (int u = 0; u < U; u++) {
        kn[u] = 0;
        const uint32_t i = u * 256 + tid;
        if (nn != END && i < nn) kn[u] = __builtin_nontemporal_load((const uint64_t*)s_dsrc[0] + i);
    }
    const int lane = tid & 63;
    for (uint32_t it = 0;; it++) {
        const uint32_t n = nn;
        if (n == END) break;
        uint64_t k[U], bkt[U];
        ulonglong2 L[U][4];
#pragma unroll
        for (int u = 0; u < U; u++) {
            k[u] = kn[u];
            const bool valid = u * 256 + tid < n;
            bkt[u] = valid ? tsq_mulhi64(tsq_mix64(k[u]), a.t.nbuckets) : 0;
            if (QUAD) {  // 4 adjacent lanes read one bucket: L[u][s] = 16 bytes (lane & 3) of the bucket of lane 16 s + (lane >> 2)
#pragma unroll
                for (int sg = 0; sg < 4; sg++) {
                    const uint64_t bq = __shfl(bkt[u], sg * 16 + (lane >> 2), 64);
                    L[u][sg] = reinterpret_cast<const ulonglong2*>(a.t.keys + bq * TSQ_BUCKET)[lane & 3];
                }
            } else if (valid) {
                const ulonglong2* line = reinterpret_cast<const ulonglong2*>(a.t.keys + bkt[u] * TSQ_BUCKET);
                L[u][0] = line[0]; L[u][1] = line[1]; L[u][2] = line[2]; L[u][3] = line[3];
            }
        }
        const int ns = (it + 1) & 1;
        nn = s_dn[ns];
#pragma unroll
        for (int u = 0; u < U; u++) {
            kn[u] = 0;
            const uint32_t i = u * 256 + tid;
            if (nn != END && i < nn) kn[u] = __builtin_nontemporal_load((const uint64_t*)s_dsrc[ns] + i);
        }
        if (tid == 0) {  // chunk it+2 -> slot it & 1 (its readers finished before the previous barrier)
            decode(tk_pending, it & 1);
            tk_pending = take();
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (QUAD) {
#pragma unroll
                for (int sg = 0; sg < 4; sg++) {
                    const int srcl = sg * 16 + (lane >> 2);
                    const uint64_t kw = __shfl(k[u], srcl, 64);
                    const bool valid = (uint32_t)(u * 256 + (tid & ~63) + srcl) < n;
                    const bool e = L[u][sg].x == TSQ_EMPTY_KEY || L[u][sg].y == TSQ_EMPTY_KEY;
                    const uint64_t em = __ballot(e);
                    if (valid) {
                        cnt += (L[u][sg].x == kw ? 1u : 0u) + (L[u][sg].y == kw ? 1u : 0u);
                        if ((lane & 3) == 0 && ((em >> (lane & ~3)) & 0xfull) == 0) {
                            const uint32_t sl = atomicAdd(&s_spn, 1u);
                            s_spk[sl] = kw;
                            s_spb[sl] = __shfl(bkt[u], srcl, 64);
                        }
                    }
                }
            } else if (u * 256 + tid < n) {
                const uint64_t kw = k[u];
                const uint64_t w[8] = {L[u][0].x, L[u][0].y, L[u][1].x, L[u][1].y, L[u][2].x, L[u][2].y, L[u][3].x, L[u][3].y};
                uint32_t cc = 0;
                bool has_empty = false;
#pragma unroll
                for (int sl2 = 0; sl2 < TSQ_BUCKET; sl2++) {
                    cc += w[sl2] == kw ? 1u : 0u;
                    has_empty |= w[sl2] == TSQ_EMPTY_KEY;
                }
                if (!has_empty) {
                    const uint32_t sl = atomicAdd(&s_spn, 1u);
                    s_spk[sl] = kw;
                    s_spb[sl] = bkt[u];
                }
                cnt += cc;
            }
        }
        __syncthreads();
        const uint32_t spn = s_spn;
        if (nn == END || spn + CH > SPCAP) {
            for (uint32_t i = tid; i < spn; i += 256) cnt += radix_probe_spill(a.t, s_spk[i], s_spb[i]);
            __syncthreads();
            if (tid == 0) s_spn = 0;
            __syncthreads();
        }
    }
    cnt = wave_sum_u64(cnt);
    if ((tid & 63) == 0 && cnt) atomicAdd(&a.counters[0], (unsigned long long)cnt);
}

template <class F>
static float time_ms(F&& launch, int reps = 3) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e30f;
    for (int r = 0; r < reps; r++) {
        CK(hipEventRecord(a));
        launch();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return best;
}

int main(int argc, char** argv) {
    const uint32_t bits = argc > 1 ? (uint32_t)atoi(argv[1]) : 10u;
    const int only = argc > 2 ? atoi(argv[2]) : -1;
    const int Jc = argc > 3 ? atoi(argv[3]) : 4;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int64_t NB = 100000000, NP = 100000000;
    const int CUS = prop.multiProcessorCount;
    uint64_t *bk, *pk;
    CK(hipMalloc(&bk, NB * 8)); CK(hipMalloc(&pk, NP * 8));
    hipLaunchKernelGGL(k_gen, dim3(CUS * 8), dim3(256), 0, 0, bk, pk, NB, NP);
    JoinTable t{};
    t.nbuckets = (uint64_t)((NB + 3) / 4);
    CK(hipMalloc(&t.keys, t.nbuckets * 64 + 256));
    CK(hipMemset(t.keys, 0x80, t.nbuckets * 64 + 256));
    unsigned long long* counters;
    CK(hipMalloc(&counters, 8192)); CK(hipMemset(counters, 0, 8192));
    hipLaunchKernelGGL(k_build, dim3(CUS * 8), dim3(256), 0, 0, bk, NB, t);
    CK(hipDeviceSynchronize());
    if (only < 0) {  // E1
        uint32_t* d; CK(hipMalloc(&d, 2048 * 4));
        hipLaunchKernelGGL(k_census, dim3(2048), dim3(256), 0, 0, d);
        std::vector<uint32_t> h(2048);
        CK(hipMemcpy(h.data(), d, 2048 * 4, hipMemcpyDeviceToHost));
        int match = 0, hist[8] = {0};
        for (int b = 0; b < 2048; b++) { match += (h[b] == (uint32_t)(b & 7)); hist[h[b]]++; }
        printf("E1 census: %d/2048 workgroups on XCD (b %% 8); per-XCD counts %d %d %d %d %d %d %d %d; first 16:", match, hist[0], hist[1], hist[2], hist[3], hist[4], hist[5], hist[6], hist[7]);
        for (int b = 0; b < 16; b++) printf(" %u", h[b]);
        printf("\n");
    }
    RadixStore st{};
    const uint32_t P = 1u << bits;
    st.bits = bits; st.R = 8;
    const double lam = (double)NP / ((double)P * 8);
    st.cap = ((uint32_t)(lam * 1.08 + 8 * sqrt(lam) + 2.0 * 16384 / 64 + 64) + 15) & ~15u;
    CK(hipMalloc(&st.keys, (size_t)P * 8 * st.cap * 8));
    CK(hipMalloc(&st.cursor, P * 8 * 4)); CK(hipMemset(st.cursor, 0, P * 8 * 4));
    CK(hipMalloc(&st.valid_end, P * 8 * 4)); CK(hipMemset(st.valid_end, 0xff, P * 8 * 4));
    CK(hipMalloc(&st.ovf_keys, (size_t)NP * 8)); CK(hipMalloc(&st.ovf_count, 4)); CK(hipMemset(st.ovf_count, 0, 4));
    st.ovf_cap = (uint32_t)NP;
    RadixSrc src{};
    src.data = pk; src.type = TSQ_I64; src.nrows = NP;
    hipLaunchKernelGGL((k_radix_partition<1024, 16, 4, 0, false>), dim3(CUS), dim3(1024), 0, 0, src, st);
    CK(hipDeviceSynchronize());
    RadixProbeArgs pa{};
    pa.st = st; pa.t = t; pa.counters = counters;
    const uint32_t J = (uint32_t)(CUS / 8 * Jc);
    auto report = [&](const char* name, float ms) {
        unsigned long long c[8];
        CK(hipMemcpy(c, counters, 64, hipMemcpyDeviceToHost)); CK(hipMemset(counters, 0, 256));
        printf("%-44s bits=%u J/CU=%d : %.3f ms  %.1f Grows/s  count=%llu\n", name, bits, Jc, ms, NP / ms / 1e6, c[0]);
    };
#define RUN(idx, name, U, MODE, depth)                                                                                        \
    if (only < 0 || only == idx) {                                                                                            \
        float ms = time_ms([&] { hipLaunchKernelGGL((k_probe_exp<U, MODE>), dim3(J * 8), dim3(256), 0, 0, pa, depth); }, 3);  \
        CK(hipGetLastError());                                                                                                \
        report(name, ms);                                                                                                     \
    }
    RUN(0, "E0 normal U=2", 2, 0, 0)
    RUN(1, "E2 folded table (all L2 hits) U=1", 1, 1, 0)
    RUN(2, "E2 folded table (all L2 hits) U=2", 2, 1, 0)
    RUN(3, "E2 folded table (all L2 hits) U=4", 4, 1, 0)
    RUN(4, "E3 prefetch depth 1 U=2", 2, 2, 1)
    RUN(5, "E3 prefetch depth 2 U=2", 2, 2, 2)
    RUN(6, "E3 prefetch depth 3 U=2", 2, 2, 3)
    RUN(7, "E3 prefetch depth 2 U=4", 4, 2, 2)
    RUN(8, "E3 prefetch depth 4 U=2", 2, 2, 4)
#define RUNB(idx, name, U, depth)                                                                                             \
    if (only < 0 || only == idx) {                                                                                            \
        float best = 1e30f;                                                                                                   \
        for (int rep = 0; rep < 3; rep++) {                                                                                   \
            CK(hipMemset(counters + 8, 0, 64));                                                                               \
            float ms = time_ms([&] { hipLaunchKernelGGL((k_probe_exp<U, 3>), dim3(J * 8), dim3(256), 0, 0, pa, depth); }, 1); \
            best = ms < best ? ms : best;                                                                                     \
        }                                                                                                                     \
        CK(hipGetLastError());                                                                                                \
        report(name, best);                                                                                                   \
    }
    RUNB(9, "E3c prefetch+team barrier depth 1 U=2", 2, 1)
    RUNB(10, "E3c prefetch+team barrier depth 2 U=2", 2, 2)
    RUNB(11, "E3c prefetch+team barrier depth 3 U=2", 2, 3)
    RUNB(12, "E3c prefetch+team barrier depth 2 U=1", 1, 2)

#define RUNQ(idx, name, U, NT, PF, depth)                                                                                     \
    if (only < 0 || only == idx) {                                                                                            \
        float best = 1e30f;                                                                                                   \
        for (int rep = 0; rep < 3; rep++) {                                                                                   \
            CK(hipMemset(counters + 64, 0, 4096));                                                                            \
            float ms = time_ms([&] { hipLaunchKernelGGL((k_probe_q<U, NT, PF>), dim3(J * 8), dim3(256), 0, 0, pa, depth, counters + 64); }, 1); \
            best = ms < best ? ms : best;                                                                                     \
        }                                                                                                                     \
        CK(hipGetLastError());                                                                                                \
        report(name, best);                                                                                                   \
    }
    RUNQ(30, "E5 queue CH=256", 1, false, false, 0)
    RUNQ(31, "E5 queue CH=512", 2, false, false, 0)
    RUNQ(32, "E5 queue CH=1024", 4, false, false, 0)
    RUNQ(33, "E5 queue CH=512 nt", 2, true, false, 0)
    RUNQ(34, "E5 queue CH=1024 nt", 4, true, false, 0)
    RUNQ(35, "E5 queue CH=512 nt pf1", 2, true, true, 1)
    RUNQ(36, "E5 queue CH=512 nt pf2", 2, true, true, 2)
    RUNQ(37, "E5 queue CH=1024 nt pf1", 4, true, true, 1)
    RUNQ(38, "E5 queue CH=1024 nt pf2", 4, true, true, 2)
    RUNQ(39, "E5 queue CH=256 nt pf2", 1, true, true, 2)
    RUNQ(40, "E5 queue CH=2048 nt", 8, true, false, 0)
    RUNQ(41, "E5 queue CH=2048 nt pf1", 8, true, true, 1)
    RUNQ(42, "E5 queue CH=1536 nt", 6, true, false, 0)

#define RUNQ2(idx, name, U, PF, depth)                                                                                        \
    if (only < 0 || only == idx) {                                                                                            \
        float best = 1e30f;                                                                                                   \
        for (int rep = 0; rep < 3; rep++) {                                                                                   \
            CK(hipMemset(counters, 0, 8192));                                                                                 \
            float ms = time_ms([&] { hipLaunchKernelGGL((k_probe_q2<U, PF>), dim3(J * 8), dim3(256), 0, 0, pa, depth, counters + 64); }, 1); \
            best = ms < best ? ms : best;                                                                                     \
        }                                                                                                                     \
        CK(hipGetLastError());                                                                                                \
        report(name, best);                                                                                                   \
    }
    RUNQ2(50, "E6 pipelined queue CH=256", 1, false, 0)
    RUNQ2(51, "E6 pipelined queue CH=512", 2, false, 0)
    RUNQ2(52, "E6 pipelined queue CH=1024", 4, false, 0)
    RUNQ2(53, "E6 pipelined queue CH=512 pf1", 2, true, 1)
    RUNQ2(54, "E6 pipelined queue CH=1024 pf1", 4, true, 1)
    RUNQ2(55, "E6 pipelined queue CH=512 pf2", 2, true, 2)

#define RUNQ7(idx, name, U, NT, PF, depth)                                                                                    \
    if (only < 0 || only == idx) {                                                                                            \
        float best = 1e30f;                                                                                                   \
        for (int rep = 0; rep < 3; rep++) {                                                                                   \
            CK(hipMemset(counters, 0, 8192));                                                                                 \
            float ms = time_ms([&] { hipLaunchKernelGGL((k_probe_q7<U, NT, PF>), dim3(J * 8), dim3(256), 0, 0, pa, depth, counters + 64); }, 1); \
            best = ms < best ? ms : best;                                                                                     \
        }                                                                                                                     \
        CK(hipGetLastError());                                                                                                \
        report(name, best);                                                                                                   \
    }
    RUNQ7(60, "E7 queue+deferred spill CH=256", 1, true, false, 0)
    RUNQ7(61, "E7 queue+deferred spill CH=512", 2, true, false, 0)
    RUNQ7(62, "E7 queue+deferred spill CH=1024", 4, true, false, 0)
    RUNQ7(63, "E7 queue+deferred spill CH=512 pf1", 2, true, true, 1)
    RUNQ7(64, "E7 queue+deferred spill CH=1024 pf1", 4, true, true, 1)
    RUNQ7(65, "E7 queue+deferred spill CH=1024 pf2", 4, true, true, 2)

#define RUNQ7T(idx, name, U, PF, depth)                                                                                       \
    if (only < 0 || only == idx) {                                                                                            \
        CK(hipMemset(counters, 0, 8192));                                                                                     \
        float ms = time_ms([&] { hipLaunchKernelGGL((k_probe_q7t<U, true, PF>), dim3(J * 8), dim3(256), 0, 0, pa, depth, counters + 64); }, 1); \
        CK(hipGetLastError());                                                                                                \
        unsigned long long c[32];                                                                                             \
        CK(hipMemcpy(c, counters, 256, hipMemcpyDeviceToHost));                                                               \
        const double n = (double)c[21];                                                                                       \
        printf("%-40s bits=%u J/CU=%d: %.3f ms; per chunk (us): ticket %.2f decode %.2f keys %.2f table %.2f compare %.2f  chunks/block %.1f\n", name, bits, Jc, ms, \
               c[16] / n / 100, c[17] / n / 100, c[18] / n / 100, c[19] / n / 100, c[20] / n / 100, n / (J * 8));            \
    }
    RUNQ7T(70, "E7t CH=512", 2, false, 0)
    RUNQ7T(71, "E7t CH=1024", 4, false, 0)
    RUNQ7T(72, "E7t CH=1024 pf1", 4, true, 1)

#define RUNQ8(idx, name, U, QUAD)                                                                                             \
    if (only < 0 || only == idx) {                                                                                            \
        float best = 1e30f;                                                                                                   \
        for (int rep = 0; rep < 3; rep++) {                                                                                   \
            CK(hipMemset(counters, 0, 8192));                                                                                 \
            float ms = time_ms([&] { hipLaunchKernelGGL((k_probe_q8<U, QUAD>), dim3(J * 8), dim3(256), 0, 0, pa, counters + 64); }, 1); \
            best = ms < best ? ms : best;                                                                                     \
        }                                                                                                                     \
        CK(hipGetLastError());                                                                                                \
        report(name, best);                                                                                                   \
    }
    RUNQ8(80, "E8 pipelined+deferred CH=256", 1, false)
    RUNQ8(81, "E8 pipelined+deferred CH=512", 2, false)
    RUNQ8(82, "E8 pipelined+deferred CH=1024", 4, false)
    RUNQ8(83, "E8 pipelined+deferred quad CH=256", 1, true)
    RUNQ8(84, "E8 pipelined+deferred quad CH=512", 2, true)
    RUNQ8(85, "E8 pipelined+deferred quad CH=1024", 4, true)
    if (only == 20) {  // T1: do concurrent readers of one fresh region share ONE fill?
        const size_t REG = 1 << 20;  // 1 MiB per repetition
        char* buf; CK(hipMalloc(&buf, (size_t)2 << 30)); CK(hipMemset(buf, 1, (size_t)2 << 30));
        for (int readers : {1, 2, 4, 8, 16, 32, 64, 128}) {
            float ms = time_ms([&] { hipLaunchKernelGGL(k_shared_read, dim3(8 * readers), dim3(256), 0, 0, (const ulonglong2*)buf, REG, 256, counters + 7); }, 2);
            printf("T1 readers/XCD0=%3d : %.3f ms for 256 x 1 MiB regions each read by every reader: %.1f GB/s per reader, %.1f GB/s aggregate\n", readers, ms, 256.0 * REG / ms / 1e6,
                   256.0 * REG * readers / ms / 1e6);
        }
    }
    return 0;
}
