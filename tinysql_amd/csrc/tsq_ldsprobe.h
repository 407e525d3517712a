// tsq_ldsprobe.h — probe of a partitioned key store against LDS COPIES of the join table's slices
// (device code; included by tsq_join.hip).
//
// Why: a probe that reads the table through the L2 — even partition by partition, with the slice resident in
// one XCD's L2 (k_radix_probe_count, tsq_radix.h) — pays one 64-byte L2 request per probe key: 1e8 requests at
// the 210-260 G/s an L2-resident table delivers are 0.4 ms before anything else happens, and the misses that
// bring the slice in arrive as random 64-byte lines at 55 G/s instead of as a stream.  Here the table is read
// from HBM exactly once, with 16-byte coalesced loads, into LDS, and every compare runs against LDS:
//
//   ticket (XCD vx)  ->  (partition p = 8 pi + vx, sub-slice s)      ordered queue per XCD, as in tsq_radix.h
//   image            =   nf consecutive table slices (tsq_jointable.h: self-contained linear-probing tables of
//                        bs buckets) = nf * bs * 64 bytes <= 128 KB, copied verbatim
//   keys             =   ALL table words of partition p (8 XCC regions of the HASHED store); a word whose slice
//                        lies outside the image belongs to one of the S - 1 sibling tickets and is dropped
//
// The S workgroups that hold the sub-slices of one partition draw consecutive tickets of the same XCD, so the
// partition's keys come from HBM once and from that XCD's L2 S - 1 times.  The kernel is bound by vector-instruction
// issue, not by memory (SQ counters: VALU busy 67 %, LDS 32 %), so the work per key is organised around that:
//   * words that pass the slice test are compacted through a per-wave LDS ring and probed 64 at a time;
//   * a probe round looks at ONE bucket per lane; a lane whose bucket is full (the chain may go on) puts its word back
//     into the ring with a hop count instead of walking on while 63 lanes wait — rounds = bucket visits / 64, not the
//     longest chain of every 64 keys (~10 buckets at load factor 0.75: 2.2 ms per 1e8 keys, 2/3 of it in that loop);
//   * matches are counted per lane on the vector unit (the CU's ONE scalar unit, shared by its 16 waves, was as busy as
//     the four SIMDs when the eight compare masks were popcounted and added there);
//   * the four 16-byte pieces of a bucket are read in an order rotated by the lane number: with the same piece order in
//     every lane the 16 lanes of an LDS access group would only ever hit 4 of the 16 four-bank columns.
// The image of the NEXT ticket is in flight (16-byte registers) while the keys of the current one are probed.
// Measured and dropped (profiles/r02_lds_probe_experiments.txt): drawing a number per 128-word CHUNK from an LDS counter (every
// draw is an LDS round trip on the load-issue path: 0.80 vs 0.73 ms with the static assignment — the shipped kernel draws per
// 512-word super-chunk, two draws ahead); software-pipelining the rounds (ring fetch of the next round behind the bucket reads
// of this one: 0.77 ms) — the kernel is not waiting for LDS latency, its SIMDs (50 %), LDS (42 %) and scalar unit (24 %) are
// all busy.
//
// Replaces (reference): join2Chunk + hashRowContainer.GetMatchedRows (executor/join.go:343-360,
// hash_table.go:110-134) for the COUNT(*) shape; same joined-row count whichever route a key takes (a probe row
// meets exactly the build rows with an equal key word, util/codec/codec.go:363-382).
// Algorithmic bytes: 8 B key + one 16 B slot per probe row; real traffic per probe row:
// 8 B (partitioned word) + table bytes / probe rows.
#ifndef TSQ_LDSPROBE_H
#define TSQ_LDSPROBE_H

#include "tsq_radix.h"

#define TSQ_LDS_IMAGE_MAX (128 * 1024)
#define TSQ_LDS_RING 128  // entries per wave ring: <= 63 left over + <= 64 new per step / continuations per round
#define TSQ_LDS_RING_BYTES (TSQ_LDS_RING * 10)  // 8-byte word + 2-byte hop count
#define TSQ_LDS_RING_BYTES_EMIT (TSQ_LDS_RING * 14)  // ... + 4-byte position in the partitioned store
#define TSQ_LDS_MAXPAY 2

typedef unsigned long long tsq_u64x2 __attribute__((ext_vector_type(2)));
// 16-byte streaming load (read once: do not keep the line in L1 / prefer eviction in L2)
__device__ __forceinline__ ulonglong2 nt_load16(const void* p) {
    const tsq_u64x2 v = __builtin_nontemporal_load(reinterpret_cast<const tsq_u64x2*>(p));
    return make_ulonglong2(v.x, v.y);
}

// Software-pipelined streaming loads.  The compiler's s_waitcnt insertion gives up on a rotating set of in-flight loads
// inside a loop with inner loops (it waits for vmcnt(0) at the loop header: every load then pays its full latency), so
// the chunk loads are issued and waited for by hand: results return in order, `younger` loads may stay in flight.
// The wait takes the destination as an in/out operand so that no use of it can be scheduled above the wait.  Waits the
// compiler inserts for ITS loads stay correct: it can only under-count what is in flight, i.e. wait for longer.
__device__ __forceinline__ void async_nt_load16(tsq_u64x2& dst, const void* p) {
#ifdef TSQ_LDS_KEYS_NT
    asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(dst) : "v"(p) : "memory");
#else
    // default cache policy: the S workgroups that share a partition's keys find them in the XCD's L2 (with the streaming
    // hint the lines were not kept: TCC hit rate 33 %, 2.9 GB fetched per 1e8 keys instead of the 1.9 GB of keys + table)
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
#endif
}
template <int YOUNGER>
__device__ __forceinline__ void async_wait(tsq_u64x2& dst) {
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(dst) : "n"(YOUNGER) : "memory");
}

struct LdsProbeArgs {
    RadixStore st;   // HASHED store, P = 2^st.bits >= 8 partitions
    JoinTable t;     // t.tb >= st.bits
    uint32_t S;      // sub-slices (tickets) per partition
    uint32_t nf;     // table slices per image (the last sub-slice of a partition may hold fewer)
    uint32_t unique; // the build saw no two equal table words: a probe word that has found its slot is done
    unsigned long long* counters;  // [0] += joined rows
    unsigned long long* prof;      // PROF kernels: [0..7] += shader cycles per phase, summed over waves (see k_lds_probe_count)
    // MODE 1 (sizing pass of the materialising join): joined rows per ticket -> tk_cnt[xcd * ntk + ticket]; MODE 2 (emit pass): the
    // exclusive scan of those counts = first output row of every ticket
    unsigned long long* tk_cnt;
    // MODE 2: joined rows are written column by column.  A probe / build output column is the key (src -1: recovered from the
    // table word, mix64 is a bijection) or payload column src (probe: st.pay[src] travels with the key through the partition;
    // build: bpay[src][slot] = the payload cell of the build row in table slot `slot`, see k_table_payload)
    uint64_t* out_probe[1 + TSQ_LDS_MAXPAY];
    uint64_t* out_build[1 + TSQ_LDS_MAXPAY];
    int32_t probe_src[1 + TSQ_LDS_MAXPAY], build_src[1 + TSQ_LDS_MAXPAY];
    int32_t n_out_probe, n_out_build;
    const uint64_t* bpay[TSQ_LDS_MAXPAY];
    const uint64_t* bcol[TSQ_LDS_MAXPAY];  // the same payload columns by build ROW (the side list of the sentinel word holds row ids)
    unsigned long long* ovf_cursor;        // k_lds_emit_ovf: next output row of the overflow list's joined rows
};

// Vector-memory results return in order and s_waitcnt counts them, so every wave issues a FIXED sequence of loads: a load
// that has nothing to fetch (past the end of a region, past the last chunk) is still issued, on a clamped address, and its
// result is ignored.  With loads under a branch the compiler can only wait for "everything" (vmcnt(0)), which serialises
// the prefetch distance away: 2.2 ms per 1e8 keys instead of 0.4.
// PROF: per-wave shader-cycle sums (s_memtime) -> a.prof: [0] image store + barrier A, [1] waits for chunk loads, [2] slice
// test + ring, [3] bucket compares, [4] drain + barrier B, [5] whole ticket loop, [6] probe calls, [7] bucket reads.
// MODE 0: COUNT(*) (one total); 1: joined rows per ticket (sizing pass of the materialising join); 2: emit the joined rows.
template <int NT, bool PROF = false, int MODE = 0>
__global__ void __launch_bounds__(NT) k_lds_probe_count(LdsProbeArgs a) {
    constexpr int NW = NT / 64;
    unsigned long long pf[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto now = [&]() -> unsigned long long { return PROF ? (unsigned long long)__builtin_readcyclecounter() : 0ull; };
    const unsigned long long t_begin = now();
    constexpr int IMG_R = TSQ_LDS_IMAGE_MAX / 16 / NT;  // 16-byte units of an image per thread
    // ... of which PRE_R are prefetched into registers during the previous ticket.  The emit pass has no registers left for the
    // whole image: with all 8 prefetched it spilled one (i.e. waited for every load in flight right after issuing them).
    constexpr int PRE_R = MODE == 2 ? IMG_R - 2 : IMG_R;
    constexpr int D = 4;                                // chunk loads in flight per wave
    static_assert(NW % 8 == 0, "NW / 8 waves per XCC region");
    extern __shared__ __align__(16) unsigned char s_dyn[];
    __shared__ uint32_t s_len[8];
    __shared__ uint32_t s_tk[2];
    __shared__ uint32_t s_next;  // next super-chunk of the current ticket
    __shared__ unsigned long long s_total;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t vx = blockIdx.x & 7u;
    const uint32_t P = 1u << a.st.bits, NP = P >> 3;
    const uint32_t tb = a.t.tb, bs = a.t.bs, sh = 64u - tb;
    const uint32_t fpp = 1u << (tb - a.st.bits);
    const uint32_t ntk = NP * a.S;
    const uint32_t img_words = a.nf * bs * TSQ_BUCKET;
    uint64_t* s_img = reinterpret_cast<uint64_t*>(s_dyn);
    uint64_t* s_ring = s_img + img_words + (size_t)wave * TSQ_LDS_RING;                                      // words waiting for a probe round
    uint16_t* s_hop = reinterpret_cast<uint16_t*>(s_img + img_words + (size_t)NW * TSQ_LDS_RING) + (size_t)wave * TSQ_LDS_RING;  // ... and how far along their chain
    uint32_t* s_gix = reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(s_img + img_words) + (size_t)NW * TSQ_LDS_RING * 10) + (size_t)wave * TSQ_LDS_RING;  // MODE 2: where the word sits in the store
    __shared__ unsigned long long s_tkcnt;  // MODE 1: joined rows of the current ticket; MODE 2: first output row of the ticket
    __shared__ uint32_t s_out;              // MODE 2: output rows handed out inside the ticket
    auto take = [&]() -> uint32_t {
        return (uint32_t)__hip_atomic_fetch_add(&a.st.queue[vx * TSQ_RADIX_QSTRIDE], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    uint32_t pend = 0;
    if (tid == 0) {
        s_total = 0;
        s_tk[0] = take();
        s_tk[1] = take();
        pend = take();
    }
    __syncthreads();
    // ticket -> first table slice, slices in the image
    auto first_slice = [&](uint32_t tk, uint32_t& nfi) -> uint32_t {
        const uint32_t pi = tk / a.S, sub = tk - pi * a.S;
        const uint32_t rest = fpp - sub * a.nf;
        nfi = rest < a.nf ? rest : a.nf;
        return (pi * 8u + vx) * fpp + sub * a.nf;
    };
    ulonglong2 pre[PRE_R];
    auto image_load = [&](uint32_t tk) {  // a ticket past the end loads the first image again (ignored)
        uint32_t nfi;
        const uint32_t f0 = first_slice(tk < ntk ? tk : 0u, nfi);
        const uint32_t n16 = nfi * bs * (TSQ_BUCKET / 2);
        const ulonglong2* src = reinterpret_cast<const ulonglong2*>(a.t.keys + (size_t)f0 * bs * TSQ_BUCKET);
#pragma unroll
        for (int j = 0; j < PRE_R; j++) {
            const uint32_t i = (uint32_t)j * NT + tid;
            pre[j] = nt_load16(src + (i < n16 ? i : 0u));
        }
    };
    uint32_t tk = (uint32_t)__builtin_amdgcn_readfirstlane(s_tk[0]);
    image_load(tk);
    uint32_t cnt = 0;  // joined rows seen by this lane (< 2^32: a lane probes a few thousand words, a word has < 2^14 matches)
    for (uint32_t it = 0; tk < ntk; it++) {
        const unsigned long long t0 = now();
        uint32_t nfi;
        const uint32_t f0 = first_slice(tk, nfi);
        const uint32_t p = (tk / a.S) * 8u + vx;
        {
            const uint32_t n16 = nfi * bs * (TSQ_BUCKET / 2);
            ulonglong2* dst = reinterpret_cast<ulonglong2*>(s_img);
            ulonglong2 late[IMG_R - PRE_R + 1];
            const ulonglong2* src = reinterpret_cast<const ulonglong2*>(a.t.keys + (size_t)f0 * bs * TSQ_BUCKET);
#pragma unroll
            for (int j = PRE_R; j < IMG_R; j++) {
                const uint32_t i = (uint32_t)j * NT + tid;
                late[j - PRE_R] = nt_load16(src + (i < n16 ? i : 0u));
            }
#pragma unroll
            for (int j = 0; j < PRE_R; j++) {
                const uint32_t i = (uint32_t)j * NT + tid;
                if (i < n16) dst[i] = pre[j];
            }
#pragma unroll
            for (int j = PRE_R; j < IMG_R; j++) {
                const uint32_t i = (uint32_t)j * NT + tid;
                if (i < n16) dst[i] = late[j - PRE_R];
            }
        }
        if (tid < 8) s_len[tid] = radix_region_len(a.st, P, p, tid);
        if (tid == 9) s_next = 0;
        if (MODE == 1 && tid == 8) s_tkcnt = 0;
        if (MODE == 2 && tid == 8) {
            s_tkcnt = a.tk_cnt[(size_t)vx * ntk + tk];
            s_out = 0;
        }
        const uint32_t cnt0 = cnt;
        __syncthreads();  // A: image and region lengths are in LDS
        if (PROF) pf[0] += now() - t0;
        const uint32_t tkn = (uint32_t)__builtin_amdgcn_readfirstlane(s_tk[(it + 1) & 1u]);
        // The words of the partition in SUPER-CHUNKS of 512 (4 chunks of 128 = the D loads a wave keeps in flight): super-chunk i =
        // region i / spr, words [(i % spr) * 512, +512) of it, spr = super-chunks of the longest region (a super-chunk past the end of
        // a shorter region is empty).  Waves DRAW super-chunk numbers from an LDS counter, so that all waves of the workgroup run dry
        // together whatever their luck with candidates and chains (with a static chunk -> wave assignment the waves spent 27 % of
        // their time waiting for the slowest one at the end of a ticket).  The draw for super-chunk n + 2 is issued when n starts
        // and picked up when n + 1 starts: an LDS round trip on the load-issue path of every CHUNK cost more than the balance bought.
        uint32_t maxlen = 0;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint32_t l = (uint32_t)__builtin_amdgcn_readfirstlane(s_len[r]);
            maxlen = l > maxlen ? l : maxlen;
        }
        const uint32_t spr = (maxlen + 511u) / 512u, nsc = spr * 8u;
        const uint64_t* pbase = a.st.keys + (size_t)p * 8u * a.st.cap;
        const uint32_t pgbase = (uint32_t)((size_t)p * 8u * a.st.cap);  // the store has < 2^32 slots (host check)
        struct Sc {  // where a super-chunk lives (wave uniform)
            uint32_t rbase, rlen, off0;
        };
        auto locate = [&](uint32_t id) -> Sc {
            uint32_t r = 0;
#pragma unroll
            for (uint32_t q = 1; q < 8; q++) r += id >= q * spr ? 1u : 0u;  // (id >= nsc lands in region 7 with off0 >= its length: empty)
            Sc x;
            x.rlen = (uint32_t)__builtin_amdgcn_readfirstlane(s_len[r]);
            x.rbase = r * a.st.cap;
            x.off0 = (id - r * spr) * 512u;
            return x;
        };
        auto chunk_load = [&](const Sc& sc, uint32_t d, tsq_u64x2& v, uint32_t& nv, uint32_t& gi) {
            const uint32_t off = sc.off0 + d * 128u + lane * 2u;
            nv = off + 1u < sc.rlen ? 2u : (off < sc.rlen ? 1u : 0u);  // nothing to fetch: the region's first line (ignored)
            gi = pgbase + sc.rbase + off;
            async_nt_load16(v, pbase + (sc.rbase + (off < sc.rlen ? off : 0u)));
        };
        auto draw = [&]() -> uint32_t {  // issue only: the answer is read later
            uint32_t g = 0;
            if (lane == 0) g = atomicAdd(&s_next, 1u);
            return g;
        };
        uint32_t cur = (uint32_t)__builtin_amdgcn_readfirstlane(draw());
        uint32_t nxt = (uint32_t)__builtin_amdgcn_readfirstlane(draw());
        uint32_t gpend = draw();
        Sc sc_cur = locate(cur), sc_nxt = locate(nxt);
        tsq_u64x2 v[D];
        uint32_t nv[D], gi[D];
#pragma unroll
        for (int d = 0; d < D; d++) chunk_load(sc_cur, (uint32_t)d, v[d], nv[d], gi[d]);
        image_load(tkn);          // in flight while this ticket's keys are probed
        uint32_t qh = 0, qt = 0;  // ring head / tail (wave uniform)
        // compaction append of (word, hop) for the lanes in `on`
        auto push = [&](bool on, uint64_t w, uint32_t hop, uint32_t gx) {
            const uint64_t m = __ballot(on);
            if (m) {
                if (on) {
                    const uint32_t pos = (qt + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))) & (TSQ_LDS_RING - 1);
                    s_ring[pos] = w;
                    s_hop[pos] = (uint16_t)hop;
                    if (MODE == 2) s_gix[pos] = gx;
                }
                qt += (uint32_t)__popcll(m);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            }
        };
        // one bucket per lane for the n (<= 64) oldest ring entries
        auto probe_round = [&](uint32_t n) {
            const unsigned long long tp = now();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const uint32_t pos = (qh + lane) & (TSQ_LDS_RING - 1);
            qh += n;
            bool go_on = false;
            uint64_t x = 0, gbkt = 0;
            uint32_t hop = 0, gx = 0, mm = 0;  // mm: which of the eight slots of this lane's bucket hold its word (MODE 2)
            uint64_t ppay[TSQ_LDS_MAXPAY] = {0, 0};
            if (lane < n) {  // (the compares below then leave the bits of idle lanes clear: no mask arithmetic)
                x = s_ring[pos];
                hop = s_hop[pos];
                if (MODE == 2) {
                    gx = s_gix[pos];
#pragma unroll
                    for (int vv = 0; vv < TSQ_LDS_MAXPAY; vv++)
                        if (a.st.pay[vv]) ppay[vv] = a.st.pay[vv][gx];  // the probe row's payload: in flight while the bucket is compared
                }
                if (x == TSQ_EMPTY_KEY) {
                    cnt += a.t.sent_count;  // the sentinel word matches the side list only
                    if (MODE == 2 && a.t.sent_count) {
                        const uint64_t key = tsq_unmix64(x);
                        const uint64_t row0 = s_tkcnt + atomicAdd(&s_out, a.t.sent_count);
                        for (uint32_t i = 0; i < a.t.sent_count; i++) {
                            const uint32_t brow = a.t.sent_rows[i];
                            for (int oc = 0; oc < a.n_out_probe; oc++) a.out_probe[oc][row0 + i] = a.probe_src[oc] < 0 ? key : ppay[a.probe_src[oc]];
                            for (int oc = 0; oc < a.n_out_build; oc++) a.out_build[oc][row0 + i] = a.build_src[oc] < 0 ? key : a.bcol[a.build_src[oc]][brow];
                        }
                    }
                } else {
                    // local bucket = mulhi(the 32 bits below the slice bits, bs): one funnel shift of (hi, lo) instead of 64-bit shifts
                    const uint32_t xh = (uint32_t)(x >> 32), xl = (uint32_t)x;
                    uint32_t lb = __umulhi(__builtin_amdgcn_alignbit(xh, xl, 32u - tb), bs) + hop;  // 1 <= tb <= 31 on this route
                    lb = lb >= bs ? lb - bs : lb;
                    const uint32_t lbkt = __umul24((xh >> (sh - 32u)) - f0, bs) + lb;  // bucket inside the image (a few slices of < 2^10 buckets)
                    const uint64_t* bk = s_img + (size_t)lbkt * TSQ_BUCKET;
                    const ulonglong2* b = reinterpret_cast<const ulonglong2*>(bk);
                    const ulonglong2 q0 = b[lane & 3u], q1 = b[(lane + 1u) & 3u], q2 = b[(lane + 2u) & 3u], q3 = b[(lane + 3u) & 3u];
                    const uint64_t last = bk[TSQ_BUCKET - 1];  // slots are claimed front to back: the bucket is full iff its LAST slot is taken
                    const uint32_t c = (uint32_t)(q0.x == x) + (uint32_t)(q0.y == x) + (uint32_t)(q1.x == x) + (uint32_t)(q1.y == x) +
                                       (uint32_t)(q2.x == x) + (uint32_t)(q2.y == x) + (uint32_t)(q3.x == x) + (uint32_t)(q3.y == x);
                    cnt += c;
                    go_on = last != TSQ_EMPTY_KEY && !(a.unique && c) && hop + 1u < bs;
                    if (MODE == 2 && c) {
                        // one output row per matching slot.  The lanes of the round share ONE block of output rows (LDS cursor +
                        // prefix sum of the per-lane match counts); a lane writes the cells of its rows column by column.  The loop
                        // runs over a lane's k-th match, so a unique table takes ONE pass: one payload load in flight per lane, the
                        // rows of neighbouring lanes next to each other.  (A loop over the eight slots instead put eight dependent
                        // global loads one after the other into every round: 5.4 ms per 1e8 rows.)
                        mm = (uint32_t)(q0.x == x) | ((uint32_t)(q0.y == x) << 1) | ((uint32_t)(q1.x == x) << 2) | ((uint32_t)(q1.y == x) << 3) |
                             ((uint32_t)(q2.x == x) << 4) | ((uint32_t)(q2.y == x) << 5) | ((uint32_t)(q3.x == x) << 6) | ((uint32_t)(q3.y == x) << 7);
                        gbkt = (uint64_t)f0 * bs + lbkt;  // bucket in the whole table
                    }
                }
            }
            if (MODE == 2) {
                const uint64_t any = __ballot(mm != 0);
                if (any) {  // wave uniform
                    const uint32_t c = (uint32_t)__popc(mm);
                    uint32_t tot;
                    const uint32_t pre = wave_excl_scan_u32(c, &tot);
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(&s_out, tot);
                    base = (uint32_t)__builtin_amdgcn_readfirstlane(base);
                    const uint64_t key = tsq_unmix64(x);
                    uint64_t row = s_tkcnt + base + pre;
                    while (mm) {
                        const uint32_t i = (uint32_t)__builtin_ctz(mm);
                        mm &= mm - 1u;
                        const uint32_t slot = 2u * ((lane + (i >> 1)) & 3u) + (i & 1u);  // bit i = piece (lane + i / 2) & 3, word i & 1
                        for (int oc = 0; oc < a.n_out_probe; oc++) a.out_probe[oc][row] = a.probe_src[oc] < 0 ? key : ppay[a.probe_src[oc]];
                        for (int oc = 0; oc < a.n_out_build; oc++)
                            a.out_build[oc][row] = a.build_src[oc] < 0 ? key : a.bpay[a.build_src[oc]][gbkt * TSQ_BUCKET + slot];
                        row++;
                    }
                }
            }
            push(go_on, x, hop + 1u, gx);
            if (PROF) {
                pf[3] += now() - tp;
                pf[6]++;
            }
        };
        auto step = [&](uint64_t w, bool valid, uint32_t gx) {
            // slice of w = its top tb <= 32 bits: a 32-bit shift of the high dword (sh >= 32)
            push(valid && (((uint32_t)(w >> 32) >> (sh - 32u)) - f0) < nfi, w, 0u, gx);
            while (qt - qh >= 64u) probe_round(64u);
        };
        while (cur < nsc) {  // the numbers a wave draws grow: cur is its oldest
#pragma unroll
            for (int d = 0; d < D; d++) {
                const unsigned long long tw = now();
                async_wait<D - 1>(v[d]);
                const unsigned long long ts = now();
                step(v[d].x, nv[d] >= 1, gi[d]);
                step(v[d].y, nv[d] >= 2, gi[d] + 1u);
                if (PROF) {
                    pf[1] += ts - tw;
                    pf[2] += now() - ts;
                }
                chunk_load(sc_nxt, (uint32_t)d, v[d], nv[d], gi[d]);  // reload only after use: no register copy, D - 1 chunks ahead
            }
            cur = nxt;
            sc_cur = sc_nxt;
            nxt = (uint32_t)__builtin_amdgcn_readfirstlane(gpend);
            gpend = draw();
            sc_nxt = locate(nxt);
        }
        const unsigned long long td = now();
        // D clamped loads of the last round are still in flight: their registers must stay allocated until they have landed
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) : : "memory");
        static_assert(D == 4, "drain lists v[0..3]");
        while (qt != qh) probe_round(qt - qh < 64u ? qt - qh : 64u);  // the last, partial rounds (and their continuations)
        if (MODE == 1) {  // joined rows of this ticket
            const uint64_t wsum = wave_sum_u64((uint64_t)(cnt - cnt0));
            if (lane == 0 && wsum) atomicAdd(&s_tkcnt, (unsigned long long)wsum);
        }
        __syncthreads();  // B: nobody reads the image or s_tk[(it + 1) & 1] any more
        if (MODE == 1 && tid == 8) a.tk_cnt[(size_t)vx * ntk + tk] = s_tkcnt;  // (thread 8 also clears it for the next ticket)
        if (PROF) pf[4] += now() - td;
        if (tid == 0) {   // ticket of iteration it + 2 (its atomic has been in flight since the previous iteration)
            s_tk[it & 1u] = pend;
            pend = take();
        }
        tk = tkn;
    }
    if (PROF && lane == 0) {
        pf[5] = now() - t_begin;
        pf[2] -= pf[3];
#pragma unroll
        for (int i = 0; i < 8; i++) atomicAdd(&a.prof[i], pf[i]);
    }
    const uint64_t ws = wave_sum_u64(cnt);
    if (lane == 0 && ws) atomicAdd(&s_total, (unsigned long long)ws);
    __syncthreads();
    if (MODE == 0 && tid == 0 && s_total) atomicAdd(&a.counters[0], s_total);  // one device atomic per workgroup
}

// the joined rows of the store's overflow list (skewed partitions): plain grid-stride probe of the table in HBM, output
// rows from one device cursor (the list is short or empty)
static __global__ void __launch_bounds__(256) k_lds_emit_ovf(LdsProbeArgs a) {
    uint32_t n = *a.st.ovf_count;
    n = n < a.st.ovf_cap ? n : a.st.ovf_cap;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const uint64_t w = a.st.ovf_keys[i], key = tsq_unmix64(w);
        uint64_t ppay[TSQ_LDS_MAXPAY] = {0, 0};
#pragma unroll
        for (int vv = 0; vv < TSQ_LDS_MAXPAY; vv++)
            if (a.st.ovf_pay[vv]) ppay[vv] = a.st.ovf_pay[vv][i];
        auto emit = [&](uint64_t bval_idx, bool by_row) {
            const unsigned long long row = atomicAdd(a.ovf_cursor, 1ull);
            for (int oc = 0; oc < a.n_out_probe; oc++) a.out_probe[oc][row] = a.probe_src[oc] < 0 ? key : ppay[a.probe_src[oc]];
            for (int oc = 0; oc < a.n_out_build; oc++)
                a.out_build[oc][row] = a.build_src[oc] < 0 ? key : (by_row ? a.bcol[a.build_src[oc]][bval_idx] : a.bpay[a.build_src[oc]][bval_idx]);
        };
        if (w == TSQ_EMPTY_KEY) {
            for (uint32_t q = 0; q < a.t.sent_count; q++) emit(a.t.sent_rows[q], true);
        } else {
            for_each_slot_w(a.t, w, [&](uint64_t slot) { emit(slot, false); });
        }
    }
}

// bpay[slot] = payload cell of the build row that sits in table slot `slot` (0 for an empty slot): the build side's payload
// column in TABLE order, so that the emit pass reads it next to the slice it is probing instead of gathering it by row id
// from all over HBM (one random 8-byte read per joined row: 42 G/s, 2.4 ms per 1e8 rows and column)
static __global__ void __launch_bounds__(256) k_table_payload(const uint64_t* keys, const uint32_t* vals, const uint64_t* col, uint64_t* out, uint64_t nslots) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < nslots; i += (uint64_t)gridDim.x * 256)
        out[i] = keys[i] != TSQ_EMPTY_KEY ? col[vals[i]] : 0ull;
}

#endif
