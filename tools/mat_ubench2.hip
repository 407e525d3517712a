// tools/mat_ubench2.hip — the product's kernels of the materialising join with the build side in LDS (csrc/tsq_damat.h), driven stand-alone:
// level 1 (k_da_partition_cols) -> level 2 (k_dm_split, build side; probe side with the semi-join filter) -> scan -> k_dm_emit.
// Times every kernel and checks the four output columns of a 1e8 x 1e8 (k, v) x (k, v) join by their sums.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-atomic-optimizer-strategy=None -I tinysql_amd/csrc -I include tools/mat_ubench2.hip -o tools/mat_ubench2
// run  : tools/mat_ubench2 [rows=100000000] [hole=0|4] [build_rows=rows] [outer=0|1]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include "tsq_damat.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

static __global__ void __launch_bounds__(256) k_gen(uint64_t* bk, uint64_t* bv, int64_t nb, uint64_t* pk, uint64_t* pv, int64_t np, uint64_t mulA, uint32_t hole) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (nb > np ? nb : np); i += (int64_t)gridDim.x * 256) {
        if (i < nb) {
            uint64_t k = ((uint64_t)i * mulA) % (uint64_t)nb;       // a permutation of [0, nb)
            if (hole && (k % hole) == 0) k = (uint64_t)nb + k;      // every hole-th key leaves the range (stays unique)
            bk[i] = k;
            bv[i] = k * 7u + 3u;
        }
        if (i < np) {
            uint64_t x = (uint64_t)i + 0x9E3779B97F4A7C15ull;
            x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
            x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
            x ^= x >> 31;
            pk[i] = x % (uint64_t)nb;
            pv[i] = (uint64_t)i;
        }
    }
}
static __global__ void __launch_bounds__(256) k_bitmap(const uint64_t* bk, int64_t nb, DaDomain dm, uint32_t* bits) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nb; i += (int64_t)gridDim.x * 256) {
        const uint32_t u = da_word(dm, bk[i]);
        if (u != TSQ_DA_NONE) atomicOr(&bits[u >> 5], 1u << (u & 31u));
    }
}
static __global__ void __launch_bounds__(256) k_sum4(const uint64_t* a, const uint64_t* b, const uint64_t* c, const uint64_t* d, int64_t n, unsigned long long* out) {
    unsigned long long s[4] = {0, 0, 0, 0};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        s[0] += a[i]; s[1] += b[i]; s[2] += c[i]; s[3] += d[i];
    }
    for (int k = 0; k < 4; k++) {
        const uint64_t w = wave_sum_u64(s[k]);
        if ((threadIdx.x & 63) == 0) atomicAdd(&out[k], (unsigned long long)w);
    }
}
static __global__ void __launch_bounds__(1024) k_scan1(unsigned long long* v, uint32_t n) {  // exclusive scan in place, one workgroup (n includes the total's slot)
    __shared__ unsigned long long s_part[1024];
    const uint32_t tid = threadIdx.x, per = (n + 1023u) / 1024u;
    unsigned long long sum = 0;
    for (uint32_t i = tid * per; i < (tid + 1) * per && i < n; i++) sum += v[i];
    s_part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
        unsigned long long run = 0;
        for (uint32_t i = 0; i < 1024; i++) { const unsigned long long x = s_part[i]; s_part[i] = run; run += x; }
    }
    __syncthreads();
    unsigned long long run = s_part[tid];
    for (uint32_t i = tid * per; i < (tid + 1) * per && i < n; i++) { const unsigned long long x = v[i]; v[i] = run; run += x; }
}

struct L1 {
    DaColStore cs;
    size_t nreg, slots;
    uint32_t cap;
};
static L1 make_l1(uint32_t pbits, uint32_t ebits, int64_t n) {
    L1 s;
    memset(&s, 0, sizeof s);
    const int T = 8192;
    const uint32_t P = 1u << pbits;
    const double tiles = ceil((double)n / T);
    const double lam = std::max((double)n / ((double)P * 8.0), ceil(tiles / 8.0) * std::min<double>(T, (double)n) / P);
    uint32_t cap = (uint32_t)(lam * 1.08 + 8.0 * sqrt(lam) + 2.0 * T / 64.0 + 64.0);
    cap = (cap + 63u) & ~63u;
    s.cap = cap;
    s.nreg = (size_t)P * 8;
    s.slots = s.nreg * cap;
    uint32_t* ctl;
    CK(hipMalloc(&s.cs.st.ent, s.slots * 2 + 256));
    CK(hipMalloc(&s.cs.pay[0], s.slots * 8 + 256));
    CK(hipMalloc(&ctl, (s.nreg + 16) * 4));
    CK(hipMalloc(&s.cs.st.valid_end, s.nreg * 4));
    CK(hipMalloc(&s.cs.st.ovf, (size_t)n * 4 + 64));
    CK(hipMalloc(&s.cs.st.ovf_idx, (size_t)n * 4 + 64));
    s.cs.st.cursor = ctl;
    s.cs.st.ovf_count = ctl + s.nreg;
    s.cs.st.miss_count = ctl + s.nreg + 1;
    s.cs.st.ovf_cap = (uint32_t)n;
    s.cs.st.bits = pbits;
    s.cs.st.ebits = ebits;
    s.cs.st.cap = cap;
    return s;
}
static void run_l1(L1& s, const uint64_t* k, const uint64_t* v, int64_t n, const DaDomain& dm) {
    CK(hipMemsetAsync(s.cs.st.cursor, 0, (s.nreg + 16) * 4, 0));
    CK(hipMemsetAsync(s.cs.st.valid_end, 0xff, s.nreg * 4, 0));
    DaColSrc src;
    memset(&src, 0, sizeof src);
    src.key.data = k;
    src.key.nrows = n;
    src.n_cols = 1;
    src.col[0] = v;
    hipLaunchKernelGGL((k_da_partition_cols<1024, 8, false>), dim3((unsigned)std::min<int64_t>((n + 8191) / 8192, 256)), dim3(1024), 0, 0, src, dm, s.cs);
}
struct L2 {
    DmStore d;
    uint32_t Q;
};
static L2 make_l2(const L1& s, uint32_t sbits) {
    L2 t;
    memset(&t, 0, sizeof t);
    const uint32_t P1 = 1u << s.cs.st.bits;
    t.d.cap1 = 8 * s.cap + 8 * (1u << sbits);
    t.d.sbits = sbits;
    t.d.ebits2 = s.cs.st.ebits - sbits;
    t.Q = P1 << sbits;
    const size_t slots = (size_t)P1 * t.d.cap1;
    CK(hipMalloc(&t.d.ent, slots * 2 + 256));
    CK(hipMalloc(&t.d.pay[0], slots * 8 + 256));
    CK(hipMalloc(&t.d.off, (size_t)t.Q * 4 + 64));
    CK(hipMalloc(&t.d.cnt, ((size_t)t.Q + 1) * 8 + 64));
    return t;
}
template <bool FILTER>
static void run_l2(const L1& s, L2& t, const uint32_t* bitmap, int wg_per_cu) {
    DmSplitArgs a;
    memset(&a, 0, sizeof a);
    a.src = s.cs;
    a.n_cols = 1;
    a.dst = t.d;
    a.bitmap = bitmap;
    const size_t lds = (size_t)512 * 8 * 10 + (FILTER ? ((size_t)1 << s.cs.st.ebits) / 8 : 0) + 16;
    CK(hipFuncSetAttribute((const void*)k_dm_split<512, FILTER>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipMemsetAsync(t.d.cnt + t.Q, 0, 8, 0));
    hipLaunchKernelGGL((k_dm_split<512, FILTER>), dim3(std::min<uint32_t>(1u << s.cs.st.bits, 256u * wg_per_cu)), dim3(512), lds, 0, a);
}

template <typename F>
static float timed(F&& f, int reps = 3) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int i = 0; i < reps; i++) {
        CK(hipEventRecord(e0, 0));
        f();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    CK(hipGetLastError());
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return best;
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    const int64_t np = argc > 1 ? atoll(argv[1]) : 100000000LL;
    const uint32_t hole = argc > 2 ? (uint32_t)atoi(argv[2]) : 0u;
    const int64_t nb = argc > 3 ? atoll(argv[3]) : np;
    const bool outer = argc > 4 && atoi(argv[4]) != 0;
    uint32_t b = 13;
    while ((((uint64_t)nb - 1) >> b) != 0) b++;
    DaDomain dm{0, (uint64_t)nb - 1, b, (b + 1) / 2, (uint32_t)((1ull << b) - 1), 0};
    uint64_t *bk, *bv, *pk, *pv, *o[4];
    uint8_t* onn[2];
    CK(hipMalloc(&bk, (size_t)nb * 8 + 256)); CK(hipMalloc(&bv, (size_t)nb * 8 + 256));
    CK(hipMalloc(&pk, (size_t)np * 8 + 256)); CK(hipMalloc(&pv, (size_t)np * 8 + 256));
    for (int c = 0; c < 4; c++) CK(hipMalloc(&o[c], (size_t)np * 8 + 256));
    for (int c = 0; c < 2; c++) CK(hipMalloc(&onn[c], (size_t)np + 256));
    unsigned long long* dsum;
    CK(hipMalloc(&dsum, 64));
    hipLaunchKernelGGL(k_gen, dim3(2048), dim3(256), 0, 0, bk, bv, nb, pk, pv, np, 61803399ull, hole);
    uint32_t* bitmap;
    CK(hipMalloc(&bitmap, ((size_t)1 << b) / 8 + 64));
    CK(hipMemset(bitmap, 0, ((size_t)1 << b) / 8));
    hipLaunchKernelGGL(k_bitmap, dim3(2048), dim3(256), 0, 0, bk, nb, dm, bitmap);
    CK(hipDeviceSynchronize());
    const uint32_t pbits = std::min(11u, b - 10u), ebits = b - pbits;
    // S: the build rows of a final partition fit the LDS table (<= 8192 rows of 8 bytes: 64 KB, two workgroups per CU)
    uint32_t sbits = 0;
    while (sbits < 3 && (double)nb / (double)((size_t)1 << (pbits + sbits)) * 1.25 + 64 > 8192.0) sbits++;
    printf("probe rows %lld, build rows %lld, key bits %u, hole %u, %s: level 1 2^%u partitions (entries of %u bits), level 2 x %u\n", (long long)np, (long long)nb, b, hole,
           outer ? "LEFT OUTER" : "inner", pbits, ebits, 1u << sbits);
    L1 b1 = make_l1(pbits, ebits, nb), p1 = make_l1(pbits, ebits, np);
    const float ms_b1 = timed([&] { run_l1(b1, bk, bv, nb, dm); });
    const float ms_p1 = timed([&] { run_l1(p1, pk, pv, np, dm); });
    uint32_t ovf[2];
    CK(hipMemcpy(&ovf[0], b1.cs.st.ovf_count, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&ovf[1], p1.cs.st.ovf_count, 4, hipMemcpyDeviceToHost));
    printf("level 1: build %.3f ms (%.2f TB/s), probe %.3f ms (%.2f TB/s), overflow rows %u / %u\n", ms_b1, 26.0 * nb / ms_b1 / 1e9, ms_p1, 26.0 * np / ms_p1 / 1e9, ovf[0], ovf[1]);
    L2 b2 = make_l2(b1, sbits), p2 = make_l2(p1, sbits);
    for (int wg : {1, 2, 3}) {
        const float ms_b2 = timed([&] { run_l2<false>(b1, b2, nullptr, wg); });
        const float ms_p2 = outer ? timed([&] { run_l2<false>(p1, p2, nullptr, wg); }) : timed([&] { run_l2<true>(p1, p2, bitmap, wg); });
        printf("level 2 (%d WG/CU): build %.3f ms (%.2f TB/s at 22 B/row), probe %s %.3f ms\n", wg, ms_b2, 22.0 * nb / ms_b2 / 1e9, outer ? "unfiltered" : "filtered", ms_p2);
    }
    const float ms_scan = timed([&] { hipLaunchKernelGGL(k_scan1, dim3(1), dim3(1024), 0, 0, p2.d.cnt, p2.Q + 1); }, 1);
    unsigned long long total;
    CK(hipMemcpy(&total, p2.d.cnt + p2.Q, 8, hipMemcpyDeviceToHost));
    // the largest final partition of the build side
    std::vector<unsigned long long> bc(b2.Q);
    CK(hipMemcpy(bc.data(), b2.d.cnt, (size_t)b2.Q * 8, hipMemcpyDeviceToHost));
    const unsigned long long bmax = *std::max_element(bc.begin(), bc.end());
    DmEmitArgs ea;
    memset(&ea, 0, sizeof ea);
    ea.bst = b2.d; ea.pst = p2.d; ea.dm = dm; ea.pbits = pbits + sbits;
    ea.tab_rows = (uint32_t)((bmax + 31) & ~31ull);
    ea.out_pkey = o[0]; ea.out_bkey = o[2]; ea.out_bkey_nn = outer ? onn[0] : nullptr;
    ea.n_probe = 1; ea.n_build = 1;
    ea.out_probe[0] = o[1]; ea.out_build[0] = o[3]; ea.out_build_nn[0] = outer ? onn[1] : nullptr;
    const size_t lds = dm_emit_lds(ea.pst.ebits2, ea.tab_rows, 1, false);
    printf("scan %.3f ms; output rows %llu; largest build partition %llu rows; emit LDS %zu bytes\n", ms_scan, total, bmax, lds);
    std::vector<uint64_t> hpk((size_t)np);
    CK(hipMemcpy(hpk.data(), pk, (size_t)np * 8, hipMemcpyDeviceToHost));
    unsigned long long want[4] = {0, 0, 0, 0}, rows = 0;
    for (int64_t i = 0; i < np; i++) {
        const uint64_t k = hpk[(size_t)i];
        const bool miss = hole && (k % hole) == 0;
        if (miss && !outer) continue;
        rows++;
        want[0] += k; want[1] += (uint64_t)i;
        if (!miss) { want[2] += k; want[3] += k * 7u + 3u; }
    }
    auto run_emit = [&](int nt, int wg) {
        for (int c = 0; c < 4; c++) CK(hipMemsetAsync(o[c], 0, (size_t)np * 8, 0));
        if (outer) for (int c = 0; c < 2; c++) CK(hipMemsetAsync(onn[c], 1, (size_t)np, 0));
        float ms;
        const dim3 grid(std::min<uint32_t>(1u << ea.pbits, 256u * wg));
#define LAUNCH(NT, OUT) do { CK(hipFuncSetAttribute((const void*)k_dm_emit<NT, OUT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        ms = timed([&] { hipLaunchKernelGGL((k_dm_emit<NT, OUT>), grid, dim3(NT), lds, 0, ea); }); } while (0)
        if (nt == 512 && outer) LAUNCH(512, true);
        else if (nt == 512) LAUNCH(512, false);
        else if (nt == 256 && outer) LAUNCH(256, true);
        else if (nt == 256) LAUNCH(256, false);
        else if (outer) LAUNCH(1024, true);
        else LAUNCH(1024, false);
        CK(hipMemset(dsum, 0, 64));
        hipLaunchKernelGGL(k_sum4, dim3(2048), dim3(256), 0, 0, o[0], o[1], o[2], o[3], (int64_t)total, dsum);
        unsigned long long got[4];
        CK(hipMemcpy(got, dsum, 32, hipMemcpyDeviceToHost));
        const bool ok = rows == total && !memcmp(got, want, 32);
        printf("  emit<%d> x %d WG/CU: %.3f ms  %.2f TB/s (10 B/build row + 10 + 32 B/row)  %s\n", nt, wg, ms, (10.0 * nb + 42.0 * (double)total) / ms / 1e9, ok ? "ok" : "MISMATCH");
    };
    run_emit(512, 2);
    run_emit(512, 1);
    run_emit(256, 4);
    run_emit(1024, 1);
    return 0;
}
