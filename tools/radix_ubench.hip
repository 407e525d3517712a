// radix_ubench.hip — sweeps the radix-partitioned probe (tinysql_amd/csrc/tsq_radix.h) against the
// unpartitioned probe on the bench workload (1e8 x 1e8 int64 keys, J-uniq-shuffled) to pick
// partition bits / tile size / region policy / probe grid.  Results feed DESIGN.md; the product
// (libtsq) uses the same kernels through tsq_join.hip.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/radix_ubench.hip -o tools/radix_ubench
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
__global__ void __launch_bounds__(256) k_probe_flat(const uint64_t* pk, int64_t np, JoinTable t, unsigned long long* counters) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    uint64_t cnt = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < np; i += stride) for_each_slot(t, pk[i], [&](uint64_t) { cnt++; });
    cnt = wave_sum_u64(cnt);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&counters[0], (unsigned long long)cnt);
}
// returning atomics on a small cursor array: the reservation pattern of the shared-region partition
__global__ void __launch_bounds__(256) k_atomic_small(uint32_t* cur, uint32_t mask, int64_t n, uint32_t R, int use_xcc, unsigned long long* out) {
    const uint32_t r = use_xcc ? tsq_xcc_id() : 0;
    uint32_t acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        acc += atomicAdd(&cur[((uint32_t)tsq_mix64((uint64_t)i) & mask) * R + r], 7u);
    if (acc == 0x1234567) atomicAdd(out, 1ull);
}
__global__ void k_sum_cursor(const uint32_t* cur, const uint32_t* ve, uint32_t n, uint32_t cap, unsigned long long* out) {
    unsigned long long s = 0, mx = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        uint32_t c = cur[i];
        c = c < ve[i] ? c : ve[i];
        c = c < cap ? c : cap;
        s += c;
        mx = c > mx ? c : mx;
    }
    atomicAdd(&out[0], s);
    atomicMax(&out[1], mx);
}

template <class F>
static float time_ms(F&& launch, int reps = 3) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e30f;
    for (int r = 0; r < reps; r++) {
        hipEventRecord(a);
        launch();
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    hipEventDestroy(a); hipEventDestroy(b);
    return best;
}

int main(int argc, char** argv) {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int64_t NB = argc > 1 ? atoll(argv[1]) : 100000000, NP = argc > 2 ? atoll(argv[2]) : 100000000;
    printf("device %s CUs=%d LDS/block=%zu KB  build=%lld probe=%lld\n", prop.gcnArchName, prop.multiProcessorCount, prop.sharedMemPerBlock >> 10, (long long)NB, (long long)NP);
    const int CUS = prop.multiProcessorCount;
    uint64_t *bk, *pk;
    CK(hipMalloc(&bk, NB * 8)); CK(hipMalloc(&pk, NP * 8));
    hipLaunchKernelGGL(k_gen, dim3(CUS * 8), dim3(256), 0, 0, bk, pk, NB, NP);
    JoinTable t{};
    t.nbuckets = (uint64_t)((NB + 3) / 4);
    CK(hipMalloc(&t.keys, t.nbuckets * 64));
    CK(hipMemset(t.keys, 0x80, t.nbuckets * 64));
    unsigned long long* counters;
    CK(hipMalloc(&counters, 64)); CK(hipMemset(counters, 0, 64));
    {
        float ms = time_ms([&] { hipLaunchKernelGGL(k_build, dim3(CUS * 8), dim3(256), 0, 0, bk, NB, t); }, 1);
        printf("build (CAS insert) %.3f ms  %.2f Grows/s  table %.0f MB\n", ms, NB / ms / 1e6, t.nbuckets * 64 / 1e6);
    }
    auto get_count = [&]() { unsigned long long c[8]; CK(hipMemcpy(c, counters, 64, hipMemcpyDeviceToHost)); CK(hipMemset(counters, 0, 64)); return c[0]; };
    {
        float ms = time_ms([&] { hipLaunchKernelGGL(k_probe_flat, dim3(CUS * 8), dim3(256), 0, 0, pk, NP, t, counters); }, 3);
        unsigned long long c = get_count();
        printf("flat probe %.3f ms  %.2f Grows/s  count=%llu (%s)\n", ms, NP / ms / 1e6, c, c == 3ull * NP ? "ok" : "BAD");
    }
    // ---- returning atomics on a small array (12.5M ops = one per 8 keys)
    {
        uint32_t* cur; CK(hipMalloc(&cur, 2048 * 8 * 4)); CK(hipMemset(cur, 0, 2048 * 8 * 4));
        const int64_t nops = 12500000;
        for (int use_xcc = 0; use_xcc < 2; use_xcc++)
            for (uint32_t P : {256u, 2048u}) {
                float ms = time_ms([&] { hipLaunchKernelGGL(k_atomic_small, dim3(CUS * 8), dim3(256), 0, 0, cur, P - 1, nops, 8u, use_xcc, counters + 4); });
                printf("atomic rtn small array P=%u per-xcc=%d: %.3f ms  %.2f Gops/s\n", P, use_xcc, ms, nops / ms / 1e6);
            }
        hipFree(cur);
    }
    // ---- partitioned store (sized for the largest variant)
    const uint32_t OVF_CAP = (uint32_t)NP;
    RadixStore st{};
    const size_t store_keys = (size_t)(NP * 1.8) + (size_t)TSQ_RADIX_MAX_P * 1024 * 64;
    CK(hipMalloc(&st.keys, store_keys * 8));
    CK(hipMalloc(&st.cursor, (size_t)TSQ_RADIX_MAX_P * 1024 * 4));
    CK(hipMalloc(&st.valid_end, (size_t)TSQ_RADIX_MAX_P * 1024 * 4));
    CK(hipMalloc(&st.ovf_keys, (size_t)OVF_CAP * 8));
    CK(hipMalloc(&st.ovf_count, 4));
    st.ovf_cap = OVF_CAP;
    unsigned long long* d_sum; CK(hipMalloc(&d_sum, 16));
    RadixSrc src{};
    src.data = pk; src.type = TSQ_I64; src.nrows = NP;

    CK(hipMalloc(&st.queue, 8 * TSQ_RADIX_QSTRIDE * 8));
    auto run_variant = [&](uint32_t bits, int NT, int K, int blocks_per_cu) {
        const uint32_t P = 1u << bits;
        const int T = NT * K;
        int grid = CUS * blocks_per_cu;
        const int64_t ntiles = (NP + T - 1) / T;
        if (grid > ntiles) grid = (int)ntiles;
        st.bits = bits;
        st.R = 8u;
        const double lam = (double)NP / ((double)P * st.R);
        st.cap = (uint32_t)(lam * 1.08 + 8 * sqrt(lam) + 2.0 * T / 8 / 8 + 64);
        st.cap = (st.cap + 15) & ~15u;  // 128-byte aligned regions
        if ((size_t)P * st.R * st.cap > store_keys) { printf("store too small\n"); return; }
        auto reset = [&] {
            CK(hipMemsetAsync(st.cursor, 0, (size_t)P * st.R * 4, 0));
            CK(hipMemsetAsync(st.valid_end, 0xff, (size_t)P * st.R * 4, 0));
            CK(hipMemsetAsync(st.ovf_count, 0, 4, 0));
        };
        auto launch_part = [&] {
#define PART(NT_, K_, W_) \
    if (NT == NT_ && K == K_) hipLaunchKernelGGL((k_radix_partition<NT_, K_, W_, 0, false>), dim3(grid), dim3(NT_), 0, 0, src, st);
            PART(256, 16, 3) PART(512, 8, 6) PART(512, 14, 4) PART(1024, 16, 4)
        };
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            reset();
            float ms = time_ms(launch_part, 1);
            best = ms < best ? ms : best;
        }
        CK(hipGetLastError());
        CK(hipMemset(d_sum, 0, 16));
        hipLaunchKernelGGL(k_sum_cursor, dim3(256), dim3(256), 0, 0, st.cursor, st.valid_end, P * st.R, st.cap, d_sum);
        unsigned long long hs[2]; uint32_t ovf;
        CK(hipMemcpy(hs, d_sum, 16, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&ovf, st.ovf_count, 4, hipMemcpyDeviceToHost));
        printf("PART bits=%2u NT=%4d K=%2d grid=%4d cap=%6u : %.3f ms  %.1f Gkeys/s  %.0f GB/s(16B/key)  stored=%llu ovf=%u maxfill=%llu %s\n", bits, NT, K, grid,
               st.cap, best, NP / best / 1e6, NP * 16.0 / best / 1e6, hs[0], ovf, hs[1], hs[0] + ovf == (unsigned long long)NP ? "ok" : "BAD");
        RadixProbeArgs pa{};
        pa.st = st; pa.t = t; pa.counters = counters;
        for (int Jc : {5, 6}) {  // workgroups per CU
            const uint32_t J = (uint32_t)(CUS / 8 * Jc);
            for (int U : {2}) {  // U + 100*PFB + 10*(lead-1)
                float bestp = 1e30f;
                unsigned long long c = 0;
                for (int rep = 0; rep < 3; rep++) {
                    CK(hipMemset(st.queue, 0, 8 * TSQ_RADIX_QSTRIDE * 8));
                    CK(hipMemset(counters, 0, 64));
                    float ms = time_ms([&] {
                        pa.pf_lead = (uint32_t)((U / 10) % 10) + 1;
                        if (U == 2) hipLaunchKernelGGL((k_radix_probe_count<2, 0>), dim3(J * 8), dim3(256), 0, 0, pa);
                        if (U == 102) hipLaunchKernelGGL((k_radix_probe_count<2, 1>), dim3((J + 1) * 8), dim3(256), 0, 0, pa);
                        if (U == 202) hipLaunchKernelGGL((k_radix_probe_count<2, 2>), dim3((J + 2) * 8), dim3(256), 0, 0, pa);
                        hipLaunchKernelGGL(k_radix_probe_ovf, dim3(256), dim3(256), 0, 0, pa);
                    }, 1);
                    bestp = ms < bestp ? ms : bestp;
                    c = get_count();
                }
                CK(hipGetLastError());
                printf("   PROBE J/CU=%d U=%d : %.3f ms  %.1f Grows/s   part+probe %.3f ms = %.1f Grows/s = %.1f%% of 8TB/s @24B  count %s\n", Jc, U, bestp, NP / bestp / 1e6, bestp + best,
                       NP / (bestp + best) / 1e6, NP * 24.0 / (bestp + best) / 1e6 / 8000 * 100, c == (unsigned long long)NP ? "ok" : "BAD");
            }
        }
    };
    const int only = argc > 3 ? atoi(argv[3]) : -1;
    int vi = 0;
#define V(...) { if (only < 0 || only == vi) run_variant(__VA_ARGS__); vi++; }
    V(10, 1024, 16, 1)

    return 0;
}
