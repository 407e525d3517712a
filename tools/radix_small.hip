// radix_small.hip — small-input harness for the radix path (bring-up / debugging).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <unordered_map>
#include "../tinysql_amd/csrc/tsq_radix.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
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
int main(int argc, char** argv) {
    const int64_t NB = argc > 1 ? atoll(argv[1]) : 547, NP = argc > 2 ? atoll(argv[2]) : 3009;
    const uint64_t D = argc > 3 ? atoll(argv[3]) : 60;
    const uint32_t bits = argc > 4 ? atoi(argv[4]) : 3;
    std::vector<uint64_t> hb(NB), hp(NP);
    std::unordered_map<uint64_t, uint64_t> mult;
    for (int64_t i = 0; i < NB; i++) { hb[i] = tsq_splitmix64(i) % D; mult[hb[i]]++; }
    uint64_t want = 0;
    for (int64_t i = 0; i < NP; i++) { hp[i] = tsq_splitmix64(1000003 + i) % (D + 5); auto it = mult.find(hp[i]); if (it != mult.end()) want += it->second; }
    uint64_t *bk, *pk;
    CK(hipMalloc(&bk, NB * 8)); CK(hipMalloc(&pk, NP * 8));
    CK(hipMemcpy(bk, hb.data(), NB * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(pk, hp.data(), NP * 8, hipMemcpyHostToDevice));
    JoinTable t{};
    t.nbuckets = (uint64_t)((NB + 3) / 4); if (t.nbuckets < 16) t.nbuckets = 16;
    CK(hipMalloc(&t.keys, t.nbuckets * 64)); CK(hipMemset(t.keys, 0x80, t.nbuckets * 64));
    hipLaunchKernelGGL(k_build, dim3(64), dim3(256), 0, 0, bk, NB, t);
    CK(hipDeviceSynchronize());
    printf("built %lld rows, %llu buckets; probe %lld rows, D=%llu bits=%u want=%llu\n", (long long)NB, (unsigned long long)t.nbuckets, (long long)NP, (unsigned long long)D, bits, (unsigned long long)want); fflush(stdout);
    RadixStore st{};
    st.bits = bits; st.R = 8;
    const uint32_t P = 1u << bits;
    const double lam = (double)NP / (P * 8.0);
    st.cap = ((uint32_t)(lam * 1.08 + 8 * sqrt(lam) + 2.0 * 16384 / 64 + 64) + 15) & ~15u;
    if (argc > 5) st.cap = (uint32_t)atoi(argv[5]);
    const size_t nreg = (size_t)P * 8;
    CK(hipMalloc(&st.keys, nreg * st.cap * 8 + 256));
    CK(hipMalloc(&st.cursor, nreg * 4)); CK(hipMemset(st.cursor, 0, nreg * 4));
    CK(hipMalloc(&st.valid_end, nreg * 4)); CK(hipMemset(st.valid_end, 0xff, nreg * 4));
    CK(hipMalloc(&st.ovf_keys, NP * 8 + 64)); CK(hipMalloc(&st.ovf_count, 4)); CK(hipMemset(st.ovf_count, 0, 4));
    CK(hipMalloc(&st.queue, 8 * TSQ_RADIX_QSTRIDE * 8)); CK(hipMemset(st.queue, 0, 8 * TSQ_RADIX_QSTRIDE * 8));
    st.ovf_cap = (uint32_t)NP;
    RadixSrc src{};
    src.data = pk; src.type = TSQ_I64; src.nrows = NP;
    const int T = 16384;
    int grid = (int)((NP + T - 1) / T); if (grid > 256) grid = 256;
    hipLaunchKernelGGL((k_radix_partition<1024, 16, 4, 0, false>), dim3(grid), dim3(1024), 0, 0, src, st);
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    std::vector<uint32_t> cur(nreg), ve(nreg); uint32_t ovf;
    CK(hipMemcpy(cur.data(), st.cursor, nreg * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(ve.data(), st.valid_end, nreg * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&ovf, st.ovf_count, 4, hipMemcpyDeviceToHost));
    uint64_t stored = 0; uint32_t mx = 0;
    for (size_t i = 0; i < nreg; i++) { uint32_t c = cur[i] < ve[i] ? cur[i] : ve[i]; c = c < st.cap ? c : st.cap; stored += c; mx = c > mx ? c : mx; }
    printf("partition done: cap=%u stored=%llu ovf=%u maxfill=%u\n", st.cap, (unsigned long long)stored, ovf, mx); fflush(stdout);
    unsigned long long* counters; CK(hipMalloc(&counters, 64)); CK(hipMemset(counters, 0, 64));
    RadixProbeArgs pa{};
    pa.st = st; pa.t = t; pa.counters = counters;
    hipLaunchKernelGGL((k_radix_probe_count<2, 0>), dim3(8 * 32 * 6), dim3(256), 0, 0, pa);
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    printf("probe done\n"); fflush(stdout);
    hipLaunchKernelGGL(k_radix_probe_ovf, dim3(256), dim3(256), 0, 0, pa);
    CK(hipDeviceSynchronize());
    unsigned long long c;
    CK(hipMemcpy(&c, counters, 8, hipMemcpyDeviceToHost));
    printf("count=%llu %s\n", c, c == want ? "ok" : "BAD");
    return 0;
}
