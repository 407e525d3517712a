// tools/partition_check.hip — differential check of the two packed partition kernels.
// Runs k_da_partition<1024,16,ET> and k_da_partition2<512,8,4,true,ET> on the same keys (a hot key: runs that overflow their region)
// and compares what each left in its store — region entries [0, len) + overflow list — with the multiset of words computed on the host.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-atomic-optimizer-strategy=None -I tinysql_amd/csrc -I include tools/partition_check.hip -o tools/partition_check
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <random>
#include <vector>
#include "tsq_dajoin.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)

template <typename ET, int VAR>
static int run_case(const char* name, uint32_t b, uint32_t pb, const std::vector<uint64_t>& keys, const uint8_t* nulls_host, int64_t kmin, uint64_t range) {
    const int64_t n = (int64_t)keys.size();
    DaDomain dm{(uint64_t)kmin, range, b, (b + 1) / 2, (uint32_t)((1ull << b) - 1), 0};
    const uint32_t ebits = b - pb, P = 1u << pb;
    const int T = 16384;
    const double tiles = ceil((double)n / T);
    const double lam = std::max((double)n / ((double)P * 8.0), ceil(tiles / 8.0) * std::min<double>(T, n) / P);
    uint32_t cap = (uint32_t)(lam * 1.08 + 8.0 * sqrt(lam) + 2.0 * T / 64.0 + 64.0);
    cap = (cap + 63u) & ~63u;
    const size_t nreg = (size_t)P * 8;
    uint64_t* dkeys; uint8_t* dnulls = nullptr; ET* dent; uint32_t *dctl, *dvend, *dovf;
    CK(hipMalloc(&dkeys, n * 8 + 64)); CK(hipMemcpy(dkeys, keys.data(), n * 8, hipMemcpyHostToDevice));
    if (nulls_host) { CK(hipMalloc(&dnulls, n / 8 + 64)); CK(hipMemcpy(dnulls, nulls_host, (n + 7) / 8, hipMemcpyHostToDevice)); }
    CK(hipMalloc(&dent, nreg * cap * sizeof(ET) + 256)); CK(hipMemset(dent, 0xEE, nreg * cap * sizeof(ET)));
    CK(hipMalloc(&dctl, (nreg + 16) * 4)); CK(hipMemset(dctl, 0, (nreg + 16) * 4));
    CK(hipMalloc(&dvend, nreg * 4)); CK(hipMemset(dvend, 0xff, nreg * 4));
    CK(hipMalloc(&dovf, n * 4 + 64));
    DaStore st; memset(&st, 0, sizeof st);
    st.ent = dent; st.cursor = dctl; st.ovf_count = dctl + nreg; st.valid_end = dvend; st.ovf = dovf; st.ovf_cap = (uint32_t)n; st.bits = pb; st.ebits = ebits; st.cap = cap;
    DaSrc src{dkeys, dnulls, n};
    const int64_t ntiles = (n + T - 1) / T;
    if (VAR == 1) hipLaunchKernelGGL((k_da_partition<1024, 16, ET>), dim3((unsigned)std::min<int64_t>(ntiles, 256)), dim3(1024), 0, 0, src, dm, st);
    else hipLaunchKernelGGL((k_da_partition2<512, 8, 4, true, ET>), dim3((unsigned)std::min<int64_t>(ntiles, 512)), dim3(512), 0, 0, src, dm, st);
    CK(hipGetLastError());
    hipError_t se = hipDeviceSynchronize();
    if (se != hipSuccess) { printf("[%s var %d] kernel FAULT: %s\n", name, VAR, hipGetErrorString(se)); return 1; }
    std::vector<uint32_t> cur(nreg + 16), vend(nreg), ovf((size_t)n);
    std::vector<ET> ent(nreg * cap);
    CK(hipMemcpy(cur.data(), dctl, (nreg + 16) * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(vend.data(), dvend, nreg * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(ent.data(), dent, nreg * cap * sizeof(ET), hipMemcpyDeviceToHost));
    const uint32_t novf = std::min<uint32_t>(cur[nreg], (uint32_t)n);
    CK(hipMemcpy(ovf.data(), dovf, (size_t)novf * 4, hipMemcpyDeviceToHost));
    std::map<uint32_t, int64_t> want, got;
    for (int64_t i = 0; i < n; i++) {
        if (nulls_host && !((nulls_host[i >> 3] >> (i & 7)) & 1)) continue;
        const uint64_t d = keys[i] - dm.kmin;
        if (d > dm.range) continue;
        want[tsq_da_mix((uint32_t)d, dm.s, dm.mask)]++;
    }
    int64_t in_regions = 0, garbage = 0;
    for (uint32_t p = 0; p < P; p++)
        for (uint32_t r = 0; r < 8; r++) {
            const uint32_t c = r * P + p;
            uint32_t len = std::min(std::min(cur[c], vend[c]), cap);
            for (uint32_t i = 0; i < len; i++) {
                const ET e = ent[(size_t)(p * 8 + r) * cap + i];
                if ((uint64_t)e >> ebits) garbage++;
                got[(p << ebits) | (uint32_t)e]++;
                in_regions++;
            }
        }
    for (uint32_t i = 0; i < novf; i++) got[ovf[i]]++;
    int64_t missing = 0, extra = 0, shown = 0;
    for (auto& kv : want) {
        const int64_t g = got.count(kv.first) ? got[kv.first] : 0;
        if (g < kv.second) { missing += kv.second - g; if (shown++ < 5) printf("   word %08x (p %u e %u): want %lld got %lld\n", kv.first, kv.first >> ebits, kv.first & ((1u << ebits) - 1), (long long)kv.second, (long long)g); }
        if (g > kv.second) { extra += g - kv.second; if (shown++ < 5) printf("   word %08x (p %u e %u): want %lld got %lld\n", kv.first, kv.first >> ebits, kv.first & ((1u << ebits) - 1), (long long)kv.second, (long long)g); }
    }
    for (auto& kv : got)
        if (!want.count(kv.first)) { extra += kv.second; if (shown++ < 8) printf("   word %08x (p %u e %u): not a key, got %lld\n", kv.first, kv.first >> ebits, kv.first & ((1u << ebits) - 1), (long long)kv.second); }
    printf("[%s var %d sizeof(ET) %zu] n %lld P %u ebits %u cap %u: in regions %lld, overflow list %u, garbage entries %lld, missing %lld, extra %lld -> %s\n", name, VAR, sizeof(ET),
           (long long)n, P, ebits, cap, (long long)in_regions, novf, (long long)garbage, (long long)missing, (long long)extra, (missing || extra) ? "MISMATCH" : "ok");
    hipFree(dkeys); if (dnulls) hipFree(dnulls); hipFree(dent); hipFree(dctl); hipFree(dvend); hipFree(dovf);
    return (missing || extra) ? 1 : 0;
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    // usage: partition_check N HOT B VARIANT — one case per process (a fault must not hide the other cases); no arguments: the list
    if (argc < 5) {
        for (long long n : {16384LL, 49153LL, 82020LL, 9830477LL})
            for (int hot : {0, 1})
                for (unsigned b : {20u, 27u, 28u, 30u})
                    for (int var : {1, 2}) printf("%lld %d %u %d\n", n, hot, b, var);
        return 0;
    }
    const int64_t n = atoll(argv[1]);
    const int hot = atoi(argv[2]);
    const uint32_t b = (uint32_t)atoi(argv[3]);
    const int var = atoi(argv[4]);
    std::mt19937_64 rng(7 + (uint64_t)n + b);
    const uint64_t span = (1ull << b) - 3;
    std::vector<uint64_t> keys((size_t)n);
    for (auto& k : keys) k = 1000 + rng() % span;
    if (hot)
        for (int64_t i = 0; i < std::min<int64_t>(n, 40 * 16384); i++)
            if (rng() % 10 < 6) keys[(size_t)i] = 1000 + 12345 % span;
    const uint32_t max_e = b > 28 ? 20u : 17u;
    uint32_t pb = std::min(11u, b - 10u);
    if (pb + max_e < b) pb = b - max_e;
    char name[96];
    snprintf(name, sizeof name, "n=%lld hot=%d b=%u", (long long)n, hot, b);
    printf("case %s var %d: pb %u ebits %u\n", name, var, pb, b - pb);
    int bad;
    if (b - pb <= 16) bad = var == 1 ? run_case<uint16_t, 1>(name, b, pb, keys, nullptr, 1000, span - 1) : run_case<uint16_t, 2>(name, b, pb, keys, nullptr, 1000, span - 1);
    else bad = var == 1 ? run_case<uint32_t, 1>(name, b, pb, keys, nullptr, 1000, span - 1) : run_case<uint32_t, 2>(name, b, pb, keys, nullptr, 1000, span - 1);
    return bad;
}
