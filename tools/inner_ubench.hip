// inner_ubench.hip — throughput of the probe kernel's inner loop alone (quad bucket reads + scalar-mask compare), on a
// table folded into 256 KB per XCD (all L2 hits), grid-stride over the raw probe keys: no queue, no barrier, no scout.
// Tells how much of k_radix_probe_count's 1.0 ms is the per-chunk machinery and how much the loop body itself.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../tinysql_amd/csrc/tsq_radix.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
__global__ void __launch_bounds__(256) k_gen(uint64_t* pk, int64_t np) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < np; i += (int64_t)gridDim.x * blockDim.x) pk[i] = tsq_splitmix64(42 ^ (uint64_t)i) % 100000000ULL;
}
template <int U, bool HASH>
__global__ void __launch_bounds__(256) k_inner(const uint64_t* __restrict__ pk, int64_t np, const uint64_t* __restrict__ tkeys, uint64_t nb, unsigned long long* out) {
    const int lane = threadIdx.x & 63;
    const uint32_t vx = blockIdx.x & 7u;
    uint64_t scnt = 0;
    const int64_t stride = (int64_t)gridDim.x * 256 * U;
    for (int64_t i0 = (int64_t)blockIdx.x * 256 * U + threadIdx.x; i0 < np; i0 += stride) {
        uint64_t k[U], bkt[U];
        ulonglong2 L[U][4];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int64_t i = i0 + u * 256;
            k[u] = i < np ? __builtin_nontemporal_load(pk + i) : 0;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t h = HASH ? tsq_mix64(k[u]) : k[u] * 0x9E3779B97F4A7C15ULL;
            bkt[u] = (radix_bucket(h, nb) & 4095u) + vx * 4096u;
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const uint64_t bq = __shfl(bkt[u], g * 16 + (lane >> 2), 64);
                L[u][g] = reinterpret_cast<const ulonglong2*>(tkeys + bq * 8)[lane & 3];
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const uint64_t kw = __shfl(k[u], g * 16 + (lane >> 2), 64);
                scnt += (uint64_t)__popcll(__ballot(L[u][g].x == kw)) + (uint64_t)__popcll(__ballot(L[u][g].y == kw));
            }
        }
    }
    if (lane == 0 && scnt) atomicAdd(out, (unsigned long long)scnt);
}
template <class F> static float time_ms(F&& f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); f(); hipDeviceSynchronize(); float best = 1e30f;
    for (int r = 0; r < 3; r++) { hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best; }
    return best;
}
int main() {
    const int64_t NP = 100000000;
    uint64_t *pk, *tk; unsigned long long* out;
    CK(hipMalloc(&pk, NP * 8)); CK(hipMalloc(&tk, 8 * 4096 * 64)); CK(hipMalloc(&out, 8)); CK(hipMemset(out, 0, 8)); CK(hipMemset(tk, 0x55, 8 * 4096 * 64));
    hipLaunchKernelGGL(k_gen, dim3(2048), dim3(256), 0, 0, pk, NP);
    for (int bpc : {4, 6, 8})
#define R(U, H) { float ms = time_ms([&] { hipLaunchKernelGGL((k_inner<U, H>), dim3(256 * bpc), dim3(256), 0, 0, pk, NP, tk, (uint64_t)25000000, out); }); printf("blocks/CU=%d U=%d hash=%s : %.3f ms  %.1f Gkeys/s\n", bpc, U, H ? "mix64" : "1 mul ", ms, NP / ms / 1e6); }
    { R(1, true) R(2, true) R(4, true) R(2, false) }
    return 0;
}
