// probe_ubench.hip — micro-benchmark of the memory access patterns a hash-join probe can use on
// MI355X: how fast can 256 CUs fetch RANDOM 8/16/64/128-byte pieces from tables of different sizes
// (L2 / Infinity Cache / HBM resident)?  Drives the table-layout decisions in DESIGN.md.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe_ubench.hip -o tools/probe_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint64_t mix64(uint64_t k) {
    k ^= k >> 33; k *= 0xFF51AFD7ED558CCDULL; k ^= k >> 33; k *= 0xC4CEB9FE1A85EC53ULL; k ^= k >> 33; return k;
}

// streaming read of 8-byte keys (the probe-key stream), 16 B per lane
__global__ void __launch_bounds__(256) k_stream(const ulonglong2* __restrict__ p, int64_t n2, unsigned long long* out) {
    uint64_t acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (int64_t)gridDim.x * blockDim.x) {
        ulonglong2 v = p[i];
        acc += v.x ^ v.y;
    }
    if (acc == 0x1234567) atomicAdd(out, 1ull);
}

// random access: each lane reads BYTES contiguous bytes at a random BYTES-aligned position; ROWS independent rows per iteration
template <int BYTES, int ROWS>
__global__ void __launch_bounds__(256) k_random(const uint64_t* __restrict__ tab, uint64_t nunits, int64_t nprobe, unsigned long long* out) {
    uint64_t acc = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * ROWS;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * ROWS; i < nprobe; i += stride) {
        uint64_t u[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; r++) u[r] = __umul64hi(mix64((uint64_t)(i + r)), nunits);
        if (BYTES == 8) {
#pragma unroll
            for (int r = 0; r < ROWS; r++) acc += tab[u[r]];
        } else {
            constexpr int V = BYTES / 16;
            ulonglong2 v[ROWS][V];
#pragma unroll
            for (int r = 0; r < ROWS; r++) {
                const ulonglong2* p = reinterpret_cast<const ulonglong2*>(tab) + u[r] * V;
#pragma unroll
                for (int k = 0; k < V; k++) v[r][k] = p[k];
            }
#pragma unroll
            for (int r = 0; r < ROWS; r++)
#pragma unroll
                for (int k = 0; k < V; k++) acc += v[r][k].x ^ v[r][k].y;
        }
    }
    if (acc == 0x1234567) atomicAdd(out, 1ull);
}

// random atomics: the aggregate / build pattern
__global__ void __launch_bounds__(256) k_atomic_add(unsigned long long* tab, uint64_t nslots, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        atomicAdd(&tab[__umul64hi(mix64((uint64_t)i), nslots)], 1ull);
}
__global__ void __launch_bounds__(256) k_atomic_cas(unsigned long long* tab, uint64_t nslots, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        atomicCAS(&tab[__umul64hi(mix64((uint64_t)i), nslots)], 0ull, (unsigned long long)i + 1);
}
__global__ void __launch_bounds__(256) k_atomic_addf(double* tab, uint64_t nslots, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        atomicAdd(&tab[__umul64hi(mix64((uint64_t)i), nslots)], 1.0);
}

template <class F>
static float time_ms(F&& launch, int reps = 3) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    launch();  // warm
    hipDeviceSynchronize();
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

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs=%d L2=%d MB clock=%d MHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.l2CacheSize >> 20, prop.clockRate / 1000);
    const int64_t NPROBE = 100000000;
    const size_t MAXB = (size_t)4 << 30;
    uint64_t* tab; unsigned long long* out;
    CK(hipMalloc(&tab, MAXB)); CK(hipMalloc(&out, 8)); CK(hipMemset(out, 0, 8));
    CK(hipMemset(tab, 1, MAXB));
    const int grid = prop.multiProcessorCount * 8;
    {
        float ms = time_ms([&] { hipLaunchKernelGGL(k_stream, dim3(grid), dim3(256), 0, 0, (const ulonglong2*)tab, (int64_t)(MAXB / 16), out); });
        printf("stream_read 4GiB: %.3f ms  %.1f GB/s\n", ms, MAXB / ms / 1e6);
        ms = time_ms([&] { hipLaunchKernelGGL(k_stream, dim3(grid), dim3(256), 0, 0, (const ulonglong2*)tab, (int64_t)(NPROBE * 8 / 16), out); });
        printf("stream_read 0.8GB (1e8 keys): %.3f ms  %.1f GB/s\n", ms, NPROBE * 8.0 / ms / 1e6);
    }
    const size_t sizes[] = {(size_t)2 << 20, (size_t)24 << 20, (size_t)160 << 20, (size_t)800 << 20, (size_t)1600 << 20, (size_t)3200 << 20};
    for (size_t sz : sizes) {
#define RUN(B, R)                                                                                                          \
    {                                                                                                                      \
        uint64_t nunits = sz / B;                                                                                          \
        float ms = time_ms([&] { hipLaunchKernelGGL((k_random<B, R>), dim3(grid), dim3(256), 0, 0, tab, nunits, NPROBE, out); }); \
        printf("random table=%5zu MB  unit=%3dB rows/lane=%d : %8.3f ms  %7.2f Grows/s  line-traffic(unit) %7.1f GB/s\n", sz >> 20, B, R, ms, \
               NPROBE / ms / 1e6, (double)NPROBE * B / ms / 1e6);                                                          \
    }
        RUN(8, 1) RUN(8, 4) RUN(16, 1) RUN(64, 1) RUN(64, 2) RUN(64, 4) RUN(128, 1) RUN(128, 2)
    }
    for (size_t sz : {(size_t)48 << 20, (size_t)1600 << 20}) {
        uint64_t nslots = sz / 8;
        CK(hipMemset(tab, 0, sz));
        float ms = time_ms([&] { hipLaunchKernelGGL(k_atomic_add, dim3(grid), dim3(256), 0, 0, (unsigned long long*)tab, nslots, NPROBE); });
        printf("atomicAdd u64 table=%5zu MB: %8.3f ms  %7.2f Gops/s\n", sz >> 20, ms, NPROBE / ms / 1e6);
        CK(hipMemset(tab, 0, sz));
        ms = time_ms([&] { hipLaunchKernelGGL(k_atomic_addf, dim3(grid), dim3(256), 0, 0, (double*)tab, nslots, NPROBE); });
        printf("atomicAdd f64 table=%5zu MB: %8.3f ms  %7.2f Gops/s\n", sz >> 20, ms, NPROBE / ms / 1e6);
        CK(hipMemset(tab, 0, sz));
        ms = time_ms([&] { hipLaunchKernelGGL(k_atomic_cas, dim3(grid), dim3(256), 0, 0, (unsigned long long*)tab, nslots, NPROBE); }, 1);
        printf("atomicCAS u64 table=%5zu MB: %8.3f ms  %7.2f Gops/s\n", sz >> 20, ms, NPROBE / ms / 1e6);
    }
    hipFree(tab); hipFree(out);
    return 0;
}
