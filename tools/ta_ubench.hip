// ta_ubench.hip — does the texture-address unit charge a random 64-byte bucket read per LANE access or
// per distinct LINE?  (a) one lane reads the whole bucket with 4 dwordx4 loads; (b) 4 adjacent lanes
// read 16 B each of the same bucket with ONE dwordx4 load.  Table small enough to sit in every L2.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)
__device__ __forceinline__ uint64_t mix64(uint64_t k) { k ^= k >> 33; k *= 0xFF51AFD7ED558CCDULL; k ^= k >> 33; k *= 0xC4CEB9FE1A85EC53ULL; k ^= k >> 33; return k; }

template <int ROWS>
__global__ void __launch_bounds__(256) k_lane(const ulonglong2* __restrict__ tab, uint64_t nb, int64_t n, unsigned long long* out) {
    uint64_t acc = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * ROWS;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * ROWS; i < n; i += stride) {
        ulonglong2 v[ROWS][4];
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            const ulonglong2* p = tab + __umul64hi(mix64((uint64_t)(i + r)), nb) * 4;
#pragma unroll
            for (int k = 0; k < 4; k++) v[r][k] = p[k];
        }
#pragma unroll
        for (int r = 0; r < ROWS; r++)
#pragma unroll
            for (int k = 0; k < 4; k++) acc += v[r][k].x ^ v[r][k].y;
    }
    if (acc == 0x1234567) atomicAdd(out, 1ull);
}
// 4 lanes per bucket: lane computes the hash of ITS key index (64 keys per wave per round), buckets are
// handed to quads with a wave shuffle; ROUNDS x 4 dwordx4 loads in flight per lane.
template <int ROUNDS>
__global__ void __launch_bounds__(256) k_quad(const ulonglong2* __restrict__ tab, uint64_t nb, int64_t n, unsigned long long* out) {
    uint64_t acc = 0;
    const int lane = threadIdx.x & 63;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * ROUNDS;
    for (int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63)) * ROUNDS; i0 < n; i0 += stride) {
        ulonglong2 v[ROUNDS][4];
#pragma unroll
        for (int r = 0; r < ROUNDS; r++) {
            const uint64_t b = __umul64hi(mix64((uint64_t)(i0 + r * 64 + lane)), nb);
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const uint64_t bq = __shfl(b, s * 16 + (lane >> 2), 64);
                v[r][s] = tab[bq * 4 + (lane & 3)];
            }
        }
#pragma unroll
        for (int r = 0; r < ROUNDS; r++)
#pragma unroll
            for (int s = 0; s < 4; s++) acc += v[r][s].x ^ v[r][s].y;
    }
    if (acc == 0x1234567) atomicAdd(out, 1ull);
}
template <class F>
static float time_ms(F&& launch, int reps = 3) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(); hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < reps; r++) { hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
    return best;
}
int main() {
    const int64_t N = 100000000;
    const size_t MAXB = (size_t)2 << 30;
    ulonglong2* tab; unsigned long long* out;
    CK(hipMalloc(&tab, MAXB)); CK(hipMalloc(&out, 8)); CK(hipMemset(out, 0, 8)); CK(hipMemset(tab, 1, MAXB));
    for (int bpc : {4, 8})
    for (size_t sz : {(size_t)256 << 10, (size_t)1 << 20, (size_t)2 << 20, (size_t)1600 << 20}) {
        const uint64_t nb = sz / 64;
        const int grid = 256 * bpc;
#define L(R) { float ms = time_ms([&] { hipLaunchKernelGGL((k_lane<R>), dim3(grid), dim3(256), 0, 0, tab, nb, N, out); }); printf("blocks/CU=%d table=%7zu KB lane x4 rows=%d : %.3f ms %.1f G/s\n", bpc, sz >> 10, R, ms, N / ms / 1e6); }
#define Q(R) { float ms = time_ms([&] { hipLaunchKernelGGL((k_quad<R>), dim3(grid), dim3(256), 0, 0, tab, nb, N, out); }); printf("blocks/CU=%d table=%7zu KB quad   rounds=%d : %.3f ms %.1f G/s\n", bpc, sz >> 10, R, ms, N / ms / 1e6); }
        L(1) L(2) Q(1) Q(2) Q(4)
    }
    return 0;
}
