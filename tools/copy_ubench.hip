// copy_ubench.hip — what streaming kernels can reach on this MI355X: read-only, write-only and copy (read + write) rates
// with 16-byte accesses, grid-stride, 2048 x 256 threads.  Context for the roofline fractions in DESIGN.md (peak 8 TB/s).
// build: hipcc -O3 --offload-arch=gfx950 tools/copy_ubench.hip -o tools/copy_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void __launch_bounds__(256) k_read(const uint4* __restrict__ a, size_t n, unsigned long long* out) {
    unsigned long long s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = a[i];
        s += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (s == 0x12345678deadbeefULL) *out = s;
}
__global__ void __launch_bounds__(256) k_write(uint4* __restrict__ a, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = make_uint4((unsigned)i, 1, 2, 3);
}
__global__ void __launch_bounds__(256) k_copy(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}

int main() {
    const size_t bytes = (size_t)2 << 30, n = bytes / 16;
    uint4 *a, *b;
    unsigned long long* out;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&out, 8);
    hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grids[] = {2048, 8192};
    for (int g : grids) {
        for (int which = 0; which < 3; which++) {
            float best = 1e30f;
            for (int rep = 0; rep < 5; rep++) {
                hipEventRecord(e0, 0);
                if (which == 0) hipLaunchKernelGGL(k_read, dim3(g), dim3(256), 0, 0, a, n, out);
                else if (which == 1) hipLaunchKernelGGL(k_write, dim3(g), dim3(256), 0, 0, b, n);
                else hipLaunchKernelGGL(k_copy, dim3(g), dim3(256), 0, 0, a, b, n);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double moved = which == 2 ? 2.0 * bytes : (double)bytes;
            printf("grid %5d %-5s %7.3f ms  %7.1f GB/s\n", g, which == 0 ? "read" : which == 1 ? "write" : "copy", best, moved / best / 1e6);
        }
    }
    return 0;
}
