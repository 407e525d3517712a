// lds_ubench.hip — what an LDS operation with 64 random addresses costs on gfx950 (cycles per wave instruction, per CU, with 8 / 16 waves
// resident): plain read, plain write, atomic add without / with return, 64-bit atomic add; addresses random over 1024 / 4096 words, or
// conflict-free (lane-linear).  The partition kernels issue 9-14 such operations per 64 rows (profiles/r06_headline_sq.txt).
// build: hipcc -O3 --offload-arch=gfx950 -o tools/lds_ubench tools/lds_ubench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int OP>
__global__ void __launch_bounds__(1024) k(uint32_t* out, const uint32_t* idx, int iters, uint32_t mask, int linear) {
    __shared__ unsigned long long s[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) s[i] = i;
    __syncthreads();
    uint32_t a = linear ? (threadIdx.x & 63u) : (idx[blockIdx.x * blockDim.x + threadIdx.x] & mask);
    uint32_t acc = 0;
    uint32_t* s32 = reinterpret_cast<uint32_t*>(s);
    for (int i = 0; i < iters; i++) {
        if (OP == 0) acc += s32[a];
        else if (OP == 1) s32[a] = acc + i;
        else if (OP == 2) atomicAdd(&s32[a], 1u);
        else if (OP == 3) acc += atomicAdd(&s32[a], 1u);
        else if (OP == 4) atomicAdd(&s[a], 1ull);
        else if (OP == 5) acc += (uint32_t)atomicAdd(&s[a], 1ull);
        a = linear ? a : ((a * 1664525u + 1013904223u + acc * 0u) & mask);  // next pseudo-random word (independent of the loaded value)
    }
    __syncthreads();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + s32[threadIdx.x];
}

template <int OP>
double run(uint32_t* out, uint32_t* idx, int nt, uint32_t mask, int linear, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(nt), 0, 0, out, idx, iters, mask, linear);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(nt), 0, 0, out, idx, iters, mask, linear);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // one workgroup per CU (256 CUs): wave instructions per CU = (nt / 64) * iters; cycles at 2.4 GHz
    return (double)ms * 1e-3 * 2.4e9 / ((double)(nt / 64) * iters);
}

int main() {
    const int n = 256 * 1024;
    std::vector<uint32_t> h(n);
    uint32_t x = 12345;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = x >> 8; }
    uint32_t *idx, *out;
    hipMalloc(&idx, n * 4);
    hipMalloc(&out, n * 4);
    hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice);
    const char* names[6] = {"ds_read_b32", "ds_write_b32", "ds_add_u32", "ds_add_rtn_u32", "ds_add_u64", "ds_add_rtn_u64"};
    printf("cycles per wave instruction and CU (2.4 GHz assumed), 2000 iterations\n%-16s %10s %10s %10s %10s\n", "op", "lin/16w", "1024w/16w", "4096w/16w", "1024w/8w");
    for (int op = 0; op < 6; op++) {
        double r[4];
        for (int c = 0; c < 4; c++) {
            const int nt = c == 3 ? 512 : 1024;
            const uint32_t mask = c == 2 ? 4095u : 1023u;
            const int lin = c == 0;
            switch (op) {
                case 0: r[c] = run<0>(out, idx, nt, mask, lin, 2000); break;
                case 1: r[c] = run<1>(out, idx, nt, mask, lin, 2000); break;
                case 2: r[c] = run<2>(out, idx, nt, mask, lin, 2000); break;
                case 3: r[c] = run<3>(out, idx, nt, mask, lin, 2000); break;
                case 4: r[c] = run<4>(out, idx, nt, mask, lin, 2000); break;
                default: r[c] = run<5>(out, idx, nt, mask, lin, 2000); break;
            }
        }
        printf("%-16s %10.1f %10.1f %10.1f %10.1f\n", names[op], r[0], r[1], r[2], r[3]);
    }
    return 0;
}
