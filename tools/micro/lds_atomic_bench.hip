#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
// LDS atomic throughput: 512 threads, each does N atomics to pseudo-random addresses in a 32 KB LDS array
template <int MODE>
__global__ void __launch_bounds__(512) k(const uint32_t *idx, float *out, int n) {
    __shared__ float acc[8192];
    for (int i = threadIdx.x; i < 8192; i += 512) acc[i] = 0.f;
    __syncthreads();
    uint32_t r = idx[threadIdx.x + blockIdx.x * 512];
    for (int i = 0; i < n; i++) {
        r = r * 1664525u + 1013904223u;
        const uint32_t a = (r >> 10) & 8191u;
        if (MODE == 0) atomicAdd(&acc[a], 1.0f);
        else if (MODE == 1) atomicAdd(reinterpret_cast<unsigned int *>(acc) + a, 1u);
        else if (MODE == 2) atomicAdd(reinterpret_cast<unsigned long long *>(acc) + (a >> 1), 1ull);
        else if (MODE == 3) acc[a] = 1.0f;      // plain store
        else if (MODE == 4) { float old = acc[a]; acc[a] = old + 1.0f; }  // non-atomic RMW
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = acc[1];
}
int main() {
    uint32_t *idx; float *out;
    hipMalloc(&idx, 4 * 512 * 2048); hipMalloc(&out, 4 * 2048);
    hipMemset(idx, 1, 4 * 512 * 2048);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int n = 64, blocks = 1024;
    auto run = [&](int mode) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(a);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(512), 0, 0, idx, out, n);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(512), 0, 0, idx, out, n);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(512), 0, 0, idx, out, n);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(512), 0, 0, idx, out, n);
            if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(512), 0, 0, idx, out, n);
            hipEventRecord(b); hipEventSynchronize(b);
        }
        float ms; hipEventElapsedTime(&ms, a, b);
        const double ops = (double)blocks * 512 * n;
        printf("mode %d: %.1f us, %.2f lane-ops/clk/CU (2.1 GHz, 256 CUs)\n", mode, ms * 1e3, ops / (ms * 1e-3) / 2.1e9 / 256);
    };
    for (int m = 0; m < 5; m++) run(m);
    return 0;
}
