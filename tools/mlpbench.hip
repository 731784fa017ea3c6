// tools/mlpbench.hip -- in-place row update (K2's access shape) with a compute phase between
// load and store, at 2..5 workgroups per CU: how many rows in flight does the memory system need?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) rowk(float4* buf, int delay_iters) {
    extern __shared__ float4 lds[];
    const size_t base = (size_t)blockIdx.x * 4096;
    float4 v[16];
#pragma unroll
    for (int a = 0; a < 16; ++a) v[a] = buf[base + a * 256 + threadIdx.x];
    for (int i = 0; i < delay_iters; ++i) {
#pragma unroll
        for (int a = 0; a < 16; ++a) { v[a].x = fmaf(v[a].x, 0.999f, 0.001f); v[a].y = fmaf(v[a].y, 0.999f, 0.001f); v[a].z = fmaf(v[a].z, 0.999f, 0.001f); v[a].w = fmaf(v[a].w, 0.999f, 0.001f); }
    }
#pragma unroll
    for (int a = 0; a < 16; ++a) buf[base + a * 256 + threadIdx.x] = v[a];
}
int main() {
    const int nwg = 48 * 256; const size_t n4 = (size_t)nwg * 4096;
    float4* A; hipMalloc(&A, n4 * 16); hipMemset(A, 0, n4 * 16);
    hipFuncSetAttribute((const void*)rowk, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int iters : {0, 10, 20, 40}) for (int lds : {65536, 49152, 32768}) {
        for (int r = 0; r < 100; ++r) rowk<<<nwg, 256, lds>>>(A, iters);   // clock ramp
        hipDeviceSynchronize();
        hipEventRecord(e0); for (int r = 0; r < 50; ++r) rowk<<<nwg, 256, lds>>>(A, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 50;
        printf("fma iters=%2d (%4d VALU/thread) lds=%5d (%d WG/CU): %.3f ms  %.2f TB/s\n", iters, iters * 64, lds, 163840 / lds, ms, 2 * n4 * 16 / 1e9 / ms);
    }
    return 0;
}
