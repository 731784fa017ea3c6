// tools/membench.hip -- access-pattern ceilings for the three pipeline kernels
// (measurement tool, not part of the library).  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

// row pattern (K2): WG handles one contiguous 64 KB row: 16 float4 per thread
template<int MODE> // 0 copy inplace, 1 copy out-of-place, 2 read only, 3 write only
__global__ void __launch_bounds__(256) row_kernel(float4* src, float4* dst, float* sink) {
    extern __shared__ float4 lds[];
    const size_t base = (size_t)blockIdx.x * 4096;
    float4 v[16];
    if (MODE != 3) {
#pragma unroll
        for (int a = 0; a < 16; ++a) v[a] = src[base + a * 256 + threadIdx.x];
    } else {
#pragma unroll
        for (int a = 0; a < 16; ++a) v[a] = make_float4(a, threadIdx.x, blockIdx.x, 1.f);
    }
    if (MODE == 2) {
        float acc = 0.f;
#pragma unroll
        for (int a = 0; a < 16; ++a) acc += v[a].x + v[a].y + v[a].z + v[a].w;
        if (acc == 123.456f) sink[0] = acc;
    } else {
        float4* o = (MODE == 1) ? dst : src;
#pragma unroll
        for (int a = 0; a < 16; ++a) { v[a].x += 1.0f; o[base + a * 256 + threadIdx.x] = v[a]; }
    }
}
// column pattern (K1 write / K3 read): WG handles 32 columns x 256 rows of a [256][8192] float2 matrix:
// thread (hi = t>>4, cp = t&15) touches rows hi + 16*k, 16 bytes at column pair cp
template<int MODE> // 0 read only, 1 write only, 2 read A(matrix) write B(matrix2) same pattern
__global__ void __launch_bounds__(256) col_kernel(float4* m, float4* m2, float* sink, int remap) {
    extern __shared__ float4 lds[];
    unsigned lin = blockIdx.x;
    unsigned slot, tile;
    if (remap) { unsigned xcd = lin & 7u, seq = lin >> 3; slot = seq >> 5; unsigned half = (seq >> 4) & 1u, tl = seq & 15u; tile = (half * 8u + xcd) * 16u + tl; }
    else { slot = lin >> 8; tile = lin & 255u; }
    const int t = threadIdx.x, hi = t >> 4, cp = t & 15;
    float4* p = m + (size_t)slot * (256 * 4096) + tile * 16 + cp;
    float4* q = m2 + (size_t)slot * (256 * 4096) + tile * 16 + cp;
    float4 v[16];
    if (MODE != 1) {
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = p[(size_t)(hi + 16 * k) * 4096];
    } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = make_float4(k, t, lin, 1.f);
    }
    if (MODE == 0) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) acc += v[k].x + v[k].y + v[k].z + v[k].w;
        if (acc == 123.456f) sink[0] = acc;
    } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) q[(size_t)(hi + 16 * k) * 4096] = v[k];
    }
}
// f32 input pattern (K1 read): two streams, 8 bytes per lane, 128-byte segments, stride 32 KB
__global__ void __launch_bounds__(256) k1in_kernel(const float* src, long long hop, float* sink) {
    const unsigned lin = blockIdx.x; const unsigned slot = lin >> 8, tile = lin & 255u;
    const int t = threadIdx.x, hi = t >> 4, cp = t & 15;
    const float2* sa = (const float2*)(src + (2ll * slot) * hop + tile * 32 + 2 * cp);
    const float2* sb = (const float2*)(src + (2ll * slot + 1) * hop + tile * 32 + 2 * cp);
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 16; ++a) { size_t off = (size_t)(a * 16 + hi) * 4096; float2 x = sa[off], y = sb[off]; acc += x.x + x.y + y.x + y.y; }
    if (acc == 123.456f) sink[0] = acc;
}
// plain grid-stride float4 copy (reference ceiling)
__global__ void __launch_bounds__(256) stream_copy(const float4* s, float4* d, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) d[i] = s[i];
}

template<typename F> float timeit(F f, int reps = 100) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 100; ++i) f();   // leave the idle power state (the first ~50 ms run at lower clocks)
    hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < reps; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}
int main() {
    const int npairs = 48; const size_t n4 = (size_t)npairs * 256 * 4096; // float4 count (805 MB)
    float4 *A, *B; float* sink; float* hay;
    CK(hipMalloc(&A, n4 * 16)); CK(hipMalloc(&B, n4 * 16)); CK(hipMalloc(&sink, 16));
    const long long hop = 1655808; CK(hipMalloc(&hay, (size_t)(96 * hop + 2097152 + 64) * 4));
    CK(hipMemset(A, 0, n4 * 16)); CK(hipMemset(B, 0, n4 * 16)); CK(hipMemset(hay, 0, (size_t)(96 * hop + 2097152 + 64) * 4));
    const double gb = n4 * 16 / 1e9;
    const int nwg = npairs * 256;
    for (int lds : {0, 32768, 65536}) {
        hipFuncSetAttribute((const void*)row_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        hipFuncSetAttribute((const void*)row_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        hipFuncSetAttribute((const void*)row_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        hipFuncSetAttribute((const void*)row_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        hipFuncSetAttribute((const void*)col_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        hipFuncSetAttribute((const void*)col_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        hipFuncSetAttribute((const void*)col_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        float t;
        t = timeit([&]{ row_kernel<0><<<nwg, 256, lds>>>(A, B, sink); }); printf("lds=%5d row copy inplace : %.3f ms  %.2f TB/s\n", lds, t, 2 * gb / t);
        t = timeit([&]{ row_kernel<1><<<nwg, 256, lds>>>(A, B, sink); }); printf("lds=%5d row copy A->B    : %.3f ms  %.2f TB/s\n", lds, t, 2 * gb / t);
        t = timeit([&]{ row_kernel<2><<<nwg, 256, lds>>>(A, B, sink); }); printf("lds=%5d row read only    : %.3f ms  %.2f TB/s\n", lds, t, gb / t);
        t = timeit([&]{ row_kernel<3><<<nwg, 256, lds>>>(A, B, sink); }); printf("lds=%5d row write only   : %.3f ms  %.2f TB/s\n", lds, t, gb / t);
        for (int remap : {0, 1}) {
            t = timeit([&]{ col_kernel<0><<<nwg, 256, lds>>>(A, B, sink, remap); }); printf("lds=%5d remap=%d col read   : %.3f ms  %.2f TB/s\n", lds, remap, t, gb / t);
            t = timeit([&]{ col_kernel<1><<<nwg, 256, lds>>>(A, B, sink, remap); }); printf("lds=%5d remap=%d col write  : %.3f ms  %.2f TB/s\n", lds, remap, t, gb / t);
            t = timeit([&]{ col_kernel<2><<<nwg, 256, lds>>>(A, B, sink, remap); }); printf("lds=%5d remap=%d col rd+wr  : %.3f ms  %.2f TB/s\n", lds, remap, t, 2 * gb / t);
        }
        t = timeit([&]{ k1in_kernel<<<nwg, 256, lds>>>(hay, hop, sink); }); printf("lds=%5d K1 f32 input read  : %.3f ms  %.2f TB/s (805 MB nominal)\n", lds, t, gb / t);
    }
    float t = timeit([&]{ stream_copy<<<2048, 256>>>(A, B, n4); }); printf("grid-stride float4 copy A->B: %.3f ms  %.2f TB/s\n", t, 2 * gb / t);
    return 0;
}
