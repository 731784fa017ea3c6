// tools/layoutbench.hip -- memory-only timing of the three kernels' access shapes for two layouts
// of the work matrix (measurement tool):
//   A (current): row-major [pair][row 256][col 8192] -- K2 streams 64 KB rows, K1/K3 touch 256 rows x 256 B
//   B (hybrid) : [pair][rowblock 16][tile 256][16 rows][32 cols] -- K1/K3 touch 16 chunks of 4 KB,
//                K2 touches 256 pieces of 256 B inside one 1 MB region
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

// float4 index of (row, col-pair cp16 within tile) for both layouts; tile = 32 cols = 16 float4
template<int LAYOUT> __device__ __forceinline__ size_t idx4(unsigned row, unsigned tile, unsigned q) {
    if (LAYOUT == 0) return (size_t)row * 4096 + tile * 16 + q;
    return ((size_t)(row >> 4) * 256 + tile) * 256 + (row & 15) * 16 + q;   // 256 float4 = 4 KB per (rowblock, tile)
}
template<int LAYOUT, int MODE>   // MODE 0: K2 in place (row), 1: K1 write (column tile), 2: K3 read (column tile)
__global__ void __launch_bounds__(256) k(float4* m, float* sink, unsigned npairs, unsigned R = 1) {
    const int t = threadIdx.x;
    float4 v[16];
    if (MODE == 0) {
        // R = 0: plain (rows of one pair in launch order); R >= 1: every XCD owns blocks of R adjacent rows and
        // walks them pair by pair (R = 1 is the library's mapping)
        const unsigned lin = blockIdx.x, xcd = lin & 7u, seq = lin >> 3;
        unsigned row, slot;
        if (R == 0) { slot = lin >> 8; row = lin & 255u; }
        else { const unsigned rin = seq % R, rest = seq / R; slot = rest % npairs; row = ((rest / npairs) * 8u + xcd) * R + rin; }
        float4* p = m + (size_t)slot * (256 * 4096);
#pragma unroll
        for (int a = 0; a < 16; ++a) { const unsigned c4 = a * 256 + t; v[a] = p[idx4<LAYOUT>(row, c4 >> 4, c4 & 15)]; }
#pragma unroll
        for (int a = 0; a < 16; ++a) { const unsigned c4 = a * 256 + t; v[a].x += 1.0f; p[idx4<LAYOUT>(row, c4 >> 4, c4 & 15)] = v[a]; }
    } else {
        const unsigned slot = blockIdx.x >> 8, tile = blockIdx.x & 255u;
        const unsigned hi = t >> 4, q = t & 15;
        float4* p = m + (size_t)slot * (256 * 4096);
        if (MODE == 1) {
#pragma unroll
            for (int b = 0; b < 16; ++b) p[idx4<LAYOUT>(hi + 16 * b, tile, q)] = make_float4(b, t, slot, 1.f);
        } else {
            float acc = 0.f;
#pragma unroll
            for (int b = 0; b < 16; ++b) v[b] = p[idx4<LAYOUT>(hi + 16 * b, tile, q)];
#pragma unroll
            for (int b = 0; b < 16; ++b) acc += v[b].x + v[b].y + v[b].z + v[b].w;
            if (acc == 123.456f) sink[0] = acc;
        }
    }
}
template<typename F> float timeit(F f) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 100; ++i) f();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a); for (int i = 0; i < 100; ++i) f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); return ms / 100;
}
int main() {
    const unsigned npairs = 48; const size_t n4 = (size_t)npairs * 256 * 4096;
    float4* A; float* sink;
    CK(hipMalloc(&A, n4 * 16)); CK(hipMalloc(&sink, 16)); CK(hipMemset(A, 0, n4 * 16));
    const int nwg = npairs * 256; const double gb = n4 * 16 / 1e9;
    float t;
    for (unsigned R : {0u, 1u, 2u, 4u, 8u, 16u, 32u}) {
        t = timeit([&]{ k<0,0><<<nwg,256>>>(A, sink, npairs, R); }); printf("layout A  K2 in place R=%2u : %.3f ms %.2f TB/s\n", R, t, 2*gb/t);
    }
    t = timeit([&]{ k<1,0><<<nwg,256>>>(A, sink, npairs); }); printf("layout B  K2 in place : %.3f ms %.2f TB/s\n", t, 2*gb/t);
    t = timeit([&]{ k<0,1><<<nwg,256>>>(A, sink, npairs); }); printf("layout A  K1 write    : %.3f ms %.2f TB/s\n", t, gb/t);
    t = timeit([&]{ k<1,1><<<nwg,256>>>(A, sink, npairs); }); printf("layout B  K1 write    : %.3f ms %.2f TB/s\n", t, gb/t);
    t = timeit([&]{ k<0,2><<<nwg,256>>>(A, sink, npairs); }); printf("layout A  K3 read     : %.3f ms %.2f TB/s\n", t, gb/t);
    t = timeit([&]{ k<1,2><<<nwg,256>>>(A, sink, npairs); }); printf("layout B  K3 read     : %.3f ms %.2f TB/s\n", t, gb/t);
    return 0;
}
