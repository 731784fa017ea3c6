// tools/ntbench.hip -- cache-policy bits (sc0 / nt / sc1) on the access shapes of the three
// pipeline kernels, and the plain streaming copy MI355X_MICROARCH.md quotes (6.29 TB/s float4
// copy, 6.5-6.8 TB/s nt streams).  Measurement tool, not part of the library.
//   hipcc -O3 --offload-arch=gfx950 -o tools/ntbench tools/ntbench.hip
// aux of the raw buffer builtins on gfx940+: bit 0 = sc0, bit 1 = nt, bit 4 = sc1.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((__vector_size__(16)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
template <int AUX> __device__ __forceinline__ f32x4 ld4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, AUX));
}
template <int AUX> __device__ __forceinline__ void st4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, soff, AUX);
    asm volatile("s_nop 1" : : "v"(v));
}

// K2 shape: one contiguous 64 KB row per workgroup, 16 x 16 B per thread, in place or A -> B
template <int LA, int SA, bool INPLACE>
__global__ void __launch_bounds__(256, 2) row_copy(float4* a, float4* b) {
    extern __shared__ float4 lds[];
    const size_t base = (size_t)blockIdx.x * 4096;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(a + base, 65536), rd = make_rsrc((INPLACE ? a : b) + base, 65536);
    f32x4 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = ld4<LA>(rs, threadIdx.x * 16u, i * 4096);
#pragma unroll
    for (int i = 0; i < 16; ++i) { v[i].x += 1.0f; st4<SA>(rd, threadIdx.x * 16u, i * 4096, v[i]); }
}
// The same row with `spin` rounds of 64 dependent FMAs per thread between the loads and the
// stores (K2 issues ~3400 VALU instructions per wave: spin = 50), at a chosen LDS footprint,
// i.e. at 2, 3 or 4 workgroups per CU: what the in-place stream gives a kernel that computes.
__global__ void __launch_bounds__(256, 2) row_work(float4* a, int spin) {
    extern __shared__ float4 lds[];
    const size_t base = (size_t)blockIdx.x * 4096;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(a + base, 65536);
    f32x4 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = ld4<0>(rs, threadIdx.x * 16u, i * 4096);
    for (int k = 0; k < spin; ++k) {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = v[i] * 1.0001f + v[(i + 1) & 15];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) st4<0>(rs, threadIdx.x * 16u, i * 4096, v[i]);
}
// K3 / K1 shape: 32 columns x 256 rows of a [256][8192] float2 matrix per workgroup
template <int LA, int SA, int MODE>   // 0 read only, 1 write only
__global__ void __launch_bounds__(256, 3) col_rw(float4* m, float* sink) {
    const unsigned lin = blockIdx.x, xcd = lin & 7u, seq = lin >> 3;
    const unsigned slot = seq >> 5, half = (seq >> 4) & 1u, tl = seq & 15u;
    const unsigned tile = (half * 8u + xcd) * 16u + tl;
    const int t = threadIdx.x, hi = t >> 4, cp = t & 15;
    const __amdgpu_buffer_rsrc_t r = make_rsrc(m + (size_t)slot * (256 * 4096) + tile * 16, 256u * 65536u - tile * 256u);
    f32x4 v[16];
    if (MODE == 0) {
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = ld4<LA>(r, (unsigned)(hi * 65536 + cp * 16), k * 16 * 65536);
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) acc += v[k].x + v[k].y + v[k].z + v[k].w;
        if (acc == 123.456f) sink[0] = acc;
    } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) { v[k] = (f32x4){(float)k, (float)t, (float)lin, 1.f}; st4<SA>(r, (unsigned)(hi * 65536 + cp * 16), k * 16 * 65536, v[k]); }
    }
}
// plain streaming copy, UNROLL independent 16-byte accesses per thread per trip
template <int NT, int UNROLL>
__global__ void __launch_bounds__(256) stream_copy(const float4* __restrict__ s, float4* __restrict__ d, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
        f32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const f32x4* p = reinterpret_cast<const f32x4*>(s) + i + u * stride;
            v[u] = NT ? __builtin_nontemporal_load(p) : *p;
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            f32x4* q = reinterpret_cast<f32x4*>(d) + i + u * stride;
            if (NT) __builtin_nontemporal_store(v[u], q); else *q = v[u];
        }
    }
    // the rest of the buffer, one access per trip: every launch moves all n elements whatever the grid
    // (round 2's version stopped at the last full unrolled trip: with 16384 workgroups x 8 accesses a
    // third of the buffer was never touched and the rate printed from the whole buffer exceeded 8 TB/s)
    for (; i < n; i += stride) {
        const f32x4* p = reinterpret_cast<const f32x4*>(s) + i;
        const f32x4 v = NT ? __builtin_nontemporal_load(p) : *p;
        f32x4* q = reinterpret_cast<f32x4*>(d) + i;
        if (NT) __builtin_nontemporal_store(v, q); else *q = v;
    }
}
template <int NT, int UNROLL>
__global__ void __launch_bounds__(256) stream_read(const float4* __restrict__ s, float* sink, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    float acc = 0.f;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const f32x4* p = reinterpret_cast<const f32x4*>(s) + i + u * stride;
            const f32x4 v = NT ? __builtin_nontemporal_load(p) : *p;
            acc += v.x + v.y + v.z + v.w;
        }
    }
    for (; i < n; i += stride) {   // the rest of the buffer (see stream_copy)
        const f32x4* p = reinterpret_cast<const f32x4*>(s) + i;
        const f32x4 v = NT ? __builtin_nontemporal_load(p) : *p;
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123.456f) sink[0] = acc;
}

template <typename F> float timeit(F f, int reps = 60) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 60; ++i) f();
    hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < reps; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); hipEventDestroy(a); hipEventDestroy(b); return ms / reps;
}
#define ROW(LA, SA, IP) do { hipFuncSetAttribute((const void*)row_copy<LA, SA, IP>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536); \
    float t = timeit([&]{ row_copy<LA, SA, IP><<<nwg, 256, 65536>>>(A, B); }); \
    printf("row %s  load aux %2d store aux %2d : %.3f ms  %.2f TB/s\n", IP ? "in place" : "A -> B  ", LA, SA, t, 2 * gb / t); } while (0)
#define COL(LA, SA, M) do { float t = timeit([&]{ col_rw<LA, SA, M><<<nwg, 256, 32768>>>(A, sink); }); \
    printf("col %s aux %2d : %.3f ms  %.2f TB/s\n", M ? "write" : "read ", M ? SA : LA, t, gb / t); } while (0)

int main() {
    const int npairs = 48; const size_t n4 = (size_t)npairs * 256 * 4096;   // float4 count (805 MB)
    float4 *A, *B; float* sink;
    CK(hipMalloc(&A, n4 * 16)); CK(hipMalloc(&B, n4 * 16)); CK(hipMalloc(&sink, 16));
    CK(hipMemset(A, 0, n4 * 16)); CK(hipMemset(B, 0, n4 * 16));
    const double gb = n4 * 16 / 1e9;
    const int nwg = npairs * 256;
    // plain streaming copy / read: grid sizes x unroll x nt
    for (int g : {1024, 2048, 4096, 8192, 16384}) {
        float t;
        t = timeit([&]{ stream_copy<0, 1><<<g, 256>>>(A, B, n4); }); printf("copy  grid %5d u1 plain: %.3f ms %.2f TB/s\n", g, t, 2 * gb / t);
        t = timeit([&]{ stream_copy<0, 4><<<g, 256>>>(A, B, n4); }); printf("copy  grid %5d u4 plain: %.3f ms %.2f TB/s\n", g, t, 2 * gb / t);
        t = timeit([&]{ stream_copy<1, 4><<<g, 256>>>(A, B, n4); }); printf("copy  grid %5d u4 nt   : %.3f ms %.2f TB/s\n", g, t, 2 * gb / t);
        t = timeit([&]{ stream_copy<0, 8><<<g, 256>>>(A, B, n4); }); printf("copy  grid %5d u8 plain: %.3f ms %.2f TB/s\n", g, t, 2 * gb / t);
        t = timeit([&]{ stream_copy<1, 8><<<g, 256>>>(A, B, n4); }); printf("copy  grid %5d u8 nt   : %.3f ms %.2f TB/s\n", g, t, 2 * gb / t);
        t = timeit([&]{ stream_read<0, 8><<<g, 256>>>(A, sink, n4); }); printf("read  grid %5d u8 plain: %.3f ms %.2f TB/s\n", g, t, gb / t);
        t = timeit([&]{ stream_read<1, 8><<<g, 256>>>(A, sink, n4); }); printf("read  grid %5d u8 nt   : %.3f ms %.2f TB/s\n", g, t, gb / t);
    }
    // K2 shape
    ROW(0, 0, true); ROW(2, 0, true); ROW(0, 2, true); ROW(2, 2, true); ROW(16, 0, true); ROW(0, 16, true); ROW(18, 18, true); ROW(1, 0, true); ROW(3, 2, true);
    ROW(0, 0, false); ROW(2, 2, false); ROW(2, 0, false); ROW(0, 2, false);
    // K2 shape with compute, by occupancy (LDS footprint 64 / 52 / 40 / 32 KB = 2 / 3 / 4 / 5 workgroups per CU)
    hipFuncSetAttribute((const void*)row_work, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int spin : {0, 25, 50}) for (int lds : {65536, 53248, 40960, 32768}) {
        float t = timeit([&]{ row_work<<<nwg, 256, lds>>>(A, spin); });
        printf("row in place + %2d x 64 FMA, LDS %5d B : %.3f ms  %.2f TB/s\n", spin, lds, t, 2 * gb / t);
    }
    // K3 read / K1 write shapes
    COL(0, 0, 0); COL(2, 0, 0); COL(16, 0, 0); COL(18, 0, 0); COL(1, 0, 0);
    COL(0, 0, 1); COL(0, 2, 1); COL(0, 16, 1); COL(0, 18, 1);
    return 0;
}
