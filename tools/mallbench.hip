// tools/mallbench.hip -- does the Infinity Cache keep freshly written data?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) wr(float4* d, size_t n, float s) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) d[i] = make_float4(s, i, 1.f, 2.f);
}
__global__ void __launch_bounds__(256) rd(const float4* d, size_t n, float* sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { float4 v = d[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) sink[0] = acc;
}
__global__ void __launch_bounds__(256) rmw(float4* d, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { float4 v = d[i]; v.x += 1.f; d[i] = v; }
}
int main() {
    float4* buf; float* sink; hipMalloc(&buf, (size_t)2 << 30); hipMalloc(&sink, 16);
    hipMemset(buf, 0, (size_t)2 << 30);
    hipEvent_t e[4]; for (auto& x : e) hipEventCreate(&x);
    for (size_t mb : {16, 32, 64, 96, 128, 192, 256, 384, 512, 1024}) {
        size_t n = mb * 1024 * 1024 / 16;
        float tw = 0, tr = 0, tm = 0; const int reps = 20;
        for (int r = 0; r < reps + 2; ++r) {
            hipEventRecord(e[0]); wr<<<2048, 256>>>(buf, n, (float)r); hipEventRecord(e[1]);
            rd<<<2048, 256>>>(buf, n, sink); hipEventRecord(e[2]);
            rmw<<<2048, 256>>>(buf, n); hipEventRecord(e[3]);
            hipEventSynchronize(e[3]);
            float a, b, c; hipEventElapsedTime(&a, e[0], e[1]); hipEventElapsedTime(&b, e[1], e[2]); hipEventElapsedTime(&c, e[2], e[3]);
            if (r >= 2) { tw += a; tr += b; tm += c; }
        }
        double gb = mb * 1.048576e-3;
        printf("%5zu MB: write %.2f TB/s | read-after-write %.2f TB/s | rmw in place %.2f TB/s (rd+wr)\n", mb, gb / (tw / reps), gb / (tr / reps), 2 * gb / (tm / reps));
    }
    return 0;
}
