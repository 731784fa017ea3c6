// tools/pkbench.hip -- is packed f32 (v_pk_fma_f32) faster than scalar v_fma_f32 on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template<int PK> __global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b) {
    f2 x0={1.f+threadIdx.x,2.f}, x1={3.f,4.f}, x2={5.f,6.f}, x3={7.f,8.f}, x4={1.5f,2.5f}, x5={3.5f,4.5f}, x6={5.5f,6.5f}, x7={7.5f,8.5f};
    f2 A={a,a*0.5f}, B={b,b*0.25f};
    for (int i=0;i<iters;++i) {
        if (PK) {
            asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                         "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(x0),"+v"(x1),"+v"(x2),"+v"(x3),"+v"(x4),"+v"(x5),"+v"(x6),"+v"(x7) : "v"(A),"v"(B));
        } else {
            asm volatile("v_fma_f32 %0, %0, %16, %17\n v_fma_f32 %1, %1, %16, %17\n v_fma_f32 %2, %2, %16, %17\n v_fma_f32 %3, %3, %16, %17\n"
                         "v_fma_f32 %4, %4, %16, %17\n v_fma_f32 %5, %5, %16, %17\n v_fma_f32 %6, %6, %16, %17\n v_fma_f32 %7, %7, %16, %17\n"
                         "v_fma_f32 %8, %8, %16, %17\n v_fma_f32 %9, %9, %16, %17\n v_fma_f32 %10, %10, %16, %17\n v_fma_f32 %11, %11, %16, %17\n"
                         "v_fma_f32 %12, %12, %16, %17\n v_fma_f32 %13, %13, %16, %17\n v_fma_f32 %14, %14, %16, %17\n v_fma_f32 %15, %15, %16, %17\n"
                         : "+v"(x0.x),"+v"(x0.y),"+v"(x1.x),"+v"(x1.y),"+v"(x2.x),"+v"(x2.y),"+v"(x3.x),"+v"(x3.y),
                           "+v"(x4.x),"+v"(x4.y),"+v"(x5.x),"+v"(x5.y),"+v"(x6.x),"+v"(x6.y),"+v"(x7.x),"+v"(x7.y) : "v"(a),"v"(b));
        }
    }
    f2 s = x0+x1+x2+x3+x4+x5+x6+x7;
    out[blockIdx.x*256+threadIdx.x] = s.x+s.y;
}
int main(){
    float* o; hipMalloc(&o, 4096*256*4);
    hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters=4000;
    for (int waves : {1,2,4}) {   // waves per SIMD: blocks of 256 threads = 4 waves = 1 per SIMD
      for (int pk=0; pk<2; ++pk) {
        int blocks = 256*waves;
        if (pk) k<1><<<blocks,256>>>(o,10,0.999f,0.001f); else k<0><<<blocks,256>>>(o,10,0.999f,0.001f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        if (pk) k<1><<<blocks,256>>>(o,iters,0.999f,0.001f); else k<0><<<blocks,256>>>(o,iters,0.999f,0.001f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms,e0,e1);
        double flops = (double)blocks*256*iters*16*2;   // 16 fma lanes-ops per thread per iter either way
        printf("waves/SIMD=%d %s: %.3f ms  %.1f TFLOP/s\n", waves, pk?"v_pk_fma_f32":"v_fma_f32  ", ms, flops/ms/1e9);
      }
    }
    return 0;
}
