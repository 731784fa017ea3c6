#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_h2(float2 v) { const __half2 h = __float22half2_rn(v); return __builtin_bit_cast(unsigned, h); }
__device__ __forceinline__ float2 unpack_h2(unsigned u) { return __half22float2(__builtin_bit_cast(__half2, u)); }
__global__ void k(unsigned* buf, float* out) {
    int t = threadIdx.x;
    float2 a = make_float2(1.0f + t, 100.0f + t), b = make_float2(-2.0f - t, -200.0f - t);
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 4096, 0x00020000);
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    u32x2 o; o.x = pack_h2(a); o.y = pack_h2(b);
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((__vector_size__(2 * sizeof(unsigned)))) unsigned, o), r, t * 8, 0, 0);
    __syncthreads();
    const f32x2 v = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, t * 8, 0, 0));
    float2 x = unpack_h2(__builtin_bit_cast(unsigned, v.x)), y = unpack_h2(__builtin_bit_cast(unsigned, v.y));
    out[t*4+0]=x.x; out[t*4+1]=x.y; out[t*4+2]=y.x; out[t*4+3]=y.y;
}
int main(){ unsigned* b; float* o; hipMalloc(&b,4096); hipMalloc(&o,64*16); k<<<1,64>>>(b,o); float h[256]; hipMemcpy(h,o,1024,hipMemcpyDeviceToHost); unsigned hb[8]; hipMemcpy(hb,b,32,hipMemcpyDeviceToHost);
 for(int t=0;t<3;++t) printf("t%d: %g %g %g %g\n",t,h[t*4],h[t*4+1],h[t*4+2],h[t*4+3]); printf("raw %08x %08x %08x %08x\n",hb[0],hb[1],hb[2],hb[3]); return 0; }
