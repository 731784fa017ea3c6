// am_fft.hip -- the correlation pipeline of libaudiomatch_amd.so (gfx950).
//
// Replaces MyConvolve::correlate / LibConvolve::correlate of the reference
// (audio_matcher.rs:297-310, 414-457): rfft(within) * conj(rfft(needle)) ->
// irfft -> scale -> crop, re-designed as overlap-save over a power-of-two
// complex transform that carries TWO real blocks at once (re = block 2g,
// im = block 2g+1; correlation with a real needle is real-linear, so
// IFFT(FFT(a + ib) * conj(H)) = corr(a,h) + i corr(b,h) and no real/complex
// untangling pass exists).
//
// The N-point transform is factored N = N1 * N2 (n = n1*N2 + n2,
// k = k1 + N1*k2) into three kernels, each of which keeps its whole sub-problem
// in LDS and touches HBM exactly once for reading and once for writing:
//
//   K1 k1_cols_fwd : 32 adjacent columns x N1 rows per workgroup; PCM->f32 load
//                    with virtual zero padding, length-N1 column FFTs.
//   K2 k2_rows     : one contiguous row (N2 points) per workgroup; twiddle
//                    W_N^(n2*k1), forward row FFT, multiply by the needle
//                    spectrum conj(H)/N, inverse row FFT, conjugate twiddle.
//                    The spectrum never exists in HBM.
//   K3 k3_cols_inv : inverse column FFTs, needle-energy scaling, real part ->
//                    scores of block 2g, imaginary part -> scores of block 2g+1.
//
// Forward transforms are decimation-in-frequency (natural in, bit-reversed
// out), inverse ones decimation-in-time (bit-reversed in, natural out), so no
// reordering pass exists anywhere: the needle spectrum is produced by the same
// K1/K2 code and therefore lives in the same permuted layout.
#include "am_kernels.h"

namespace am {

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
// a * conj(b)
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) {
    return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
__device__ __forceinline__ float2 mul_neg_i(float2 a) { return make_float2(a.y, -a.x); }
__device__ __forceinline__ float2 mul_pos_i(float2 a) { return make_float2(-a.y, a.x); }

// Forward DIF transform of length 2^logL along the slow axis of
// s[i * 2^BL + c] (2^BL independent columns c).  tw[k] = W_L^k, k < L/2.
// Each radix-4 step is two fused radix-2 DIF stages, so the result is in
// plain bit-reversed order.
template <int BL>
__device__ void lds_fft_fwd(float2* s, int logL, const float2* __restrict__ tw, int tid, int nthr) {
    const int L = 1 << logL;
    int lm = logL;
    while (lm >= 2) {
        const int q = 1 << (lm - 2);
        const int tws = logL - lm;
        const int st = q << BL;
        const int total = (L >> 2) << BL;
        for (int t = tid; t < total; t += nthr) {
            const int c = t & ((1 << BL) - 1);
            const int bf = t >> BL;
            const int j = bf & (q - 1);
            const int g = bf >> (lm - 2);
            float2* p = s + (((g << lm) + j) << BL) + c;
            const float2 x0 = p[0], x1 = p[st], x2 = p[2 * st], x3 = p[3 * st];
            const float2 w1 = tw[j << tws];
            const float2 w2 = tw[(2 * j) << tws];
            const float2 t0 = cadd(x0, x2), t1 = csub(x0, x2);
            const float2 t2 = cadd(x1, x3), t3 = mul_neg_i(csub(x1, x3));
            p[0] = cadd(t0, t2);
            p[st] = cmul(csub(t0, t2), w2);
            p[2 * st] = cmul(cadd(t1, t3), w1);
            p[3 * st] = cmul(cmul(csub(t1, t3), w1), w2);
        }
        __syncthreads();
        lm -= 2;
    }
    if (lm == 1) {
        const int total = (L >> 1) << BL;
        for (int t = tid; t < total; t += nthr) {
            const int c = t & ((1 << BL) - 1);
            const int bf = t >> BL;
            float2* p = s + ((bf * 2) << BL) + c;
            const float2 a = p[0], b = p[1 << BL];
            p[0] = cadd(a, b);
            p[1 << BL] = csub(a, b);
        }
        __syncthreads();
    }
}

// Inverse DIT transform (bit-reversed in, natural out, unnormalised), the exact
// mirror of lds_fft_fwd.
template <int BL>
__device__ void lds_fft_inv(float2* s, int logL, const float2* __restrict__ tw, int tid, int nthr) {
    const int L = 1 << logL;
    int lm = 2;
    if (logL & 1) {
        const int total = (L >> 1) << BL;
        for (int t = tid; t < total; t += nthr) {
            const int c = t & ((1 << BL) - 1);
            const int bf = t >> BL;
            float2* p = s + ((bf * 2) << BL) + c;
            const float2 a = p[0], b = p[1 << BL];
            p[0] = cadd(a, b);
            p[1 << BL] = csub(a, b);
        }
        __syncthreads();
        lm = 3;
    }
    for (; lm <= logL; lm += 2) {
        const int q = 1 << (lm - 2);
        const int tws = logL - lm;
        const int st = q << BL;
        const int total = (L >> 2) << BL;
        for (int t = tid; t < total; t += nthr) {
            const int c = t & ((1 << BL) - 1);
            const int bf = t >> BL;
            const int j = bf & (q - 1);
            const int g = bf >> (lm - 2);
            float2* p = s + (((g << lm) + j) << BL) + c;
            const float2 x0 = p[0], x1 = p[st], x2 = p[2 * st], x3 = p[3 * st];
            const float2 wB = tw[j << tws];        // W_m^j      (conjugated below)
            const float2 wA = tw[(2 * j) << tws];  // W_(m/2)^j
            const float2 a1 = cmulc(x1, wA), a3 = cmulc(x3, wA);
            const float2 u0 = cadd(x0, a1), u1 = csub(x0, a1);
            const float2 u2 = cadd(x2, a3), u3 = csub(x2, a3);
            const float2 b2 = cmulc(u2, wB);
            const float2 b3 = mul_pos_i(cmulc(u3, wB));
            p[0] = cadd(u0, b2);
            p[2 * st] = csub(u0, b2);
            p[st] = cadd(u1, b3);
            p[3 * st] = csub(u1, b3);
        }
        __syncthreads();
    }
}

__device__ __forceinline__ float load_padded(const float* __restrict__ src, long long i, long long len) {
    return (i >= 0 && i < len) ? src[i] : 0.0f;
}

// ---------------------------------------------------------------------------
// K1: PCM/f32 window load (pad(), audio_matcher.rs:232-235, 422) + column FFTs.
template <int BL>
__global__ void __launch_bounds__(kFftThreads)
k1_cols_fwd(Job job, float2* __restrict__ work, PlanDev pl) {
    extern __shared__ float2 s[];
    const int N1 = 1 << pl.logN1, N2 = 1 << pl.logN2;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int n2_0 = blockIdx.x << BL;
    const int pair = job.first_pair + blockIdx.y;
    const long long blkA = 2ll * pair, blkB = blkA + 1;
    const bool validB = blkB < job.nblocks;
    const long long baseA = blkA * job.hop - job.lead;
    const long long baseB = blkB * job.hop - job.lead;
    const int total = N1 << BL;
    for (int idx = tid; idx < total; idx += nthr) {
        const int r = idx >> BL, c = idx & ((1 << BL) - 1);
        const long long n = (long long)r * N2 + n2_0 + c;
        const float a = load_padded(job.src, baseA + n, job.src_len);
        const float b = validB ? load_padded(job.src, baseB + n, job.src_len) : 0.0f;
        s[idx] = make_float2(a, b);
    }
    __syncthreads();
    lds_fft_fwd<BL>(s, pl.logN1, pl.tw1, tid, nthr);
    float2* out = work + ((size_t)blockIdx.y << pl.logN) + n2_0;
    for (int idx = tid; idx < total; idx += nthr) {
        const int p = idx >> BL, c = idx & ((1 << BL) - 1);
        out[(size_t)p * N2 + c] = s[idx];
    }
}

// ---------------------------------------------------------------------------
// K2: one row.  SPECTRUM = true writes conj(FFT)/N of the needle block instead
// of correlating (fft_b + the conj of pairwise_mult_in_place + the 1/len of
// audio_matcher.rs:430-442 folded into one table).
template <bool SPECTRUM>
__global__ void __launch_bounds__(kFftThreads)
k2_rows(float2* __restrict__ work, const float2* __restrict__ hc, float2* __restrict__ hc_out, PlanDev pl) {
    extern __shared__ float2 s[];
    const int N2 = 1 << pl.logN2;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int p = blockIdx.x;
    const unsigned k1 = pl.logN1 ? (__brev((unsigned)p) >> (32 - pl.logN1)) : 0u;
    float2* row = work + ((size_t)blockIdx.y << pl.logN) + (size_t)p * N2;
    const unsigned maskN = (1u << pl.logN) - 1u, maskLo = (1u << pl.logLo) - 1u;
    for (int n2 = tid; n2 < N2; n2 += nthr) {
        const unsigned m = ((unsigned)n2 * k1) & maskN;
        const float2 w = cmul(pl.twhi[m >> pl.logLo], pl.twlo[m & maskLo]);
        s[n2] = cmul(row[n2], w);
    }
    __syncthreads();
    lds_fft_fwd<0>(s, pl.logN2, pl.tw2, tid, nthr);
    const size_t hoff = (size_t)p * N2;
    if (SPECTRUM) {
        const float invN = 1.0f / (float)(1u << pl.logN);
        for (int q = tid; q < N2; q += nthr) {
            const float2 v = s[q];
            hc_out[hoff + q] = make_float2(v.x * invN, -v.y * invN);
        }
        return;
    }
    for (int q = tid; q < N2; q += nthr) s[q] = cmul(s[q], hc[hoff + q]);
    __syncthreads();
    lds_fft_inv<0>(s, pl.logN2, pl.tw2, tid, nthr);
    for (int n2 = tid; n2 < N2; n2 += nthr) {
        const unsigned m = ((unsigned)n2 * k1) & maskN;
        const float2 w = cmul(pl.twhi[m >> pl.logLo], pl.twlo[m & maskLo]);
        row[n2] = cmulc(s[n2], w);
    }
}

// ---------------------------------------------------------------------------
// K3: inverse column FFTs, scaling (scale_slice, audio_matcher.rs:246-252,
// 306-308) and the crop to the block's valid lags (centered(), :460-464).
template <int BL>
__global__ void __launch_bounds__(kFftThreads)
k3_cols_inv(Job job, const float2* __restrict__ work, PlanDev pl, float out_scale) {
    extern __shared__ float2 s[];
    const int N1 = 1 << pl.logN1, N2 = 1 << pl.logN2;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int n2_0 = blockIdx.x << BL;
    const int pair = job.first_pair + blockIdx.y;
    const long long blkA = 2ll * pair, blkB = blkA + 1;
    const bool validB = blkB < job.nblocks;
    const int total = N1 << BL;
    const float2* in = work + ((size_t)blockIdx.y << pl.logN) + n2_0;
    for (int idx = tid; idx < total; idx += nthr) {
        const int p = idx >> BL, c = idx & ((1 << BL) - 1);
        s[idx] = in[(size_t)p * N2 + c];
    }
    __syncthreads();
    lds_fft_inv<BL>(s, pl.logN1, pl.tw1, tid, nthr);
    const long long outA = blkA * job.hop, outB = blkB * job.hop;
    for (int idx = tid; idx < total; idx += nthr) {
        const int r = idx >> BL, c = idx & ((1 << BL) - 1);
        const long long n = (long long)r * N2 + n2_0 + c;
        if (n >= job.hop) continue;
        const float2 v = s[idx];
        if (outA + n < job.out_count) job.dst[outA + n] = v.x * out_scale;
        if (validB && outB + n < job.out_count) job.dst[outB + n] = v.y * out_scale;
    }
}

// ---------------------------------------------------------------------------
static constexpr int kMaxLds = 160 * 1024;

hipError_t fft_kernels_init() {
    hipError_t e;
    e = hipFuncSetAttribute((const void*)k1_cols_fwd<kColsLog>, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)k3_cols_inv<kColsLog>, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)k2_rows<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)k2_rows<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds);
    return e;
}

hipError_t launch_k1(hipStream_t st, const Job& job, int npairs, float2* work, const PlanDev& pl) {
    const dim3 grid((1u << pl.logN2) >> kColsLog, npairs);
    const size_t lds = (sizeof(float2) << pl.logN1) << kColsLog;
    hipLaunchKernelGGL(k1_cols_fwd<kColsLog>, grid, dim3(kFftThreads), lds, st, job, work, pl);
    return hipGetLastError();
}

hipError_t launch_k2(hipStream_t st, int npairs, float2* work, const float2* hc, const PlanDev& pl) {
    const dim3 grid(1u << pl.logN1, npairs);
    const size_t lds = sizeof(float2) << pl.logN2;
    hipLaunchKernelGGL(k2_rows<false>, grid, dim3(kFftThreads), lds, st, work, hc, (float2*)nullptr, pl);
    return hipGetLastError();
}

hipError_t launch_k2_spectrum(hipStream_t st, float2* work, float2* hc_out, const PlanDev& pl) {
    const dim3 grid(1u << pl.logN1, 1);
    const size_t lds = sizeof(float2) << pl.logN2;
    hipLaunchKernelGGL(k2_rows<true>, grid, dim3(kFftThreads), lds, st, work, (const float2*)nullptr, hc_out, pl);
    return hipGetLastError();
}

hipError_t launch_k3(hipStream_t st, const Job& job, int npairs, const float2* work,
                     const PlanDev& pl, float out_scale) {
    const dim3 grid((1u << pl.logN2) >> kColsLog, npairs);
    const size_t lds = (sizeof(float2) << pl.logN1) << kColsLog;
    hipLaunchKernelGGL(k3_cols_inv<kColsLog>, grid, dim3(kFftThreads), lds, st, job, work, pl, out_scale);
    return hipGetLastError();
}

}  // namespace am
