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
// on chip and touches HBM exactly once for reading and once for writing:
//
//   K1 k1_cols_fwd : 32 adjacent columns x N1 rows per workgroup; f32 load with
//                    virtual zero padding, length-N1 column FFTs, twiddle
//                    W_N^(n2*k1).
//   K2 k2_rows     : one contiguous row (N2 points) per workgroup; forward row
//                    FFT, multiply by the needle spectrum conj(H)/N, inverse
//                    row FFT.  The spectrum never exists in HBM.
//   K3 k3_cols_inv : conjugate twiddle, inverse column FFTs, needle-energy
//                    scaling, real part -> scores of block 2g, imaginary part
//                    -> scores of block 2g+1, plus a (min,max) summary per 32
//                    scores for the peak pick.
//
// Four families of kernels exist (all with N2 = 8192-point rows except *_gen):
//   *_c512 : K1/K3 for N = 2^22 = 512 x 8192 (needles from 3.2 s up, the headline): 512-thread column
//            kernels, 512-point column transform as 16 x 32.
//   *_r16  : N1 = 256 (N = 2^21; short needles).  Radix-16/32 butterflies held in VGPRs, LDS used
//            only for the exchanges between passes (conflict-free accesses, XOR-swizzled rows), 16-byte
//            coalesced HBM accesses.
//   *_c1024: K1/K3 for N = 2^23 = 1024 x 8192 (needles above 34 s; needle partitioning above 2^22 samples).
//   *_gen  : any N = 2^10 .. 2^23 (small inputs, forced plans): in-LDS radix-4 passes.
// K2 of the register plans is k2_rows_r16_planes (the row crosses LDS one 32 KB plane at a time: four
// workgroups per CU -- the row kernel is bound by VALU issue), k2_rows_r16_group_planes
// for several needles, k2_rows_h16 with packed-f16 butterflies.  Twiddle powers come from TWO table entries
// per pass boundary (w and w^4, twiddle_apply / twiddle_chain below): a power e of one rounded entry carries
// e times its rounding error.
// The needle spectrum is produced by the same K1/K2 code of the same flavour
// and therefore always lives in the layout the multiply expects.
#include "am_kernels.h"

#include <float.h>
#include <stdlib.h>
#include <algorithm>
#include <atomic>
#include <type_traits>
#include <hip/hip_fp16.h>

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

// W_N^m through the two-level table of the plan (m already reduced mod N)
__device__ __forceinline__ float2 tw_big(const PlanDev& pl, unsigned m) {
    return cmul(pl.twhi[m >> pl.logLo], pl.twlo[m & ((1u << pl.logLo) - 1u)]);
}

// W_N^m and W_N^(4m) through the tables that hold every entry's fourth power beside it: the same two fetches
__device__ __forceinline__ void tw_big_pair(const PlanDev& pl, unsigned m, float2& w, float2& w4) {
    const float4 h = pl.twhi4[m >> pl.logLo], l = pl.twlo4[m & ((1u << pl.logLo) - 1u)];
    w = cmul(make_float2(h.x, h.y), make_float2(l.x, l.y));
    w4 = cmul(make_float2(h.z, h.w), make_float2(l.z, l.w));
}

// ===========================================================================
// Register-resident radix-R DIF butterflies (R <= 32), natural order in,
// bit-reversed order out: X[k] ends up in x[brev(k)].
// ===========================================================================
__device__ constexpr float kCos32[16] = {
    1.0f, 0.9807852804032304f, 0.9238795325112867f, 0.8314696123025452f, 0.7071067811865476f,
    0.5555702330196023f, 0.38268343236508984f, 0.19509032201612833f, 0.0f, -0.1950903220161282f,
    -0.3826834323650897f, -0.555570233019602f, -0.7071067811865475f, -0.8314696123025453f,
    -0.9238795325112867f, -0.9807852804032304f};
__device__ constexpr float kSin32[16] = {
    0.0f, 0.19509032201612825f, 0.3826834323650898f, 0.5555702330196022f, 0.7071067811865475f,
    0.8314696123025452f, 0.9238795325112867f, 0.9807852804032304f, 1.0f, 0.9807852804032304f,
    0.9238795325112867f, 0.8314696123025455f, 0.7071067811865476f, 0.5555702330196022f,
    0.3826834323650899f, 0.1950903220161286f};

template <int R>
__host__ __device__ constexpr int brev(int i) {
    int r = 0;
    for (int b = 1; b < R; b <<= 1) { r = (r << 1) | (i & 1); i >>= 1; }
    return r;
}

// d * W_32^idx (forward) or d * conj(W_32^idx) (inverse); idx is a constant
// after unrolling, so the trivial cases fold away.
template <bool INV>
__device__ __forceinline__ float2 mul_w32(float2 d, int idx) {
    if (idx == 0) return d;
    if (idx == 8) return INV ? mul_pos_i(d) : mul_neg_i(d);
    const float c = kCos32[idx], s = kSin32[idx];
    // forward twiddle = (c, -s); inverse = (c, +s)
    if (INV) return make_float2(d.x * c - d.y * s, d.x * s + d.y * c);
    return make_float2(d.x * c + d.y * s, d.y * c - d.x * s);
}

// Half-precision complex arithmetic (option half_pipeline = 2: K2's butterflies in packed f16).
// One complex point is one 32-bit register; a complex add is one v_pk_add_f16, a complex
// multiply two packed instructions (the swap and the signs ride on op_sel / neg modifiers).
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ h2 cadd(h2 a, h2 b) { return a + b; }
__device__ __forceinline__ h2 csub(h2 a, h2 b) { return a - b; }
__device__ __forceinline__ h2 cmul(h2 a, h2 b) {
    const h2 t = a * b.xx;
    return __builtin_elementwise_fma(a.yx, (h2){-b.y, b.y}, t);
}
__device__ __forceinline__ h2 mul_neg_i(h2 a) { return (h2){a.y, -a.x}; }
__device__ __forceinline__ h2 mul_pos_i(h2 a) { return (h2){-a.y, a.x}; }
__device__ __forceinline__ h2 to_h2(float2 v) { return (h2){(_Float16)v.x, (_Float16)v.y}; }
__device__ __forceinline__ float2 to_f2(h2 v) { return make_float2((float)v.x, (float)v.y); }
__device__ __forceinline__ unsigned h2_bits(h2 v) { return __builtin_bit_cast(unsigned, v); }
__device__ __forceinline__ h2 bits_h2(unsigned u) { return __builtin_bit_cast(h2, u); }
template <bool INV>
__device__ __forceinline__ h2 mul_w32(h2 d, int idx) {
    if (idx == 0) return d;
    if (idx == 8) return INV ? mul_pos_i(d) : mul_neg_i(d);
    const _Float16 c = (_Float16)kCos32[idx], s = (_Float16)kSin32[idx];
    const h2 t = d * (h2){c, c};
    if (INV) return __builtin_elementwise_fma(d.yx, (h2){-s, s}, t);
    return __builtin_elementwise_fma(d.yx, (h2){s, -s}, t);
}

// Packed single precision: a complex point as one 64-bit register pair, so that a complex add is
// one v_pk_add_f32 and a complex multiply one v_pk_mul_f32 + one v_pk_fma_f32 (the swap and the
// signs ride on op_sel / neg modifiers) -- half the VALU instructions of the scalar form.  Measured,
// not adopted (AM_K3_PK): a packed instruction takes about 1.75x the issue time of a scalar one.
typedef float p2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ p2 cadd(p2 a, p2 b) { return a + b; }
__device__ __forceinline__ p2 csub(p2 a, p2 b) { return a - b; }
__device__ __forceinline__ p2 cmul(p2 a, p2 b) {
    const p2 t = a * b.xx;
    return __builtin_elementwise_fma(a.yx, (p2){-b.y, b.y}, t);
}
__device__ __forceinline__ p2 mul_neg_i(p2 a) { return (p2){a.y, -a.x}; }
__device__ __forceinline__ p2 mul_pos_i(p2 a) { return (p2){-a.y, a.x}; }
__device__ __forceinline__ p2 to_p2(float2 v) { return (p2){v.x, v.y}; }
// u + sgn * v (the radix-2 stage of the 512-point column transform)
__device__ __forceinline__ p2 add_signed(float2 u, float2 v, float sgn, p2) {
    return __builtin_elementwise_fma(to_p2(v), (p2){sgn, sgn}, to_p2(u));
}
__device__ __forceinline__ float2 add_signed(float2 u, float2 v, float sgn, float2) {
    return make_float2(fmaf(sgn, v.x, u.x), fmaf(sgn, v.y, u.y));
}
template <bool INV>
__device__ __forceinline__ p2 mul_w32(p2 d, int idx) {
    if (idx == 0) return d;
    if (idx == 8) return INV ? mul_pos_i(d) : mul_neg_i(d);
    const float c = kCos32[idx], s = kSin32[idx];
    const p2 t = d * (p2){c, c};
    if (INV) return __builtin_elementwise_fma(d.yx, (p2){-s, s}, t);
    return __builtin_elementwise_fma(d.yx, (p2){s, -s}, t);
}

template <int R, bool INV, typename T>
__device__ __forceinline__ void dif(T* x) {
    if constexpr (R >= 2) {
#pragma unroll
        for (int i = 0; i < R / 2; ++i) {
            const T a = x[i], b = x[i + R / 2];
            x[i] = cadd(a, b);
            x[i + R / 2] = mul_w32<INV>(csub(a, b), i * (32 / R));
        }
        dif<R / 2, INV>(x);
        dif<R / 2, INV>(x + R / 2);
    }
}

// x[brev(e)] *= w^e (or conj(w)^e), e = 1..R-1.  Powers come from balanced
// products pw[e] = pw[ceil(e/2)] * pw[floor(e/2)] (depth log2 R, so the table
// entry's rounding error is amplified at most R times); each power is applied
// as soon as it exists so that only pw[1 .. R/2] stay live.
template <int R, bool CONJ, bool BREV>
__device__ __forceinline__ void twiddle_apply(float2* x, float2 w) {
    float2 pw[R / 2 + 1];
    if (CONJ) w.y = -w.y;
    pw[1] = w;
    x[BREV ? brev<R>(1) : 1] = cmul(x[BREV ? brev<R>(1) : 1], w);
#pragma unroll
    for (int e = 2; e < R; ++e) {
        const float2 v = cmul(pw[(e + 1) / 2], pw[e / 2]);
        if (e <= R / 2) pw[e] = v;
        x[BREV ? brev<R>(e) : e] = cmul(x[BREV ? brev<R>(e) : e], v);
    }
}
// the same on half-precision points: the powers are formed in f32 (a chain of f16 products would
// carry several roundings into every twiddle) and rounded once
template <int R, bool CONJ, bool BREV>
__device__ __forceinline__ void twiddle_apply(h2* x, float2 w) {
    float2 pw[R / 2 + 1];
    if (CONJ) w.y = -w.y;
    pw[1] = w;
    x[BREV ? brev<R>(1) : 1] = cmul(x[BREV ? brev<R>(1) : 1], to_h2(w));
#pragma unroll
    for (int e = 2; e < R; ++e) {
        const float2 v = cmul(pw[(e + 1) / 2], pw[e / 2]);
        if (e <= R / 2) pw[e] = v;
        x[BREV ? brev<R>(e) : e] = cmul(x[BREV ? brev<R>(e) : e], to_h2(v));
    }
}
// packed single precision: powers and products in packed form
template <int R, bool CONJ, bool BREV>
__device__ __forceinline__ void twiddle_apply(p2* x, float2 w) {
    p2 pw[R / 2 + 1];
    if (CONJ) w.y = -w.y;
    pw[1] = to_p2(w);
    x[BREV ? brev<R>(1) : 1] = cmul(x[BREV ? brev<R>(1) : 1], pw[1]);
#pragma unroll
    for (int e = 2; e < R; ++e) {
        const p2 v = cmul(pw[(e + 1) / 2], pw[e / 2]);
        if (e <= R / 2) pw[e] = v;
        x[BREV ? brev<R>(e) : e] = cmul(x[BREV ? brev<R>(e) : e], v);
    }
}
template <int R, bool CONJ, bool BREV>
__device__ __forceinline__ void twiddle_chain(p2* x, float2 base, float2 step) {
    if (CONJ) { base.y = -base.y; step.y = -step.y; }
    p2 c = to_p2(base);
    const p2 st = to_p2(step);
    x[0] = cmul(x[0], c);
#pragma unroll
    for (int e = 1; e < R; ++e) {
        c = cmul(c, st);
        x[BREV ? brev<R>(e) : e] = cmul(x[BREV ? brev<R>(e) : e], c);
    }
}
// two half-precision columns that take the same twiddles (the column passes: the pass twiddle does
// not depend on the column)
template <int R, bool CONJ, bool BREV>
__device__ __forceinline__ void twiddle_apply2(h2* x0, h2* x1, float2 w) {
    float2 pw[R / 2 + 1];
    if (CONJ) w.y = -w.y;
    pw[1] = w;
    {
        const h2 wh = to_h2(w);
        x0[BREV ? brev<R>(1) : 1] = cmul(x0[BREV ? brev<R>(1) : 1], wh);
        x1[BREV ? brev<R>(1) : 1] = cmul(x1[BREV ? brev<R>(1) : 1], wh);
    }
#pragma unroll
    for (int e = 2; e < R; ++e) {
        const float2 v = cmul(pw[(e + 1) / 2], pw[e / 2]);
        if (e <= R / 2) pw[e] = v;
        const h2 vh = to_h2(v);
        x0[BREV ? brev<R>(e) : e] = cmul(x0[BREV ? brev<R>(e) : e], vh);
        x1[BREV ? brev<R>(e) : e] = cmul(x1[BREV ? brev<R>(e) : e], vh);
    }
}
template <int R, bool CONJ, bool BREV>
__device__ __forceinline__ void twiddle_chain(h2* x, float2 base, float2 step) {
    if (CONJ) { base.y = -base.y; step.y = -step.y; }
    float2 c = base;
    x[0] = cmul(x[0], to_h2(c));
#pragma unroll
    for (int e = 1; e < R; ++e) {
        c = cmul(c, step);
        x[BREV ? brev<R>(e) : e] = cmul(x[BREV ? brev<R>(e) : e], to_h2(c));
    }
}
// x[idx(e)] *= base * step^e (or the conjugates), e = 0..R-1, as one running
// product c[e] = c[e-1] * step: R-1 products instead of a separate base pass.
template <int R, bool CONJ, bool BREV>
__device__ __forceinline__ void twiddle_chain(float2* x, float2 base, float2 step) {
    if (CONJ) { base.y = -base.y; step.y = -step.y; }
    float2 c = base;
    x[0] = cmul(x[0], c);
#pragma unroll
    for (int e = 1; e < R; ++e) {
        c = cmul(c, step);
        x[BREV ? brev<R>(e) : e] = cmul(x[BREV ? brev<R>(e) : e], c);
    }
}
template <int R, bool CONJ, typename T>
__device__ __forceinline__ void twiddle_brev(T* x, float2 w) { twiddle_apply<R, CONJ, true>(x, w); }
template <int R, bool CONJ, typename T>
__device__ __forceinline__ void twiddle_nat(T* x, float2 w) { twiddle_apply<R, CONJ, false>(x, w); }

// The same from TWO table entries, w and w4 = w^4: w^e = (w^4)^(e >> 2) * w^(e & 3).  A table entry is
// rounded to f32 (0.3 eps rms per component) and a power e carries e times that error, on top of the
// products' own roundings: from w alone the 15 powers are 3.6 eps rms off (16 eps at worst), from w and
// w^4 1.1 eps (5 at worst) -- and the twiddles' error is what bounds the transforms' (DESIGN.md section 3,
// the reference's correlation KAT).  13 products instead of 14, four live powers instead of eight.
__device__ __forceinline__ float2 tw_as(float2 v, float2) { return v; }
__device__ __forceinline__ h2 tw_as(float2 v, h2) { return to_h2(v); }
__device__ __forceinline__ p2 tw_as(float2 v, p2) { return to_p2(v); }
template <int R, bool CONJ, bool BREV, typename T>
__device__ __forceinline__ void twiddle_apply(T* x, float2 w, float2 w4) {
    static_assert(R == 16, "powers 1..15 from w and w^4");
    if (CONJ) { w.y = -w.y; w4.y = -w4.y; }
    float2 lo[4];
    lo[1] = w; lo[2] = cmul(w, w); lo[3] = cmul(lo[2], w);
    float2 g = w4;   // w^(4 * grp)
#pragma unroll
    for (int e = 1; e < R; ++e) {
        if ((e & 3) == 0 && e > 4) g = cmul(g, w4);
        const float2 v = e < 4 ? lo[e] : (e & 3) == 0 ? g : cmul(g, lo[e & 3]);
        x[BREV ? brev<R>(e) : e] = cmul(x[BREV ? brev<R>(e) : e], tw_as(v, T{}));
    }
}
template <int R, bool CONJ, typename T>
__device__ __forceinline__ void twiddle_brev(T* x, float2 w, float2 w4) { twiddle_apply<R, CONJ, true>(x, w, w4); }
template <int R, bool CONJ, typename T>
__device__ __forceinline__ void twiddle_nat(T* x, float2 w, float2 w4) { twiddle_apply<R, CONJ, false>(x, w, w4); }
// x[idx(e)] *= base * step^e with step4 = step^4 from the table as well: four runs of three products
// off the bases base * step4^k instead of one run of fifteen (the same 15 products)
template <int R, bool CONJ, bool BREV, typename T>
__device__ __forceinline__ void twiddle_chain(T* x, float2 base, float2 step, float2 step4) {
    static_assert(R == 16, "four runs of four");
    if (CONJ) { base.y = -base.y; step.y = -step.y; step4.y = -step4.y; }
    float2 g = base;
#pragma unroll
    for (int k = 0; k < R; k += 4) {
        if (k) g = cmul(g, step4);
        float2 c = g;
        x[BREV ? brev<R>(k) : k] = cmul(x[BREV ? brev<R>(k) : k], tw_as(c, T{}));
#pragma unroll
        for (int j = 1; j < 4; ++j) {
            c = cmul(c, step);
            x[BREV ? brev<R>(k + j) : k + j] = cmul(x[BREV ? brev<R>(k + j) : k + j], tw_as(c, T{}));
        }
    }
}

// 16-byte buffer accesses: one VGPR of address for a whole unrolled sequence
// (per-access offsets live in SGPRs / immediates).
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
// Cache policy of the work-matrix stream (aux of the raw buffer builtins on gfx940+: bit 0 =
// sc0, bit 1 = nt, bit 4 = sc1).  The work matrix is read once and written once per kernel; the
// needle-spectrum rows and the twiddle tables are what should stay in L2.  The defaults are the
// measured best (tools/ntbench.hip); the macros exist for A/B builds.
#ifndef AM_K2_LOAD_AUX
#define AM_K2_LOAD_AUX 0
#endif
#ifndef AM_K2_STORE_AUX
#define AM_K2_STORE_AUX 0
#endif
#ifndef AM_K1_STORE_NT
#define AM_K1_STORE_NT 0
#endif
#ifndef AM_K3_LOAD_NT
#define AM_K3_LOAD_NT 1   // K3's once-read column loads: 0.180 -> 0.156 ms per 1 h haystack (profiles/r02/nt_ab.txt)
#endif
#ifndef AM_K3H_LOAD_AUX
#define AM_K3H_LOAD_AUX 2   // K3's column loads from a half-storage work matrix: nt, as the f32 form's
#endif
#ifndef AM_K1_LOAD_NT
#define AM_K1_LOAD_NT 1   // K1's sample loads: 0.269 -> 0.261 ms with the 512-row kernel (profiles/r02/nt_ab_c512.txt)
#endif
template <int AUX = 0>
__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, AUX));
    return make_float4(v.x, v.y, v.z, v.w);
}
// 16-byte global accesses with an optional nt bit (column passes: K1's stores, K3's loads)
template <int NT>
__device__ __forceinline__ float4 load_f4(const float4* p) {
    if (NT) {
        const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
        return make_float4(v.x, v.y, v.z, v.w);
    }
    return *p;
}
template <int NT>
__device__ __forceinline__ void store_f4(float4* p, float4 v) {
    if (NT) {
        f32x4 o; o.x = v.x; o.y = v.y; o.z = v.z; o.w = v.w;
        __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(p));
    } else {
        *p = v;
    }
}
// A 16-byte store reads its four data VGPRs over several cycles after issue.  hipcc (ROCm 7.2)
// keeps the documented wait state before a VALU write of one of them only when the store has no
// SGPR soffset; with one (all of ours: soff is a multiple of 4096) it may place the overwrite
// directly behind the store, and on MI355X the last lanes of each 16-lane group then store the
// NEW value of the upper dwords now and then (the intermittent wide-plan failure of DESIGN.md
// section 3: `buffer_store_dwordx4 v[2:5], ..., s30 offen` followed by `v_sub_f32 v4, ...`).
// The empty-looking asm reads the data registers after the store, so nothing can overwrite them
// before two wait states have passed.  tools/check_store_hazard.py checks the ISA for this.
template <int AUX = 0>
__device__ __forceinline__ void buf_store4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float4 v) {
    f32x4 o; o.x = v.x; o.y = v.y; o.z = v.z; o.w = v.w;
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, o), r, voff, soff, AUX);
    asm volatile("s_nop 1" : : "v"(o));
}

// Half-precision STORAGE of the work matrix (option "half_pipeline", BASELINE
// config 5): every point travels through HBM as one __half2 (4 bytes instead of
// 8); butterflies, twiddles and the spectrum multiply stay in f32 registers.
__device__ __forceinline__ unsigned pack_h2(float2 v) {
    const __half2 h = __float22half2_rn(v);
    return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ float2 unpack_h2(unsigned u) {
    return __half22float2(__builtin_bit_cast(__half2, u));
}
template <int AUX = 0>
__device__ __forceinline__ uint2 buf_load_u2(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    const f32x2 v = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, AUX));
    // element-wise through scalars: __builtin_bit_cast on a vector ELEMENT reads the
    // vector's first lane for every element (observed with ROCm 7.2's clang)
    const float lo = v.x, hi = v.y;
    return make_uint2(__float_as_uint(lo), __float_as_uint(hi));
}
__device__ __forceinline__ void buf_store_u2(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, uint2 v) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    u32x2 o; o.x = v.x; o.y = v.y;
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((__vector_size__(2 * sizeof(unsigned)))) unsigned, o), r, voff, soff, 0);
}

// ===========================================================================
// Production kernels: N1 = 256 (16 x 16), N2 = 8192 (16 x 16 x 32), 256 threads
// ===========================================================================
constexpr int kR16LogN1 = 8, kR16LogN2 = 13;
constexpr int kN2 = 1 << kR16LogN2;

// Input samples: KIND 0 = f32 mono, KIND 1 = interleaved i16 stereo frames,
// down-mixed on the fly exactly as mp3_reader.rs:12, 28-37:
// (l as f32 + r as f32) * 0.5 * (1 / 65535), every step rounded to f32.
// Three instructions per frame instead of five, same bits: l + r is exact as an integer (17 bits) and as a sum of two
// f32 values, so one conversion of the integer sum gives the reference's f32 sum; a multiplication by 0.5 is exact
// (nothing here is near the denormals), so rounding once after multiplying with half the constant gives what rounding
// after the second of two multiplications gives.
__device__ __forceinline__ float downmix_s16(short2 lr) {
    return __fmul_rn((float)((int)lr.x + (int)lr.y), 0.5f * (1.0f / 65535.0f));
}
template <int KIND>
__device__ __forceinline__ float load_sample(const void* __restrict__ src, long long i) {
    if (KIND == 0) return static_cast<const float*>(src)[i];
    return downmix_s16(static_cast<const short2*>(src)[i]);
}
// two consecutive samples out of the 8 bytes a 64-bit load returned (an f32 sample and an i16 stereo frame
// are both 4 bytes)
template <int KIND>
__device__ __forceinline__ float2 decode_sample2(uint2 raw) {
    if (KIND == 0) return make_float2(__uint_as_float(raw.x), __uint_as_float(raw.y));
    short2 a, b;
    a.x = (short)(raw.x & 0xffffu); a.y = (short)(raw.x >> 16);
    b.x = (short)(raw.y & 0xffffu); b.y = (short)(raw.y >> 16);
    return make_float2(downmix_s16(a), downmix_s16(b));
}
// two consecutive samples from an 8-byte aligned position
template <int KIND>
__device__ __forceinline__ float2 load_sample2(const void* __restrict__ src, long long i) {
    if (KIND == 0) {
        if (AM_K1_LOAD_NT) {
            const f32x2 v = __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(static_cast<const float*>(src) + i));
            return make_float2(v.x, v.y);
        }
        return *reinterpret_cast<const float2*>(static_cast<const float*>(src) + i);
    }
    typedef int i32x2 __attribute__((ext_vector_type(2)));
    const i32x2* p = reinterpret_cast<const i32x2*>(static_cast<const short2*>(src) + i);
    const i32x2 raw = AM_K1_LOAD_NT ? __builtin_nontemporal_load(p) : *p;   // (read once, like the f32 samples)
    const int lo = raw.x, hi = raw.y;
    return make_float2(downmix_s16(__builtin_bit_cast(short2, lo)), downmix_s16(__builtin_bit_cast(short2, hi)));
}
template <int KIND>
__device__ __forceinline__ float2 load2_padded(const void* __restrict__ src, long long i, long long len) {
    float2 v;
    v.x = (i >= 0 && i < len) ? load_sample<KIND>(src, i) : 0.0f;
    v.y = (i + 1 >= 0 && i + 1 < len) ? load_sample<KIND>(src, i + 1) : 0.0f;
    return v;
}

// the table lookups of one tile: W_256^b for the pass boundary and W_N^(n2*a'), W_N^(16*n2)
// for the pipeline twiddle (k1 = a' + 16*b', a' = hi), requested before the samples so that
// their latency overlaps the streaming loads
// (w256q, step0q, step1q: the fourth powers, see twiddle_apply(x, w, w4))
struct K1Twiddles { float2 w256, w256q, base0, base1, step0, step1, step0q, step1q; };
__device__ __forceinline__ K1Twiddles k1_twiddles(const PlanDev& pl, long long col, int hi) {
    const unsigned maskN = (unsigned)((1ll << pl.logN) - 1);
    K1Twiddles w;
    w.w256 = pl.tw1[hi];
    w.w256q = pl.tw1[4 * hi];
    w.base0 = tw_big(pl, ((unsigned)col * (unsigned)hi) & maskN);
    w.base1 = tw_big(pl, (((unsigned)col + 1u) * (unsigned)hi) & maskN);
    tw_big_pair(pl, ((unsigned)col * 16u) & maskN, w.step0, w.step0q);
    tw_big_pair(pl, (((unsigned)col + 1u) * 16u) & maskN, w.step1, w.step1q);
    return w;
}
// pass 1 ownership: rows n1 = a*16 + hi of columns col, col+1 (sample index n1 * in_stride + col
// in both packed blocks): real part = block A, imaginary part = block B
template <int KIND>
__device__ __forceinline__ void k1_load(const Job& job, long long col, int in_stride, int hi, long long baseA, long long baseB,
                                        bool validB, bool fast, float2 (&x0)[16], float2 (&x1)[16]) {
    if (fast) {
#pragma unroll
        for (int a = 0; a < 16; ++a) {
            const long long off = (long long)(a * 16 + hi) * in_stride + col;
            const float2 va = load_sample2<KIND>(job.src, baseA + off), vb = load_sample2<KIND>(job.src, baseB + off);
            x0[a] = make_float2(va.x, vb.x);
            x1[a] = make_float2(va.y, vb.y);
        }
    } else {
#pragma unroll
        for (int a = 0; a < 16; ++a) {
            const long long n = (long long)(a * 16 + hi) * in_stride + col;
            const float2 va = load2_padded<KIND>(job.src, baseA + n, job.src_len);
            const float2 vb = validB ? load2_padded<KIND>(job.src, baseB + n, job.src_len) : make_float2(0.f, 0.f);
            x0[a] = make_float2(va.x, vb.x);
            x1[a] = make_float2(va.y, vb.y);
        }
    }
}
// 256-point column FFT (two passes with one LDS exchange) + pipeline twiddle W_N^(col*k1).
// On return x0/x1[brev(b')] hold row k1 = hi + 16*b' of columns col, col+1.
__device__ __forceinline__ void k1_transform(const K1Twiddles& w, float2* lds2, int hi, int cp, float2 (&x0)[16], float2 (&x1)[16]) {
    dif<16, false>(x0);
    dif<16, false>(x1);
    twiddle_brev<16, false>(x0, w.w256, w.w256q);   // W_256^(b*a')
    twiddle_brev<16, false>(x1, w.w256, w.w256q);
    // Exchange between the two passes, one column of the pair at a time so that a
    // workgroup needs 34 KB of LDS (row stride 17 keeps the 8-byte reads of rows
    // 16 apart on disjoint banks): pass 2 owns a' = hi, b = 0..15.
#pragma unroll
    for (int ap = 0; ap < 16; ++ap) lds2[(ap * 16 + hi) * 17 + cp] = x0[brev<16>(ap)];
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 16; ++b) x0[b] = lds2[(hi * 16 + b) * 17 + cp];
    __syncthreads();
#pragma unroll
    for (int ap = 0; ap < 16; ++ap) lds2[(ap * 16 + hi) * 17 + cp] = x1[brev<16>(ap)];
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 16; ++b) x1[b] = lds2[(hi * 16 + b) * 17 + cp];
    dif<16, false>(x0);
    dif<16, false>(x1);
    // k1 = a' + 16*b';  W_N^(n2*k1) = W_N^(n2*a') * (W_N^(16*n2))^b'
    twiddle_chain<16, false, true>(x0, w.base0, w.step0, w.step0q);
    twiddle_chain<16, false, true>(x1, w.base1, w.step1, w.step1q);
}
// K1: f32 window load (pad(), audio_matcher.rs:232-235, 422) + 256-point column
// FFTs of 32 adjacent columns + twiddle W_N^(n2*k1); row k1 of the work matrix
// holds frequency k1 in natural order.
template <int KIND>
__device__ __forceinline__ void k1_tile(const Job& job, const PlanDev& pl, float2* lds2, long long col, int in_stride,
                                        int hi, int cp, long long baseA, long long baseB, bool validB, bool fast,
                                        float2 (&x0)[16], float2 (&x1)[16]) {
    const K1Twiddles w = k1_twiddles(pl, col, hi);
    k1_load<KIND>(job, col, in_stride, hi, baseA, baseB, validB, fast, x0, x1);
    k1_transform(w, lds2, hi, cp, x0, x1);
}

// 256-point column FFT + pipeline twiddle on packed half-precision points (option half_pipeline = 2):
// k1_transform's steps, both columns of the pair through LDS in one exchange
__device__ __forceinline__ void k1_transform_h16(const K1Twiddles& w, uint2* ldsu, int hi, int cp, h2 (&x0)[16], h2 (&x1)[16]) {
    dif<16, false>(x0);
    dif<16, false>(x1);
    twiddle_apply2<16, false, true>(x0, x1, w.w256);
#pragma unroll
    for (int ap = 0; ap < 16; ++ap) ldsu[(ap * 16 + hi) * 17 + cp] = make_uint2(h2_bits(x0[brev<16>(ap)]), h2_bits(x1[brev<16>(ap)]));
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 16; ++b) {
        const uint2 v = ldsu[(hi * 16 + b) * 17 + cp];
        x0[b] = bits_h2(v.x);
        x1[b] = bits_h2(v.y);
    }
    dif<16, false>(x0);
    dif<16, false>(x1);
    twiddle_chain<16, false, true>(x0, w.base0, w.step0);
    twiddle_chain<16, false, true>(x1, w.base1, w.step1);
}

template <int KIND, int HALF>   // HALF: 0 = f32 work matrix, 1 = f16 storage, 2 = f16 storage and f16 butterflies
__device__ __forceinline__ void k1_cols_fwd_r16_tile(const Job& job, float2* __restrict__ work, const PlanDev& pl) {
    extern __shared__ float4 lds4[];
    const int t = threadIdx.x;
    const int hi = t >> 4, cp = t & 15;
    const int n2_0 = blockIdx.x << kColsLog;
    const int pair = job.first_pair + blockIdx.y;
    const long long blkA = 2ll * pair, blkB = blkA + 1;
    const bool validB = blkB < job.nblocks;
    const long long N = 1ll << pl.logN;
    const long long baseA = blkA * job.hop - job.lead;
    const long long baseB = blkB * job.hop - job.lead;
    const bool fast = ((reinterpret_cast<uintptr_t>(job.src) & 7) == 0) && ((baseA & 1) == 0) && ((baseB & 1) == 0) &&
                      baseA >= 0 && baseA + N <= job.src_len && validB && baseB + N <= job.src_len;
    float2 x0[16], x1[16];
    if constexpr (HALF == 2) {
        const long long col = (long long)n2_0 + 2 * cp;
        const K1Twiddles w = k1_twiddles(pl, col, hi);
        k1_load<KIND>(job, col, kN2, hi, baseA, baseB, validB, fast, x0, x1);
        h2 hx0[16], hx1[16];
#pragma unroll
        for (int a = 0; a < 16; ++a) { hx0[a] = to_h2(x0[a]); hx1[a] = to_h2(x1[a]); }
        k1_transform_h16(w, reinterpret_cast<uint2*>(lds4), hi, cp, hx0, hx1);
        uint2* __restrict__ out2 = reinterpret_cast<uint2*>(reinterpret_cast<unsigned*>(work) + ((size_t)blockIdx.y << pl.logN) + n2_0) + cp;
#pragma unroll
        for (int bp = 0; bp < 16; ++bp) {
            const size_t k1 = (size_t)(hi + 16 * bp);
            out2[k1 * (kN2 / 2)] = make_uint2(h2_bits(hx0[brev<16>(bp)]), h2_bits(hx1[brev<16>(bp)]));
        }
        return;
    }
    k1_tile<KIND>(job, pl, reinterpret_cast<float2*>(lds4), (long long)n2_0 + 2 * cp, kN2, hi, cp, baseA, baseB, validB, fast, x0, x1);
    if (HALF) {
        uint2* __restrict__ out2 = reinterpret_cast<uint2*>(reinterpret_cast<unsigned*>(work) + ((size_t)blockIdx.y << pl.logN) + n2_0) + cp;
#pragma unroll
        for (int bp = 0; bp < 16; ++bp) {
            const size_t k1 = (size_t)(hi + 16 * bp);
            out2[k1 * (kN2 / 2)] = make_uint2(pack_h2(x0[brev<16>(bp)]), pack_h2(x1[brev<16>(bp)]));
        }
        return;
    }
    float4* __restrict__ out4 = reinterpret_cast<float4*>(work + ((size_t)blockIdx.y << pl.logN) + n2_0) + cp;
#pragma unroll
    for (int bp = 0; bp < 16; ++bp) {
        const size_t k1 = (size_t)(hi + 16 * bp);
        store_f4<AM_K1_STORE_NT>(out4 + k1 * (kN2 / 2), make_float4(x0[brev<16>(bp)].x, x0[brev<16>(bp)].y,
                                                                    x1[brev<16>(bp)].x, x1[brev<16>(bp)].y));
    }
}
template <int KIND, int HALF>
__global__ void __launch_bounds__(256, 3)
k1_cols_fwd_r16(Job job, float2* __restrict__ work, PlanDev pl) {
    k1_cols_fwd_r16_tile<KIND, HALF>(job, work, pl);
}
// The odd last blocks of several haystacks in one launch (TailBatch): entry z = blockIdx.y is pair 0 of its own job
// and fills work slot z (the tile function takes its slot from blockIdx.y and its pair from first_pair + blockIdx.y).
template <int KIND, int HALF>
__global__ void __launch_bounds__(256, 3)
tail_cols_fwd_r16(TailBatch tb, int hop, float2* __restrict__ work, PlanDev pl) {
    const unsigned z = blockIdx.y;
    Job job;
    job.src = tb.src[z]; job.src_len = tb.src_len[z]; job.lead = 0; job.dst = nullptr; job.out_count = tb.out_count[z];
    job.hop = hop; job.nblocks = (int)((tb.out_count[z] + hop - 1) / hop); job.first_pair = -(int)z; job.src_kind = KIND;
    k1_cols_fwd_r16_tile<KIND, HALF>(job, work, pl);
}

// Ordering point for an LDS exchange whose writers and readers are lanes of the
// same wavefront: LDS serves one wave's instructions in issue order, so only the
// compiler has to be kept from moving the accesses across it.
__device__ __forceinline__ void wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// K2: one 8192-point row: forward FFT (16 x 16 x 32), multiply by conj(H)/N,
// inverse FFT.  The three stages below are shared by the single-needle kernel
// and the needle-group kernel.
struct K2Lane {
    int t, hi, cp;
    unsigned voff;               // byte offset of the thread's first float4 in a row
    float2 wj0, wj1, wc0, wc1;   // twiddle seeds of both pass boundaries
    float2 qj0, qj1, qc0, qc1;   // and their fourth powers (see twiddle_apply(x, w, w4))
};
__device__ __forceinline__ K2Lane k2_lane(const PlanDev& pl, int t = threadIdx.x) {
    K2Lane k;
    k.t = t; k.hi = k.t >> 4; k.cp = k.t & 15;
    k.voff = (unsigned)k.t * 16u;
    // fetched beside the row so that their (L2) latency is not exposed in the middle of the transform:
    // W_8192^(2t), ^(2t+1), ^(32 cp), ^(32 cp + 16) and their fourth powers, four 16-byte fetches (PlanDev::k2j, k2c)
    const float4 j = pl.k2j[2 * k.t], jq = pl.k2j[2 * k.t + 1], c = pl.k2c[2 * k.cp], cq = pl.k2c[2 * k.cp + 1];
    k.wj0 = make_float2(j.x, j.y); k.wj1 = make_float2(j.z, j.w);
    k.qj0 = make_float2(jq.x, jq.y); k.qj1 = make_float2(jq.z, jq.w);
    k.wc0 = make_float2(c.x, c.y); k.wc1 = make_float2(c.z, c.w);
    k.qc0 = make_float2(cq.x, cq.y); k.qc1 = make_float2(cq.z, cq.w);
    return k;
}

// row load + forward passes 1 and 2; leaves the row in LDS in the layout pass 3 reads
template <bool HALF>
__device__ __forceinline__ void k2_forward12(const K2Lane& k, __amdgpu_buffer_rsrc_t rrow, float4* lds4) {
    const int t = k.t, hi = k.hi, cp = k.cp;
    float2 x0[16], x1[16];
#pragma unroll
    for (int a = 0; a < 16; ++a) {   // elements a*512 + 2t, +1
        if (HALF) {
            const uint2 v = buf_load_u2(rrow, k.voff / 2, a * 2048);
            x0[a] = unpack_h2(v.x);
            x1[a] = unpack_h2(v.y);
        } else {
            const float4 v = buf_load4<AM_K2_LOAD_AUX>(rrow, k.voff, a * 4096);
            x0[a] = make_float2(v.x, v.y);
            x1[a] = make_float2(v.z, v.w);
        }
    }
    // ---- pass 1 over a (stride 512), twiddle W_8192^(j*a'), j = 2t, 2t+1 ----
    dif<16, false>(x0);
    dif<16, false>(x1);
    twiddle_brev<16, false>(x0, k.wj0, k.qj0);
    twiddle_brev<16, false>(x1, k.wj1, k.qj1);
#pragma unroll
    for (int ap = 0; ap < 16; ++ap)   // L1[a'][j]
        lds4[ap * 256 + t] = make_float4(x0[brev<16>(ap)].x, x0[brev<16>(ap)].y,
                                         x1[brev<16>(ap)].x, x1[brev<16>(ap)].y);
    __syncthreads();
    // ---- pass 2 over b (stride 32): a' = hi, c = 2cp, 2cp+1; twiddle W_512^(c*b') ----
#pragma unroll
    for (int b = 0; b < 16; ++b) {
        const float4 v = lds4[hi * 256 + b * 16 + cp];
        x0[b] = make_float2(v.x, v.y);
        x1[b] = make_float2(v.z, v.w);
    }
    dif<16, false>(x0);
    dif<16, false>(x1);
    twiddle_brev<16, false>(x0, k.wc0, k.qc0);
    twiddle_brev<16, false>(x1, k.wc1, k.qc1);
    wave_sync_lds();   // this exchange stays inside one wavefront (rows 64w .. 64w+63 <-> threads of wave w)
#pragma unroll
    for (int bp = 0; bp < 16; ++bp)   // L2 row u = a'*16 + b', slot cp ^ b'
        lds4[(hi * 16 + bp) * 16 + (cp ^ bp)] = make_float4(x0[brev<16>(bp)].x, x0[brev<16>(bp)].y,
                                                            x1[brev<16>(bp)].x, x1[brev<16>(bp)].y);
    wave_sync_lds();   // this exchange stays inside one wavefront (rows 64w .. 64w+63 <-> threads of wave w)
}

// ---- pass 3 over c (32 contiguous): thread owns row u = t; z[r] = frequency brev(r) of that row ----
__device__ __forceinline__ void k2_forward3(const K2Lane& k, const float4* lds4, float2 (&z)[32]) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float4 v = lds4[k.t * 16 + (i ^ k.cp)];
        z[2 * i] = make_float2(v.x, v.y);
        z[2 * i + 1] = make_float2(v.z, v.w);
    }
    dif<32, false>(z);
}

// pointwise multiply (pairwise_mult_in_place, audio_matcher.rs:432-438)
// q = z * h in the order the inverse wants (z[r] holds frequency brev(r); the inverse takes natural order)
template <bool HALF>
__device__ __forceinline__ void k2_multiply(const float2 (&z)[32], const float4 (&h)[16], float hscale, float2 (&q)[32]) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        // hscale: half storage only (it keeps the stored values well inside f16's range); the f32 form
        // does not spend 64 multiplications by one on it
        const float s = HALF ? hscale : 1.0f;
        q[brev<32>(2 * i)] = cmul(z[2 * i], HALF ? make_float2(h[i].x * s, h[i].y * s) : make_float2(h[i].x, h[i].y));
        q[brev<32>(2 * i + 1)] = cmul(z[2 * i + 1], HALF ? make_float2(h[i].z * s, h[i].w * s) : make_float2(h[i].z, h[i].w));
    }
}
// The same with the spectrum row fetched a quarter at a time (needle-group kernel: z
// stays live across needles, so z, q and a whole spectrum row do not fit the register
// file together).  The fetch of a quarter is in flight while the previous one is used;
// `first` is quarter 0, requested by the caller ahead of time.
__device__ __forceinline__ void k2_fetch_quarter(__amdgpu_buffer_rsrc_t rh, unsigned voff, int part, float4 (&h)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) h[i] = buf_load4(rh, voff, (part * 4 + i) * 4096);
}
__device__ __forceinline__ void k2_multiply_quarter(const float2 (&z)[32], const float4 (&h)[4], int part, float2 (&q)[32]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = part * 4 + i;
        q[brev<32>(2 * e)] = cmul(z[2 * e], make_float2(h[i].x, h[i].y));
        q[brev<32>(2 * e + 1)] = cmul(z[2 * e + 1], make_float2(h[i].z, h[i].w));
    }
}
__device__ __forceinline__ void k2_multiply_fetch(const float2 (&z)[32], __amdgpu_buffer_rsrc_t rh, unsigned voff,
                                                  const float4 (&first)[4], float2 (&q)[32]) {
    float4 ha[4], hb[4];
    k2_fetch_quarter(rh, voff, 1, ha);
    __builtin_amdgcn_sched_barrier(0);
    k2_multiply_quarter(z, first, 0, q);
    k2_fetch_quarter(rh, voff, 2, hb);
    __builtin_amdgcn_sched_barrier(0);
    k2_multiply_quarter(z, ha, 1, q);
    k2_fetch_quarter(rh, voff, 3, ha);
    __builtin_amdgcn_sched_barrier(0);
    k2_multiply_quarter(z, hb, 2, q);
    k2_multiply_quarter(z, ha, 3, q);
}

// inverse passes 3, 2, 1 of the product q and the row store
template <bool HALF, int SAUX = AM_K2_STORE_AUX>
__device__ __forceinline__ void k2_inverse(const K2Lane& k, float2 (&q)[32], float4* lds4, __amdgpu_buffer_rsrc_t rdst) {
    const int t = k.t, hi = k.hi, cp = k.cp;
    // ---- inverse pass 3 over c' ----
    dif<32, true>(q);   // time index c at q[brev(c)]
#pragma unroll
    for (int i = 0; i < 16; ++i)   // own row again, no barrier needed before
        lds4[t * 16 + (i ^ cp)] = make_float4(q[brev<32>(2 * i)].x, q[brev<32>(2 * i)].y,
                                              q[brev<32>(2 * i + 1)].x, q[brev<32>(2 * i + 1)].y);
    wave_sync_lds();   // this exchange stays inside one wavefront (rows 64w .. 64w+63 <-> threads of wave w)
    // ---- inverse pass 2 over b': conj twiddle first, then butterflies ----
    float2 x0[16], x1[16];
#pragma unroll
    for (int bp = 0; bp < 16; ++bp) {
        const float4 v = lds4[(hi * 16 + bp) * 16 + (cp ^ bp)];
        x0[bp] = make_float2(v.x, v.y);
        x1[bp] = make_float2(v.z, v.w);
    }
    twiddle_nat<16, true>(x0, k.wc0, k.qc0);
    twiddle_nat<16, true>(x1, k.wc1, k.qc1);
    dif<16, true>(x0);
    dif<16, true>(x1);
    wave_sync_lds();   // this exchange stays inside one wavefront (rows 64w .. 64w+63 <-> threads of wave w)
#pragma unroll
    for (int b = 0; b < 16; ++b)
        lds4[hi * 256 + b * 16 + cp] = make_float4(x0[brev<16>(b)].x, x0[brev<16>(b)].y,
                                                   x1[brev<16>(b)].x, x1[brev<16>(b)].y);
    __syncthreads();
    // ---- inverse pass 1 over a' ----
#pragma unroll
    for (int ap = 0; ap < 16; ++ap) {
        const float4 v = lds4[ap * 256 + t];
        x0[ap] = make_float2(v.x, v.y);
        x1[ap] = make_float2(v.z, v.w);
    }
    twiddle_nat<16, true>(x0, k.wj0, k.qj0);
    twiddle_nat<16, true>(x1, k.wj1, k.qj1);
    dif<16, true>(x0);
    dif<16, true>(x1);
    if (HALF) {
#pragma unroll
        for (int a = 0; a < 16; ++a)
            buf_store_u2(rdst, k.voff / 2, a * 2048, make_uint2(pack_h2(x0[brev<16>(a)]), pack_h2(x1[brev<16>(a)])));
        return;
    }
#pragma unroll
    for (int a = 0; a < 16; ++a)
        buf_store4<SAUX>(rdst, k.voff, a * 4096, make_float4(x0[brev<16>(a)].x, x0[brev<16>(a)].y,
                                                       x1[brev<16>(a)].x, x1[brev<16>(a)].y));
}

// Workgroups are dealt round-robin over the 8 XCDs (speed only, never
// correctness): give every XCD whole rows, so that the needle-spectrum row
// shared by all pairs is fetched into that XCD's L2 once.
__device__ __forceinline__ void k2_place(unsigned npairs, unsigned& row, unsigned& slot) {
    const unsigned lin = blockIdx.x, xcd = lin & 7u, seq = lin >> 3;
    row = (seq / npairs) * 8u + xcd;
    slot = seq % npairs;
}

// SPECTRUM = true stores conj(FFT)/N of the needle instead (fft_b, the conj of
// pairwise_mult_in_place and the 1/len of audio_matcher.rs:430-442 folded into
// one table).
template <bool SPECTRUM, bool HALF>
__global__ void __launch_bounds__(256, 2)
k2_rows_r16(float2* __restrict__ work, const float2* __restrict__ hc, float2* __restrict__ hc_out, PlanDev pl,
            unsigned npairs, float hscale) {
    extern __shared__ float4 lds4[];
    unsigned row, slot;
    k2_place(npairs, row, slot);
    const size_t row_off = ((size_t)slot << pl.logN) + (size_t)row * kN2;
    // one point is 8 bytes (float2) or, with half storage, 4 bytes (__half2)
    const __amdgpu_buffer_rsrc_t rrow = HALF ? make_rsrc(reinterpret_cast<unsigned*>(work) + row_off, kN2 * 4)
                                             : make_rsrc(work + row_off, kN2 * 8);
    const size_t hoff4 = (size_t)row * (kN2 / 2);
    const K2Lane k = k2_lane(pl);
    k2_forward12<HALF>(k, rrow, lds4);
    // the needle-spectrum row (L2-resident) is requested here, where only the 32
    // points of pass 3 are live, so that its latency hides behind that pass
    float4 h[16];
    if (!SPECTRUM) {
        __builtin_amdgcn_sched_barrier(0);
        const __amdgpu_buffer_rsrc_t rh = make_rsrc(reinterpret_cast<const float4*>(hc) + hoff4, kN2 * 8);
#pragma unroll
        for (int i = 0; i < 16; ++i) h[i] = buf_load4(rh, k.voff, i * 4096);
        __builtin_amdgcn_sched_barrier(0);
    }
    float2 z[32];
    k2_forward3(k, lds4, z);
    if (SPECTRUM) {
        const float invN = 1.0f / (float)(1u << pl.logN);
        const __amdgpu_buffer_rsrc_t rho = make_rsrc(reinterpret_cast<float4*>(hc_out) + hoff4, kN2 * 8);
#pragma unroll
        for (int i = 0; i < 16; ++i)
            buf_store4(rho, k.voff, i * 4096, make_float4(z[2 * i].x * invN, -z[2 * i].y * invN,
                                                         z[2 * i + 1].x * invN, -z[2 * i + 1].y * invN));
        return;
    }
    // in place, or into a second work matrix (hc_out doubles as that destination)
    const __amdgpu_buffer_rsrc_t rdst = !hc_out ? rrow
        : HALF ? make_rsrc(reinterpret_cast<unsigned*>(hc_out) + row_off, kN2 * 4) : make_rsrc(hc_out + row_off, kN2 * 8);
    float2 q[32];
    k2_multiply<HALF>(z, h, hscale, q);
    k2_inverse<HALF>(k, q, lds4, rdst);
}

// K2 with half-precision butterflies (option half_pipeline = 2, BASELINE config 5's "f16 FFT
// pipeline"): the same three passes on h2 points.  A row is 32 KB in LDS (8 bytes per pair of
// points), a thread's 32 points are 32 registers.  Range: the stored row is multiplied by `pre`
// (2^-7) on the way in, so that the forward spectrum of full-scale input stays below f16's 65504,
// and the needle spectrum arrives scaled to an rms of 1/8 per bin (`hscale`); K3 divides both out
// in f32.  LDS layout: the f32 kernel's, in 8-byte elements, with 16 elements of padding per
// 256 so that the two 16-lane halves of a 32-lane access group sit 32 banks apart.
constexpr int kK2hSlab = 272;                  // 8-byte elements per a' (256 + 16)
constexpr int kK2hLds = 16 * kK2hSlab * 8;     // 34 816 bytes

#ifndef AM_K2H_WGS
#define AM_K2H_WGS 3   // waves per SIMD the register allocation has to allow (3 and 4 measure the same; 4 spills)
#endif
__device__ __forceinline__ void k2_rows_h16_row(unsigned* __restrict__ work, const unsigned* __restrict__ hc16, unsigned* __restrict__ dst,
                                                const PlanDev& pl, unsigned npairs, float pre) {
    extern __shared__ float4 lds4[];
    uint2* ldsu = reinterpret_cast<uint2*>(lds4);
    unsigned row, slot;
    k2_place(npairs, row, slot);
    const size_t row_off = ((size_t)slot << pl.logN) + (size_t)row * kN2;
    const __amdgpu_buffer_rsrc_t rrow = make_rsrc(work + row_off, kN2 * 4);
    const __amdgpu_buffer_rsrc_t rdst = dst ? make_rsrc(dst + row_off, kN2 * 4) : rrow;   // in place, or a second work matrix
    const K2Lane k = k2_lane(pl);
    const int t = k.t, hi = k.hi, cp = k.cp;
    h2 x0[16], x1[16];
    {
        const h2 p2 = (h2){(_Float16)pre, (_Float16)pre};
#pragma unroll
        for (int a = 0; a < 16; ++a) {   // elements a*512 + 2t, +1
            const uint2 v = buf_load_u2(rrow, k.voff / 2, a * 2048);
            x0[a] = bits_h2(v.x) * p2;
            x1[a] = bits_h2(v.y) * p2;
        }
    }
    // ---- pass 1 over a ----
    dif<16, false>(x0);
    dif<16, false>(x1);
    twiddle_brev<16, false>(x0, k.wj0, k.qj0);
    twiddle_brev<16, false>(x1, k.wj1, k.qj1);
#pragma unroll
    for (int ap = 0; ap < 16; ++ap)
        ldsu[ap * kK2hSlab + t] = make_uint2(h2_bits(x0[brev<16>(ap)]), h2_bits(x1[brev<16>(ap)]));
    __syncthreads();
    // ---- pass 2 over b ----
#pragma unroll
    for (int b = 0; b < 16; ++b) {
        const uint2 v = ldsu[hi * kK2hSlab + b * 16 + cp];
        x0[b] = bits_h2(v.x);
        x1[b] = bits_h2(v.y);
    }
    dif<16, false>(x0);
    dif<16, false>(x1);
    twiddle_brev<16, false>(x0, k.wc0, k.qc0);
    twiddle_brev<16, false>(x1, k.wc1, k.qc1);
    wave_sync_lds();   // wave-local exchange: slab hi belongs to the wavefront of the threads with that hi
#pragma unroll
    for (int bp = 0; bp < 16; ++bp)   // row u = hi*16 + b' of the slab, slot cp ^ b'
        ldsu[hi * kK2hSlab + bp * 16 + (cp ^ bp)] = make_uint2(h2_bits(x0[brev<16>(bp)]), h2_bits(x1[brev<16>(bp)]));
    wave_sync_lds();
    // the needle-spectrum row, requested where only the 32 points of pass 3 are live
    uint2 h[16];
    __builtin_amdgcn_sched_barrier(0);
    {
        const __amdgpu_buffer_rsrc_t rh = make_rsrc(hc16 + (size_t)row * kN2, kN2 * 4);
#pragma unroll
        for (int i = 0; i < 16; ++i) h[i] = buf_load_u2(rh, k.voff / 2, i * 2048);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- pass 3 over c: thread owns row u = t ----
    h2 z[32];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint2 v = ldsu[hi * kK2hSlab + cp * 16 + (i ^ cp)];
        z[2 * i] = bits_h2(v.x);
        z[2 * i + 1] = bits_h2(v.y);
    }
    dif<32, false>(z);
    // ---- multiply (pairwise_mult_in_place, audio_matcher.rs:432-438) ----
    h2 q[32];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        q[brev<32>(2 * i)] = cmul(z[2 * i], bits_h2(h[i].x));
        q[brev<32>(2 * i + 1)] = cmul(z[2 * i + 1], bits_h2(h[i].y));
    }
    // ---- inverse pass 3 ----
    dif<32, true>(q);
#pragma unroll
    for (int i = 0; i < 16; ++i)
        ldsu[hi * kK2hSlab + cp * 16 + (i ^ cp)] = make_uint2(h2_bits(q[brev<32>(2 * i)]), h2_bits(q[brev<32>(2 * i + 1)]));
    wave_sync_lds();
    // ---- inverse pass 2 ----
#pragma unroll
    for (int bp = 0; bp < 16; ++bp) {
        const uint2 v = ldsu[hi * kK2hSlab + bp * 16 + (cp ^ bp)];
        x0[bp] = bits_h2(v.x);
        x1[bp] = bits_h2(v.y);
    }
    twiddle_nat<16, true>(x0, k.wc0, k.qc0);
    twiddle_nat<16, true>(x1, k.wc1, k.qc1);
    dif<16, true>(x0);
    dif<16, true>(x1);
    wave_sync_lds();
#pragma unroll
    for (int b = 0; b < 16; ++b)
        ldsu[hi * kK2hSlab + b * 16 + cp] = make_uint2(h2_bits(x0[brev<16>(b)]), h2_bits(x1[brev<16>(b)]));
    __syncthreads();
    // ---- inverse pass 1 ----
#pragma unroll
    for (int ap = 0; ap < 16; ++ap) {
        const uint2 v = ldsu[ap * kK2hSlab + t];
        x0[ap] = bits_h2(v.x);
        x1[ap] = bits_h2(v.y);
    }
    twiddle_nat<16, true>(x0, k.wj0, k.qj0);
    twiddle_nat<16, true>(x1, k.wj1, k.qj1);
    dif<16, true>(x0);
    dif<16, true>(x1);
#pragma unroll
    for (int a = 0; a < 16; ++a)
        buf_store_u2(rdst, k.voff / 2, a * 2048, make_uint2(h2_bits(x0[brev<16>(a)]), h2_bits(x1[brev<16>(a)])));
}
__global__ void __launch_bounds__(256, AM_K2H_WGS)
k2_rows_h16(unsigned* __restrict__ work, const unsigned* __restrict__ hc16, unsigned* __restrict__ dst, PlanDev pl, unsigned npairs,
            float pre) {
    k2_rows_h16_row(work, hc16, dst, pl, npairs, pre);
}
// (the same kernel for a haystack's odd last block on the smaller plan -- launch_k2(..., tail) -- under a name of its
// own: a profile's average for k2_rows_h16 stays that of the main pass's launches)
__global__ void __launch_bounds__(256, AM_K2H_WGS)
tail_rows_h16(unsigned* __restrict__ work, const unsigned* __restrict__ hc16, unsigned* __restrict__ dst, PlanDev pl, unsigned npairs,
              float pre) {
    k2_rows_h16_row(work, hc16, dst, pl, npairs, pre);
}

// K2 in f32 with the row exchanged one PLANE at a time (the points a thread holds as x0, then those it
// holds as x1: every exchange of the kernel keeps the two apart, see k2_forward12 / k2_inverse), in the
// 8-byte layout of k2_rows_h16: 34 KB of LDS instead of 64 and, with the needle-spectrum row fetched by
// halves, at most 168 registers -- three workgroups per CU instead of two.  Why that matters: K2 is bound
// by VALU issue, not by its stream (SQ counters, DESIGN.md section 5: 3292 VALU instructions per wave, the
// SIMDs' VALU busy in 90 % of the kernel's cycles at one instruction per four cycles); one wave issues a
// VALU instruction at most every eight cycles, two waves per SIMD reach one per four, three and more one
// per 3.2 - 3.5 (tools/pkbench).  The two cross-wave exchanges cost three barriers each instead of one.
#ifndef AM_K2_PLANES
#define AM_K2_PLANES 1
#endif
#ifndef AM_K2P_WAVES
#define AM_K2P_WAVES 4   // (both forms fit 124 registers: four workgroups per CU, 136 KB of LDS)
#endif
// row load, forward passes 1 and 2 and the exchange into pass 3's layout: z[r] = point r of the thread's row u = t
template <bool HALF>
__device__ __forceinline__ void k2p_forward(const K2Lane& k, __amdgpu_buffer_rsrc_t rrow, float2* lds2, float2 (&z)[32]) {
    const int t = k.t, hi = k.hi, cp = k.cp;
    float2 x0[16], x1[16];
#pragma unroll
    for (int a = 0; a < 16; ++a) {   // elements a*512 + 2t, +1
        if (HALF) {
            const uint2 v = buf_load_u2(rrow, k.voff / 2, a * 2048);
            x0[a] = unpack_h2(v.x);
            x1[a] = unpack_h2(v.y);
        } else {
            const float4 v = buf_load4<AM_K2_LOAD_AUX>(rrow, k.voff, a * 4096);
            x0[a] = make_float2(v.x, v.y);
            x1[a] = make_float2(v.z, v.w);
        }
    }
    // ---- pass 1 over a (stride 512), twiddle W_8192^(j*a'), j = 2t, 2t+1; exchange L1[a'][j] ----
    dif<16, false>(x0);
    twiddle_brev<16, false>(x0, k.wj0, k.qj0);
#pragma unroll
    for (int ap = 0; ap < 16; ++ap) lds2[ap * kK2hSlab + t] = x0[brev<16>(ap)];
    dif<16, false>(x1);
    twiddle_brev<16, false>(x1, k.wj1, k.qj1);
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 16; ++b) x0[b] = lds2[hi * kK2hSlab + b * 16 + cp];
    __syncthreads();
#pragma unroll
    for (int ap = 0; ap < 16; ++ap) lds2[ap * kK2hSlab + t] = x1[brev<16>(ap)];
    // ---- pass 2 over b (stride 32): a' = hi, c = 2cp, 2cp+1; twiddle W_512^(c*b') ----
    dif<16, false>(x0);
    twiddle_brev<16, false>(x0, k.wc0, k.qc0);
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 16; ++b) x1[b] = lds2[hi * kK2hSlab + b * 16 + cp];
    // from here to the last exchange a slab (a' = hi) is touched by the wavefront of the threads with that hi only
    wave_sync_lds();
#pragma unroll
    for (int bp = 0; bp < 16; ++bp) lds2[hi * kK2hSlab + bp * 16 + (cp ^ bp)] = x0[brev<16>(bp)];   // row u = hi*16 + b', slot cp ^ b'
    dif<16, false>(x1);
    twiddle_brev<16, false>(x1, k.wc1, k.qc1);
    wave_sync_lds();
    // ---- pass 3 over c (32 contiguous): thread owns row u = t ----
#pragma unroll
    for (int i = 0; i < 16; ++i) z[2 * i] = lds2[hi * kK2hSlab + cp * 16 + (i ^ cp)];
    wave_sync_lds();
#pragma unroll
    for (int bp = 0; bp < 16; ++bp) lds2[hi * kK2hSlab + bp * 16 + (cp ^ bp)] = x1[brev<16>(bp)];
    wave_sync_lds();
#pragma unroll
    for (int i = 0; i < 16; ++i) z[2 * i + 1] = lds2[hi * kK2hSlab + cp * 16 + (i ^ cp)];
}

// inverse passes 3, 2, 1 of the product q and the row store
// LATE_J: the first boundary's seeds are fetched behind pass 2, where they are wanted, instead of with the second
// boundary's at the start: the needle-group kernel, which carries the forward spectrum through this function, gets
// from 12 spilled registers to 5 with it (1.46 -> 1.37 ms per launch); the single-needle kernel has the registers
// and measures 2 % slower with the later fetch.
template <bool HALF, int SAUX = AM_K2_STORE_AUX, bool LATE_J = false>
__device__ __forceinline__ void k2p_inverse(const PlanDev& pl, float2 (&q)[32], float2* lds2, __amdgpu_buffer_rsrc_t rdst) {
    // the lane's twiddle seeds are fetched again (L1 / L2 hits, in flight behind pass 3) instead of kept alive
    // across the product: sixteen registers the multiply, the kernel's widest point, does not have
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    const int hi = t >> 4, cp = t & 15;
    K2Lane ki;
    ki.t = t; ki.hi = hi; ki.cp = cp; ki.voff = (unsigned)t * 16u;
    {
        const float4 c = pl.k2c[2 * cp], cq = pl.k2c[2 * cp + 1];
        ki.wc0 = make_float2(c.x, c.y); ki.wc1 = make_float2(c.z, c.w);
        ki.qc0 = make_float2(cq.x, cq.y); ki.qc1 = make_float2(cq.z, cq.w);
    }
    if (!LATE_J) {
        const float4 j = pl.k2j[2 * t], jq = pl.k2j[2 * t + 1];
        ki.wj0 = make_float2(j.x, j.y); ki.wj1 = make_float2(j.z, j.w);
        ki.qj0 = make_float2(jq.x, jq.y); ki.qj1 = make_float2(jq.z, jq.w);
    }
    const K2Lane& k = ki;
    float2 x0[16], x1[16];
    // ---- inverse pass 3 over c' ----
    dif<32, true>(q);   // time index c at q[brev(c)]
    wave_sync_lds();
#pragma unroll
    for (int i = 0; i < 16; ++i) lds2[hi * kK2hSlab + cp * 16 + (i ^ cp)] = q[brev<32>(2 * i)];
    wave_sync_lds();
#pragma unroll
    for (int bp = 0; bp < 16; ++bp) x0[bp] = lds2[hi * kK2hSlab + bp * 16 + (cp ^ bp)];
    wave_sync_lds();
#pragma unroll
    for (int i = 0; i < 16; ++i) lds2[hi * kK2hSlab + cp * 16 + (i ^ cp)] = q[brev<32>(2 * i + 1)];
    // ---- inverse pass 2 over b': conj twiddle first, then butterflies ----
    twiddle_nat<16, true>(x0, ki.wc0, ki.qc0);
    dif<16, true>(x0);
    wave_sync_lds();
#pragma unroll
    for (int bp = 0; bp < 16; ++bp) x1[bp] = lds2[hi * kK2hSlab + bp * 16 + (cp ^ bp)];
    wave_sync_lds();
#pragma unroll
    for (int b = 0; b < 16; ++b) lds2[hi * kK2hSlab + b * 16 + cp] = x0[brev<16>(b)];
    if (LATE_J) {
        __builtin_amdgcn_sched_barrier(0);
        const float4 j = pl.k2j[2 * t], jq = pl.k2j[2 * t + 1];
        ki.wj0 = make_float2(j.x, j.y); ki.wj1 = make_float2(j.z, j.w);
        ki.qj0 = make_float2(jq.x, jq.y); ki.qj1 = make_float2(jq.z, jq.w);
        __builtin_amdgcn_sched_barrier(0);
    }
    twiddle_nat<16, true>(x1, ki.wc1, ki.qc1);
    dif<16, true>(x1);
    __syncthreads();
    // ---- inverse pass 1 over a' ----
#pragma unroll
    for (int ap = 0; ap < 16; ++ap) x0[ap] = lds2[ap * kK2hSlab + t];
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 16; ++b) lds2[hi * kK2hSlab + b * 16 + cp] = x1[brev<16>(b)];
    twiddle_nat<16, true>(x0, ki.wj0, ki.qj0);
    dif<16, true>(x0);
    __syncthreads();
#pragma unroll
    for (int ap = 0; ap < 16; ++ap) x1[ap] = lds2[ap * kK2hSlab + t];
    twiddle_nat<16, true>(x1, ki.wj1, ki.qj1);
    dif<16, true>(x1);
    if (HALF) {
#pragma unroll
        for (int a = 0; a < 16; ++a)
            buf_store_u2(rdst, k.voff / 2, a * 2048, make_uint2(pack_h2(x0[brev<16>(a)]), pack_h2(x1[brev<16>(a)])));
        return;
    }
#pragma unroll
    for (int a = 0; a < 16; ++a)
        buf_store4<SAUX>(rdst, k.voff, a * 4096, make_float4(x0[brev<16>(a)].x, x0[brev<16>(a)].y,
                                                           x1[brev<16>(a)].x, x1[brev<16>(a)].y));
}

template <bool HALF>   // HALF: the work matrix holds __half2 points (half_pipeline = 1), the arithmetic stays f32
__device__ __forceinline__ void k2_rows_r16_planes_row(float2* __restrict__ work, const float2* __restrict__ hc, float2* __restrict__ dst,
                                                       const PlanDev& pl, unsigned npairs, float hscale) {
    extern __shared__ float4 lds4[];
    float2* lds2 = reinterpret_cast<float2*>(lds4);
    unsigned row, slot;
    k2_place(npairs, row, slot);
    const size_t row_off = ((size_t)slot << pl.logN) + (size_t)row * kN2;
    // one point is 8 bytes (float2) or, with half storage, 4 bytes (__half2)
    const __amdgpu_buffer_rsrc_t rrow = HALF ? make_rsrc(reinterpret_cast<unsigned*>(work) + row_off, kN2 * 4)
                                             : make_rsrc(work + row_off, kN2 * 8);
    // in place, or into a second work matrix
    const __amdgpu_buffer_rsrc_t rdst = !dst ? rrow
        : HALF ? make_rsrc(reinterpret_cast<unsigned*>(dst) + row_off, kN2 * 4) : make_rsrc(dst + row_off, kN2 * 8);
    const __amdgpu_buffer_rsrc_t rh = make_rsrc(reinterpret_cast<const float4*>(hc) + (size_t)row * (kN2 / 2), kN2 * 8);
    const K2Lane k = k2_lane(pl);
    float2 z[32];
    k2p_forward<HALF>(k, rrow, lds2, z);
    // the needle-spectrum row (L2-resident) by quarters: two requested where only the 32 points of pass 3 are
    // live, the others while the earlier ones are used (z, q and the whole row together do not fit 168 registers)
    float4 ha[4], hb[4], hc4[4];
    __builtin_amdgcn_sched_barrier(0);
    k2_fetch_quarter(rh, k.voff, 0, ha);
    k2_fetch_quarter(rh, k.voff, 1, hb);
    __builtin_amdgcn_sched_barrier(0);
    dif<32, false>(z);   // z[r] = frequency brev(r) of the row
    __builtin_amdgcn_sched_barrier(0);
    k2_fetch_quarter(rh, k.voff, 2, hc4);
    __builtin_amdgcn_sched_barrier(0);
    // ---- multiply (pairwise_mult_in_place, audio_matcher.rs:432-438): q in the order the inverse wants ----
    // (hscale: half storage only -- it keeps the stored values well inside f16's range)
    auto scaled = [&](float4 (&h)[4]) {
        if (HALF) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { h[i].x *= hscale; h[i].y *= hscale; h[i].z *= hscale; h[i].w *= hscale; }
        }
    };
    float2 q[32];
    scaled(ha);
    k2_multiply_quarter(z, ha, 0, q);
    k2_fetch_quarter(rh, k.voff, 3, ha);
    __builtin_amdgcn_sched_barrier(0);
    scaled(hb);
    k2_multiply_quarter(z, hb, 1, q);
    scaled(hc4);
    k2_multiply_quarter(z, hc4, 2, q);
    scaled(ha);
    k2_multiply_quarter(z, ha, 3, q);
    k2p_inverse<HALF>(pl, q, lds2, rdst);
}
template <bool HALF>
__global__ void __launch_bounds__(256, AM_K2P_WAVES)
k2_rows_r16_planes(float2* __restrict__ work, const float2* __restrict__ hc, float2* __restrict__ dst, PlanDev pl, unsigned npairs,
                   float hscale) {
    k2_rows_r16_planes_row<HALF>(work, hc, dst, pl, npairs, hscale);
}
template <bool HALF>   // (a haystack's odd last block: see tail_rows_h16)
__global__ void __launch_bounds__(256, AM_K2P_WAVES)
tail_rows_r16_planes(float2* __restrict__ work, const float2* __restrict__ hc, float2* __restrict__ dst, PlanDev pl, unsigned npairs,
                     float hscale) {
    k2_rows_r16_planes_row<HALF>(work, hc, dst, pl, npairs, hscale);
}

// The needle-group kernel (below) on the same plane-by-plane exchanges.
#ifndef AM_K2G_PLANES
#define AM_K2G_PLANES 1
#endif
#ifndef AM_K2GP_WAVES
#define AM_K2GP_WAVES 3
#endif
#ifndef AM_K2G_STORE_AUX
// cache policy of the group kernels' eight write streams: 2 = nt.  With the plane-wise kernel at three waves per
// SIMD nt stores take 6.05 ms per call of 32 needles x 22 pairs against 6.6 (and against 6.5 for the 64 KB form at
// two waves per SIMD, where nt made no difference: profiles/r03/k2_group_ab.txt)
#define AM_K2G_STORE_AUX 2
#endif
__global__ void __launch_bounds__(256, AM_K2GP_WAVES)
k2_rows_r16_group_planes(const float2* __restrict__ work, K2Group grp, PlanDev pl, unsigned npairs) {
    extern __shared__ float4 lds4[];
    float2* lds2 = reinterpret_cast<float2*>(lds4);
    unsigned row, slot;
    k2_place(npairs, row, slot);
    const size_t row_off = ((size_t)slot << pl.logN) + (size_t)row * kN2;
    const __amdgpu_buffer_rsrc_t rrow = make_rsrc(work + row_off, kN2 * 8);
    const size_t hoff4 = (size_t)row * (kN2 / 2);
    const K2Lane k = k2_lane(pl);
    float2 z[32];
    k2p_forward<false>(k, rrow, lds2, z);
    dif<32, false>(z);
#pragma unroll 1
    for (int j = 0; j < grp.n; ++j) {
        const __amdgpu_buffer_rsrc_t rh = make_rsrc(reinterpret_cast<const float4*>(grp.hc[j]) + hoff4, kN2 * 8);
        float4 hq[4];
        k2_fetch_quarter(rh, k.voff, 0, hq);
        float2 q[32];
        k2_multiply_fetch(z, rh, k.voff, hq, q);
        k2p_inverse<false, AM_K2G_STORE_AUX, true>(pl, q, lds2, make_rsrc(grp.dst[j] + row_off, kN2 * 8));
        __syncthreads();   // the last pass read slabs of every wave: finish before the next needle's exchanges overwrite them
    }
}

__global__ void __launch_bounds__(256) spectrum_to_half_kernel(const float2* __restrict__ hc, long long n, float scale, unsigned* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        out[i] = h2_bits(to_h2(make_float2(hc[i].x * scale, hc[i].y * scale)));
}
hipError_t launch_spectrum_to_half(hipStream_t st, const float2* hc, long long n, float scale, unsigned* out) {
    hipLaunchKernelGGL(spectrum_to_half_kernel, dim3(2048), dim3(256), 0, st, hc, n, scale, out);
    return hipGetLastError();
}

// ===========================================================================
// K2 with the 16- and 32-point butterflies on the matrix cores (option "k2_mfma", half_pipeline = 2): an A/B
// experiment against k2_rows_h16, whose 2000 packed-f16 VALU instructions per wave are what bounds that kernel
// (SQ counters, profiles/r04: VALU busy 0.83 at three waves per SIMD for half the f32 kernel's bytes).
//
// A 16-point DFT of 16 independent columns is one complex 16 x 16 matrix product, i.e. two real ones of shape
// 16 x 32 x 16: v_mfma_f32_16x16x32_f16 with the DFT matrix as the A operand ([Re F | -Im F] for the real parts of
// the outputs, [Im F | Re F] for the imaginary parts, k = 2 p + {re, im} of input point p) and the data as the B
// operand: lane (n = lane & 15, g = lane >> 4) holds input points p = 4g .. 4g+3 of column n as four h2 registers
// (exactly the operand's eight f16 k-values) and receives output points m = 4g .. 4g+3 of column n in f32.  The
// 32-point pass is the same with two k-steps and two blocks of output rows (eight instructions per 16 columns).
// The products are accumulated in f32 -- the packed butterflies round to f16 after every radix-2 stage.
//
// The row transform is the one of the other K2 kernels (n = a 512 + b 32 + c, k = a' + 16 b' + 256 c'; pass 1 over
// a with the twiddle W_8192^(j a'), j = 32 b + c; pass 2 over b with W_512^(c b'); pass 3 over c), the inverse is
// run as conj o forward o conj: the same DFT matrices and the same (forward) twiddles, applied to the inputs of
// inverse passes 2 and 1; the conjugations at the two ends ride on the spectrum multiply (the spectrum is stored
// conjugated, reordered to [a'][b'][c']: spectrum_to_half_mfma) and on the final conversion to f16.  Every twiddle a
// lane needs is a per-lane constant: 32 + 8 h2 registers, fetched once per workgroup -- the workgroups are
// persistent (a grid of four per CU walks the rows), so neither twiddle arithmetic nor seeds exist per row.
//
// Exchanges between the passes (one 33 KB LDS buffer, four workgroups per CU):
//   E1 (pass 1 -> 2): 16 planes b of 512 rows (a', c), plane stride 516 dwords; written 8 bytes at a time (both
//       columns 2n, 2n+1 of a lane), read as single dwords (a lane's four inputs b = 4g .. 4g+3 lie in four planes).
//   E2 (pass 2 -> 3) and E3 (inverse pass 3 -> 2): private to a wave, one 2 KB block per a': E2 = [c][b'] with
//       16-byte writes (b' = 4g .. 4g+3), E3 = [b'][c] likewise; XOR-swizzled so that the 16-byte writes and the
//       dword reads are conflict-free.
//   E4 (inverse pass 2 -> 1): 16 planes a' of 512 rows j, plane stride 520; dword writes, 8-byte reads.
// E1 and E4 cross wavefronts: five workgroup barriers per row (write -> read of E1 and E4, and before a buffer
// that other waves still read is overwritten).
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
constexpr int kM16Ps1 = 516, kM16Ps4 = 520;
constexpr int kK2mLds = 16 * kM16Ps4 * 4;   // 33 280 bytes
// offsets (in dwords) of the constant tables behind PlanDev::mf (built by the host, am_api.hip build_mfma_tables)
constexpr int kMfA16 = 0, kMfA32 = 2 * 64 * 4, kMfT1 = kMfA32 + 8 * 64 * 4, kMfT2 = kMfT1 + 256 * 32, kMfTotal = kMfT2 + 256 * 8;

__device__ __forceinline__ half8 as_half8(const h2 (&v)[4]) {
    u32x4v u;
    u.x = h2_bits(v[0]); u.y = h2_bits(v[1]); u.z = h2_bits(v[2]); u.w = h2_bits(v[3]);
    return __builtin_bit_cast(half8, u);
}
__device__ __forceinline__ half8 load_half8(const unsigned* p) {
    const uint4 v = *reinterpret_cast<const uint4*>(p);
    u32x4v u; u.x = v.x; u.y = v.y; u.z = v.z; u.w = v.w;
    return __builtin_bit_cast(half8, u);
}
// 16-point DFT of the lane's column: inputs p = 4g .. 4g+3 in b, outputs m = 4g .. 4g+3 in y
__device__ __forceinline__ void dft16_mfma(const half8& are, const half8& aim, const h2 (&b)[4], float2 (&y)[4]) {
    const half8 bv = as_half8(b);
    const f32x4v zero = {0.f, 0.f, 0.f, 0.f};
    const f32x4v dre = __builtin_amdgcn_mfma_f32_16x16x32_f16(are, bv, zero, 0, 0, 0);
    const f32x4v dim = __builtin_amdgcn_mfma_f32_16x16x32_f16(aim, bv, zero, 0, 0, 0);
    y[0] = make_float2(dre.x, dim.x); y[1] = make_float2(dre.y, dim.y);
    y[2] = make_float2(dre.z, dim.z); y[3] = make_float2(dre.w, dim.w);
}
// conj(a) * b
__device__ __forceinline__ h2 cmul_conj_a(h2 a, h2 b) {
    const h2 t = a * b.xx;
    return __builtin_elementwise_fma(a.yx, b.yy, (h2){t.x, -t.y});
}
__device__ __forceinline__ int swap_bits02(int c) { return (c & ~5) | ((c & 1) << 2) | ((c >> 2) & 1); }

// Measured (profiles/r04/k2_mfma_ab.txt): 869 VALU instructions per wave and row against the packed-f16 kernel's 2007,
// and still slower -- 0.241 ms per launch against 0.210: the six passes of a row are one dependent chain with five
// workgroup barriers, the kernel needs 216 registers (two waves per SIMD), and forms with fewer registers (the pass-1
// twiddles fetched twice per row: 128 / 168 registers) or with prefetches ran at 0.28 - 0.36 ms.  Kept as an opt-in.
#ifndef AM_K2M_WAVES
#define AM_K2M_WAVES 2
#endif
__global__ void __launch_bounds__(256, AM_K2M_WAVES)
k2_rows_m16(unsigned* __restrict__ work, const unsigned* __restrict__ hcm, unsigned* __restrict__ dst, PlanDev pl, unsigned npairs,
            float pre) {
    extern __shared__ float4 lds4[];
    unsigned* lds = reinterpret_cast<unsigned*>(lds4);
    const int t = threadIdx.x, w = t >> 6, l = t & 63, n = l & 15, g = l >> 4;
    // ---- per-lane constants: the DFT-16 operands and every twiddle this lane ever applies ----
    const half8 a16re = load_half8(pl.mf + kMfA16 + l * 4), a16im = load_half8(pl.mf + kMfA16 + 256 + l * 4);
    h2 tw1[4][2][4], tw2[2][4];
    {
        const uint4* p1 = reinterpret_cast<const uint4*>(pl.mf + kMfT1 + t * 32);
#pragma unroll
        for (int gl = 0; gl < 4; ++gl)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const uint4 v = p1[gl * 2 + e];
                tw1[gl][e][0] = bits_h2(v.x); tw1[gl][e][1] = bits_h2(v.y); tw1[gl][e][2] = bits_h2(v.z); tw1[gl][e][3] = bits_h2(v.w);
            }
        const uint4* p2 = reinterpret_cast<const uint4*>(pl.mf + kMfT2 + t * 8);
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            const uint4 v = p2[ch];
            tw2[ch][0] = bits_h2(v.x); tw2[ch][1] = bits_h2(v.y); tw2[ch][2] = bits_h2(v.z); tw2[ch][3] = bits_h2(v.w);
        }
    }
    const h2 prescale = (h2){(_Float16)pre, (_Float16)pre};
    // LDS addresses (dwords)
    const int e1w = (4 * w) * kM16Ps1 + (4 * g) * 32 + 2 * n;            // + gl * Ps1 + r * 32        (8-byte writes)
    const int e1r = (4 * g) * kM16Ps1 + (4 * w) * 32 + n;                // + r * Ps1 + q * 32 + 16 ch  (dword reads)
    const int blk = (4 * w) * 512;                                         // + q * 512: the wave's private blocks
    const int e2w = blk + swap_bits02(n) * 16 + ((g ^ (n & 3)) * 4);       // + q * 512 + 256 ch          (16-byte writes)
    const int e2r = blk + (g & 1) * 16 + (g >> 1) * 128 + (n & 3);         // + q * 512 + ks * 256 + (r >> 1) * 32 + (r & 1) * 64 + ((n >> 2) ^ r) * 4
    const int e3w = blk + n * 32 + ((g ^ (n & 7)) * 4);                    // + q * 512, ^ 16 for the second block of rows (16-byte writes)
    const int e3r = blk + g * 128 + (n & 3);                               // + q * 512 + r * 32 + (ch ^ (g & 1)) * 16 + ((n >> 2) ^ r) * 4
    const int e4w = (4 * w) * kM16Ps4 + g * 128 + n;                       // + q * Ps4 + r * 32 + 16 ch  (dword writes)
    const int e4r = (4 * g) * kM16Ps4 + w * 128 + 2 * n;                   // + r * Ps4 + gl * 32         (8-byte reads)
    const unsigned voff = (unsigned)(((4 * g) * 512 + 128 * w + 2 * n) * 4);   // + (r * 512 + 32 gl) * 4: the row's elements a 512 + 32 G + 2n
    const unsigned xcd = blockIdx.x & 7u;
    const unsigned nseq = (npairs << pl.logN1) >> 3, sstep = gridDim.x >> 3;
    for (unsigned seq = blockIdx.x >> 3; seq < nseq; seq += sstep) {
        const unsigned row = (seq / npairs) * 8u + xcd, slot = seq % npairs;
        const size_t row_off = ((size_t)slot << pl.logN) + (size_t)row * kN2;
        const __amdgpu_buffer_rsrc_t rrow = make_rsrc(work + row_off, kN2 * 4);
        const __amdgpu_buffer_rsrc_t rdst = dst ? make_rsrc(dst + row_off, kN2 * 4) : rrow;
        // ---- pass 1: columns j = 32 G + 2n + e (G = 4w + gl), DFT over a ----
        uint2 raw[4][4];
#pragma unroll
        for (int gl = 0; gl < 4; ++gl)
#pragma unroll
            for (int r = 0; r < 4; ++r) raw[gl][r] = buf_load_u2(rrow, voff, (unsigned)((r * 512 + 32 * gl) * 4));
#pragma unroll
        for (int gl = 0; gl < 4; ++gl) {
            h2 b0[4], b1[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { b0[r] = bits_h2(raw[gl][r].x) * prescale; b1[r] = bits_h2(raw[gl][r].y) * prescale; }
            float2 y0[4], y1[4];
            dft16_mfma(a16re, a16im, b0, y0);
            dft16_mfma(a16re, a16im, b1, y1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const h2 v0 = cmul(to_h2(y0[r]), tw1[gl][0][r]), v1 = cmul(to_h2(y1[r]), tw1[gl][1][r]);
                *reinterpret_cast<uint2*>(lds + e1w + gl * kM16Ps1 + r * 32) = make_uint2(h2_bits(v0), h2_bits(v1));
            }
        }
        __syncthreads();
        // ---- pass 2: columns (a' = 4w + q, c = 16 ch + n), DFT over b ----
        h2 bb[4][2][4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int ch = 0; ch < 2; ++ch)
#pragma unroll
                for (int r = 0; r < 4; ++r) bb[q][ch][r] = bits_h2(lds[e1r + r * kM16Ps1 + q * 32 + 16 * ch]);
        __syncthreads();   // E1 has been read by everybody: the waves' private blocks may overwrite it
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                float2 y[4];
                dft16_mfma(a16re, a16im, bb[q][ch], y);
                uint4 o;
                o.x = h2_bits(cmul(to_h2(y[0]), tw2[ch][0])); o.y = h2_bits(cmul(to_h2(y[1]), tw2[ch][1]));
                o.z = h2_bits(cmul(to_h2(y[2]), tw2[ch][2])); o.w = h2_bits(cmul(to_h2(y[3]), tw2[ch][3]));
                *reinterpret_cast<uint4*>(lds + e2w + q * 512 + 256 * ch) = o;
            }
        wave_sync_lds();
        // ---- pass 3 (DFT-32 over c), the spectrum multiply and inverse pass 3, one a' at a time ----
        {
            // (fetched per row -- 8 KB shared by every wave of the chip, L1-resident -- instead of living in 32 registers
            // across the whole loop: the pointer is laundered so that the loads stay here)
            const unsigned* a32p = pl.mf + kMfA32 + l * 4;
            asm volatile("" : "+v"(a32p));
            half8 a32[2][2][2];   // [block of output rows][k-step][re, im]
#pragma unroll
            for (int i = 0; i < 8; ++i) a32[i >> 2][(i >> 1) & 1][i & 1] = load_half8(a32p + i * 256);
            const unsigned* hrow = hcm + ((size_t)row * 256 + (size_t)(4 * w) * 16 + n) * 32 + 4 * g;   // + q * 512 + 16 mb
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint4 hv0 = *reinterpret_cast<const uint4*>(hrow + q * 512), hv1 = *reinterpret_cast<const uint4*>(hrow + q * 512 + 16);
                h2 bk[2][4];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        bk[ks][r] = bits_h2(lds[e2r + q * 512 + ks * 256 + (r >> 1) * 32 + (r & 1) * 64 + (((n >> 2) ^ r) * 4)]);
                f32x4v d[2][2];
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) { d[mb][0] = (f32x4v){0.f, 0.f, 0.f, 0.f}; d[mb][1] = d[mb][0]; }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const half8 bv = as_half8(bk[ks]);
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb) {
                        d[mb][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a32[mb][ks][0], bv, d[mb][0], 0, 0, 0);
                        d[mb][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a32[mb][ks][1], bv, d[mb][1], 0, 0, 0);
                    }
                }
                // product with the (conjugated, scaled) needle spectrum at k = a' + 16 b' + 256 c', c' = 16 mb + 4g + r:
                // conj(z) * conj(h) = conj(z h) -- the inverse below is conj o forward o conj
                h2 qv[2][4];
                {
                    const unsigned hh[2][4] = {{hv0.x, hv0.y, hv0.z, hv0.w}, {hv1.x, hv1.y, hv1.z, hv1.w}};
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb) {
                        const float zr[4] = {d[mb][0].x, d[mb][0].y, d[mb][0].z, d[mb][0].w};
                        const float zi[4] = {d[mb][1].x, d[mb][1].y, d[mb][1].z, d[mb][1].w};
#pragma unroll
                        for (int r = 0; r < 4; ++r) qv[mb][r] = cmul_conj_a(to_h2(make_float2(zr[r], zi[r])), bits_h2(hh[mb][r]));
                    }
                }
                // inverse pass 3: the same DFT-32 on the conjugated product; its inputs c' sit where the outputs were
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) { d[mb][0] = (f32x4v){0.f, 0.f, 0.f, 0.f}; d[mb][1] = d[mb][0]; }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const half8 bv = as_half8(qv[ks]);
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb) {
                        d[mb][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a32[mb][ks][0], bv, d[mb][0], 0, 0, 0);
                        d[mb][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a32[mb][ks][1], bv, d[mb][1], 0, 0, 0);
                    }
                }
                wave_sync_lds();   // (block q of E2 has been read; E3 takes its place)
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {   // c = 16 mb + 4g + r of column (a', b' = n)
                    uint4 o;
                    o.x = h2_bits(to_h2(make_float2(d[mb][0].x, d[mb][1].x))); o.y = h2_bits(to_h2(make_float2(d[mb][0].y, d[mb][1].y)));
                    o.z = h2_bits(to_h2(make_float2(d[mb][0].z, d[mb][1].z))); o.w = h2_bits(to_h2(make_float2(d[mb][0].w, d[mb][1].w)));
                    *reinterpret_cast<uint4*>(lds + ((e3w + q * 512) ^ (mb * 16))) = o;
                }
            }
        }
        wave_sync_lds();
        // ---- inverse pass 2: columns (a', c = 16 ch + n), twiddle W_512^(c b') on the inputs b' = 4g + r, DFT over b' ----
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int ch = 0; ch < 2; ++ch)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    bb[q][ch][r] = bits_h2(lds[e3r + q * 512 + r * 32 + ((ch ^ (g & 1)) * 16) + (((n >> 2) ^ r) * 4)]);
        __syncthreads();   // every wave has read its blocks: E4 (all planes, all waves) may overwrite them
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                h2 bt[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) bt[r] = cmul(bb[q][ch][r], tw2[ch][r]);
                float2 y[4];
                dft16_mfma(a16re, a16im, bt, y);
#pragma unroll
                for (int r = 0; r < 4; ++r) lds[e4w + q * kM16Ps4 + r * 32 + 16 * ch] = h2_bits(to_h2(y[r]));   // b = 4g + r
            }
        __syncthreads();
        // ---- inverse pass 1: columns j = 32 G + 2n + e, twiddle W_8192^(j a') on the inputs a' = 4g + r, DFT over a' ----
#pragma unroll
        for (int gl = 0; gl < 4; ++gl)
#pragma unroll
            for (int r = 0; r < 4; ++r) raw[gl][r] = *reinterpret_cast<const uint2*>(lds + e4r + r * kM16Ps4 + gl * 32);
        __syncthreads();   // E4 has been read: the next row's pass 1 may write E1
#pragma unroll
        for (int gl = 0; gl < 4; ++gl) {
            h2 b0[4], b1[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { b0[r] = cmul(bits_h2(raw[gl][r].x), tw1[gl][0][r]); b1[r] = cmul(bits_h2(raw[gl][r].y), tw1[gl][1][r]); }
            float2 y0[4], y1[4];
            dft16_mfma(a16re, a16im, b0, y0);
            dft16_mfma(a16re, a16im, b1, y1);
#pragma unroll
            for (int r = 0; r < 4; ++r)   // the final conjugation rides on the conversion
                buf_store_u2(rdst, voff, (unsigned)((r * 512 + 32 * gl) * 4),
                             make_uint2(h2_bits(to_h2(make_float2(y0[r].x, -y0[r].y))), h2_bits(to_h2(make_float2(y1[r].x, -y1[r].y)))));
        }
    }
}

// the needle spectrum for k2_rows_m16: conjugated, scaled, as h2 points in [row][a'][b'][c'] order (hc: the f32
// spectrum in the register order of k2_rows_r16: float2 index 2t + 512 i + e of a row holds frequency
// (a' = t >> 4, b' = t & 15, c' = brev32(2i + e)))
__global__ void __launch_bounds__(256) spectrum_to_half_mfma_kernel(const float2* __restrict__ hc, long long n, float scale, unsigned* __restrict__ out) {
    for (long long o = (long long)blockIdx.x * 256 + threadIdx.x; o < n; o += (long long)gridDim.x * 256) {
        const long long row = o >> kR16LogN2;
        const int within = (int)(o & (kN2 - 1)), tt = within >> 5, cp = within & 31;
        const int rr = brev<32>(cp), i = rr >> 1, e = rr & 1;
        const float2 v = hc[row * kN2 + 2 * tt + 512 * i + e];
        out[o] = h2_bits(to_h2(make_float2(v.x * scale, -v.y * scale)));
    }
}
hipError_t launch_spectrum_to_half_mfma(hipStream_t st, const float2* hc, long long n, float scale, unsigned* out) {
    hipLaunchKernelGGL(spectrum_to_half_mfma_kernel, dim3(2048), dim3(256), 0, st, hc, n, scale, out);
    return hipGetLastError();
}

// K2 for a group of needles against one haystack (am_match_multi_device, BASELINE
// config 4): the row is read and transformed ONCE; every needle of the group then
// multiplies that spectrum with its own and runs its own inverse transform into its
// own work matrix.  Per needle the row costs 8/n + 8 bytes of HBM traffic instead of
// 16 and the forward half of the arithmetic is shared.
#ifndef AM_K2G_PREFETCH
// 1 = request the first quarter of the next needle's spectrum row one needle ahead.  Measured and dropped
// (profiles/r03/k2_group_ab.txt): the 16 registers it holds across the inverse passes push the kernel
// from 230 to 256 VGPRs plus 15 spilled ones (64 bytes of scratch per lane, reloaded inside the needle
// loop): 2.16 ms per launch of 8 needles x 22 pairs with it, 1.60 ms without.
#define AM_K2G_PREFETCH 0
#endif
__global__ void __launch_bounds__(256, 2)
k2_rows_r16_group(const float2* __restrict__ work, K2Group grp, PlanDev pl, unsigned npairs) {
    extern __shared__ float4 lds4[];
    unsigned row, slot;
    k2_place(npairs, row, slot);
    const size_t row_off = ((size_t)slot << pl.logN) + (size_t)row * kN2;
    const __amdgpu_buffer_rsrc_t rrow = make_rsrc(work + row_off, kN2 * 8);
    const size_t hoff4 = (size_t)row * (kN2 / 2);
    const K2Lane k = k2_lane(pl);
    k2_forward12<false>(k, rrow, lds4);
    float2 z[32];
    k2_forward3(k, lds4, z);
    float4 hq[4];
    if (AM_K2G_PREFETCH) k2_fetch_quarter(make_rsrc(reinterpret_cast<const float4*>(grp.hc[0]) + hoff4, kN2 * 8), k.voff, 0, hq);
#pragma unroll 1
    for (int j = 0; j < grp.n; ++j) {
        const __amdgpu_buffer_rsrc_t rh = make_rsrc(reinterpret_cast<const float4*>(grp.hc[j]) + hoff4, kN2 * 8);
        if (!AM_K2G_PREFETCH) k2_fetch_quarter(rh, k.voff, 0, hq);
        // The twiddle powers of the inverse passes depend on the lane only; left alone
        // the compiler computes them once before the loop and keeps ~120 values alive
        // across it (in scratch).  Recomputing them per needle is far cheaper.
        K2Lane kj = k;
        asm volatile("" : "+v"(kj.wj0.x), "+v"(kj.wj0.y), "+v"(kj.wj1.x), "+v"(kj.wj1.y),
                          "+v"(kj.wc0.x), "+v"(kj.wc0.y), "+v"(kj.wc1.x), "+v"(kj.wc1.y));
        asm volatile("" : "+v"(kj.qj0.x), "+v"(kj.qj0.y), "+v"(kj.qj1.x), "+v"(kj.qj1.y),
                          "+v"(kj.qc0.x), "+v"(kj.qc0.y), "+v"(kj.qc1.x), "+v"(kj.qc1.y));
        float2 q[32];
        k2_multiply_fetch(z, rh, k.voff, hq, q);
        if (AM_K2G_PREFETCH) {
            const int jn = j + 1 < grp.n ? j + 1 : j;   // (the last needle refetches its own quarter: harmless)
            k2_fetch_quarter(make_rsrc(reinterpret_cast<const float4*>(grp.hc[jn]) + hoff4, kN2 * 8), k.voff, 0, hq);
            __builtin_amdgcn_sched_barrier(0);
        }
        k2_inverse<false, AM_K2G_STORE_AUX>(kj, q, lds4, make_rsrc(grp.dst[j] + row_off, kN2 * 8));
        __syncthreads();   // the last pass read rows of every wave: finish before the next needle overwrites them
    }
}

// v_min3 / v_max3 without the input canonicalisation the compiler puts in front of
// fminf / fmaxf on values it cannot prove canonical (anything that came out of LDS)
__device__ __forceinline__ float min3_raw(float a, float b, float c) {
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float max3_raw(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// Minimum over the wavefront's 64 lanes through DPP (no LDS round trips): four row_shr steps leave the
// minimum of each 16-lane row in its last lane, row_bcast:15 / :31 carry it across the rows into
// lane 63, which is read back as a scalar.  A lane without a source keeps its own value (`old`).
__device__ __forceinline__ float wave_min_f(float v) {
    // v_min_f32 with a DPP source: one instruction per step (through the builtin the compiler emits a copy, the DPP
    // move, a canonicalising max and the min).  dst = src1 = v, so a lane without a source (its write is
    // suppressed) keeps its value.  s_nop 1: the two wait states between a VALU write of a register and a DPP
    // read of it, which nobody inserts inside an asm block.
    asm("s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Score-scan exchange of K3: the workgroup's 256 x 32 scores of one block go through LDS
// once so that every thread ends up with the whole 32-score run of one row; (min, max) of a
// run then cost 16 + 16 three-input operations.
// The column owner (hi, cp) holds rows a*16 + hi, a = 0..15, and a wavefront holds four
// values of hi, so the 32 scores of a row are written by 16 lanes of ONE wavefront.  The row is
// therefore read back by a lane of that same wavefront -- lane l of wave w takes row
// r = (l >> 2) * 16 + 4 w + (l & 3) (* 32 in the 512-row kernels) -- and the exchange needs no workgroup barrier, only the
// wavefront's own ordering (wave_sync_lds).  16-byte chunk c of row r sits at chunk position
// c ^ s(r), s(r) = (a & 3) << 1 | ((r & 3) >> 1), a = the row's register index: with it both the 8-byte writes of the
// column owners and the 16-byte reads of the row owners are conflict-free (every 16-lane group
// of a ds_read_b128 meets all 16 chunk columns once).
// HB = log2 of the number of hi values (rows per register index a): 4 for the 256-row kernels
// (256 threads), 5 for the 512-row ones (512 threads).  Row = a * 2^HB + hi.
template <int HB>
__device__ __forceinline__ unsigned scan_swizzle(unsigned row) { return (((row >> HB) & 3u) << 1) | ((row & 3u) >> 1); }
template <int HB>
__device__ __forceinline__ int scan_row_of(int t) { return ((t & 63) >> 2) * (1 << HB) + (t >> 6) * 4 + (t & 3); }
template <int HB>
__device__ __forceinline__ void scan_put(float2* lds2, int hi, int cp, const float (&c0)[16], const float (&c1)[16]) {
    const unsigned half = (unsigned)cp & 1u, chunk = (unsigned)cp >> 1, hb = ((unsigned)hi & 3u) >> 1;
#pragma unroll
    for (int a = 0; a < 16; ++a) {
        const unsigned pos = chunk ^ ((((unsigned)a & 3u) << 1) | hb);
        lds2[a * (16 << HB) + hi * 16 + (int)((pos << 1) | half)] = make_float2(c0[a], c1[a]);
    }
}
// row = the run's row; whole = every score of the run is valid, otherwise scores at run offsets
// >= nvalid are ignored
template <int HB>
__device__ __forceinline__ void scan_row_minmax(const float4* lds4, int row, bool whole, int nvalid, float& mn, float& mx) {
    const unsigned k = scan_swizzle<HB>((unsigned)row);
    float4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = lds4[row * 8 + (int)((unsigned)i ^ k)];
    if (!whole) {
        mn = FLT_MAX; mx = -FLT_MAX;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c0 = i << 2;   // first run offset held by chunk i
            const float e[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (c0 + j < nvalid) { mn = fminf(mn, e[j]); mx = fmaxf(mx, e[j]); }
        }
        return;
    }
    mn = min3_raw(v[0].x, v[0].y, v[0].z); mx = max3_raw(v[0].x, v[0].y, v[0].z);
    mn = min3_raw(mn, v[0].w, v[1].x);     mx = max3_raw(mx, v[0].w, v[1].x);
#pragma unroll
    for (int i = 1; i < 8; ++i) {
        if (i > 1) { mn = min3_raw(mn, v[i - 1].w, v[i].x); mx = max3_raw(mx, v[i - 1].w, v[i].x); }
        mn = min3_raw(mn, v[i].y, v[i].z); mx = max3_raw(mx, v[i].y, v[i].z);
    }
    mn = fminf(mn, v[7].w); mx = fmaxf(mx, v[7].w);
}

// Does the 32-score run [lo, lo+32) hold the first (i*c) or the last
// (i*c + d) score of a reference chunk?  (audio_matcher.rs:104, 119: chunk i
// covers scores [i*c, i*c + d].)  Such runs are only partly inside a chunk, so
// the peak pick needs their raw scores, not just their summary.
__device__ __forceinline__ long long mod_recip(long long x, long long c, double inv_c) {
    // x mod c for 0 <= x < 2^50 through one f64 multiply and a fix-up
    long long q = (long long)((double)x * inv_c);
    long long r = x - q * c;
    if (r < 0) r += c;
    else if (r >= c) r -= c;
    return r;
}
__device__ __forceinline__ bool run_has_chunk_edge(long long lo, long long c, long long d, double inv_c) {
    if (c <= 0) return true;
    // a multiple of c in [lo, lo+31]  <=>  (lo + 31) mod c <= 31
    if (mod_recip(lo + 31, c, inv_c) <= 31) return true;
    const long long hi2 = lo + 31 - d;   // i*c + d in [lo, lo+31]  <=>  i*c in [lo-d, lo-d+31]
    if (hi2 < 0) return false;
    return mod_recip(hi2, c, inv_c) <= 31;
}

// K3: conjugate twiddle, inverse 256-point column FFTs, scaling (scale_slice,
// audio_matcher.rs:246-252, 306-308), crop to the block's valid lags
// (centered(), :460-464) and the per-32-score (min,max) summary.
//
// k3_tile: everything K3 does with one column tile once its rows k1 = hi + 16*b' (natural
// b' order, columns col, col+1) are in registers: conjugate pipeline twiddle,
// inverse 256-point column FFT, scaling, fused score scan, conditional raw-score
// store.  Output index of row n1, column c is n1 * out_stride + c.
struct K3Edges {
    long long outA, outB;
    int relcA, reldA, relcB, reldB;   // one_edge: the block's chunk edges relative to its first score (INT_MAX: none in reach)
    bool one_edge;
};
// Chunk edges (scores i*c and i*c + d, audio_matcher.rs:104, 119).  With chunks longer
// than a block (the usual case) each block holds at most one edge of either kind;
// their positions depend on the block only and are worked out while the tile's loads
// are still in flight.
__device__ __forceinline__ K3Edges k3_edges(const Job& job, const ScanCfg& scan, long long blkA, long long blkB) {
    K3Edges e;
    e.outA = blkA * job.hop; e.outB = blkB * job.hop;
    e.one_edge = scan.stats32 != nullptr && scan.seg_c >= (long long)job.hop + 32;
    e.relcA = e.reldA = e.relcB = e.reldB = 0x7fffffff;
    if (scan.edges_n > 0) {   // from the host (a scalar fetch out of the kernel arguments instead of f64 arithmetic in every wave)
        const int s = (int)(blkA >> 1) - job.first_pair;
        e.relcA = scan.edge_rel[s][0]; e.reldA = scan.edge_rel[s][1];
        e.relcB = scan.edge_rel[s][2]; e.reldB = scan.edge_rel[s][3];
        return e;
    }
    if (e.one_edge) {
        const long long c = scan.seg_c, d = scan.seg_d;
        const long long m = mod_recip(e.outA, c, scan.inv_c);
        const long long ecA = m == 0 ? e.outA : e.outA + (c - m);      // smallest i*c >= outA
        long long edA = d;                                             // smallest i*c + d >= outA, i >= 0
        if (e.outA > d) {
            const long long m2 = mod_recip(e.outA - d, c, scan.inv_c);
            edA = m2 == 0 ? e.outA : e.outA + (c - m2);
        }
        // block B starts hop < c later: its first edge is the same one or the next
        const long long ecB = ecA >= e.outB ? ecA : ecA + c;
        const long long edB = edA >= e.outB ? edA : edA + c;
        auto rel = [](long long edge, long long start) { const long long r = edge - start; return r < 0x7fffffffll ? (int)r : 0x7fffffff; };
        e.relcA = rel(ecA, e.outA); e.reldA = rel(edA, e.outA);
        e.relcB = rel(ecB, e.outB); e.reldB = rel(edB, e.outB);
    }
    return e;
}

// The end of K3 for one column tile: x0 / x1[brev(a)] hold the correlation values of rows
// n1 = a * 2^HB + hi (columns col, col+1; real part = block A, imaginary part = block B):
// scaling, fused score scan, block vote, conditional raw-score store.
// ACC: the scores are added to what job.dst already holds (needle partitioning: the correlation with a long
// needle is the sum of the correlations with its segments, each on a shifted source).  Every score is written
// and the fused scan is left out -- scan and accumulation together do not fit the 128 registers of the
// 1024-thread kernel (over a thousand spilled values); the sums get their summary from tile_stats instead.
// SCALED: the caller has folded out_scale into the transform (its first twiddle), the values are scores already.
template <int HB, typename T, bool ACC = false, bool SCALED = false>
__device__ __forceinline__ void k3_finish(const Job& job, const ScanCfg& scan, const K3Edges& ed, float2* lds2,
                                          int n2_0, int out_stride, int t, long long blkA, long long blkB,
                                          float out_scale, const T (&x0)[16], const T (&x1)[16]) {
    const int hi = t >> 4, cp = t & 15;
    const long long col = n2_0 + 2 * cp;
    const long long outA = ed.outA, outB = ed.outB;
    const bool one_edge = ed.one_edge;
    const bool dst8 = ((reinterpret_cast<uintptr_t>(job.dst) & 7) == 0) && ((job.hop & 1) == 0);
    long long limA = job.out_count - outA; if (limA > job.hop) limA = job.hop;
    long long limB = blkB < job.nblocks ? job.out_count - outB : 0; if (limB > job.hop) limB = job.hop;
    // scores of row n1 = a*16 + hi: block A columns col, col+1 = (sa0, sa1), block B = (sb0, sb1)
    float sa0[16], sa1[16], sb0[16], sb1[16];
#pragma unroll
    for (int a = 0; a < 16; ++a) {
        const T v0 = x0[brev<16>(a)], v1 = x1[brev<16>(a)];
        if (SCALED) {
            sa0[a] = v0.x; sa1[a] = v1.x;
            sb0[a] = v0.y; sb1[a] = v1.y;
        } else {
            sa0[a] = v0.x * out_scale; sa1[a] = v1.x * out_scale;
            sb0[a] = v0.y * out_scale; sb1[a] = v1.y * out_scale;
        }
    }
    // which of this thread's 16 rows leave the chip as raw scores, per block: bit 4 * (a & 7) of word a >> 3 = row
    // a * 2^HB + hi (the layout the wavefront's ballot has once it is shifted by hi & 3, see below)
    unsigned wantA[2] = {0x11111111u, 0x11111111u}, wantB[2] = {0x11111111u, 0x11111111u};
    if (!ACC && scan.stats32 != nullptr) {   // (the accumulating form writes plain scores: its sums are summarised by tile_stats)
        // ---- fused score scan: (min,max) per 32 consecutive scores ------------
        // this thread owns the run of row n1 = row (scores row*out_stride + n2_0 .. +31 of both blocks)
        const int row = scan_row_of<HB>(t);
        const int rowrun = row * out_stride + n2_0;   // (below the block length: 32-bit arithmetic from here on)
        // Every run is wholly valid or wholly invalid, except in the block that holds
        // the end of the score array.
        const int leftA = (int)limA - rowrun, leftB = (int)limB - rowrun;
        const bool wholeA = leftA >= 32 || leftA <= 0, wholeB = leftB >= 32 || leftB <= 0;
        const float4* lds4 = reinterpret_cast<const float4*>(lds2);
        float rmnA, rmxA, rmnB, rmxB;
        __syncthreads();   // the column exchange above is finished with the tile (it crosses wavefronts)
        scan_put<HB>(lds2, hi, cp, sa0, sa1);
        wave_sync_lds();   // a row is written and read by lanes of one wavefront
        scan_row_minmax<HB>(lds4, row, wholeA, leftA < 32 ? leftA : 32, rmnA, rmxA);
        wave_sync_lds();
        scan_put<HB>(lds2, hi, cp, sb0, sb1);
        wave_sync_lds();
        scan_row_minmax<HB>(lds4, row, wholeB, leftB < 32 ? leftB : 32, rmnB, rmxB);
        // Raw scores leave the chip only for RUNS that can matter to the peak pick: a run whose maximum
        // reaches the tile's write threshold, or one that straddles a chunk edge.  The threshold is the
        // tile's own: its 2^(HB+4) runs are spread evenly over the whole block (one every out_stride
        // scores), so their minimum is a good estimate of the minimum of the chunks in that block, and
        // the threshold sits `margin` (half a prominence) above it.  The pick checks, per chunk, that no
        // tile's threshold was too high (the certificate in peaks_kernel): estimate and decision need
        // no history and no second pass.
        float tminA = wave_min_f(leftA > 0 ? rmnA : FLT_MAX), tminB = wave_min_f(leftB > 0 ? rmnB : FLT_MAX);
        // every wavefront leaves its two minima in the first words of a scan row of its own (its lanes are
        // done reading it: the reads above were waited for before the minima could be formed), so the
        // kernel needs no LDS beyond the tile
        float* slots = reinterpret_cast<float*>(lds2 + scan_row_of<HB>(t & ~63) * 16);   // row of the wave's lane 0
        wave_sync_lds();
        if ((t & 63) == 0) { slots[0] = tminA; slots[1] = tminB; }
        __syncthreads();
#pragma unroll
        for (int w = 0; w < (1 << (HB - 2)); ++w) {
            const float* v = reinterpret_cast<const float*>(lds2 + scan_row_of<HB>(w * 64) * 16);
            tminA = fminf(tminA, v[0]);
            tminB = fminf(tminB, v[1]);
        }
        // (hist_min: the lowest chunk minimum of the needle's last few haystacks, or FLT_MAX -- it keeps a
        // chunk whose minimum lies in another block pair than this tile's scores inside its certificate)
        const bool dense = scan.margin < 0.0f;
        const float thA = dense ? -FLT_MAX : fminf(fminf(tminA, tminB), scan.hist_min) + scan.margin, thB = thA;
        bool edgeA, edgeB;
        if (one_edge) {   // edge positions relative to the block start, INT_MAX when beyond any run of the block
            edgeA = leftA > 0 && ((unsigned)(ed.relcA - rowrun) <= 31u || (unsigned)(ed.reldA - rowrun) <= 31u);
            edgeB = leftB > 0 && ((unsigned)(ed.relcB - rowrun) <= 31u || (unsigned)(ed.reldB - rowrun) <= 31u);
        } else {
            edgeA = leftA > 0 && run_has_chunk_edge(outA + rowrun, scan.seg_c, scan.seg_d, scan.inv_c);
            edgeB = leftB > 0 && run_has_chunk_edge(outB + rowrun, scan.seg_c, scan.seg_d, scan.inv_c);
        }
        const bool pa = (leftA > 0 && rmxA >= thA) || edgeA, pb = (leftB > 0 && rmxB >= thB) || edgeB;
        // The owner of row a * 2^HB + hi is lane (a << 2) | (hi & 3) of the wavefront that holds the column
        // owners (hi, *) (scan_row_of): the row's 16 writers read its decision out of a ballot.
        const unsigned long long ba = __ballot(pa), bb = __ballot(pb);
        const unsigned sh = (unsigned)hi & 3u;
        wantA[0] = ((unsigned)ba >> sh) & 0x11111111u; wantA[1] = ((unsigned)(ba >> 32) >> sh) & 0x11111111u;
        wantB[0] = ((unsigned)bb >> sh) & 0x11111111u; wantB[1] = ((unsigned)(bb >> 32) >> sh) & 0x11111111u;
        // what was written, for the peak pick: the threshold per (block, tile) and the wavefront's ballot
        // (bit (a << 2) | j = row a * 2^HB + 4 w + j of the tile) per (block, tile, wavefront)
        const long long tileA = blkA * (out_stride >> kColsLog) + ((unsigned)n2_0 >> kColsLog), tileB = tileA + (out_stride >> kColsLog);
        if (t == 0 && scan.tile_theta != nullptr) {
            scan.tile_theta[tileA] = thA;
            if (blkB < job.nblocks) scan.tile_theta[tileB] = thB;
        }
        if ((t & 63) == 0 && scan.wbits != nullptr) {
            scan.wbits[(tileA << (HB - 2)) + (t >> 6)] = ba;
            if (blkB < job.nblocks) scan.wbits[(tileB << (HB - 2)) + (t >> 6)] = bb;
        }
        if (leftA > 0) scan.stats32[(outA + rowrun) >> 5] = make_float2(rmnA, rmxA);
        if (leftB > 0) scan.stats32[(outB + rowrun) >> 5] = make_float2(rmnB, rmxB);
    }
    if (wantA[0] | wantA[1]) {
#pragma unroll
        for (int a = 0; a < 16; ++a) {
            if (!((wantA[a >> 3] >> ((a & 7) * 4)) & 1u)) continue;
            const long long n = (long long)(a * (1 << HB) + hi) * out_stride + col;
            // hop and out offsets are even whenever this kernel is used, so a pair is
            // valid or invalid as a whole except at the very end of the score array
            // (ACC: read-modify-write right here -- the sums are needed nowhere else, see above)
            if (dst8 && n + 1 < limA) {
                float2* ptr = reinterpret_cast<float2*>(job.dst + outA + n);
                float2 v = make_float2(sa0[a], sa1[a]);
                if (ACC) { const float2 o = *ptr; v.x += o.x; v.y += o.y; }
                *ptr = v;
            } else {
                if (n < limA) job.dst[outA + n] = sa0[a] + (ACC ? job.dst[outA + n] : 0.0f);
                if (n + 1 < limA) job.dst[outA + n + 1] = sa1[a] + (ACC ? job.dst[outA + n + 1] : 0.0f);
            }
            if (ACC && (a & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // (keeps the read-modify-writes from being hoisted all at once)
        }
    }
    if (wantB[0] | wantB[1]) {
#pragma unroll
        for (int a = 0; a < 16; ++a) {
            if (!((wantB[a >> 3] >> ((a & 7) * 4)) & 1u)) continue;
            const long long n = (long long)(a * (1 << HB) + hi) * out_stride + col;
            if (dst8 && n + 1 < limB) {
                float2* ptr = reinterpret_cast<float2*>(job.dst + outB + n);
                float2 v = make_float2(sb0[a], sb1[a]);
                if (ACC) { const float2 o = *ptr; v.x += o.x; v.y += o.y; }
                *ptr = v;
            } else {
                if (n < limB) job.dst[outB + n] = sb0[a] + (ACC ? job.dst[outB + n] : 0.0f);
                if (n + 1 < limB) job.dst[outB + n + 1] = sb1[a] + (ACC ? job.dst[outB + n + 1] : 0.0f);
            }
            if (ACC && (a & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
    }
}


// k3_tile: everything the 256-row K3 kernels do with one column tile once its rows
// k1 = hi + 16*b' (natural b' order, columns col, col+1) are in registers: conjugate
// pipeline twiddle, inverse 256-point column FFT, then k3_finish.  Output index of row n1,
// column c is n1 * out_stride + c.
template <bool ACC = false>
// `geo` (needle-group launch): where the chunk edges come from -- the kernel argument itself, whose edge table is
// indexed with a run-time slot; `scan` is then a per-needle copy whose pointers differ (a copy that were indexed
// like that would have to live in scratch memory).
__device__ __forceinline__ void k3_tile(const Job& job, const PlanDev& pl, const ScanCfg& scan, float2* lds2,
                                        int n2_0, int out_stride, int t, long long blkA, long long blkB,
                                        float out_scale, float2 (&x0)[16], float2 (&x1)[16], const ScanCfg* geo = nullptr) {
    const int hi = t >> 4, cp = t & 15;
    const long long N = 1ll << pl.logN;
    const long long col = n2_0 + 2 * cp;
    const K3Edges ed = k3_edges(job, geo ? *geo : scan, blkA, blkB);
    const unsigned maskN = (unsigned)(N - 1);
    const float2 w256 = pl.tw1[hi], w256q = pl.tw1[4 * hi];   // (fourth powers: see twiddle_apply(x, w, w4))
    {
        const unsigned n2 = (unsigned)col;
        float2 base0 = tw_big(pl, (n2 * (unsigned)hi) & maskN);
        float2 base1 = tw_big(pl, ((n2 + 1) * (unsigned)hi) & maskN);
        base0.x *= out_scale; base0.y *= out_scale;   // (out_scale rides on the pipeline twiddle, as in the 512-row kernel)
        base1.x *= out_scale; base1.y *= out_scale;
        float2 step0, step1, step0q, step1q;
        tw_big_pair(pl, (n2 * 16u) & maskN, step0, step0q);
        tw_big_pair(pl, ((n2 + 1) * 16u) & maskN, step1, step1q);
        // (column 0 through its first pass and into LDS before column 1 is touched, its second pass while column 1
        // crosses: arithmetic between every two barriers, as in the 512-row kernel)
        twiddle_chain<16, true, false>(x0, base0, step0, step0q);
        dif<16, true>(x0);   // b at x[brev(b)]
        // exchange, one column of the pair at a time (32 KB of LDS per workgroup);
        // afterwards ownership is b = hi, a' = 0..15
#pragma unroll
        for (int b = 0; b < 16; ++b) lds2[(hi * 16 + b) * 16 + cp] = x0[brev<16>(b)];
        twiddle_chain<16, true, false>(x1, base1, step1, step1q);
    }
    dif<16, true>(x1);
    __syncthreads();
#pragma unroll
    for (int ap = 0; ap < 16; ++ap) x0[ap] = lds2[(ap * 16 + hi) * 16 + cp];
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 16; ++b) lds2[(hi * 16 + b) * 16 + cp] = x1[brev<16>(b)];
    twiddle_nat<16, true>(x0, w256, w256q);
    dif<16, true>(x0);   // a at x[brev(a)], n1 = a*16 + b
    __syncthreads();
#pragma unroll
    for (int ap = 0; ap < 16; ++ap) x1[ap] = lds2[(ap * 16 + hi) * 16 + cp];
    twiddle_nat<16, true>(x1, w256, w256q);
    dif<16, true>(x1);
    k3_finish<4, float2, ACC, true>(job, scan, ed, lds2, n2_0, out_stride, t, blkA, blkB, out_scale, x0, x1);
}

// The same with the first pass (pipeline twiddle, 16-point transform over b') and the exchange on
// packed half-precision points (option half_pipeline = 2): both columns of the pair cross LDS in one
// exchange; the second pass and everything behind it stay f32, so that no score is rounded to f16.
__device__ __forceinline__ void k3_tile_h16(const Job& job, const PlanDev& pl, const ScanCfg& scan, float2* lds2,
                                            int n2_0, int out_stride, int t, long long blkA, long long blkB,
                                            float out_scale, h2 (&hx0)[16], h2 (&hx1)[16]) {
    const int hi = t >> 4, cp = t & 15;
    const long long N = 1ll << pl.logN;
    const long long col = n2_0 + 2 * cp;
    const K3Edges ed = k3_edges(job, scan, blkA, blkB);
    const unsigned maskN = (unsigned)(N - 1);
    const float2 w256 = pl.tw1[hi];
    {
        const unsigned n2 = (unsigned)col;
        const float2 base0 = tw_big(pl, (n2 * (unsigned)hi) & maskN);
        const float2 base1 = tw_big(pl, ((n2 + 1) * (unsigned)hi) & maskN);
        const float2 step0 = tw_big(pl, (n2 * 16u) & maskN);
        const float2 step1 = tw_big(pl, ((n2 + 1) * 16u) & maskN);
        twiddle_chain<16, true, false>(hx0, base0, step0);
        twiddle_chain<16, true, false>(hx1, base1, step1);
    }
    dif<16, true>(hx0);
    dif<16, true>(hx1);
    uint2* ldsu = reinterpret_cast<uint2*>(lds2);
#pragma unroll
    for (int b = 0; b < 16; ++b) ldsu[(hi * 16 + b) * 16 + cp] = make_uint2(h2_bits(hx0[brev<16>(b)]), h2_bits(hx1[brev<16>(b)]));
    __syncthreads();
    float2 x0[16], x1[16];
#pragma unroll
    for (int ap = 0; ap < 16; ++ap) {
        const uint2 v = ldsu[(ap * 16 + hi) * 16 + cp];
        x0[ap] = to_f2(bits_h2(v.x));
        x1[ap] = to_f2(bits_h2(v.y));
    }
    twiddle_nat<16, true>(x0, w256);
    twiddle_nat<16, true>(x1, w256);
    dif<16, true>(x0);
    dif<16, true>(x1);
    k3_finish<4>(job, scan, ed, lds2, n2_0, out_stride, t, blkA, blkB, out_scale, x0, x1);
}

#ifndef AM_K3_WGS
#define AM_K3_WGS 3   // waves per SIMD the register allocation has to allow (= workgroups per CU for 256 threads; it uses 118 VGPRs: four fit)
#endif
// One column tile of the 256-row K3; lin = the tile's number in the launch (blockIdx.x in the pipeline's launches).
template <int HALF, bool ACC>   // 0 = f32 work matrix, 1 = f16 storage, 2 = f16 storage and an f16 first pass
__device__ __forceinline__ void k3_cols_inv_r16_tile(unsigned lin, float4* lds4, const Job& job, const float2* __restrict__ work,
                                                     const PlanDev& pl, float out_scale, const ScanCfg& scan, const ScanCfg* geo = nullptr) {
    const int t = threadIdx.x;
    const int hi = t >> 4, cp = t & 15;
    // XCD-aware placement (speed only): the 16 adjacent column tiles that share
    // one 128-byte line of stats32 run on the same XCD, so the line is merged in
    // that L2 before it is written back.  256 tiles per pair = 8 XCDs x 2 x 16.
    const unsigned xcd = lin & 7u, seq = lin >> 3;
    const unsigned slot = seq >> 5, half = (seq >> 4) & 1u, tl = seq & 15u;
    const int n2_0 = (int)(((half * 8u + xcd) * 16u + tl) << kColsLog);
    const int pair = job.first_pair + (int)slot;
    if (scan.only_pairs != nullptr && scan.only_pairs[pair] == 0) return;   // device-side redo: flagged pairs only
    const long long blkA = 2ll * pair, blkB = blkA + 1;
    if constexpr (HALF == 2) {
        const uint2* __restrict__ in2 = reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned*>(work) + ((size_t)slot << pl.logN) + n2_0) + cp;
        h2 hx0[16], hx1[16];
#pragma unroll
        for (int bp = 0; bp < 16; ++bp) {
            const uint2 v = in2[(size_t)(hi + 16 * bp) * (kN2 / 2)];
            hx0[bp] = bits_h2(v.x);
            hx1[bp] = bits_h2(v.y);
        }
        k3_tile_h16(job, pl, scan, reinterpret_cast<float2*>(lds4), n2_0, kN2, t, blkA, blkB, out_scale, hx0, hx1);
        return;
    }
    float2 x0[16], x1[16];
    if (HALF) {
        const uint2* __restrict__ in2 = reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned*>(work) + ((size_t)slot << pl.logN) + n2_0) + cp;
#pragma unroll
        for (int bp = 0; bp < 16; ++bp) {
            const uint2 v = in2[(size_t)(hi + 16 * bp) * (kN2 / 2)];
            x0[bp] = unpack_h2(v.x);
            x1[bp] = unpack_h2(v.y);
        }
    } else {
        // (buffer loads off one address register, as in the 512-row kernel)
        const __amdgpu_buffer_rsrc_t rin = make_rsrc(work + ((size_t)slot << pl.logN) + n2_0, (unsigned)((((size_t)1 << pl.logN) - n2_0) * 8));
        const unsigned voff = (unsigned)hi * (kN2 * 8u) + (unsigned)cp * 16u;
#pragma unroll
        for (int bp = 0; bp < 16; ++bp) {   // rows k1 = a' + 16*b', a' = hi
            const float4 v = buf_load4<AM_K3_LOAD_NT ? 2 : 0>(rin, voff, (unsigned)bp * (16u * kN2 * 8u));
            x0[bp] = make_float2(v.x, v.y);
            x1[bp] = make_float2(v.z, v.w);
        }
    }
    k3_tile<ACC>(job, pl, scan, reinterpret_cast<float2*>(lds4), n2_0, kN2, t, blkA, blkB, out_scale, x0, x1, geo);
}
// The kernel: one tile per workgroup.  REDO = the device-side redo's instantiation (a name of its own in kernel
// traces): a SMALL grid walks the launch's tiles and runs those of the flagged pairs (scan.only_pairs) -- most
// launches of it find nothing to do, and 512 workgroups that look at 22 flags are gone sooner than 5632 that each
// need a workgroup's worth of LDS and registers to find out.
template <int HALF, bool ACC = false, int REDO = 0>
__global__ void __launch_bounds__(256, REDO ? 4 : AM_K3_WGS)
k3_cols_inv_r16(Job job, const float2* __restrict__ work, PlanDev pl, float out_scale, ScanCfg scan) {
    extern __shared__ float4 lds4[];
    if constexpr (REDO != 0) {
        for (unsigned lin = blockIdx.x; lin < (unsigned)scan.redo_tiles; lin += gridDim.x) {
            k3_cols_inv_r16_tile<HALF, ACC>(lin, lds4, job, work, pl, out_scale, scan);
            __syncthreads();   // the next tile's exchanges overwrite the scan rows other wavefronts may still read
        }
    } else {
        k3_cols_inv_r16_tile<HALF, ACC>(blockIdx.x, lds4, job, work, pl, out_scale, scan);
    }
}

// the same for the 256-row plan (see k3_cols_inv_c512_group)
__global__ void __launch_bounds__(256, AM_K3_WGS)
k3_cols_inv_r16_group(Job job, PlanDev pl, ScanCfg scan, K3Group grp) {
    extern __shared__ float4 lds4[];
    const unsigned z = blockIdx.y;
    job.dst = grp.dst[z];
    ScanCfg mine;
    mine.stats32 = grp.stats32[z]; mine.wbits = grp.wbits[z]; mine.tile_theta = grp.tile_theta[z]; mine.hist_min = grp.hist_min[z];
    mine.margin = scan.margin; mine.seg_c = scan.seg_c; mine.seg_d = scan.seg_d; mine.inv_c = scan.inv_c;
    mine.only_pairs = nullptr; mine.redo_tiles = 0; mine.edges_n = 0;
    k3_cols_inv_r16_tile<0, false>(blockIdx.x, lds4, job, grp.work[z], pl, grp.out_scale[z], mine, &scan);
}

// ... and for the odd last blocks of several haystacks (TailBatch): entry z is one pair of a job of its own, every run
// written, the summary at the place the entry names.
template <int HALF>
__global__ void __launch_bounds__(256, AM_K3_WGS)
tail_cols_inv_r16(TailBatch tb, int hop, const float2* __restrict__ work, PlanDev pl, float out_scale) {
    extern __shared__ float4 lds4[];
    const unsigned z = blockIdx.y;
    Job job;
    job.src = nullptr; job.src_len = 0; job.lead = 0; job.dst = tb.dst[z]; job.out_count = tb.out_count[z];
    job.hop = hop; job.nblocks = (int)((tb.out_count[z] + hop - 1) / hop); job.first_pair = 0; job.src_kind = 0;
    ScanCfg mine;
    mine.stats32 = tb.stats32[z]; mine.wbits = nullptr; mine.tile_theta = nullptr; mine.margin = -1.0f; mine.hist_min = FLT_MAX;
    mine.seg_c = 0; mine.seg_d = 0; mine.inv_c = 0.0; mine.only_pairs = nullptr; mine.redo_tiles = 0; mine.edges_n = 0;
    // (one point of the work matrix is 8 bytes, or 4 with half storage)
    const float2* mywork = HALF ? reinterpret_cast<const float2*>(reinterpret_cast<const unsigned*>(work) + ((size_t)z << pl.logN))
                                : work + ((size_t)z << pl.logN);
    k3_cols_inv_r16_tile<HALF, false>(blockIdx.x, lds4, job, mywork, pl, out_scale, mine);
}

// A tail's scores and summary into the score-side buffers of its haystack, the main layout's ballots and thresholds
// of that block to "every run written" (TailBatch; am_api.hip, match_many)
__global__ void __launch_bounds__(256)
tail_commit_kernel(const float* __restrict__ tsc, float* __restrict__ sc, long long n, const float2* __restrict__ tst, float2* __restrict__ st32,
                   unsigned long long* __restrict__ wbits, long long words, float* __restrict__ theta, int tiles) {
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x, nth = (long long)gridDim.x * blockDim.x;
    const long long n4 = n >> 2;   // (both score pointers are 16-byte aligned: the tail starts at a multiple of 1024)
    const float4* __restrict__ t4 = reinterpret_cast<const float4*>(tsc);
    float4* __restrict__ s4 = reinterpret_cast<float4*>(sc);
    for (long long i = tid; i < n4; i += nth) s4[i] = t4[i];
    for (long long i = (n4 << 2) + tid; i < n; i += nth) sc[i] = tsc[i];
    const long long n32 = (n + 31) >> 5;
    for (long long i = tid; i < n32; i += nth) st32[i] = tst[i];
    if (wbits != nullptr)
        for (long long i = tid; i < words; i += nth) wbits[i] = ~0ull;
    if (theta != nullptr)
        for (long long i = tid; i < tiles; i += nth) theta[i] = -FLT_MAX;
}

__global__ void __launch_bounds__(256)
tail_preset_group_kernel(K3Group grp, long long blk, int log_n1, int log_n2) {
    const unsigned z = blockIdx.y;
    const long long tiles = 1ll << (log_n2 - kColsLog), words = tiles << (log_n1 - 6);
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x, nth = (long long)gridDim.x * blockDim.x;
    unsigned long long* __restrict__ wb = grp.wbits[z] + blk * words;
    float* __restrict__ th = grp.tile_theta[z] + blk * tiles;
    for (long long i = tid; i < words; i += nth) wb[i] = ~0ull;
    for (long long i = tid; i < tiles; i += nth) th[i] = -FLT_MAX;
}

// ===========================================================================
// N = 2^22 = 512 x 8192: column kernels with 512 threads (two waves
// per SIMD; two workgroups = 16 waves per CU, like the 256-thread kernels at four) that hold 16
// points of two columns per thread exactly as the 256-row kernels do.  The 512-point column
// transform is 16 x 32: n1 = a*32 + b, k1 = a' + 16*b',
//   W_512^(n1*k1) = W_16^(a*a') * W_512^(b*a') * W_32^(b*b'),
// a 16-point pass over a in registers, the twiddle W_512^(b*a'), an LDS exchange, and the
// 32-point pass over b as one radix-2 stage + a 16-point pass: thread (a', half) reads all 32
// values of its a' and forms x[b] + x[b+16] (half 0: even b') or (x[b] - x[b+16]) W_32^b (half 1:
// odd b').  K2 sees 512 ordinary 8192-point rows per pair; K3 reads every row once.
// ===========================================================================
constexpr int kC512Slab = 560;   // float2 per a' slab: 32 rows of 17 + 16, so that consecutive a' sit 32 banks apart
__device__ __forceinline__ int c512_idx(int ap, int b, int cp) { return ap * kC512Slab + b * 17 + cp; }
constexpr int kC512Lds = (15 * kC512Slab + 31 * 17 + 16) * 8;
static_assert(kC512Lds >= 512 * 16 * 8, "K3's score scan needs 512 rows x 32 scores");
// K3's exchange runs the other way round (written per a' slab by lanes that differ in a', read by
// lanes that differ in b): rows of 16 put consecutive b 32 banks apart for the reads, and a slab of
// 32 * 16 + 16 does the same for consecutive a' in the writes.  (With K1's row stride of 17 a
// quarter of K3's LDS cycles were bank conflicts: the two rows of a 32-lane group overlapped in
// two banks.)
constexpr int kC512Slab3 = 32 * 16 + 16;
__device__ __forceinline__ int c512_idx3(int ap, int b, int cp) { return ap * kC512Slab3 + b * 16 + cp; }
static_assert((15 * kC512Slab3 + 31 * 16 + 16) * 8 <= kC512Lds, "K3's exchange fits the kernel's LDS");

// LN2: log2 of the row length (13: N = 2^22 = 512 x 8192, the production plan; 14: N = 2^23 = 512 x 16384, see
// am_debug_column_bench -- the column kernels do not care how long a row is, only the strides change)
template <int KIND, int HALF, int LN2 = kR16LogN2>   // HALF: 0 = f32 work matrix, 1 = f16 storage, 2 = f16 storage and f16 butterflies
// (second argument: waves per SIMD = two workgroups per CU: held to 128 VGPRs -- left alone the allocator takes
// 130 - 150 and the kernel runs alone on its CU)
__global__ void __launch_bounds__(512, 4)
k1_cols_fwd_c512(Job job, float2* __restrict__ work, PlanDev pl) {
    constexpr int kN2 = 1 << LN2;   // (shadows the 8192 of the other plans)
    extern __shared__ float4 lds4[];
    float2* lds2 = reinterpret_cast<float2*>(lds4);
    const int t = threadIdx.x;
    const int hi = t >> 4, cp = t & 15;             // pass 1: b = hi (0..31)
    const int ap = hi & 15, half = hi >> 4;         // pass 2: a' and the parity of b'
    const int k10 = ap + 16 * half;                 // k1 = k10 + 32 * beta
    const int n2_0 = blockIdx.x << kColsLog;
    const int pair = job.first_pair + blockIdx.y;
    const long long blkA = 2ll * pair, blkB = blkA + 1;
    const bool validB = blkB < job.nblocks;
    const long long N = 1ll << pl.logN;
    const long long baseA = blkA * job.hop - job.lead;
    const long long baseB = blkB * job.hop - job.lead;
    const bool fast = ((reinterpret_cast<uintptr_t>(job.src) & 7) == 0) && ((baseA & 1) == 0) && ((baseB & 1) == 0) &&
                      baseA >= 0 && baseA + N <= job.src_len && validB && baseB + N <= job.src_len;
    const long long col = (long long)n2_0 + 2 * cp;
    const unsigned maskN = (unsigned)(N - 1);
    const float2 w512 = pl.tw1[hi];                 // W_512^b
    const float2 w512q = pl.tw1[4 * hi];            // (fourth powers: see twiddle_apply(x, w, w4))
    const float2 base0 = tw_big(pl, ((unsigned)col * (unsigned)k10) & maskN);
    const float2 base1 = tw_big(pl, (((unsigned)col + 1u) * (unsigned)k10) & maskN);
    // W_N^(32 n2) per column, and its fourth power: the f32 forms fetch the two behind the last pass (held from the
    // start of the kernel like the other table values they push it past 128 registers)
    float2 step0, step1, step0q, step1q;
    if constexpr (HALF == 2) {
        step0 = tw_big(pl, ((unsigned)col * 32u) & maskN);
        step1 = tw_big(pl, (((unsigned)col + 1u) * 32u) & maskN);
    }
    float2 x0[16], x1[16];
    if (fast && KIND == 1) {
        // i16 stereo: buffer loads off one address register -- row a's offset (a MB) and block B's (hop frames) ride
        // in SGPRs, so the 32 requests leave back to back instead of behind a 64-bit address addition each.  K1 with
        // the down-mix and the f16 butterflies 0.255 -> 0.246 ms; the f32 form, which waits on memory either way,
        // measures the same with both kinds of load and keeps the plain ones below
        // (profiles/r03/k2_planes_k3_diet_ab.txt).
        const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(static_cast<const char*>(job.src) + 4 * baseA, (unsigned)(4 * (N + job.hop)));
        const unsigned voff = 4u * ((unsigned)hi * kN2 + (unsigned)col), offB = 4u * (unsigned)job.hop;
#pragma unroll
        for (int a = 0; a < 16; ++a) {
            const float2 va = decode_sample2<KIND>(buf_load_u2<AM_K1_LOAD_NT ? 2 : 0>(rsrc, voff, (unsigned)a * (4u * 32u * kN2)));
            const float2 vb = decode_sample2<KIND>(buf_load_u2<AM_K1_LOAD_NT ? 2 : 0>(rsrc, voff, (unsigned)a * (4u * 32u * kN2) + offB));
            x0[a] = make_float2(va.x, vb.x);
            x1[a] = make_float2(va.y, vb.y);
        }
    } else if (fast) {
#pragma unroll
        for (int a = 0; a < 16; ++a) {
            const long long off = (long long)(a * 32 + hi) * kN2 + col;
            const float2 va = load_sample2<KIND>(job.src, baseA + off), vb = load_sample2<KIND>(job.src, baseB + off);
            x0[a] = make_float2(va.x, vb.x);
            x1[a] = make_float2(va.y, vb.y);
        }
    } else {
#pragma unroll
        for (int a = 0; a < 16; ++a) {
            const long long n = (long long)(a * 32 + hi) * kN2 + col;
            const float2 va = load2_padded<KIND>(job.src, baseA + n, job.src_len);
            const float2 vb = validB ? load2_padded<KIND>(job.src, baseB + n, job.src_len) : make_float2(0.f, 0.f);
            x0[a] = make_float2(va.x, vb.x);
            x1[a] = make_float2(va.y, vb.y);
        }
    }
    if constexpr (HALF == 2) {
        // the same transform on packed half-precision points; both columns of the pair cross LDS
        // together (8 bytes per element, as one f32 column does): one exchange instead of two
        h2 hx0[16], hx1[16];
#pragma unroll
        for (int a = 0; a < 16; ++a) { hx0[a] = to_h2(x0[a]); hx1[a] = to_h2(x1[a]); }
        dif<16, false>(hx0);
        dif<16, false>(hx1);
        twiddle_apply2<16, false, true>(hx0, hx1, w512);
        uint2* ldsu = reinterpret_cast<uint2*>(lds4);
#pragma unroll
        for (int a2 = 0; a2 < 16; ++a2) ldsu[c512_idx(a2, hi, cp)] = make_uint2(h2_bits(hx0[brev<16>(a2)]), h2_bits(hx1[brev<16>(a2)]));
        __syncthreads();
#pragma unroll
        for (int b = 0; b < 16; ++b) {
            const uint2 lo = ldsu[c512_idx(ap, b, cp)], up = ldsu[c512_idx(ap, b + 16, cp)];
            hx0[b] = half ? mul_w32<false>(csub(bits_h2(lo.x), bits_h2(up.x)), b) : cadd(bits_h2(lo.x), bits_h2(up.x));
            hx1[b] = half ? mul_w32<false>(csub(bits_h2(lo.y), bits_h2(up.y)), b) : cadd(bits_h2(lo.y), bits_h2(up.y));
        }
        dif<16, false>(hx0);
        dif<16, false>(hx1);
        twiddle_chain<16, false, true>(hx0, base0, step0);
        twiddle_chain<16, false, true>(hx1, base1, step1);
        uint2* __restrict__ out2 = reinterpret_cast<uint2*>(reinterpret_cast<unsigned*>(work) + ((size_t)blockIdx.y << pl.logN) + n2_0) + cp;
#pragma unroll
        for (int bt = 0; bt < 16; ++bt) {
            const size_t k1 = (size_t)(k10 + 32 * bt);
            out2[k1 * (kN2 / 2)] = make_uint2(h2_bits(hx0[brev<16>(bt)]), h2_bits(hx1[brev<16>(bt)]));
        }
        return;
    }
    dif<16, false>(x0);
    dif<16, false>(x1);
    twiddle_brev<16, false>(x0, w512, w512q);   // W_512^(b*a')
    twiddle_brev<16, false>(x1, w512, w512q);
    // exchange, one column of the pair at a time; afterwards thread (a', half) holds
    // w[b] = x[b] + x[b+16] or (x[b] - x[b+16]) W_32^b, b = 0..15
    float2 y0[16], y1[16];
#pragma unroll
    for (int a2 = 0; a2 < 16; ++a2) lds2[c512_idx(a2, hi, cp)] = x0[brev<16>(a2)];
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 16; ++b) {
        const float2 lo = lds2[c512_idx(ap, b, cp)], up = lds2[c512_idx(ap, b + 16, cp)];
        y0[b] = half ? mul_w32<false>(csub(lo, up), b) : cadd(lo, up);
    }
    __syncthreads();
#pragma unroll
    for (int a2 = 0; a2 < 16; ++a2) lds2[c512_idx(a2, hi, cp)] = x1[brev<16>(a2)];
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 16; ++b) {
        const float2 lo = lds2[c512_idx(ap, b, cp)], up = lds2[c512_idx(ap, b + 16, cp)];
        y1[b] = half ? mul_w32<false>(csub(lo, up), b) : cadd(lo, up);
    }
    __builtin_amdgcn_sched_barrier(0);
    tw_big_pair(pl, ((unsigned)col * 32u) & maskN, step0, step0q);   // (in flight behind the last pass)
    tw_big_pair(pl, (((unsigned)col + 1u) * 32u) & maskN, step1, step1q);
    __builtin_amdgcn_sched_barrier(0);
    dif<16, false>(y0);   // beta at y[brev(beta)], b' = 2*beta + half
    dif<16, false>(y1);
    // W_N^(n2*k1) = W_N^(n2*k10) * (W_N^(32*n2))^beta
    twiddle_chain<16, false, true>(y0, base0, step0, step0q);
    twiddle_chain<16, false, true>(y1, base1, step1, step1q);
    if (HALF) {
        uint2* __restrict__ out2 = reinterpret_cast<uint2*>(reinterpret_cast<unsigned*>(work) + ((size_t)blockIdx.y << pl.logN) + n2_0) + cp;
#pragma unroll
        for (int bt = 0; bt < 16; ++bt) {
            const size_t k1 = (size_t)(k10 + 32 * bt);
            out2[k1 * (kN2 / 2)] = make_uint2(pack_h2(y0[brev<16>(bt)]), pack_h2(y1[brev<16>(bt)]));
        }
        return;
    }
    float4* __restrict__ out4 = reinterpret_cast<float4*>(work + ((size_t)blockIdx.y << pl.logN) + n2_0) + cp;
#pragma unroll
    for (int bt = 0; bt < 16; ++bt) {
        const size_t k1 = (size_t)(k10 + 32 * bt);
        store_f4<AM_K1_STORE_NT>(out4 + k1 * (kN2 / 2), make_float4(y0[brev<16>(bt)].x, y0[brev<16>(bt)].y,
                                                                    y1[brev<16>(bt)].x, y1[brev<16>(bt)].y));
    }
}

template <int HALF, bool ACC, int LN2 = kR16LogN2>   // as in k1_cols_fwd_c512
__device__ __forceinline__ void k3_cols_inv_c512_tile(unsigned lin, float4* lds4, const Job& job, const float2* __restrict__ work,
                                                      const PlanDev& pl, float out_scale, const ScanCfg& scan, const ScanCfg* geo = nullptr) {
    constexpr int kN2 = 1 << LN2;
    float2* lds2 = reinterpret_cast<float2*>(lds4);
    const int t = threadIdx.x;
    const int hi = t >> 4, cp = t & 15;
    const int ap = hi & 15, half = hi >> 4;
    const int k10 = ap + 16 * half;
    // placement as in k3_cols_inv_r16: the 16 column tiles that share a line of the summary on one XCD
    // (256 tiles per pair = 8 XCDs x 2 x 16; with rows of 16384 points 512 = 8 x 4 x 16)
    const unsigned xcd = lin & 7u, seq = lin >> 3;
    const unsigned slot = seq >> (LN2 - 8), hf = (seq >> 4) & ((1u << (LN2 - 12)) - 1u), tl = seq & 15u;
    const int n2_0 = (int)(((hf * 8u + xcd) * 16u + tl) << kColsLog);
    const int pair = job.first_pair + (int)slot;
    if (scan.only_pairs != nullptr && scan.only_pairs[pair] == 0) return;   // device-side redo: flagged pairs only
    const long long blkA = 2ll * pair, blkB = blkA + 1;
    const long long N = 1ll << pl.logN;
    const unsigned maskN = (unsigned)(N - 1);
    const unsigned n2 = (unsigned)n2_0 + 2u * (unsigned)cp;
#ifndef AM_K3_INTERLEAVE
#define AM_K3_INTERLEAVE 1
#endif
#ifndef AM_K3_PK
#define AM_K3_PK 0   // 1 = the 512-row K3's butterflies in packed f32: 40 % fewer VALU instructions, the same 0.170 ms
                     // (tools/pkbench: v_pk_fma_f32 delivers 1.14x the flops of v_fma_f32 at 4 waves per SIMD, not 2x)
#endif
    using T = typename std::conditional<AM_K3_PK != 0, p2, float2>::type;
    T x0[16], x1[16];
    if constexpr (HALF == 2) {
        // first pass (pipeline twiddle, 16-point transform over beta, the W_32 branch factors) and the
        // exchange on packed half-precision points, both columns in one exchange; the second pass and
        // everything behind it in f32, so that no score is rounded to f16 on the way out
        // (buffer loads: one address register for all 16, see the f32 form below)
        const __amdgpu_buffer_rsrc_t rin = make_rsrc(reinterpret_cast<const unsigned*>(work) + ((size_t)slot << pl.logN) + n2_0,
                                                     (unsigned)((N - n2_0) * 4));
        const unsigned voff = (unsigned)k10 * (kN2 * 4u) + (unsigned)cp * 8u;
        h2 hx0[16], hx1[16];
#pragma unroll
        for (int bt = 0; bt < 16; ++bt) {
            const uint2 v = buf_load_u2<AM_K3H_LOAD_AUX>(rin, voff, (unsigned)bt * (32u * kN2 * 4u));
            hx0[bt] = bits_h2(v.x);
            hx1[bt] = bits_h2(v.y);
        }
        const K3Edges ed = k3_edges(job, scan, blkA, blkB);
        const float2 w512 = pl.tw1[hi];
        {
            const float2 base0 = tw_big(pl, (n2 * (unsigned)k10) & maskN);
            const float2 base1 = tw_big(pl, ((n2 + 1u) * (unsigned)k10) & maskN);
            const float2 step0 = tw_big(pl, (n2 * 32u) & maskN);
            const float2 step1 = tw_big(pl, ((n2 + 1u) * 32u) & maskN);
            twiddle_chain<16, true, false>(hx0, base0, step0);
            twiddle_chain<16, true, false>(hx1, base1, step1);
        }
        dif<16, true>(hx0);
        dif<16, true>(hx1);
        if (half) {
#pragma unroll
            for (int b = 1; b < 16; ++b) {
                hx0[brev<16>(b)] = mul_w32<true>(hx0[brev<16>(b)], b);
                hx1[brev<16>(b)] = mul_w32<true>(hx1[brev<16>(b)], b);
            }
        }
        const float sgn = hi >= 16 ? -1.0f : 1.0f;
        const int bb = hi & 15;
        uint2* ldsu = reinterpret_cast<uint2*>(lds4);
#pragma unroll
        for (int b = 0; b < 16; ++b) ldsu[c512_idx3(ap, b + 16 * half, cp)] = make_uint2(h2_bits(hx0[brev<16>(b)]), h2_bits(hx1[brev<16>(b)]));
        __syncthreads();
#pragma unroll
        for (int a2 = 0; a2 < 16; ++a2) {
            const uint2 u = ldsu[c512_idx3(a2, bb, cp)], v = ldsu[c512_idx3(a2, bb + 16, cp)];
            const float2 u0 = to_f2(bits_h2(u.x)), v0 = to_f2(bits_h2(v.x)), u1 = to_f2(bits_h2(u.y)), v1 = to_f2(bits_h2(v.y));
            x0[a2] = T{fmaf(sgn, v0.x, u0.x), fmaf(sgn, v0.y, u0.y)};
            x1[a2] = T{fmaf(sgn, v1.x, u1.x), fmaf(sgn, v1.y, u1.y)};
        }
        twiddle_nat<16, true>(x0, w512);
        twiddle_nat<16, true>(x1, w512);
        dif<16, true>(x0);
        dif<16, true>(x1);
        k3_finish<5>(job, scan, ed, lds2, n2_0, kN2, t, blkA, blkB, out_scale, x0, x1);
        return;
    }
    if (HALF) {
        const __amdgpu_buffer_rsrc_t rin = make_rsrc(reinterpret_cast<const unsigned*>(work) + ((size_t)slot << pl.logN) + n2_0,
                                                     (unsigned)((N - n2_0) * 4));
        const unsigned voff = (unsigned)k10 * (kN2 * 4u) + (unsigned)cp * 8u;
#pragma unroll
        for (int bt = 0; bt < 16; ++bt) {
            const uint2 v = buf_load_u2<AM_K3H_LOAD_AUX>(rin, voff, (unsigned)bt * (32u * kN2 * 4u));
            const float2 f0 = unpack_h2(v.x), f1 = unpack_h2(v.y);
            x0[bt] = T{f0.x, f0.y};
            x1[bt] = T{f1.x, f1.y};
        }
    } else {
        // buffer loads: one VGPR of address for all 16 (the row offsets, multiples of 2 MB, ride in SGPRs), so
        // that the requests leave back to back instead of behind a 64-bit address addition each
        const __amdgpu_buffer_rsrc_t rin = make_rsrc(work + ((size_t)slot << pl.logN) + n2_0, (unsigned)((N - n2_0) * 8));
        const unsigned voff = (unsigned)k10 * (kN2 * 8u) + (unsigned)cp * 16u;
#pragma unroll
        for (int bt = 0; bt < 16; ++bt) {   // rows k1 = k10 + 32*beta
            const float4 v = buf_load4<AM_K3_LOAD_NT ? 2 : 0>(rin, voff, (unsigned)bt * (32u * kN2 * 8u));
            x0[bt] = T{v.x, v.y};
            x1[bt] = T{v.z, v.w};
        }
    }
    const K3Edges ed = k3_edges(job, geo ? *geo : scan, blkA, blkB);
    const float2 w512 = pl.tw1[hi], w512q = pl.tw1[4 * hi];   // (fourth powers: see twiddle_apply(x, w, w4))
    {
        // out_scale rides on the pipeline twiddle: everything behind it is linear, the scan sees scores
        float2 base0 = tw_big(pl, (n2 * (unsigned)k10) & maskN);
        float2 base1 = tw_big(pl, ((n2 + 1u) * (unsigned)k10) & maskN);
        base0.x *= out_scale; base0.y *= out_scale;
        base1.x *= out_scale; base1.y *= out_scale;
        float2 step0, step1, step0q, step1q;
        tw_big_pair(pl, (n2 * 32u) & maskN, step0, step0q);
        tw_big_pair(pl, ((n2 + 1u) * 32u) & maskN, step1, step1q);
#if AM_K3_INTERLEAVE
        // column 0 goes through its first pass and into LDS before column 1 is touched, and comes out into its
        // second pass while column 1 is on its way: arithmetic between every two barriers
        twiddle_chain<16, true, false>(x0, base0, step0, step0q);
        dif<16, true>(x0);   // inverse over beta: branch value b at x[brev(b)], b = 0..15
        if (half) {          // the odd-b' branch carries conj(W_32^b)
#pragma unroll
            for (int b = 1; b < 16; ++b) x0[brev<16>(b)] = mul_w32<true>(x0[brev<16>(b)], b);
        }
#pragma unroll
        for (int b = 0; b < 16; ++b) lds2[c512_idx3(ap, b + 16 * half, cp)] = make_float2(x0[brev<16>(b)].x, x0[brev<16>(b)].y);
        twiddle_chain<16, true, false>(x1, base1, step1, step1q);
    }
    dif<16, true>(x1);
    if (half) {
#pragma unroll
        for (int b = 1; b < 16; ++b) x1[brev<16>(b)] = mul_w32<true>(x1[brev<16>(b)], b);
    }
    // exchange, one column of the pair at a time: afterwards thread b = hi (0..31) holds
    // z[a'] = u[a'][b & 15] +- v[a'][b & 15]
    const float sgn = hi >= 16 ? -1.0f : 1.0f;
    const int bb = hi & 15;
    __syncthreads();
#pragma unroll
    for (int a2 = 0; a2 < 16; ++a2) {
        const float2 u = lds2[c512_idx3(a2, bb, cp)], v = lds2[c512_idx3(a2, bb + 16, cp)];
        x0[a2] = add_signed(u, v, sgn, T{});
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 16; ++b) lds2[c512_idx3(ap, b + 16 * half, cp)] = make_float2(x1[brev<16>(b)].x, x1[brev<16>(b)].y);
    twiddle_nat<16, true>(x0, w512, w512q);   // conj(W_512^(b*a'))
    dif<16, true>(x0);   // a at x[brev(a)], n1 = a*32 + b
    __syncthreads();
#pragma unroll
    for (int a2 = 0; a2 < 16; ++a2) {
        const float2 u = lds2[c512_idx3(a2, bb, cp)], v = lds2[c512_idx3(a2, bb + 16, cp)];
        x1[a2] = add_signed(u, v, sgn, T{});
    }
    twiddle_nat<16, true>(x1, w512, w512q);
    dif<16, true>(x1);
#else
        twiddle_chain<16, true, false>(x0, base0, step0, step0q);
        twiddle_chain<16, true, false>(x1, base1, step1, step1q);
    }
    dif<16, true>(x0);   // inverse over beta: branch value b at x[brev(b)], b = 0..15
    dif<16, true>(x1);
    if (half) {          // the odd-b' branch carries conj(W_32^b)
#pragma unroll
        for (int b = 1; b < 16; ++b) {
            x0[brev<16>(b)] = mul_w32<true>(x0[brev<16>(b)], b);
            x1[brev<16>(b)] = mul_w32<true>(x1[brev<16>(b)], b);
        }
    }
    // exchange, one column of the pair at a time: afterwards thread b = hi (0..31) holds
    // z[a'] = u[a'][b & 15] +- v[a'][b & 15]
    const float sgn = hi >= 16 ? -1.0f : 1.0f;
    const int bb = hi & 15;
#pragma unroll
    for (int b = 0; b < 16; ++b) lds2[c512_idx3(ap, b + 16 * half, cp)] = make_float2(x0[brev<16>(b)].x, x0[brev<16>(b)].y);
    __syncthreads();
#pragma unroll
    for (int a2 = 0; a2 < 16; ++a2) {
        const float2 u = lds2[c512_idx3(a2, bb, cp)], v = lds2[c512_idx3(a2, bb + 16, cp)];
        x0[a2] = add_signed(u, v, sgn, T{});
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 16; ++b) lds2[c512_idx3(ap, b + 16 * half, cp)] = make_float2(x1[brev<16>(b)].x, x1[brev<16>(b)].y);
    __syncthreads();
#pragma unroll
    for (int a2 = 0; a2 < 16; ++a2) {
        const float2 u = lds2[c512_idx3(a2, bb, cp)], v = lds2[c512_idx3(a2, bb + 16, cp)];
        x1[a2] = add_signed(u, v, sgn, T{});
    }
    twiddle_nat<16, true>(x0, w512, w512q);   // conj(W_512^(b*a'))
    twiddle_nat<16, true>(x1, w512, w512q);
    dif<16, true>(x0);   // a at x[brev(a)], n1 = a*32 + b
    dif<16, true>(x1);
#endif
    k3_finish<5, T, ACC, true>(job, scan, ed, lds2, n2_0, kN2, t, blkA, blkB, out_scale, x0, x1);
}
template <int HALF, bool ACC = false, int REDO = 0>   // (REDO: see k3_cols_inv_r16)
__global__ void __launch_bounds__(512, REDO ? 4 : 2)   // (113 - 119 VGPRs, two workgroups per CU, without being told; the loop of the REDO form has to be held to 128)
k3_cols_inv_c512(Job job, const float2* __restrict__ work, PlanDev pl, float out_scale, ScanCfg scan) {
    extern __shared__ float4 lds4[];
    if constexpr (REDO != 0) {
        for (unsigned lin = blockIdx.x; lin < (unsigned)scan.redo_tiles; lin += gridDim.x) {
            k3_cols_inv_c512_tile<HALF, ACC>(lin, lds4, job, work, pl, out_scale, scan);
            __syncthreads();
        }
    } else {
        k3_cols_inv_c512_tile<HALF, ACC>(blockIdx.x, lds4, job, work, pl, out_scale, scan);
    }
}

// The same tile for rows of 16384 points (N = 2^23 = 512 x 16384; f32 work matrix; am_debug_column_bench)
__global__ void __launch_bounds__(512, 2)
k3_cols_inv_c512w(Job job, const float2* __restrict__ work, PlanDev pl, float out_scale, ScanCfg scan) {
    extern __shared__ float4 lds4[];
    k3_cols_inv_c512_tile<0, false, 14>(blockIdx.x, lds4, job, work, pl, out_scale, scan);
}

// The needles of a group in one launch (launch_k3_group): blockIdx.y picks the needle, everything else is the
// single-needle tile.  One launch instead of eight per group: no launch boundary -- and no drain of the previous
// launch's dirty lines -- between the needles' K3s.
__global__ void __launch_bounds__(512, 2)
k3_cols_inv_c512_group(Job job, PlanDev pl, ScanCfg scan, K3Group grp) {
    extern __shared__ float4 lds4[];
    const unsigned z = blockIdx.y;
    job.dst = grp.dst[z];
    ScanCfg mine;   // (the scalar fields k3_finish reads; the edge table stays in the kernel argument: `geo`)
    mine.stats32 = grp.stats32[z]; mine.wbits = grp.wbits[z]; mine.tile_theta = grp.tile_theta[z]; mine.hist_min = grp.hist_min[z];
    mine.margin = scan.margin; mine.seg_c = scan.seg_c; mine.seg_d = scan.seg_d; mine.inv_c = scan.inv_c;
    mine.only_pairs = nullptr; mine.redo_tiles = 0; mine.edges_n = 0;
    k3_cols_inv_c512_tile<0, false>(blockIdx.x, lds4, job, grp.work[z], pl, grp.out_scale[z], mine, &scan);
}

// ===========================================================================
// N = 2^23 = 1024 x 8192: column kernels with 1024 threads (one workgroup = 16 waves per CU, the
// same wave count as two 512-thread workgroups), 16 points of two columns per thread as in every
// other column kernel.  The 1024-point column transform is 16 x 64: n1 = a*64 + b, k1 = a' + 16*b',
//   W_1024^(n1*k1) = W_16^(a*a') * W_1024^(b*a') * W_64^(b*b'),
// a 16-point pass over a in registers, the twiddle W_1024^(b*a'), an LDS exchange, and the 64-point
// pass over b as one radix-4 stage + a 16-point pass: with b = b0 + 16 m and b' = 4 beta + q,
//   W_64^(b*b') = W_4^(m*q) * W_64^(b0*q) * W_16^(b0*beta),
// so thread (a', q) reads all 64 values of its a', forms y[b0] = (sum_m x[b0 + 16 m] (-i)^(m q)) W_64^(b0 q)
// and transforms y over b0.  q = t >> 8 is uniform over a wavefront: the four cases do not diverge.
// With one workgroup per CU every wave of a CU is in the same phase of its tile (two 512-thread workgroups
// drift apart and overlap one's loads with the other's arithmetic): per point K1 costs 1.33x and K3 1.8x
// their 512-row forms (profiles/r03/needle_sweep.txt), so the plan pays from about 36 s of 44.1 kHz needle
// up, where the longer hop outweighs that (60 s: 1.26 ms per hour of audio against 1.74).
// ===========================================================================
constexpr int kC1024Slab = 64 * 17 + 16;   // float2 per a' slab: 64 rows of 17 + 16 (consecutive a' 32 banks apart)
__device__ __forceinline__ int c1024_idx(int ap, int b, int cp) { return ap * kC1024Slab + b * 17 + cp; }
constexpr int kC1024Lds = 16 * kC1024Slab * 8;   // 141 312 bytes: one workgroup per CU
constexpr int kC1024Slab3 = 64 * 16 + 16;  // K3's exchange runs the other way round (see c512_idx3)
__device__ __forceinline__ int c1024_idx3(int ap, int b, int cp) { return ap * kC1024Slab3 + b * 16 + cp; }
static_assert(16 * kC1024Slab3 * 8 <= kC1024Lds && 1024 * 16 * 8 <= kC1024Lds, "K3's exchange and score scan fit the kernel's LDS");

// sum_m v[m] * w^(m*q), w = -i (forward) or +i (inverse), q uniform over the wavefront
template <bool INV>
__device__ __forceinline__ float2 radix4_branch(float2 v0, float2 v1, float2 v2, float2 v3, int q) {
    const float2 s02 = cadd(v0, v2), d02 = csub(v0, v2), s13 = cadd(v1, v3), d13 = csub(v1, v3);
    if (q == 0) return cadd(s02, s13);
    if (q == 2) return csub(s02, s13);
    const bool minus_i = (q == 1) != INV;        // forward q = 1 and inverse q = 3: d02 - i d13
    return minus_i ? cadd(d02, mul_neg_i(d13)) : cadd(d02, mul_pos_i(d13));
}

template <int KIND>
__global__ void __launch_bounds__(1024)
k1_cols_fwd_c1024(Job job, float2* __restrict__ work, PlanDev pl) {
    extern __shared__ float4 lds4[];
    float2* lds2 = reinterpret_cast<float2*>(lds4);
    const int t = threadIdx.x;
    const int hi = t >> 4, cp = t & 15;             // pass 1: b = hi (0..63)
    const int ap = hi & 15, q = hi >> 4;            // pass 2: a' and b' mod 4
    const int k10 = ap + 16 * q;                    // k1 = k10 + 64 * beta
    const int n2_0 = blockIdx.x << kColsLog;
    const int pair = job.first_pair + blockIdx.y;
    const long long blkA = 2ll * pair, blkB = blkA + 1;
    const bool validB = blkB < job.nblocks;
    const long long N = 1ll << pl.logN;
    const long long baseA = blkA * job.hop - job.lead;
    const long long baseB = blkB * job.hop - job.lead;
    const bool fast = ((reinterpret_cast<uintptr_t>(job.src) & 7) == 0) && ((baseA & 1) == 0) && ((baseB & 1) == 0) &&
                      baseA >= 0 && baseA + N <= job.src_len && validB && baseB + N <= job.src_len;
    const long long col = (long long)n2_0 + 2 * cp;
    const unsigned maskN = (unsigned)(N - 1);
    const float2 w1024 = pl.tw1[hi];                // W_1024^b
    const float2 w64q = pl.tw1[16 * q];             // W_64^q
    float2 x0[16], x1[16];
    if (fast) {
#pragma unroll
        for (int a = 0; a < 16; ++a) {
            const long long off = (long long)(a * 64 + hi) * kN2 + col;
            const float2 va = load_sample2<KIND>(job.src, baseA + off), vb = load_sample2<KIND>(job.src, baseB + off);
            x0[a] = make_float2(va.x, vb.x);
            x1[a] = make_float2(va.y, vb.y);
        }
    } else {
#pragma unroll
        for (int a = 0; a < 16; ++a) {
            const long long n = (long long)(a * 64 + hi) * kN2 + col;
            const float2 va = load2_padded<KIND>(job.src, baseA + n, job.src_len);
            const float2 vb = validB ? load2_padded<KIND>(job.src, baseB + n, job.src_len) : make_float2(0.f, 0.f);
            x0[a] = make_float2(va.x, vb.x);
            x1[a] = make_float2(va.y, vb.y);
        }
    }
    dif<16, false>(x0);
    dif<16, false>(x1);
    twiddle_brev<16, false>(x0, w1024);   // W_1024^(b*a')
    twiddle_brev<16, false>(x1, w1024);
    // exchange, one column of the pair at a time; afterwards thread (a', q) holds the radix-4 branch q
    float2 y0[16], y1[16];
#pragma unroll
    for (int a2 = 0; a2 < 16; ++a2) lds2[c1024_idx(a2, hi, cp)] = x0[brev<16>(a2)];
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 16; ++b) {
        y0[b] = radix4_branch<false>(lds2[c1024_idx(ap, b, cp)], lds2[c1024_idx(ap, b + 16, cp)],
                                     lds2[c1024_idx(ap, b + 32, cp)], lds2[c1024_idx(ap, b + 48, cp)], q);
        if ((b & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // (64 LDS reads in flight at once would not fit the register file)
    }
    __syncthreads();
#pragma unroll
    for (int a2 = 0; a2 < 16; ++a2) lds2[c1024_idx(a2, hi, cp)] = x1[brev<16>(a2)];
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 16; ++b) {
        y1[b] = radix4_branch<false>(lds2[c1024_idx(ap, b, cp)], lds2[c1024_idx(ap, b + 16, cp)],
                                     lds2[c1024_idx(ap, b + 32, cp)], lds2[c1024_idx(ap, b + 48, cp)], q);
        if ((b & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // (64 LDS reads in flight at once would not fit the register file)
    }
    // the pipeline twiddle's table entries are requested here, where both columns' first-pass registers
    // have been released and the second pass's arithmetic hides their latency (requested before the sample
    // loads, as in the 512-row kernel, they push this 128-register kernel into scratch)
    const float2 base0 = tw_big(pl, ((unsigned)col * (unsigned)k10) & maskN);
    const float2 base1 = tw_big(pl, (((unsigned)col + 1u) * (unsigned)k10) & maskN);
    const float2 step0 = tw_big(pl, ((unsigned)col * 64u) & maskN);
    const float2 step1 = tw_big(pl, (((unsigned)col + 1u) * 64u) & maskN);
    if (q) {                              // W_64^(b0*q)
        twiddle_nat<16, false>(y0, w64q);
        twiddle_nat<16, false>(y1, w64q);
    }
    dif<16, false>(y0);   // beta at y[brev(beta)], b' = 4*beta + q
    dif<16, false>(y1);
    // W_N^(n2*k1) = W_N^(n2*k10) * (W_N^(64*n2))^beta
    twiddle_chain<16, false, true>(y0, base0, step0);
    twiddle_chain<16, false, true>(y1, base1, step1);
    float4* __restrict__ out4 = reinterpret_cast<float4*>(work + ((size_t)blockIdx.y << pl.logN) + n2_0) + cp;
#pragma unroll
    for (int bt = 0; bt < 16; ++bt) {
        const size_t k1 = (size_t)(k10 + 64 * bt);
        store_f4<AM_K1_STORE_NT>(out4 + k1 * (kN2 / 2), make_float4(y0[brev<16>(bt)].x, y0[brev<16>(bt)].y,
                                                                    y1[brev<16>(bt)].x, y1[brev<16>(bt)].y));
    }
}

template <bool ACC>   // ACC: add to what job.dst holds (needle partitioning, see k3_finish)
__device__ __forceinline__ void k3_cols_inv_c1024_tile(unsigned lin, float4* lds4, const Job& job, const float2* __restrict__ work,
                                                       const PlanDev& pl, float out_scale, const ScanCfg& scan) {
    float2* lds2 = reinterpret_cast<float2*>(lds4);
    const int t = threadIdx.x;
    const int hi = t >> 4, cp = t & 15;
    const int ap = hi & 15, q = hi >> 4;
    const int k10 = ap + 16 * q;
    // placement as in k3_cols_inv_r16: the 16 column tiles that share a line of the summary on one XCD
    const unsigned xcd = lin & 7u, seq = lin >> 3;
    const unsigned slot = seq >> 5, hf = (seq >> 4) & 1u, tl = seq & 15u;
    const int n2_0 = (int)(((hf * 8u + xcd) * 16u + tl) << kColsLog);
    const int pair = job.first_pair + (int)slot;
    if (scan.only_pairs != nullptr && scan.only_pairs[pair] == 0) return;   // device-side redo: flagged pairs only
    const long long blkA = 2ll * pair, blkB = blkA + 1;
    const long long N = 1ll << pl.logN;
    const unsigned maskN = (unsigned)(N - 1);
    const unsigned n2 = (unsigned)n2_0 + 2u * (unsigned)cp;
    float2 x0[16], x1[16];
    {   // (buffer loads off one address register, as in the 512-row kernel)
        const __amdgpu_buffer_rsrc_t rin = make_rsrc(work + ((size_t)slot << pl.logN) + n2_0, (unsigned)((N - n2_0) * 8));
        const unsigned voff = (unsigned)k10 * (kN2 * 8u) + (unsigned)cp * 16u;
#pragma unroll
        for (int bt = 0; bt < 16; ++bt) {   // rows k1 = k10 + 64*beta
            const float4 v = buf_load4<AM_K3_LOAD_NT ? 2 : 0>(rin, voff, (unsigned)bt * (64u * kN2 * 8u));
            x0[bt] = make_float2(v.x, v.y);
            x1[bt] = make_float2(v.z, v.w);
        }
    }
    const K3Edges ed = k3_edges(job, scan, blkA, blkB);
    const float2 w1024 = pl.tw1[hi];
    const float2 w64q = pl.tw1[16 * q];
    {
        float2 base0 = tw_big(pl, (n2 * (unsigned)k10) & maskN);
        float2 base1 = tw_big(pl, ((n2 + 1u) * (unsigned)k10) & maskN);
        base0.x *= out_scale; base0.y *= out_scale;   // (out_scale rides on the pipeline twiddle, as in the 512-row kernel)
        base1.x *= out_scale; base1.y *= out_scale;
        const float2 step0 = tw_big(pl, (n2 * 64u) & maskN);
        const float2 step1 = tw_big(pl, ((n2 + 1u) * 64u) & maskN);
        twiddle_chain<16, true, false>(x0, base0, step0);
        twiddle_chain<16, true, false>(x1, base1, step1);
    }
    // (column 0 through its first pass and into LDS before column 1's, its second pass while column 1 crosses:
    // arithmetic between every two barriers, as in the 512-row kernel -- here all 16 waves of the CU meet at each)
    // exchange, one column of the pair at a time: afterwards thread b = hi (0..63) holds
    // z[a'] = sum_q u_q[a'][b & 15] * i^((b >> 4) * q)
    const int bb = hi & 15, m = hi >> 4;
    dif<16, true>(x0);   // inverse over beta: b0 at x[brev(b0)]
    if (q) twiddle_brev<16, true>(x0, w64q);   // conj(W_64^(b0*q))
#pragma unroll
    for (int b = 0; b < 16; ++b) lds2[c1024_idx3(ap, b + 16 * q, cp)] = x0[brev<16>(b)];
    dif<16, true>(x1);
    if (q) twiddle_brev<16, true>(x1, w64q);
    __syncthreads();
#pragma unroll
    for (int a2 = 0; a2 < 16; ++a2)
        x0[a2] = radix4_branch<true>(lds2[c1024_idx3(a2, bb, cp)], lds2[c1024_idx3(a2, bb + 16, cp)],
                                     lds2[c1024_idx3(a2, bb + 32, cp)], lds2[c1024_idx3(a2, bb + 48, cp)], m);
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 16; ++b) lds2[c1024_idx3(ap, b + 16 * q, cp)] = x1[brev<16>(b)];
    twiddle_nat<16, true>(x0, w1024);   // conj(W_1024^(b*a'))
    dif<16, true>(x0);   // a at x[brev(a)], n1 = a*64 + b
    __syncthreads();
#pragma unroll
    for (int a2 = 0; a2 < 16; ++a2)
        x1[a2] = radix4_branch<true>(lds2[c1024_idx3(a2, bb, cp)], lds2[c1024_idx3(a2, bb + 16, cp)],
                                     lds2[c1024_idx3(a2, bb + 32, cp)], lds2[c1024_idx3(a2, bb + 48, cp)], m);
    twiddle_nat<16, true>(x1, w1024);
    dif<16, true>(x1);
    k3_finish<6, float2, ACC, true>(job, scan, ed, lds2, n2_0, kN2, t, blkA, blkB, out_scale, x0, x1);
}
template <bool ACC, int REDO = 0>   // (REDO: see k3_cols_inv_r16)
__global__ void __launch_bounds__(1024)
k3_cols_inv_c1024(Job job, const float2* __restrict__ work, PlanDev pl, float out_scale, ScanCfg scan) {
    extern __shared__ float4 lds4[];
    if constexpr (REDO != 0) {
        for (unsigned lin = blockIdx.x; lin < (unsigned)scan.redo_tiles; lin += gridDim.x) {
            k3_cols_inv_c1024_tile<ACC>(lin, lds4, job, work, pl, out_scale, scan);
            __syncthreads();
        }
    } else {
        k3_cols_inv_c1024_tile<ACC>(blockIdx.x, lds4, job, work, pl, out_scale, scan);
    }
}

// ===========================================================================
// Generic kernels: any N1 x N2, transforms done by in-LDS radix-4 passes.
// ===========================================================================
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

template <int KIND>
__device__ __forceinline__ float load_padded(const void* __restrict__ src, long long i, long long len) {
    return (i >= 0 && i < len) ? load_sample<KIND>(src, i) : 0.0f;
}

// K1 generic: row p of the work matrix holds frequency k1 = bitrev(p).
template <int BL, int KIND>
__global__ void __launch_bounds__(kFftThreads)
k1_cols_fwd_gen(Job job, float2* __restrict__ work, PlanDev pl) {
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
        const float a = load_padded<KIND>(job.src, baseA + n, job.src_len);
        const float b = validB ? load_padded<KIND>(job.src, baseB + n, job.src_len) : 0.0f;
        s[idx] = make_float2(a, b);
    }
    __syncthreads();
    lds_fft_fwd<BL>(s, pl.logN1, pl.tw1, tid, nthr);
    float2* out = work + ((size_t)blockIdx.y << pl.logN) + n2_0;
    const unsigned maskN = (1u << pl.logN) - 1u;
    for (int idx = tid; idx < total; idx += nthr) {
        const int p = idx >> BL, c = idx & ((1 << BL) - 1);
        const unsigned k1 = __brev((unsigned)p) >> (32 - pl.logN1);
        const float2 w = tw_big(pl, ((unsigned)(n2_0 + c) * k1) & maskN);
        out[(size_t)p * N2 + c] = cmul(s[idx], w);
    }
}

template <bool SPECTRUM>
__global__ void __launch_bounds__(kFftThreads)
k2_rows_gen(float2* __restrict__ work, const float2* __restrict__ hc, float2* __restrict__ hc_out, PlanDev pl) {
    extern __shared__ float2 s[];
    const int N2 = 1 << pl.logN2;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int p = blockIdx.x;
    float2* row = work + ((size_t)blockIdx.y << pl.logN) + (size_t)p * N2;
    for (int n2 = tid; n2 < N2; n2 += nthr) s[n2] = row[n2];
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
    float2* orow = hc_out ? hc_out + ((size_t)blockIdx.y << pl.logN) + (size_t)p * N2 : row;
    for (int n2 = tid; n2 < N2; n2 += nthr) orow[n2] = s[n2];
}

template <int BL>
__global__ void __launch_bounds__(kFftThreads)
k3_cols_inv_gen(Job job, const float2* __restrict__ work, PlanDev pl, float out_scale, int acc) {
    extern __shared__ float2 s[];
    const int N1 = 1 << pl.logN1, N2 = 1 << pl.logN2;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int n2_0 = blockIdx.x << BL;
    const int pair = job.first_pair + blockIdx.y;
    const long long blkA = 2ll * pair, blkB = blkA + 1;
    const bool validB = blkB < job.nblocks;
    const int total = N1 << BL;
    const float2* in = work + ((size_t)blockIdx.y << pl.logN) + n2_0;
    const unsigned maskN = (1u << pl.logN) - 1u;
    for (int idx = tid; idx < total; idx += nthr) {
        const int p = idx >> BL, c = idx & ((1 << BL) - 1);
        const unsigned k1 = __brev((unsigned)p) >> (32 - pl.logN1);
        const float2 w = tw_big(pl, ((unsigned)(n2_0 + c) * k1) & maskN);
        s[idx] = cmulc(in[(size_t)p * N2 + c], w);
    }
    __syncthreads();
    lds_fft_inv<BL>(s, pl.logN1, pl.tw1, tid, nthr);
    const long long outA = blkA * job.hop, outB = blkB * job.hop;
    for (int idx = tid; idx < total; idx += nthr) {
        const int r = idx >> BL, c = idx & ((1 << BL) - 1);
        const long long n = (long long)r * N2 + n2_0 + c;
        if (n >= job.hop) continue;
        const float2 v = s[idx];
        if (outA + n < job.out_count) job.dst[outA + n] = v.x * out_scale + (acc ? job.dst[outA + n] : 0.0f);
        if (validB && outB + n < job.out_count) job.dst[outB + n] = v.y * out_scale + (acc ? job.dst[outB + n] : 0.0f);
    }
}

// ===========================================================================
// Tiny needles (at most kDirectMaxNeedle samples): direct summation.  A transform
// of at least 2^10 points would spend ten rounding stages on a sum of a few
// products; summed directly the reference's own known-answer test
// (audio_matcher.rs:490-517, integer data) comes out exact.
// ===========================================================================
template <int KIND>
__global__ void __launch_bounds__(256)
correlate_direct(Job job, const float* __restrict__ needle, int s, float out_scale) {
    __shared__ float nd[kDirectMaxNeedle];
    if ((int)threadIdx.x < s) nd[threadIdx.x] = needle[threadIdx.x];
    __syncthreads();
    for (long long j = (long long)blockIdx.x * 256 + threadIdx.x; j < job.out_count; j += (long long)gridDim.x * 256) {
        float acc = 0.0f;
        for (int n = 0; n < s; ++n) acc = fmaf(load_padded<KIND>(job.src, j + n - job.lead, job.src_len), nd[n], acc);
        job.dst[j] = acc * out_scale;
    }
}

hipError_t launch_direct(hipStream_t st, const Job& job, const float* needle, int s, float out_scale) {
    if (s < 1 || s > kDirectMaxNeedle || job.out_count <= 0) return hipErrorInvalidValue;
    long long blocks = (job.out_count + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (job.src_kind == 1) hipLaunchKernelGGL(correlate_direct<1>, dim3((unsigned)blocks), dim3(256), 0, st, job, needle, s, out_scale);
    else hipLaunchKernelGGL(correlate_direct<0>, dim3((unsigned)blocks), dim3(256), 0, st, job, needle, s, out_scale);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
static constexpr int kMaxLds = 160 * 1024;
static constexpr int kR16Lds = 64 * 1024;       // K2: one 8192-point row
static constexpr int kR16LdsK1 = 256 * 17 * 8;   // K1: one column of the pair at a time, padded rows
static constexpr int kR16LdsK3 = 256 * 16 * 8;

bool plan_is_r16(const PlanDev& pl) { return pl.logN1 == kR16LogN1 && pl.logN2 == kR16LogN2; }
bool plan_is_c512(const PlanDev& pl) { return pl.logN1 == 9 && pl.logN2 == kR16LogN2; }
bool plan_is_c1024(const PlanDev& pl) { return pl.logN1 == 10 && pl.logN2 == kR16LogN2; }
bool plan_is_c512w(const PlanDev& pl) { return pl.logN1 == 9 && pl.logN2 == 14; }   // 512 x 16384 (column kernels only, so far)
bool plan_has_scan(const PlanDev& pl) { return plan_is_r16(pl) || plan_is_c512(pl) || plan_is_c1024(pl); }
// the row kernel only needs 8192-point rows; it serves any N1 (its rows are independent)
bool plan_k2_is_r16(const PlanDev& pl) { return pl.logN2 == kR16LogN2 && pl.logN1 >= 3; }

hipError_t fft_kernels_init() {
    hipError_t e;
#define AM_SET_LDS(fn, bytes)                                                                     \
    e = hipFuncSetAttribute((const void*)(fn), hipFuncAttributeMaxDynamicSharedMemorySize, bytes); \
    if (e != hipSuccess) return e;
    AM_SET_LDS((k1_cols_fwd_gen<kColsLog, 0>), kMaxLds)
    AM_SET_LDS((k1_cols_fwd_gen<kColsLog, 1>), kMaxLds)
    AM_SET_LDS(k3_cols_inv_gen<kColsLog>, kMaxLds)
    AM_SET_LDS(k2_rows_gen<false>, kMaxLds)
    AM_SET_LDS(k2_rows_gen<true>, kMaxLds)
    AM_SET_LDS((k1_cols_fwd_r16<0, 0>), kR16LdsK1)
    AM_SET_LDS((k1_cols_fwd_r16<1, 0>), kR16LdsK1)
    AM_SET_LDS((k1_cols_fwd_r16<0, 1>), kR16LdsK1)
    AM_SET_LDS((k1_cols_fwd_r16<1, 1>), kR16LdsK1)
    AM_SET_LDS((k1_cols_fwd_r16<0, 2>), kR16LdsK1)
    AM_SET_LDS((k1_cols_fwd_r16<1, 2>), kR16LdsK1)
    AM_SET_LDS((k1_cols_fwd_c512<0, 0>), kC512Lds)
    AM_SET_LDS((k1_cols_fwd_c512<1, 0>), kC512Lds)
    AM_SET_LDS((k1_cols_fwd_c512<0, 1>), kC512Lds)
    AM_SET_LDS((k1_cols_fwd_c512<1, 1>), kC512Lds)
    AM_SET_LDS((k1_cols_fwd_c512<0, 2>), kC512Lds)
    AM_SET_LDS((k1_cols_fwd_c512<1, 2>), kC512Lds)
    AM_SET_LDS(k1_cols_fwd_c1024<0>, kC1024Lds)
    AM_SET_LDS(k1_cols_fwd_c1024<1>, kC1024Lds)
    AM_SET_LDS(k3_cols_inv_c1024<false>, kC1024Lds)
    AM_SET_LDS((k3_cols_inv_c1024<false, 1>), kC1024Lds)
    AM_SET_LDS((k3_cols_inv_c512<0, false, 1>), kC512Lds)
    AM_SET_LDS((k3_cols_inv_c512<1, false, 1>), kC512Lds)
    AM_SET_LDS((k3_cols_inv_c512<2, false, 1>), kC512Lds)
    AM_SET_LDS((k3_cols_inv_r16<0, false, 1>), kR16LdsK3)
    AM_SET_LDS((k3_cols_inv_r16<1, false, 1>), kR16LdsK3)
    AM_SET_LDS((k3_cols_inv_r16<2, false, 1>), kR16LdsK3)
    AM_SET_LDS(k3_cols_inv_c1024<true>, kC1024Lds)
    AM_SET_LDS(k3_cols_inv_c512_group, kC512Lds)
    AM_SET_LDS(k3_cols_inv_c512w, kC512Lds)
    AM_SET_LDS((k1_cols_fwd_c512<0, 0, 14>), kC512Lds)
    AM_SET_LDS(k3_cols_inv_r16_group, kR16LdsK3)
    AM_SET_LDS(tail_cols_inv_r16<0>, kR16LdsK3)
    AM_SET_LDS(tail_cols_inv_r16<1>, kR16LdsK3)
    AM_SET_LDS(tail_cols_inv_r16<2>, kR16LdsK3)
    AM_SET_LDS((tail_cols_fwd_r16<0, 0>), kR16LdsK1)
    AM_SET_LDS((tail_cols_fwd_r16<1, 0>), kR16LdsK1)
    AM_SET_LDS((tail_cols_fwd_r16<0, 1>), kR16LdsK1)
    AM_SET_LDS((tail_cols_fwd_r16<1, 1>), kR16LdsK1)
    AM_SET_LDS((tail_cols_fwd_r16<0, 2>), kR16LdsK1)
    AM_SET_LDS((tail_cols_fwd_r16<1, 2>), kR16LdsK1)
    AM_SET_LDS(k3_cols_inv_c512<0>, kC512Lds)
    AM_SET_LDS((k3_cols_inv_c512<0, true>), kC512Lds)
    AM_SET_LDS((k3_cols_inv_r16<0, true>), kR16LdsK3)
    AM_SET_LDS(k3_cols_inv_c512<1>, kC512Lds)
    AM_SET_LDS(k3_cols_inv_c512<2>, kC512Lds)
    AM_SET_LDS(k3_cols_inv_r16<0>, kR16LdsK3)
    AM_SET_LDS(k3_cols_inv_r16<1>, kR16LdsK3)
    AM_SET_LDS(k3_cols_inv_r16<2>, kR16LdsK3)
    AM_SET_LDS((k2_rows_r16<false, false>), kR16Lds)
    AM_SET_LDS((k2_rows_r16<false, true>), kR16Lds)
    AM_SET_LDS((k2_rows_r16<true, false>), kR16Lds)
    AM_SET_LDS(k2_rows_r16_group, kR16Lds)
    AM_SET_LDS(k2_rows_h16, kK2hLds)
    AM_SET_LDS(tail_rows_h16, kK2hLds)
    AM_SET_LDS(k2_rows_m16, kK2mLds)
#undef AM_SET_LDS
    return hipSuccess;
}

hipError_t launch_k1(hipStream_t st, const Job& job, int npairs, float2* work, const PlanDev& pl, int half) {
    const dim3 grid((1u << pl.logN2) >> kColsLog, npairs);
    const bool pcm = job.src_kind == 1;
    if (plan_is_c512w(pl)) {
        if (half || pcm) return hipErrorInvalidValue;
        hipLaunchKernelGGL((k1_cols_fwd_c512<0, 0, 14>), grid, dim3(512), kC512Lds, st, job, work, pl);
    } else if (plan_is_c1024(pl)) {
        if (half) return hipErrorInvalidValue;   // (f32 work matrix only)
        if (pcm) hipLaunchKernelGGL(k1_cols_fwd_c1024<1>, grid, dim3(1024), kC1024Lds, st, job, work, pl);
        else hipLaunchKernelGGL(k1_cols_fwd_c1024<0>, grid, dim3(1024), kC1024Lds, st, job, work, pl);
    } else if (plan_is_c512(pl)) {
        if (half == 2) {
            if (pcm) hipLaunchKernelGGL((k1_cols_fwd_c512<1, 2>), grid, dim3(512), kC512Lds, st, job, work, pl);
            else hipLaunchKernelGGL((k1_cols_fwd_c512<0, 2>), grid, dim3(512), kC512Lds, st, job, work, pl);
        } else if (half) {
            if (pcm) hipLaunchKernelGGL((k1_cols_fwd_c512<1, 1>), grid, dim3(512), kC512Lds, st, job, work, pl);
            else hipLaunchKernelGGL((k1_cols_fwd_c512<0, 1>), grid, dim3(512), kC512Lds, st, job, work, pl);
        } else {
            if (pcm) hipLaunchKernelGGL((k1_cols_fwd_c512<1, 0>), grid, dim3(512), kC512Lds, st, job, work, pl);
            else hipLaunchKernelGGL((k1_cols_fwd_c512<0, 0>), grid, dim3(512), kC512Lds, st, job, work, pl);
        }
    } else if (plan_is_r16(pl)) {
        if (half == 2) {
            if (pcm) hipLaunchKernelGGL((k1_cols_fwd_r16<1, 2>), grid, dim3(256), kR16LdsK1, st, job, work, pl);
            else hipLaunchKernelGGL((k1_cols_fwd_r16<0, 2>), grid, dim3(256), kR16LdsK1, st, job, work, pl);
        } else if (half) {
            if (pcm) hipLaunchKernelGGL((k1_cols_fwd_r16<1, 1>), grid, dim3(256), kR16LdsK1, st, job, work, pl);
            else hipLaunchKernelGGL((k1_cols_fwd_r16<0, 1>), grid, dim3(256), kR16LdsK1, st, job, work, pl);
        } else {
            if (pcm) hipLaunchKernelGGL((k1_cols_fwd_r16<1, 0>), grid, dim3(256), kR16LdsK1, st, job, work, pl);
            else hipLaunchKernelGGL((k1_cols_fwd_r16<0, 0>), grid, dim3(256), kR16LdsK1, st, job, work, pl);
        }
    } else {
        const size_t lds = (sizeof(float2) << pl.logN1) << kColsLog;
        if (pcm) hipLaunchKernelGGL((k1_cols_fwd_gen<kColsLog, 1>), grid, dim3(kFftThreads), lds, st, job, work, pl);
        else hipLaunchKernelGGL((k1_cols_fwd_gen<kColsLog, 0>), grid, dim3(kFftThreads), lds, st, job, work, pl);
    }
    return hipGetLastError();
}

// (the environment variable AM_K2_MFMA sets the initial value, so that whole test runs can be repeated with it)
static int k2_mfma_initial() { const char* e = getenv("AM_K2_MFMA"); return e && atoi(e) ? 1 : 0; }
static std::atomic<int> g_k2_mfma{k2_mfma_initial()};
void set_k2_mfma(int on) { g_k2_mfma.store(on ? 1 : 0, std::memory_order_relaxed); }
bool k2_mfma_enabled() { return g_k2_mfma.load(std::memory_order_relaxed) != 0; }
int k2_mfma_table_dwords() { return kMfTotal; }

hipError_t launch_k2(hipStream_t st, int npairs, float2* work, const float2* hc, const PlanDev& pl, float2* dst,
                     int half, float hscale, float pre, bool tail) {
    const dim3 grid(1u << pl.logN1, npairs);
    if (tail && plan_k2_is_r16(pl) && AM_K2_PLANES) {
        const dim3 rows((unsigned)npairs << pl.logN1);
        if (half == 2) hipLaunchKernelGGL(tail_rows_h16, rows, dim3(256), kK2hLds, st, reinterpret_cast<unsigned*>(work),
                                          reinterpret_cast<const unsigned*>(hc), reinterpret_cast<unsigned*>(dst), pl, (unsigned)npairs, pre);
        else if (half) hipLaunchKernelGGL(tail_rows_r16_planes<true>, rows, dim3(256), kK2hLds, st, work, hc, dst, pl, (unsigned)npairs, hscale);
        else hipLaunchKernelGGL(tail_rows_r16_planes<false>, rows, dim3(256), kK2hLds, st, work, hc, dst, pl, (unsigned)npairs, 1.0f);
        return hipGetLastError();
    }
    if (plan_k2_is_r16(pl)) {
        if (half == 2 && k2_mfma_enabled() && pl.mf != nullptr) {
            // persistent workgroups: four per CU walk the rows (the grid stays a multiple of the 8 XCDs)
            const unsigned rows = (unsigned)npairs << pl.logN1;
            const unsigned grid = std::min<unsigned>(rows, 256u * 4u) & ~7u;
            hipLaunchKernelGGL(k2_rows_m16, dim3(grid ? grid : 8u), dim3(256), kK2mLds, st,
                               reinterpret_cast<unsigned*>(work), reinterpret_cast<const unsigned*>(hc), reinterpret_cast<unsigned*>(dst),
                               pl, (unsigned)npairs, pre);
        } else if (half == 2) {
            hipLaunchKernelGGL(k2_rows_h16, dim3((unsigned)npairs << pl.logN1), dim3(256), kK2hLds, st,
                               reinterpret_cast<unsigned*>(work), reinterpret_cast<const unsigned*>(hc), reinterpret_cast<unsigned*>(dst),
                               pl, (unsigned)npairs, pre);
        } else if (half && AM_K2_PLANES) hipLaunchKernelGGL(k2_rows_r16_planes<true>, dim3((unsigned)npairs << pl.logN1), dim3(256), kK2hLds, st, work, hc,
                                                          dst, pl, (unsigned)npairs, hscale);
        else if (half) hipLaunchKernelGGL((k2_rows_r16<false, true>), dim3((unsigned)npairs << pl.logN1), dim3(256), kR16Lds, st, work, hc,
                                     dst, pl, (unsigned)npairs, hscale);
        else if (AM_K2_PLANES) hipLaunchKernelGGL(k2_rows_r16_planes<false>, dim3((unsigned)npairs << pl.logN1), dim3(256), kK2hLds, st, work, hc,
                                                  dst, pl, (unsigned)npairs, 1.0f);
        else hipLaunchKernelGGL((k2_rows_r16<false, false>), dim3((unsigned)npairs << pl.logN1), dim3(256), kR16Lds, st, work, hc,
                                dst, pl, (unsigned)npairs, 1.0f);
    } else {
        const size_t lds = sizeof(float2) << pl.logN2;
        hipLaunchKernelGGL(k2_rows_gen<false>, grid, dim3(kFftThreads), lds, st, work, hc, dst, pl);
    }
    return hipGetLastError();
}

bool plan_k2_has_group(const PlanDev& pl) { return plan_k2_is_r16(pl); }

hipError_t launch_k2_group(hipStream_t st, int npairs, const float2* work, const K2Group& grp, const PlanDev& pl) {
    if (!plan_k2_has_group(pl) || grp.n < 1 || grp.n > kMaxNeedleGroup) return hipErrorInvalidValue;
    if (AM_K2G_PLANES) hipLaunchKernelGGL(k2_rows_r16_group_planes, dim3((unsigned)npairs << pl.logN1), dim3(256), kK2hLds, st, work, grp, pl,
                                          (unsigned)npairs);
    else hipLaunchKernelGGL(k2_rows_r16_group, dim3((unsigned)npairs << pl.logN1), dim3(256), kR16Lds, st, work, grp, pl,
                            (unsigned)npairs);
    return hipGetLastError();
}

hipError_t launch_k2_spectrum(hipStream_t st, float2* work, float2* hc_out, const PlanDev& pl) {
    const dim3 grid(1u << pl.logN1, 1);
    if (plan_k2_is_r16(pl)) {
        hipLaunchKernelGGL((k2_rows_r16<true, false>), dim3(1u << pl.logN1), dim3(256), kR16Lds, st, work, (const float2*)nullptr,
                           hc_out, pl, 1u, 1.0f);
    } else {
        const size_t lds = sizeof(float2) << pl.logN2;
        hipLaunchKernelGGL(k2_rows_gen<true>, grid, dim3(kFftThreads), lds, st, work, (const float2*)nullptr, hc_out, pl);
    }
    return hipGetLastError();
}

// scan.stats32 != nullptr only for the plans with a fused scan (plan_has_scan) and a 1024-aligned hop
// the chunk edges of the launch's block pairs (ScanCfg::edge_rel; mirrors k3_edges)
static void fill_edges(const Job& job, int npairs, ScanCfg& scan) {
    scan.edges_n = 0;
    const long long c = scan.seg_c, d = scan.seg_d, hop = job.hop;
    if (scan.stats32 == nullptr || npairs > ScanCfg::kMaxEdgeSlots || c < hop + 32) return;
    auto rel = [](long long edge, long long start) { const long long r = edge - start; return r < 0x7fffffffll ? (int)r : 0x7fffffff; };
    for (int s = 0; s < npairs; ++s) {
        const long long outA = 2ll * (job.first_pair + s) * hop, outB = outA + hop;
        const long long m = outA % c;
        const long long ecA = m == 0 ? outA : outA + (c - m);   // smallest i*c >= outA
        long long edA = d;                                      // smallest i*c + d >= outA, i >= 0
        if (outA > d) {
            const long long m2 = (outA - d) % c;
            edA = m2 == 0 ? outA : outA + (c - m2);
        }
        const long long ecB = ecA >= outB ? ecA : ecA + c, edB = edA >= outB ? edA : edA + c;
        scan.edge_rel[s][0] = rel(ecA, outA); scan.edge_rel[s][1] = rel(edA, outA);
        scan.edge_rel[s][2] = rel(ecB, outB); scan.edge_rel[s][3] = rel(edB, outB);
    }
    scan.edges_n = npairs;
}

bool plan_k3_has_group(const PlanDev& pl) { return plan_is_c512(pl) || plan_is_r16(pl); }

hipError_t launch_tail_batch_k1(hipStream_t st, const TailBatch& tb, int hop, int src_kind, float2* work, const PlanDev& pl, int half) {
    if (!plan_is_r16(pl) || tb.n < 1 || tb.n > kMaxTailBatch) return hipErrorInvalidValue;
    const dim3 grid((1u << pl.logN2) >> kColsLog, (unsigned)tb.n);
#define AM_TAIL_K1(K, H) hipLaunchKernelGGL((tail_cols_fwd_r16<K, H>), grid, dim3(256), kR16LdsK1, st, tb, hop, work, pl)
    if (src_kind == 1) { if (half == 2) AM_TAIL_K1(1, 2); else if (half) AM_TAIL_K1(1, 1); else AM_TAIL_K1(1, 0); }
    else { if (half == 2) AM_TAIL_K1(0, 2); else if (half) AM_TAIL_K1(0, 1); else AM_TAIL_K1(0, 0); }
#undef AM_TAIL_K1
    return hipGetLastError();
}
hipError_t launch_tail_batch_k3(hipStream_t st, const TailBatch& tb, int hop, const float2* work, const PlanDev& pl, float out_scale, int half) {
    if (!plan_is_r16(pl) || tb.n < 1 || tb.n > kMaxTailBatch) return hipErrorInvalidValue;
    const dim3 grid(kN2 >> kColsLog, (unsigned)tb.n);
    if (half == 2) hipLaunchKernelGGL(tail_cols_inv_r16<2>, grid, dim3(256), kR16LdsK3, st, tb, hop, work, pl, out_scale);
    else if (half) hipLaunchKernelGGL(tail_cols_inv_r16<1>, grid, dim3(256), kR16LdsK3, st, tb, hop, work, pl, out_scale);
    else hipLaunchKernelGGL(tail_cols_inv_r16<0>, grid, dim3(256), kR16LdsK3, st, tb, hop, work, pl, out_scale);
    return hipGetLastError();
}
hipError_t launch_tail_preset_group(hipStream_t st, const K3Group& grp, long long blk, int log_n1, int log_n2) {
    if (grp.n < 1 || grp.n > kMaxNeedleGroup) return hipErrorInvalidValue;
    for (int z = 0; z < grp.n; ++z) if (!grp.wbits[z] || !grp.tile_theta[z]) return hipErrorInvalidValue;
    hipLaunchKernelGGL(tail_preset_group_kernel, dim3(8, (unsigned)grp.n), dim3(256), 0, st, grp, blk, log_n1, log_n2);
    return hipGetLastError();
}
hipError_t launch_tail_commit(hipStream_t st, const float* tail_scores, float* scores, long long n, const float2* tail_stats32, float2* stats32,
                              unsigned long long* wbits, long long words, float* theta, int tiles) {
    const unsigned grid = (unsigned)std::min<long long>(512, std::max<long long>(1, (n / 4 + 255) / 256));
    hipLaunchKernelGGL(tail_commit_kernel, dim3(grid), dim3(256), 0, st, tail_scores, scores, n, tail_stats32, stats32, wbits, words, theta, tiles);
    return hipGetLastError();
}

hipError_t launch_k3_group(hipStream_t st, const Job& job, int npairs, const K3Group& grp, const PlanDev& pl, const ScanCfg& scan_in) {
    if (!plan_k3_has_group(pl) || grp.n < 1 || grp.n > kMaxNeedleGroup || scan_in.stats32 == nullptr) return hipErrorInvalidValue;
    ScanCfg scan = scan_in;
    scan.only_pairs = nullptr;
    fill_edges(job, npairs, scan);
    const dim3 grid((unsigned)npairs * (kN2 >> kColsLog), (unsigned)grp.n);
    if (plan_is_c512(pl)) hipLaunchKernelGGL(k3_cols_inv_c512_group, grid, dim3(512), kC512Lds, st, job, pl, scan, grp);
    else hipLaunchKernelGGL(k3_cols_inv_r16_group, grid, dim3(256), kR16LdsK3, st, job, pl, scan, grp);
    return hipGetLastError();
}

hipError_t launch_k3(hipStream_t st, const Job& job, int npairs, const float2* work,
                     const PlanDev& pl, float out_scale, const ScanCfg& scan_in, int half, bool accumulate) {
    const dim3 grid((1u << pl.logN2) >> kColsLog, npairs);
    ScanCfg scan = scan_in;
    fill_edges(job, npairs, scan);
    if (scan.only_pairs != nullptr && !accumulate && plan_has_scan(pl)) {
        // the device-side redo: the same kernels under names of their own
        // (a grid of one round of resident workgroups -- 1, 2, 4 per CU -- walks the tiles)
        scan.redo_tiles = npairs * (kN2 >> kColsLog);
        const unsigned per_cu = plan_is_c1024(pl) ? 1u : plan_is_c512(pl) ? 2u : 4u;
        const dim3 g1(std::min<unsigned>((unsigned)scan.redo_tiles, 256u * per_cu));
        if (plan_is_c1024(pl)) {
            if (half) return hipErrorInvalidValue;
            hipLaunchKernelGGL((k3_cols_inv_c1024<false, 1>), g1, dim3(1024), kC1024Lds, st, job, work, pl, out_scale, scan);
        } else if (plan_is_c512(pl)) {
            if (half == 2) hipLaunchKernelGGL((k3_cols_inv_c512<2, false, 1>), g1, dim3(512), kC512Lds, st, job, work, pl, out_scale, scan);
            else if (half) hipLaunchKernelGGL((k3_cols_inv_c512<1, false, 1>), g1, dim3(512), kC512Lds, st, job, work, pl, out_scale, scan);
            else hipLaunchKernelGGL((k3_cols_inv_c512<0, false, 1>), g1, dim3(512), kC512Lds, st, job, work, pl, out_scale, scan);
        } else {
            if (half == 2) hipLaunchKernelGGL((k3_cols_inv_r16<2, false, 1>), g1, dim3(256), kR16LdsK3, st, job, work, pl, out_scale, scan);
            else if (half) hipLaunchKernelGGL((k3_cols_inv_r16<1, false, 1>), g1, dim3(256), kR16LdsK3, st, job, work, pl, out_scale, scan);
            else hipLaunchKernelGGL((k3_cols_inv_r16<0, false, 1>), g1, dim3(256), kR16LdsK3, st, job, work, pl, out_scale, scan);
        }
        return hipGetLastError();
    }
    if (plan_is_c512w(pl)) {
        if (half || accumulate || scan.only_pairs != nullptr) return hipErrorInvalidValue;
        hipLaunchKernelGGL(k3_cols_inv_c512w, dim3((unsigned)npairs * ((1u << pl.logN2) >> kColsLog)), dim3(512), kC512Lds, st, job, work, pl, out_scale, scan);
        return hipGetLastError();
    }
    if (accumulate && half) return hipErrorInvalidValue;   // (the accumulating forms exist for the f32 work matrix only)
    // (a score pointer that is only 4-byte aligned, or an odd hop, takes k3_finish's scalar read-modify-writes: every
    // score belongs to one thread, so the accumulating form needs no alignment either)
    if (plan_is_c1024(pl)) {
        if (half) return hipErrorInvalidValue;
        if (accumulate) hipLaunchKernelGGL(k3_cols_inv_c1024<true>, dim3((unsigned)npairs * (kN2 >> kColsLog)), dim3(1024), kC1024Lds, st, job, work,
                                           pl, out_scale, scan);
        else hipLaunchKernelGGL(k3_cols_inv_c1024<false>, dim3((unsigned)npairs * (kN2 >> kColsLog)), dim3(1024), kC1024Lds, st, job, work,
                                pl, out_scale, scan);
    } else if (accumulate && plan_is_c512(pl)) {
        hipLaunchKernelGGL((k3_cols_inv_c512<0, true>), dim3((unsigned)npairs * (kN2 >> kColsLog)), dim3(512), kC512Lds, st, job, work,
                           pl, out_scale, scan);
    } else if (accumulate && plan_is_r16(pl)) {
        hipLaunchKernelGGL((k3_cols_inv_r16<0, true>), dim3((unsigned)npairs * (kN2 >> kColsLog)), dim3(256), kR16LdsK3, st, job, work,
                           pl, out_scale, scan);
    } else if (plan_is_c512(pl)) {
        if (half == 2) hipLaunchKernelGGL(k3_cols_inv_c512<2>, dim3((unsigned)npairs * (kN2 >> kColsLog)), dim3(512), kC512Lds, st, job, work,
                                          pl, out_scale, scan);
        else if (half) hipLaunchKernelGGL(k3_cols_inv_c512<1>, dim3((unsigned)npairs * (kN2 >> kColsLog)), dim3(512), kC512Lds, st, job, work,
                                          pl, out_scale, scan);
        else hipLaunchKernelGGL(k3_cols_inv_c512<0>, dim3((unsigned)npairs * (kN2 >> kColsLog)), dim3(512), kC512Lds, st, job, work,
                                pl, out_scale, scan);
    } else if (plan_is_r16(pl)) {
        if (half == 2) hipLaunchKernelGGL(k3_cols_inv_r16<2>, dim3((unsigned)npairs * (kN2 >> kColsLog)), dim3(256), kR16LdsK3, st, job, work,
                                          pl, out_scale, scan);
        else if (half) hipLaunchKernelGGL(k3_cols_inv_r16<1>, dim3((unsigned)npairs * (kN2 >> kColsLog)), dim3(256), kR16LdsK3, st, job, work,
                                          pl, out_scale, scan);
        else hipLaunchKernelGGL(k3_cols_inv_r16<0>, dim3((unsigned)npairs * (kN2 >> kColsLog)), dim3(256), kR16LdsK3, st, job, work,
                                pl, out_scale, scan);
    } else {
        const size_t lds = (sizeof(float2) << pl.logN1) << kColsLog;
        hipLaunchKernelGGL(k3_cols_inv_gen<kColsLog>, grid, dim3(kFftThreads), lds, st, job, work, pl, out_scale, accumulate ? 1 : 0);
    }
    return hipGetLastError();
}

}  // namespace am
