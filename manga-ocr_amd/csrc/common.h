// Shared device/host helpers for the Manga-OCR MI355X engine (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short bf16_t;  // raw bfloat16 bits

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define MOCR_WAVE 64

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN stays NaN
    return *reinterpret_cast<bf16_t*>(&b);
}

// two floats -> one dword of two bf16 (ONE v_cvt_pk_bf16_f32; lo in the low half)
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    const bf16x2_t v = __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t);
    return *reinterpret_cast<const unsigned*>(&v);
}

// ---- typed element access: T is float (parity mode) or bf16_t (bf16 storage) ------------------
template <typename T> struct elem;
template <> struct elem<float> {
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
    static __device__ __forceinline__ void ld4(const float* p, float* o) {
        float4 v = *reinterpret_cast<const float4*>(p);
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    }
    static __device__ __forceinline__ void st4(float* p, const float* v) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    }
    static __device__ __forceinline__ void ld8(const float* p, float* o) { ld4(p, o); ld4(p + 4, o + 4); }
    static __device__ __forceinline__ void st8(float* p, const float* v) { st4(p, v); st4(p + 4, v + 4); }
};
template <> struct elem<bf16_t> {
    static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
    static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
    static __device__ __forceinline__ void ld4(const bf16_t* p, float* o) {
        uint2 v = *reinterpret_cast<const uint2*>(p);
        o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
        o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
    }
    static __device__ __forceinline__ void st4(bf16_t* p, const float* v) {
        uint2 u;
        u.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
        u.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
        *reinterpret_cast<uint2*>(p) = u;
    }
    static __device__ __forceinline__ void ld8(const bf16_t* p, float* o) {
        uint4 v = *reinterpret_cast<const uint4*>(p);
        o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
        o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
        o[4] = __uint_as_float(v.z << 16); o[5] = __uint_as_float(v.z & 0xffff0000u);
        o[6] = __uint_as_float(v.w << 16); o[7] = __uint_as_float(v.w & 0xffff0000u);
    }
    static __device__ __forceinline__ void st8(bf16_t* p, const float* v) {
        uint4 u;
        u.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
        u.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
        u.z = (unsigned)f2bf(v[4]) | ((unsigned)f2bf(v[5]) << 16);
        u.w = (unsigned)f2bf(v[6]) | ((unsigned)f2bf(v[7]) << 16);
        *reinterpret_cast<uint4*>(p) = u;
    }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// exact-erf GELU (hidden_act="gelu"): 0.5 x (1 + erf(x / sqrt(2)))
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// bf16 storage: x * Phi(x) with Phi(x) ~ 1 / (1 + exp(-x (a + b x^2 + c x^4))): |error| <= 2.6e-5 against the erf form
// over the whole line (minimax fit of the three coefficients, r03) - 1/150 of a bf16 half-ulp at |y| ~ 1 - in 9 VALU
// instructions, two of them transcendental (r01-r02 used erf from Abramowitz & Stegun 7.1.26: 16 instructions; at batch
// 256 the FC1 epilogue spent 4.3 us per 256 x 256 tile on it).  x^2 is clamped at 64: the odd polynomial turns over at
// |x| ~ 10, and at |x| = 8 the sigmoid already is 0 or 1 in fp32.  The constants carry the -log2(e) of exp2.  EVERY bf16
// kernel with a GELU epilogue uses this one function, so a value does not depend on which kernel computed it; the fp32
// parity mode keeps erff.
__device__ __forceinline__ float gelu_fast(float x) {
    const float x2 = fminf(x * x, 64.0f);
    float t = fmaf(x2, 1.0142628e-3f, -1.0677572e-1f);      // -log2e * (c x^2 + b)
    t = fmaf(x2, t, -2.3011213f);                            // -log2e * a
    const float e = __builtin_amdgcn_exp2f(x * t);
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}
template <typename T> __device__ __forceinline__ float gelu_for(float x) {
    if constexpr (sizeof(T) == 2) return gelu_fast(x); else return gelu_erf(x);
}

// four floats -> four OCP e4m3 bytes (v_cvt_pk_fp8_f32: round to nearest even); callers keep |v| <= 448
__device__ __forceinline__ unsigned pack4_fp8(float a, float b, float c, float d) {
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return (unsigned)w;
}

// 16-byte non-temporal global accesses (streams that are written / read once: see the users)
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st16_nt(void* p, uint4 v) {
    u32x4_t t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    __builtin_nontemporal_store(t, reinterpret_cast<u32x4_t*>(p));
}
__device__ __forceinline__ uint4 ld16_nt(const void* p) {
    const u32x4_t t = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
    return make_uint4(t.x, t.y, t.z, t.w);
}

// Async global -> LDS copy, 16 bytes per lane.  `lds_wave_base` must be wave-uniform: the
// hardware writes lane i's 16 bytes at lds_wave_base + 16*i (cdna_hip_programming.md §5).
__device__ __forceinline__ void glds16(const void* gsrc_lane, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc_lane,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
// The same with the non-temporal cache policy (aux = 2, `nt`): for bytes one CU reads once per launch
// (MI355X_MICROARCH.md, nt-weights row)
__device__ __forceinline__ void glds16_nt(const void* gsrc_lane, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc_lane,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 2);
}
// the latent attention's key stream is such a stream (r02: +2.7 % crops/s at 2560-row batches, -7 % on an isolated
// 320..1024-row batch); -DMOCR_LAT_NO_NT builds the default-policy form for A/B runs
#ifdef MOCR_LAT_NO_NT
#define LAT_GLDS glds16
#else
#define LAT_GLDS glds16_nt
#endif
