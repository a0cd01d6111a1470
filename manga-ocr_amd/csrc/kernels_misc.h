// Memory-bound helper kernels of the encoder: pixel normalisation + patchify, CLS rows,
// LayerNorm.  All are bandwidth kernels: 16-byte accesses per lane, one wave per row for the
// reductions (wavefront shuffles, no LDS).
#pragma once
#include "common.h"

// RGB -> L with Pillow's fixed-point weights (libImaging/Convert.c):
//   L = (19595 R + 38470 G + 7471 B + 0x8000) >> 16
__global__ void rgb_to_l_kernel(const uint8_t* __restrict__ rgb, uint8_t* __restrict__ gray, long long npix) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const unsigned r = rgb[3 * i], g = rgb[3 * i + 1], b = rgb[3 * i + 2];
    gray[i] = (uint8_t)((r * 19595u + g * 38470u + b * 7471u + 0x8000u) >> 16);
}

// uint8 luminance [B][IMG][IMG] -> normalised patch matrix A[B*G*G][P*P] (T).
// The three input channels of the ViT are identical after convert('L').convert('RGB'), so the
// patch-embedding weight is summed over channels on the host and K = P*P = 256.
// lut[u] = float32(float64(u)/255) - 0.5) / 0.5 exactly as the HF image processor computes it.
// One thread = one 16-pixel patch row: a 16-byte load, 16 outputs.
template <typename T>
__global__ void patchify_kernel(const uint8_t* __restrict__ gray, const float* __restrict__ lut,
                                T* __restrict__ A, int B, int IMG, int P) {
    const int G = IMG / P;
    const long long total = (long long)B * IMG * G;   // (b, y, px)
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int px = (int)(i % G);
    const int y = (int)((i / G) % IMG);
    const int b = (int)(i / ((long long)G * IMG));
    const int py = y / P, ky = y - py * P;
    const uint4 raw = *reinterpret_cast<const uint4*>(gray + ((size_t)b * IMG + y) * IMG + px * P);
    const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = lut[(w[j >> 2] >> (8 * (j & 3))) & 0xff];
    T* dst = A + ((size_t)b * G * G + py * G + px) * (P * P) + ky * P;
    elem<T>::st8(dst, v);
    elem<T>::st8(dst + 8, v + 8);
}

// X[b*tokens + 0][:] = cls + pos[0]
__global__ void cls_rows_kernel(const float* __restrict__ cls, const float* __restrict__ pos,
                                float* __restrict__ X, int B, int tokens, int D) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * D) return;
    const int b = i / D, d = i - b * D;
    X[(size_t)b * tokens * D + d] = cls[d] + pos[d];
}

// The encoder of a few crops splits its O-proj / FC2 GEMMs over K (engine.hip, run_encoder): the LayerNorm that follows
// first finishes them - x += bias + sum of the fp32 slabs (fixed order: deterministic), written back in place as the
// residual stream - and then normalises as layernorm_kernel does.
template <typename TO, int D>
__global__ __launch_bounds__(256) void layernorm_slab_kernel(float* __restrict__ x, const float* __restrict__ slabs, int nslab,
                                                             long long slab_stride, const float* __restrict__ bias,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             TO* __restrict__ out, int M, float eps) {
    static_assert(D % 256 == 0, "row = k * 64 lanes * 4");
    constexpr int V = D / 256;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= M) return;
    float* xr = x + (size_t)row * D;
    const float* sr = slabs + (size_t)row * D;
    float v[V * 4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int c = i * 256 + lane * 4;
        float4 t = *reinterpret_cast<const float4*>(xr + c);
        const float4 bb = *reinterpret_cast<const float4*>(bias + c);
        float4 acc = *reinterpret_cast<const float4*>(sr + c);
        for (int k = 1; k < nslab; ++k) {
            const float4 u = *reinterpret_cast<const float4*>(sr + (size_t)k * slab_stride + c);
            acc.x += u.x; acc.y += u.y; acc.z += u.z; acc.w += u.w;
        }
        t.x += acc.x + bb.x; t.y += acc.y + bb.y; t.z += acc.z + bb.z; t.w += acc.w + bb.w;
        *reinterpret_cast<float4*>(xr + c) = t;
        v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w;
        s += (t.x + t.y) + (t.z + t.w);
    }
    const float mean = wave_sum(s) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < V * 4; ++i) { const float d = v[i] - mean; q += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / D) + eps);
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int c = i * 256 + lane * 4;
        const float4 g = *reinterpret_cast<const float4*>(gamma + c);
        const float4 b = *reinterpret_cast<const float4*>(beta + c);
        float o[4] = {(v[4 * i] - mean) * rstd * g.x + b.x, (v[4 * i + 1] - mean) * rstd * g.y + b.y,
                      (v[4 * i + 2] - mean) * rstd * g.z + b.z, (v[4 * i + 3] - mean) * rstd * g.w + b.w};
        elem<TO>::st4(out + (size_t)row * D + c, o);
    }
}

// LayerNorm over rows of 768 fp32 (eps 1e-12: two-pass statistics in fp32, TF/models/vit/
// configuration_vit.py:58).  One wave per row, 12 elements per lane as 3 float4.
template <typename TO, int D>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, TO* __restrict__ out,
                                                        int M, float eps) {
    static_assert(D % 256 == 0, "row = k * 64 lanes * 4");
    constexpr int V = D / 256;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= M) return;
    const float* xr = x + (size_t)row * D;
    float v[V * 4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const float4 t = *reinterpret_cast<const float4*>(xr + i * 256 + lane * 4);
        v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w;
        s += (t.x + t.y) + (t.z + t.w);
    }
    const float mean = wave_sum(s) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < V * 4; ++i) { const float d = v[i] - mean; q += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / D) + eps);
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int c = i * 256 + lane * 4;
        const float4 g = *reinterpret_cast<const float4*>(gamma + c);
        const float4 b = *reinterpret_cast<const float4*>(beta + c);
        float o[4] = {(v[4 * i] - mean) * rstd * g.x + b.x, (v[4 * i + 1] - mean) * rstd * g.y + b.y,
                      (v[4 * i + 2] - mean) * rstd * g.z + b.z, (v[4 * i + 3] - mean) * rstd * g.w + b.w};
        elem<TO>::st4(out + (size_t)row * D + c, o);
    }
}

// LayerNorm folded into the encoder GEMMs (gemm_pers_kernel LNF): what the fp32-residual GEMMs emit next to x for every
// layer, for the rows the patch embedding wrote: x as bf16 (the next GEMM's A operand - x itself, not LN(x)) and the row's
// (sum, sum of squares) in partial 0 of ln_part[row][4][2] (partials 1..3 zero).  One wave per row.
template <int D>
__global__ __launch_bounds__(256) void ln_prep_kernel(const float* __restrict__ x, bf16_t* __restrict__ xb, float* __restrict__ part, int M) {
    static_assert(D % 256 == 0, "row = k * 64 lanes * 4");
    constexpr int V = D / 256;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= M) return;
    const float* xr = x + (size_t)row * D;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int c = i * 256 + lane * 4;
        const float4 t = *reinterpret_cast<const float4*>(xr + c);
        uint2 u;
        u.x = pack_bf16x2(t.x, t.y); u.y = pack_bf16x2(t.z, t.w);
        *reinterpret_cast<uint2*>(xb + (size_t)row * D + c) = u;
        s1 += (t.x + t.y) + (t.z + t.w);
        s2 += fmaf(t.x, t.x, t.y * t.y) + fmaf(t.z, t.z, t.w * t.w);
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (lane < 2) {
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane == 0) { o.x = s1; o.y = s2; }
        *reinterpret_cast<float4*>(part + (size_t)row * 8 + 4 * lane) = o;
    }
}

