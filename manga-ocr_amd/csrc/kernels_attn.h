// Attention kernels.
//   enc_attn_simple_kernel<T> : encoder self-attention on the VALU, fp32 math (parity mode, and the
//                               on-device cross-check of the MFMA kernel)
//   enc_attn_mfma_kernel      : encoder self-attention on the matrix cores (bf16), whole K/V of one
//                               head resident in LDS, softmax in registers
//   dec_attn_kernel<T,SELF>   : single-query attention of the decode step over the self cache
//                               (<= 300 keys, appends the new K/V) or the cross K/V (197 keys).
//                               HBM-bound: streams each key/value row exactly once, 1 KiB per
//                               wave-instruction.
// Encoder sequence = 197 tokens, 12 heads x 64 (TF/models/vit/modeling_vit.py:164-238); no mask.
#pragma once
#include "common.h"

// LDS byte address (what ds_* instructions take) of a pointer into the dynamic LDS array
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const char*)p;
}

// ------------------------------------------------------------------------------------------------
// Encoder attention, VALU.  One block per (image, head); K as fp32 [S][65] and V as fp32 [S][64]
// in LDS.  Each wave owns every 4th query row: lane j scores keys j, j+64, ...; lane d accumulates
// output dim d.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void enc_attn_simple_kernel(const T* __restrict__ qkv, T* __restrict__ ctx,
                                                              int S, int H, int ld_qkv, int ld_ctx, float scale) {
    constexpr int DH = 64, KP = 65, SMAX = 200;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sK = reinterpret_cast<float*>(smem);                 // [S][65]
    float* sV = sK + SMAX * KP;                                 // [S][64]
    float* sQ = sV + SMAX * DH;                                 // [4][64]
    float* sP = sQ + 4 * DH;                                    // [4][256]
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const T* base = qkv + (size_t)b * S * ld_qkv + h * DH;
    const int D = H * DH;
    for (int i = tid; i < S * (DH / 4); i += 256) {
        const int s = i / (DH / 4), c = (i % (DH / 4)) * 4;
        float k4[4], v4[4];
        elem<T>::ld4(base + (size_t)s * ld_qkv + D + c, k4);
        elem<T>::ld4(base + (size_t)s * ld_qkv + 2 * D + c, v4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { sK[s * KP + c + e] = k4[e]; sV[s * DH + c + e] = v4[e]; }
    }
    __syncthreads();
    float* q = sQ + wave * DH;
    float* pr = sP + wave * 256;
    for (int qi = wave; qi < S; qi += 4) {
        q[lane] = elem<T>::ld(base + (size_t)qi * ld_qkv + lane);
        __builtin_amdgcn_wave_barrier();
        float sc[4];
        float mx = -INFINITY;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int key = lane + 64 * kk;
            float a = 0.f;
            if (key < S) {
                const float* kr = sK + key * KP;
#pragma unroll 16
                for (int d = 0; d < DH; ++d) a += q[d] * kr[d];
                a *= scale;
                mx = fmaxf(mx, a);
            }
            sc[kk] = a;
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int key = lane + 64 * kk;
            const float e = key < S ? expf(sc[kk] - mx) : 0.f;
            pr[key] = e;
            sum += e;
        }
        sum = wave_sum(sum);
        __builtin_amdgcn_wave_barrier();
        float o = 0.f;
        for (int key = 0; key < S; ++key) o += pr[key] * sV[key * DH + lane];
        elem<T>::st(ctx + ((size_t)b * S + qi) * ld_ctx + h * DH + lane, o / sum);
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------------------------------------
// enc_attn_f32_kernel (r04): the PARITY mode's encoder attention on the f32-input matrix cores.
//   The VALU kernel above spends 3.0 ms per layer at batch 256 - 36 ms of the fp32 mode's 262 ms per 256-crop batch, more than
//   any of its GEMMs.  v_mfma_f32_16x16x4_f32 is an exact fp32 fma chain (MI355X_MICROARCH.md: 32 cycles per SIMD, the fp32
//   vector rate), so the products stay fp32-exact; only the order of the additions differs from the VALU kernel's (both
//   are fp32 orderings of the same sums: the encoder rows stay within the 2e-4 the parity tests allow, measured ~1e-5).
//   One block (4 waves) per (image, head); K and V as fp32 [208][68] in LDS (rows 197 .. 207 zero).  Per 16-query unit:
//     S^T[key][q] = K . Q^T   13 key tiles x 16 MFMAs; the k index of an MFMA is a free bijection as long as both operands use
//                             it: k-step s of lane group g stands for dim 16 g + s, so a lane's sixteen k values are 16
//                             CONSECUTIVE floats of its key row (four ds_read_b128) / of its query row (four global loads);
//     softmax                 lane (q = lane & 15, g) holds keys 16 kt + 4 g + r of every key tile: lane-local + two exchanges;
//     O^T[d][q] = V^T . P^T   k-step (kt, r) of lane group g stands for key 16 kt + 4 g + r - where the lane's probability
//                             register already is; A = V[key][16 dt + (lane & 15)]: one ds_read_b32 per MFMA.
// ------------------------------------------------------------------------------------------------
#ifndef ENC_S
#define ENC_S 197
#endif
#define EAF_ROWS 208
#define EAF_LD 68
#define EAF_LDS (2 * EAF_ROWS * EAF_LD * 4)

__global__ __launch_bounds__(256) void enc_attn_f32_kernel(const float* __restrict__ qkv, float* __restrict__ ctx, int H, int ld_qkv,
                                                          int ld_ctx) {
    constexpr int DH = 64, KT = 13;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const sK = reinterpret_cast<float*>(smem);
    float* const sV = sK + EAF_ROWS * EAF_LD;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int D = H * DH;
    const float* base = qkv + (size_t)b * ENC_S * ld_qkv + h * DH;
    for (int i = tid; i < EAF_ROWS * (DH / 4); i += 256) {
        const int r = i >> 4, c = (i & 15) * 4;
        float4 k4 = make_float4(0.f, 0.f, 0.f, 0.f), v4 = k4;
        if (r < ENC_S) {
            k4 = *reinterpret_cast<const float4*>(base + (size_t)r * ld_qkv + D + c);
            v4 = *reinterpret_cast<const float4*>(base + (size_t)r * ld_qkv + 2 * D + c);
        }
        *reinterpret_cast<float4*>(sK + r * EAF_LD + c) = k4;
        *reinterpret_cast<float4*>(sV + r * EAF_LD + c) = v4;
    }
    __syncthreads();
    for (int unit = (wave + (int)blockIdx.x) & 3; unit < KT; unit += 4) {      // (the slot a wave takes rotates with the block)
        const int q = unit * 16 + l15;
        // this lane's query row, dims 16 g .. 16 g + 15 (k-steps 0 .. 15 of lane group g)
        float qf[16];
        {
            const float* qrow = base + (size_t)(q < ENC_S ? q : ENC_S - 1) * ld_qkv + 16 * g;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 t = *reinterpret_cast<const float4*>(qrow + 4 * j);
                qf[4 * j] = t.x; qf[4 * j + 1] = t.y; qf[4 * j + 2] = t.z; qf[4 * j + 3] = t.w;
            }
        }
        // ---- S^T[key][q]: lane (q = l15, g) ends up with keys 16 kt + 4 g + r
        f32x4 st[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            const float* krow = sK + (16 * kt + l15) * EAF_LD + 16 * g;
            float kf[16];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 t = *reinterpret_cast<const float4*>(krow + 4 * j);
                kf[4 * j] = t.x; kf[4 * j + 1] = t.y; kf[4 * j + 2] = t.z; kf[4 * j + 3] = t.w;
            }
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[s], qf[s], acc, 0, 0, 0);
            st[kt] = acc;
        }
        // ---- softmax over the 197 keys of this lane's query (52 here, the rest in the lanes l15 + 16, + 32, + 48)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (192 + 4 * g + r >= ENC_S) st[12][r] = -INFINITY;
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[kt][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = expf((st[kt][r] - mx) * 0.125f);      // exp(-inf) = 0 for the padded keys
                st[kt][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        // ---- O^T[d][q] += V[key][d] P[key][q]: k-step (kt, r), lane group g <-> key 16 kt + 4 g + r
        f32x4 oacc[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float* vrow = sV + (16 * kt + 4 * g + r) * EAF_LD + l15;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vrow[16 * dt], st[kt][r], oacc[dt], 0, 0, 0);
            }
        // lane (q = l15, g) holds d = 16 dt + 4 g + r of its query
        if (q < ENC_S) {
            const float inv = 1.0f / sum;
            float* orow = ctx + ((size_t)b * ENC_S + q) * ld_ctx + h * DH + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                *reinterpret_cast<float4*>(orow + 16 * dt) = make_float4(oacc[dt][0] * inv, oacc[dt][1] * inv, oacc[dt][2] * inv, oacc[dt][3] * inv);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Encoder attention on MFMA (bf16).  One block (4 waves) per (image, head).
//   LDS:  sK  [224 keys][64 d] bf16, 128-B rows, 16-B chunks XOR-swizzled like the GEMM tiles
//         sVt [64 d][228 keys] bf16 (V transposed; 228 = conflict-free ds_read_b64 row stride)
//   A wave takes 32 queries at a time.  S^T = K.Q^T is computed "swapped" (keys on the MFMA rows,
//   the query on the lane), so each lane holds all 224 scores of ITS query in registers: the row
//   softmax is lane-local plus one exchange with lane^32.  The probability tile is then, without
//   any data movement, the B operand of O^T = V^T.P^T (cdna_hip_programming.md §3, "An accumulator
//   tile as the next MFMA's operand"): registers 8s..8s+7 -> bf16 = k-step s, whose k order is
//   key = 16s + 8(j>>2) + 4(lane>>5) + (j&3); the V^T fragment is gathered in that same order.
// ------------------------------------------------------------------------------------------------
#define ENC_S 197
#ifdef MOCR_EXPERIMENTS      // r01-r02 kernel, kept for A/B (impl code 2 of mocr_op_enc_attention in the experiments build)
#define ENC_SP 224
#define ENC_VT_LD 228

__global__ __launch_bounds__(256, 2) void enc_attn_mfma_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ ctx,
                                                            int H, int ld_qkv, int ld_ctx, int ablate_arg = 0) {
    constexpr int DH = 64;
#ifdef MOCR_EXPERIMENTS
    const int ablate = ablate_arg;          // diagnostics (MOCR_ENC_ATTN_ABLATE): 1 = no query tiles (staging only), 2 = no K / V staging
#else
    constexpr int ablate = 0;
#endif
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sK = smem;                                              // 224 * 128 B
    bf16_t* sVt = reinterpret_cast<bf16_t*>(smem + ENC_SP * 128); // 64 * 228 * 2 B
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r32 = lane & 31, hh = lane >> 5;
    const int D = H * DH;
    const bf16_t* base = qkv + (size_t)b * ENC_S * ld_qkv + h * DH;

    // ---- the queries of this wave's two tiles, requested before anything else (r02: they used to be loaded at the head of
    // each tile, a full global-load latency in front of the tile's first MFMA)
    bf16x8 qpre[2][4];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
        const int qq = (wave + 4 * ti) * 32 + r32;
        const int qc = qq < ENC_S ? qq : ENC_S - 1;
#pragma unroll
        for (int s = 0; s < 4; ++s)
            qpre[ti][s] = *reinterpret_cast<const bf16x8*>(base + (size_t)qc * ld_qkv + s * 16 + hh * 8);
    }
    if (!(ablate & 2))
    // ---- stage K (swizzled rows) and V^T; zero the padding keys 197..223(227).  ALL global loads of a thread are issued
    // before the first LDS store (r02): left as a rolled loop the seven + four load -> store round trips ran one
    // after the other, a global-load latency each (126 -> 97 us per 3072-block launch at batch 256)
    {
        uint4 kv[7];
#pragma unroll
        for (int it = 0; it < 7; ++it) {
            const int i = tid + 256 * it, s_ = i >> 3, c = i & 7;
            kv[it] = make_uint4(0, 0, 0, 0);
            if (s_ < ENC_S) kv[it] = *reinterpret_cast<const uint4*>(base + (size_t)s_ * ld_qkv + D + c * 8);
        }
        uint4 v0[4], v1[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int i = tid + 256 * it, s_ = (i >> 3) * 2, c = i & 7;
            v0[it] = make_uint4(0, 0, 0, 0); v1[it] = make_uint4(0, 0, 0, 0);
            if (i < (ENC_VT_LD / 2) * 8) {
                if (s_ < ENC_S) v0[it] = *reinterpret_cast<const uint4*>(base + (size_t)s_ * ld_qkv + 2 * D + c * 8);
                if (s_ + 1 < ENC_S) v1[it] = *reinterpret_cast<const uint4*>(base + (size_t)(s_ + 1) * ld_qkv + 2 * D + c * 8);
            }
        }
#pragma unroll
        for (int it = 0; it < 7; ++it) {
            const int i = tid + 256 * it, s_ = i >> 3, c = i & 7;
            *reinterpret_cast<uint4*>(sK + s_ * 128 + ((c ^ ((s_ >> 1) & 7)) << 4)) = kv[it];
        }
        // V^T: a thread takes the same 8 d-values of TWO adjacent keys and writes (key, key+1) pairs as dwords
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int i = tid + 256 * it, s_ = (i >> 3) * 2, c = i & 7;
            if (i < (ENC_VT_LD / 2) * 8) {
                const unsigned w0[4] = {v0[it].x, v0[it].y, v0[it].z, v0[it].w}, w1[4] = {v1[it].x, v1[it].y, v1[it].z, v1[it].w};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const unsigned lo = (w0[e >> 1] >> (16 * (e & 1))) & 0xffffu, hi = (w1[e >> 1] >> (16 * (e & 1))) & 0xffffu;
                    *reinterpret_cast<unsigned*>(sVt + (c * 8 + e) * ENC_VT_LD + s_) = lo | (hi << 16);
                }
            }
        }
    }
    __syncthreads();

    // gridDim.y = 2 (a few crops: n * H blocks would leave most CUs idle): block y takes the query tiles 4 y .. 4 y + 3
    for (int qt = wave + (gridDim.y > 1 ? 4 * (int)blockIdx.y : 0); qt < ((ablate & 1) ? 0 : ENC_SP / 32); qt += 4 * (int)gridDim.y) {
        const int q = qt * 32 + r32;                // (padded queries are computed from a clamped row, not stored)
        // B operand of S^T = K.Q^T : lane holds Q[q][16s + 8hh .. +7]
        bf16x8 qf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = qt < 4 ? qpre[0][s] : qpre[1][s];
        f32x16 st[7];
#pragma unroll
        for (int kt = 0; kt < 7; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kt][r] = 0.f;
            const int row = kt * 32 + r32;
            const int sw = (row >> 1) & 7;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(sK + row * 128 + (((2 * s + hh) ^ sw) << 4));
                st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[s], st[kt], 0, 0, 0);
            }
            if (kt & 1) __builtin_amdgcn_sched_barrier(0);   // bound the fragment prefetch depth (registers: 2 waves/SIMD)
        }
        // ---- softmax over the 224 keys of this lane's query (112 here, 112 in lane^32)
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 7; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (kt == 6 && key >= ENC_S) st[kt][r] = -INFINITY;
                mx = fmaxf(mx, st[kt][r]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float c0 = 0.125f * 1.44269504088896340736f;   // scale * log2(e)
        const float mb = mx * c0;
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 7; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __builtin_amdgcn_exp2f(st[kt][r] * c0 - mb);   // v_exp_f32; exp2(-inf) = 0 for the padded keys
                st[kt][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 32, 64);
        // ---- O^T[d][q] = sum_key V^T[d][key] P^T[key][q]
        f32x16 oacc[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
#pragma unroll
        for (int kt = 0; kt < 7; ++kt) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bf16x8 pf;
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[j] = (__bf16)st[kt][8 * s + j];
                const int key0 = kt * 32 + 16 * s + 4 * hh;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const bf16_t* vr = sVt + (dt * 32 + r32) * ENC_VT_LD + key0;
                    const uint2 lo = *reinterpret_cast<const uint2*>(vr);       // keys key0 .. +3   (j = 0..3)
                    const uint2 hi = *reinterpret_cast<const uint2*>(vr + 8);   // keys key0+8 .. +11 (j = 4..7)
                    union { uint4 u; bf16x8 v; } cvt;
                    cvt.u = make_uint4(lo.x, lo.y, hi.x, hi.y);
                    oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cvt.v, pf, oacc[dt], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (q < ENC_S) {
            const float inv = 1.0f / sum;
            bf16_t* orow = ctx + ((size_t)b * ENC_S + q) * ld_ctx + h * DH;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float o4[4] = {oacc[dt][4 * g] * inv, oacc[dt][4 * g + 1] * inv, oacc[dt][4 * g + 2] * inv,
                                   oacc[dt][4 * g + 3] * inv};
                    elem<bf16_t>::st4(orow + dt * 32 + 8 * g + 4 * hh, o4);   // d = 32dt + 8g + 4hh + (r&3)
                }
        }
    }
}

#endif  // MOCR_EXPERIMENTS

// ------------------------------------------------------------------------------------------------
// enc_attn2_kernel (r03): encoder self-attention whose K and V arrive by LDS-DMA.
//   r02's kernel above staged K and V through registers (V with a scatter transpose: 32 ds_write_b32 per thread) and
//   spent 37 of its 94 us per batch-256 launch there, NOT overlapped with the other block's query tiles - staging and
//   softmax both live on the VALU (r03 ablations: staging alone 37 us, query tiles alone 64 us, both 94 us).  Here:
//   * K and V of the head are copied ROW-MAJOR ([208 keys][128 B], 16-byte chunk c of row r at chunk c ^ ((r>>1)&7),
//     applied on the source address) by global_load_lds: 13 one-KiB requests per wave, no register, no VALU; 52 KiB per
//     block, so three blocks share a CU and one block's copy runs under the others' query tiles;
//   * S^T = K.Q^T on v_mfma_f32_16x16x32_bf16 per 16-query unit (13 units for 197 queries, 13 key tiles of 16 for 197
//     keys: 5 % padding each way instead of 14 %): a lane ends up with one query's scores for keys 4g .. 4g+3 of every
//     key tile - softmax is lane-local plus two exchanges across the four lane groups;
//   * O^T = V^T.P^T: the probabilities of two key tiles, as they lie in the registers, ARE the B operand (its 32 k-slots
//     then stand for keys {4g+j of tile 2u, 4g+j of tile 2u+1}); the A operand follows the same key order by reading V
//     column-wise with ds_read_b64_tr_b16: rows 32u + 4g .. +3 and 32u + 16 + 4g .. +3 of the row-major image - no
//     transposed copy of V anywhere.
//   Keys 197 .. 207: rows 197-199 are copies of key 196 (the copy's source row is clamped: nothing behind an image's 197
//   rows is read), rows 200-207 are zeroed once; their scores are set to -inf, so their probabilities are exactly 0.
// ------------------------------------------------------------------------------------------------
#define EA2_ROWS 208
#define EA2_LDS (2 * EA2_ROWS * 128)

__global__ __launch_bounds__(256, 3) void enc_attn2_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ ctx,
                                                        int H, int ld_qkv, int ld_ctx, int ablate_arg = 0) {
    constexpr int DH = 64, KT = 13;
#ifdef MOCR_EXPERIMENTS
    const int ablate = ablate_arg;      // diagnostics (MOCR_ENC_ATTN_ABLATE): 1 no S product, 2 no exp, 4 no P.V, 8 no K / V copy, 16 no stores, 32 no query loads after the first
#else
    constexpr int ablate = 0;
#endif
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const sK = smem;
    char* const sV = smem + EA2_ROWS * 128;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int D = H * DH;
    const bf16_t* base = qkv + (size_t)b * ENC_S * ld_qkv + h * DH;
    if (ablate >> 8) {                      // experiment: the three blocks a CU starts with begin (ablate >> 8) x 0.5 us apart
        const int phase = (int)(blockIdx.x >> 8);
        if (phase < 3)
            for (int i = 0; i < phase * (ablate >> 8); ++i) __builtin_amdgcn_s_sleep(16);
    }

    // (r03, measured and dropped: all of K requested first and the score products started behind a counted wait for it, V
    // waited for - and the block synchronised a second time - in front of the first P.V: 82 -> 85-86 us per batch-256
    // launch.  Ablations of the same launch, us: as is 82.1, no score product 73.8, no exp 82.6, no P.V 74.4, no K / V copy
    // 64.2, no stores 68.8, none of the products 66.3, neither products nor copies 40.0, nothing but the softmax's VALU and
    // the blocks' turnover 34.9: the copies and the stores are what three blocks per CU do not hide.)
    // ---- K and V: 25 pieces of 8 rows x 128 B each (rows 0 .. 199), lane -> row lane>>3, physical chunk lane&7
    {
        const int prow = lane >> 3, pch = lane & 7;
        for (int pc = wave; pc < ((ablate & 8) ? 0 : 25); pc += 4) {
            const int row = pc * 8 + prow;
            const int c = pch ^ ((row >> 1) & 7);
            // rows 197 .. 199 of the LDS image: a second copy of key 196 (their scores are -inf, their probabilities exactly 0, but
            // 0 x whatever FOLLOWS the image's rows in memory must stay finite - and for the last image nothing follows)
            const bf16_t* src = base + (size_t)(row < ENC_S ? row : ENC_S - 1) * ld_qkv + c * 8;
            glds16(src + D, sK + pc * 1024);
            glds16(src + 2 * D, sV + pc * 1024);
        }
        if (tid < 64) {                                              // rows 200 .. 207: never copied
            *reinterpret_cast<uint4*>(sK + 200 * 128 + tid * 16) = make_uint4(0, 0, 0, 0);
            *reinterpret_cast<uint4*>(sV + 200 * 128 + tid * 16) = make_uint4(0, 0, 0, 0);
        }
    }
    // this wave's 16-query units: slot, slot + 4 * gridDim.y, ... (the unit count per slot differs by one: the slot a
    // wave takes rotates with the block, so that the SIMDs of a CU - which host the same wave number of three blocks -
    // get equal work)
    const int nslots = 4 * (int)gridDim.y;
    const int slot = ((wave + (int)blockIdx.x) & 3) + 4 * (int)blockIdx.y;
    // first unit's queries: requested before the wait for the copies
    auto load_q = [&](int unit, bf16x8 (&qf)[2]) {
        const int qq = unit * 16 + l15;
        const bf16_t* qrow = base + (size_t)(qq < ENC_S ? qq : ENC_S - 1) * ld_qkv + 8 * g;
        qf[0] = *reinterpret_cast<const bf16x8*>(qrow);
        qf[1] = *reinterpret_cast<const bf16x8*>(qrow + 32);
    };
    bf16x8 qf[2];
    if (slot < KT) load_q(slot, qf);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // transposed block reads of V: lane 4q+p of a 16-lane group supplies row q, columns 4p .. 4p+3 of a 4 x 16 block.
    // Row 32u + 4g + q (and the same + 16): its swizzle term ((row >> 1) & 7) does not depend on u, so a lane needs ONE
    // address per 16-column tile for the whole kernel; u and the + 16 rows are immediate offsets of the instruction.
    const int q4 = l15 >> 2, p4 = l15 & 3;
    unsigned vaddr[4];
    {
        const int row = 4 * g + q4, sw = (row >> 1) & 7;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
            vaddr[dt] = lds_addr(sV) + row * 128 + (((2 * dt + (p4 >> 1)) ^ sw) << 4) + (p4 & 1) * 8;
    }
    // K rows 16 kt + l15: likewise one swizzle term per lane; kt is an immediate offset
    const int ksw = (l15 >> 1) & 7;
    const char* const kp0 = sK + l15 * 128 + ((g ^ ksw) << 4);
    const char* const kp1 = sK + l15 * 128 + (((4 + g) ^ ksw) << 4);
    for (int unit = slot; unit < KT; unit += nslots) {
        const int q = unit * 16 + l15;
        // ---- S^T[key][q] over the 13 key tiles: lane (q = l15, g) gets keys 16 kt + 4g + r
        f32x4 st[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(kp0 + kt * 2048);
            const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(kp1 + kt * 2048);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (ablate & 1) { st[kt] = f32x4{(float)kt, 1.f, 2.f, (float)l15}; continue; }
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, qf[0], acc, 0, 0, 0);
            st[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, qf[1], acc, 0, 0, 0);
        }
        // the next unit's queries: requested now, used after this unit's softmax and P.V
        bf16x8 qn[2] = {qf[0], qf[1]};
        if (unit + nslots < KT && !(ablate & 32)) load_q(unit + nslots, qn);
        // ---- softmax over the 197 keys of this lane's query (52 here, the rest in the lanes l15 + 16, + 32, + 48)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (192 + 4 * g + r >= ENC_S) st[12][r] = -INFINITY;
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[kt][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float c0 = 0.125f * 1.44269504088896340736f;   // scale * log2(e)
        const float mb = mx * c0;
        float sum = 0.f;
        unsigned pk[KT][2];                                   // probabilities as bf16 pairs
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            float e[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                e[r] = (ablate & 2) ? st[kt][r] * c0 - mb : __builtin_amdgcn_exp2f(st[kt][r] * c0 - mb);      // exp2(-inf) = 0 for the padded keys
                sum += e[r];
            }
            pk[kt][0] = pack_bf16x2(e[0], e[1]);
            pk[kt][1] = pack_bf16x2(e[2], e[3]);
        }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        // ---- O^T[d][q] += V^T[d][key] P^T[key][q], two key tiles (32 k-slots) per MFMA
        f32x4 oacc[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < ((ablate & 4) ? 0 : 7); ++u) {
            union { uint4 w; bf16x8 v; } pf;
            pf.w = make_uint4(pk[2 * u][0], pk[2 * u][1], u < 6 ? pk[2 * u + 1][0] : 0u, u < 6 ? pk[2 * u + 1][1] : 0u);
            uint2 lo[4], hi[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo[dt]) : "v"(vaddr[dt]), "i"(4096 * u) : "memory");
                if (u < 6) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi[dt]) : "v"(vaddr[dt]), "i"(4096 * u + 2048) : "memory");
                else hi[dt] = make_uint2(0u, 0u);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3]));
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                union { uint4 w; bf16x8 v; } vf;
                vf.w = make_uint4(lo[dt].x, lo[dt].y, hi[dt].x, hi[dt].y);
                oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf.v, pf.v, oacc[dt], 0, 0, 0);
            }
        }
        // ---- ctx[q][64 h + d] = O^T / sum.  A lane holds d = 16 dt + 4 g + r of its query; the four lanes of a query
        // (l15 + 16 g) exchange 4 x 4 blocks (dt <-> g: v_permlane32_swap, then v_permlane16_swap) so that lane g ends up
        // with d = 16 g .. 16 g + 15: two 16-byte stores per lane instead of four 8-byte ones (r03: the 8-byte stores -
        // 16 rows x 32 B per instruction - cost 22 of the launch's 89 us at batch 256)
        {
            const float inv = 1.0f / sum;
            unsigned x[4][2];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                x[dt][0] = pack_bf16x2(oacc[dt][0] * inv, oacc[dt][1] * inv);
                x[dt][1] = pack_bf16x2(oacc[dt][2] * inv, oacc[dt][3] * inv);
            }
#pragma unroll
            for (int w = 0; w < 2; ++w) {
#pragma unroll
                for (int d0 = 0; d0 < 2; ++d0) {          // lanes 32-63 of x[d0] <-> lanes 0-31 of x[d0 + 2]
                    auto r = __builtin_amdgcn_permlane32_swap(x[d0][w], x[d0 + 2][w], false, false);
                    x[d0][w] = r[0]; x[d0 + 2][w] = r[1];
                }
#pragma unroll
                for (int d0 = 0; d0 < 4; d0 += 2) {       // odd 16-lane rows of x[d0] <-> even rows of x[d0 + 1]
                    auto r = __builtin_amdgcn_permlane16_swap(x[d0][w], x[d0 + 1][w], false, false);
                    x[d0][w] = r[0]; x[d0 + 1][w] = r[1];
                }
            }
            if (q < ENC_S && !(ablate & 16)) {
                bf16_t* orow = ctx + ((size_t)b * ENC_S + q) * ld_ctx + h * DH + 16 * g;
                *reinterpret_cast<uint4*>(orow) = make_uint4(x[0][0], x[0][1], x[1][0], x[1][1]);
                *reinterpret_cast<uint4*>(orow + 8) = make_uint4(x[2][0], x[2][1], x[3][0], x[3][1]);
            }
        }
        qf[0] = qn[0]; qf[1] = qn[1];
    }
}

// ------------------------------------------------------------------------------------------------
// Decode-step attention (query length 1).  Block = 4 waves = 4 heads of one sequence.
// Wave layout: lane = 8*g + c; the 8 lanes of group g read one 64-element K (or V) row, 8
// elements (16 B in bf16) each, so one wave-instruction streams 8 whole rows = 1 KiB contiguous.
//   q (and for SELF the new k, v) = sum of the split-K slabs of the QKV projection + bias.
//   SELF : keys 0..L-2 from the cache, key L-1 is the new token (kept in registers and appended to
//          the cache); L = step[b] + 1.       (TF/models/bert/modeling_bert.py:139-203)
//   CROSS: keys 0..196 from the cross K/V computed once per crop (modeling_bert.py:206-279).
// ------------------------------------------------------------------------------------------------
struct DecAttnParams {
    const float* slabs;      // [nslab][rows_pad][ldq] fp32 partial projections
    int nslab;
    long long slab_stride;
    int ldq;                 // 2304 (self: q|k|v) or 768 (cross: q)
    const float* bias;       // [ldq]
    const void* kbase;       // self: K cache [B][H][Lmax][64] ; cross: CKV + k column offset
    const void* vbase;
    long long kv_batch_stride;  // elements between sequences
    long long kv_head_stride;   // elements between heads
    long long kv_row_stride;    // elements between keys
    void* ctx;               // [rows][768] T
    const int* step;         // [B] current position of every row (self)
    int cross_len;           // 197 (cross)
    int H;
    float scale;
    int nt;                  // stream the K/V rows with the non-temporal policy (batches whose K/V outgrow the Infinity Cache)
    const int* rowmap;       // decode slot -> row of the batch whose K/V it attends to (identity until a batch is compacted); null: identity
};

// raw 8-element row chunk: 16 B in bf16, 32 B in fp32; loads issue without being consumed
template <typename T> struct raw8;
template <> struct raw8<bf16_t> {
    uint4 v;
    __device__ __forceinline__ void load(const bf16_t* p) { v = *reinterpret_cast<const uint4*>(p); }
    __device__ __forceinline__ void load_nt(const bf16_t* p) {      // non-temporal: a key row is read once per launch
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
        v = make_uint4(t.x, t.y, t.z, t.w);
    }
    __device__ __forceinline__ void unpack(float* o) const {
        o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
        o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
        o[4] = __uint_as_float(v.z << 16); o[5] = __uint_as_float(v.z & 0xffff0000u);
        o[6] = __uint_as_float(v.w << 16); o[7] = __uint_as_float(v.w & 0xffff0000u);
    }
};
template <> struct raw8<float> {
    float4 a, b;
    __device__ __forceinline__ void load(const float* p) {
        a = *reinterpret_cast<const float4*>(p); b = *reinterpret_cast<const float4*>(p + 4);
    }
    __device__ __forceinline__ void load_nt(const float* p) { load(p); }
    __device__ __forceinline__ void unpack(float* o) const {
        o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
    }
};

// sum over lane groups g (lanes 8g+c share c): every lane ends with the total
__device__ __forceinline__ float group_sum(float v) {
    v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
    return v;
}

// One block per (sequence, head); the keys are split over the block's 4 waves, and a wave issues
// EVERY K and V row load of its share (NG groups of 8 rows) before consuming any: one memory round
// trip per step instead of one per 8 rows.  The partial softmaxes are merged through LDS
// (running-max form).  The split-K slabs of the query projection are summed with slab s on lane
// group s mod 8, so that sum is one round trip as well.
template <typename T, bool SELF, int NG>
__global__ __launch_bounds__(256) void dec_attn_kernel(DecAttnParams p) {
    constexpr int DH = 64;
    __shared__ float s_m[4], s_l[4];
    __shared__ float s_acc[4][DH];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = lane >> 3, c = lane & 7;
    const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
    const int D = p.H * DH;
    const int col = h * DH + c * 8;

    const int L = SELF ? p.step[b] + 1 : p.cross_len;
    const int Lc = SELF ? L - 1 : L;                       // keys that live in memory
    int per = (Lc + 3) >> 2;
    per = (per + 7) & ~7;
    const int k_begin = wave * per;
    const int k_end = min(Lc, k_begin + per);
    const int br = p.rowmap ? p.rowmap[b] : b;             // K/V caches and cross K/V stay where the row was encoded
    const T* kb = reinterpret_cast<const T*>(p.kbase) + (size_t)br * p.kv_batch_stride + (size_t)h * p.kv_head_stride + c * 8;
    const T* vb = reinterpret_cast<const T*>(p.vbase) + (size_t)br * p.kv_batch_stride + (size_t)h * p.kv_head_stride + c * 8;

    // ---- issue every K/V load of this wave (clamped rows are valid addresses and get weight 0)
    raw8<T> kr[NG], vr[NG];
    const int last = Lc > 0 ? Lc - 1 : 0;
    if (Lc > 0 && p.nt) {
#pragma unroll
        for (int u = 0; u < NG; ++u) {
            const int key = min(k_begin + 8 * u + g, last);
            kr[u].load_nt(kb + (size_t)key * p.kv_row_stride);
        }
#pragma unroll
        for (int u = 0; u < NG; ++u) {
            const int key = min(k_begin + 8 * u + g, last);
            vr[u].load_nt(vb + (size_t)key * p.kv_row_stride);
        }
    } else if (Lc > 0) {
#pragma unroll
        for (int u = 0; u < NG; ++u) {
            const int key = min(k_begin + 8 * u + g, last);
            kr[u].load(kb + (size_t)key * p.kv_row_stride);
        }
#pragma unroll
        for (int u = 0; u < NG; ++u) {
            const int key = min(k_begin + 8 * u + g, last);
            vr[u].load(vb + (size_t)key * p.kv_row_stride);
        }
    }
    // ---- query (and for SELF the new key/value) = sum of split-K slabs + bias
    float q[8], kn[8], vn[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { q[e] = 0.f; kn[e] = 0.f; vn[e] = 0.f; }
    for (int s = g; s < p.nslab; s += 8) {
        const float* r = p.slabs + (size_t)s * p.slab_stride + (size_t)b * p.ldq + col;
        const float4 x0 = *reinterpret_cast<const float4*>(r), x1 = *reinterpret_cast<const float4*>(r + 4);
        q[0] += x0.x; q[1] += x0.y; q[2] += x0.z; q[3] += x0.w; q[4] += x1.x; q[5] += x1.y; q[6] += x1.z; q[7] += x1.w;
        if (SELF) {
            const float4 y0 = *reinterpret_cast<const float4*>(r + D), y1 = *reinterpret_cast<const float4*>(r + D + 4);
            const float4 z0 = *reinterpret_cast<const float4*>(r + 2 * D), z1 = *reinterpret_cast<const float4*>(r + 2 * D + 4);
            kn[0] += y0.x; kn[1] += y0.y; kn[2] += y0.z; kn[3] += y0.w; kn[4] += y1.x; kn[5] += y1.y; kn[6] += y1.z; kn[7] += y1.w;
            vn[0] += z0.x; vn[1] += z0.y; vn[2] += z0.z; vn[3] += z0.w; vn[4] += z1.x; vn[5] += z1.y; vn[6] += z1.z; vn[7] += z1.w;
        }
    }
    {
        const float4 a0 = *reinterpret_cast<const float4*>(p.bias + col), a1 = *reinterpret_cast<const float4*>(p.bias + col + 4);
        const float bq[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) q[e] = group_sum(q[e]) + bq[e];
        if (SELF) {
            const float4 c0 = *reinterpret_cast<const float4*>(p.bias + D + col), c1 = *reinterpret_cast<const float4*>(p.bias + D + col + 4);
            const float4 d0 = *reinterpret_cast<const float4*>(p.bias + 2 * D + col), d1 = *reinterpret_cast<const float4*>(p.bias + 2 * D + col + 4);
            const float bk[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
            const float bv[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) { kn[e] = group_sum(kn[e]) + bk[e]; vn[e] = group_sum(vn[e]) + bv[e]; }
            if (sizeof(T) == 2) {   // the cache holds bf16: attend to exactly the values it stores
#pragma unroll
                for (int e = 0; e < 8; ++e) { kn[e] = bf2f(f2bf(kn[e])); vn[e] = bf2f(f2bf(vn[e])); }
            }
            if (wave == 0 && g == 0) {
                elem<T>::st8(const_cast<T*>(kb) + (size_t)(L - 1) * p.kv_row_stride, kn);
                elem<T>::st8(const_cast<T*>(vb) + (size_t)(L - 1) * p.kv_row_stride, vn);
            }
        }
    }
    // ---- scores of this wave's keys
    float sc[NG];
    float mx = -INFINITY;
    if (Lc > 0) {
#pragma unroll
        for (int u = 0; u < NG; ++u) {
            float kv[8];
            kr[u].unpack(kv);
            float part = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) part += q[e] * kv[e];
            part += __shfl_xor(part, 1, 64);
            part += __shfl_xor(part, 2, 64);
            part += __shfl_xor(part, 4, 64);
            const bool valid = (k_begin + 8 * u + g) < k_end;
            sc[u] = valid ? part * p.scale : -INFINITY;
            mx = fmaxf(mx, sc[u]);
        }
    } else {
#pragma unroll
        for (int u = 0; u < NG; ++u) sc[u] = -INFINITY;
    }
    float s_new = -INFINITY;
    if (SELF && wave == 0) {
        float part = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) part += q[e] * kn[e];
        part += __shfl_xor(part, 1, 64);
        part += __shfl_xor(part, 2, 64);
        part += __shfl_xor(part, 4, 64);
        s_new = part * p.scale;
        mx = fmaxf(mx, s_new);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 8, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    // ---- partial softmax numerator / denominator of this wave
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float lsum = 0.f;
    if (mx > -INFINITY) {
        if (Lc > 0) {
#pragma unroll
            for (int u = 0; u < NG; ++u) {
                const float pk = expf(sc[u] - mx);        // exp(-inf) = 0 for clamped rows
                float vv[8];
                vr[u].unpack(vv);
                lsum += pk;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += pk * vv[e];
            }
        }
        lsum = group_sum(lsum);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = group_sum(acc[e]);
        if (SELF && wave == 0) {
            const float pn = expf(s_new - mx);
            lsum += pn;
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += pn * vn[e];
        }
    }
    if (g == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) s_acc[wave][c * 8 + e] = acc[e];
        if (c == 0) { s_m[wave] = mx; s_l[wave] = lsum; }
    }
    __syncthreads();
    if (wave == 0) {
        const float m = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float f = expf(s_m[w] - m);             // 0 for a wave without keys
            num += f * s_acc[w][lane];
            den += f * s_l[w];
        }
        elem<T>::st(reinterpret_cast<T*>(p.ctx) + (size_t)b * D + h * DH + lane, num / den);
    }
}
