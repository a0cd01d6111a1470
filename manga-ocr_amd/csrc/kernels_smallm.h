// Decode-step projections of SMALL batches (<= 32 rows, bf16): one launch per projection, no split-K slabs, and the
// LayerNorm that follows a projection folded into the PROLOGUE of its consumers (r02).
//
// At <= 64 rows every launch of a decode step costs 3.7-5.2 us whatever it does, and the generic
// path spends 9 of its 28 launches on slab-sum + bias (+ residual) + LayerNorm / GELU kernels that exist only because
// the GEMMs are split over K to fill the chip.  With M <= 32 a block can instead own ALL rows of a 16-column slice of
// the output over the WHOLE K (no slabs), so
//   * bias, residual and GELU are the GEMM's epilogue,
//   * the post-LN of BERT (TF/models/bert/modeling_bert.py:472-489: LayerNorm(dense(x) + residual)) is not a launch of
//     its own: a projection writes the PRE-LayerNorm sum (fp32), and every projection that consumes the normalised rows
//     normalises them itself while staging its A operand (each block re-does the <= 32 x 768 statistics: 48-96 KB of
//     L2 reads - less than the launch it replaces); block 0 also publishes (mean, rstd) per row, from which a later
//     epilogue rebuilds the normalised row as its residual.
// 19 launches per step instead of 28.  A block = 8 waves; wave w takes the K-steps w, w+8, ... (16x16x32 MFMA), the eight
// partial tiles are summed through LDS in wave order (deterministic), operands come straight from L2 (a lane's 16 bytes
// of a weight row / an activation row per K-step).  A launch this small is one memory round trip long or it is not worth
// having: EVERY global load of a block - weight fragments, rows, gamma / beta, and the epilogue's bias / residual /
// statistics - is requested before the first barrier.
// Arithmetic: the same fp32 accumulation of bf16 products as gemm_kernel, in a different order (no slabs), the same
// LayerNorm expression as dec_add_ln_kernel, the same bf16 rounding points (normalised rows, GELU output).
#pragma once
#include "common.h"
#include "kernels_latent.h"      // row16_sum

enum { SM_PRO_PLAIN = 0, SM_PRO_LN = 1 };
enum { SM_EPI_RAW = 0, SM_EPI_SUM = 1, SM_EPI_GELU_BF16 = 2, SM_EPI_GELU_F32 = 3 };

#define SM_NT 16                                 // output columns per block
#define SM_NW 8                                  // waves per block = K split
#define SM_MAX_ROWS 32                           // two 16-row tiles
#define SM_A_STRIDE (768 * 2 + 16)               // LDS row stride of the normalised A rows (bank spread, as LAT_OUT_HS)
#define SM_LDS(pro, mt) (((pro) == SM_PRO_LN ? (mt) * 16 * SM_A_STRIDE : 0) + SM_NW * (mt) * 1024)

struct SmallMParams {
    const bf16_t* a_bf16;      // PRO_PLAIN: [rows][K]
    const float* a_f32;        // PRO_LN: [rows][768] pre-LayerNorm sums
    const float* ln_g;         // PRO_LN: gamma / beta [768]
    const float* ln_b;
    float* stats_out;          // PRO_LN: [rows][2] (mean, rstd), written by block 0; may be null
    const bf16_t* w;           // [N][K], row n = output feature
    const float* bias;         // [N] (EPI_SUM, EPI_GELU_*)
    const float* resid;        // EPI_SUM: [rows][N] fp32 - the residual itself, or pre-LayerNorm sums when resid_stats != null
    const float* resid_stats;  //          [rows][2] of those sums
    const float* resid_g;      //          their gamma / beta [N]
    const float* resid_b;
    void* out;                 // [rows][ldo]: fp32 (RAW / SUM / GELU_F32) or bf16 (GELU_BF16)
    int ldo;
    int rows;                  // rows to compute, <= 16 MT
    int K, N;                  // K = 768 (PRO_LN: always) or a multiple of 256 up to 3072
    float eps;
};

// MT = 16-row tiles (1 or 2); KS = K-steps per wave = K / 32 / 8 (3 for K = 768, 12 for K = 3072)
template <int PRO, int EPI, int MT, int KS>
__global__ __launch_bounds__(512) void smallm_gemm_kernel(SmallMParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * SM_NT;
    const int K = p.K, last = p.rows - 1;
    float* const red = reinterpret_cast<float*>(smem + (PRO == SM_PRO_LN ? MT * 16 * SM_A_STRIDE : 0));   // [8][MT][16][16]

    // ---- every global load of the block, up front
    // (1) weight fragments of this wave's K-steps: lane (n = l15, k = 8 g ..) reads 16 bytes of weight row n0 + l15
    uint4 wf[KS];
    {
        const bf16_t* const wrow = p.w + (size_t)(n0 + l15) * K + 8 * g;
#pragma unroll
        for (int j = 0; j < KS; ++j) wf[j] = *reinterpret_cast<const uint4*>(wrow + 32 * (wave + SM_NW * j));
    }
    // (2) the epilogue's operands of this thread's output element(s): element e = tid + 512 k -> (row e >> 4, column e & 15)
    constexpr int NE = (MT * 256 + 511) / 512;
    float e_bias[NE], e_res[NE], e_mean[NE], e_rstd[NE], e_g[NE], e_b[NE];
#pragma unroll
    for (int k = 0; k < NE; ++k) {
        const int e = tid + 512 * k, m = e >> 4, col = n0 + (e & 15);
        const int mc = m < last ? m : last;
        e_bias[k] = e_res[k] = e_mean[k] = e_g[k] = e_b[k] = 0.f; e_rstd[k] = 1.f;
        if constexpr (EPI != SM_EPI_RAW) e_bias[k] = p.bias[col];
        if constexpr (EPI == SM_EPI_SUM) {
            e_res[k] = p.resid[(size_t)mc * p.N + col];
            if (p.resid_stats) {
                e_mean[k] = p.resid_stats[2 * mc]; e_rstd[k] = p.resid_stats[2 * mc + 1];
                e_g[k] = p.resid_g[col]; e_b[k] = p.resid_b[col];
            }
        }
    }
    f32x4 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;

    if constexpr (PRO == SM_PRO_LN) {
        // ---- prologue: rows -> LayerNorm -> bf16 in LDS.  A 16-lane group owns a row (768 = 16 lanes x 12 float4), so the
        // statistics are DPP row reductions; the 8 waves x 4 groups cover 32 rows in one pass
        const int row = 4 * wave + g;
        if (row < 16 * MT) {
            const float* xr = p.a_f32 + (size_t)(row < last ? row : last) * 768 + 4 * l15;
            float4 v[12], gm[12], bt[12];
#pragma unroll
            for (int j = 0; j < 12; ++j) v[j] = *reinterpret_cast<const float4*>(xr + 64 * j);
#pragma unroll
            for (int j = 0; j < 12; ++j) {
                gm[j] = *reinterpret_cast<const float4*>(p.ln_g + 4 * l15 + 64 * j);
                bt[j] = *reinterpret_cast<const float4*>(p.ln_b + 4 * l15 + 64 * j);
            }
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 12; ++j) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
            const float mean = row16_sum(s) * (1.0f / 768);
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < 12; ++j) {
                const float d0 = v[j].x - mean, d1 = v[j].y - mean, d2 = v[j].z - mean, d3 = v[j].w - mean;
                q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            }
            const float rstd = 1.0f / sqrtf(row16_sum(q) * (1.0f / 768) + p.eps);
            if (p.stats_out && blockIdx.x == 0 && l15 == 0 && row < p.rows) {
                p.stats_out[2 * row] = mean;
                p.stats_out[2 * row + 1] = rstd;
            }
            char* const dst = smem + row * SM_A_STRIDE + 8 * l15;
#pragma unroll
            for (int j = 0; j < 12; ++j) {
                uint2 u;
                u.x = (unsigned)f2bf((v[j].x - mean) * rstd * gm[j].x + bt[j].x) | ((unsigned)f2bf((v[j].y - mean) * rstd * gm[j].y + bt[j].y) << 16);
                u.y = (unsigned)f2bf((v[j].z - mean) * rstd * gm[j].z + bt[j].z) | ((unsigned)f2bf((v[j].w - mean) * rstd * gm[j].w + bt[j].w) << 16);
                *reinterpret_cast<uint2*>(dst + 128 * j) = u;
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            const int s = wave + SM_NW * j;
            union { uint4 u; bf16x8 v; } bw;
            bw.u = wf[j];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                union { uint4 u; bf16x8 v; } aw;
                aw.u = *reinterpret_cast<const uint4*>(smem + (16 * i + l15) * SM_A_STRIDE + (32 * s + 8 * g) * 2);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aw.v, bw.v, acc[i], 0, 0, 0);
            }
        }
    } else {
        // ---- PLAIN: activation fragments straight from memory as well (rows beyond the batch re-read its last row:
        // finite, their results are not stored)
        uint4 af[MT][KS];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int row = 16 * i + l15;
            const bf16_t* const arow = p.a_bf16 + (size_t)(row < last ? row : last) * K + 8 * g;
#pragma unroll
            for (int j = 0; j < KS; ++j) af[i][j] = *reinterpret_cast<const uint4*>(arow + 32 * (wave + SM_NW * j));
        }
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            union { uint4 u; bf16x8 v; } bw;
            bw.u = wf[j];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                union { uint4 u; bf16x8 v; } aw;
                aw.u = af[i][j];
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aw.v, bw.v, acc[i], 0, 0, 0);
            }
        }
    }
    // ---- the eight waves' partial tiles -> LDS; C/D map of the 16x16 MFMA: row = 4 (lane >> 4) + reg, col = lane & 15
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[((wave * MT + i) * 16 + 4 * g + r) * 16 + l15] = acc[i][r];
    __syncthreads();
    // ---- sum in wave order, epilogue, store: 16 consecutive threads cover one row's 16 columns
#pragma unroll
    for (int k = 0; k < NE; ++k) {
        const int e = tid + 512 * k, m = e >> 4, col = n0 + (e & 15);
        if (e >= MT * 256 || m >= p.rows) continue;
        float v = 0.f;
#pragma unroll
        for (int w8 = 0; w8 < SM_NW; ++w8) v += red[w8 * MT * 256 + e];
        if constexpr (EPI == SM_EPI_RAW) {
            reinterpret_cast<float*>(p.out)[(size_t)m * p.ldo + col] = v;
        } else if constexpr (EPI == SM_EPI_SUM) {
            v += e_bias[k];
            float r = e_res[k];
            if (p.resid_stats) r = (r - e_mean[k]) * e_rstd[k] * e_g[k] + e_b[k];
            reinterpret_cast<float*>(p.out)[(size_t)m * p.ldo + col] = v + r;
        } else if constexpr (EPI == SM_EPI_GELU_BF16) {
            reinterpret_cast<bf16_t*>(p.out)[(size_t)m * p.ldo + col] = f2bf(gelu_fast(v + e_bias[k]));
        } else {
            reinterpret_cast<float*>(p.out)[(size_t)m * p.ldo + col] = gelu_fast(v + e_bias[k]);
        }
    }
}
