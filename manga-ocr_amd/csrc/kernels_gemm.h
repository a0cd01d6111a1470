// MFMA GEMM for the recogniser's dense layers:  C[M,N] = A[M,K] . W[N,K]^T  (+ epilogue)
//
// Both operands are K-contiguous (activations row-major, weights in torch.nn.Linear's
// [out,in] layout), so A and W tiles are staged the same way.  One kernel template serves
//   T = bf16_t : v_mfma_f32_32x32x16_bf16, fp32 accumulate             (throughput mode)
//   T = float  : v_mfma_f32_32x32x2_f32, an exact k-ordered fmaf chain  (parity mode)
// because a K-tile is defined in BYTES: 128 B per row (64 bf16 / 32 f32).
//
// Structure (cdna_hip_programming.md §5, "2-buffer glds" form):
//   * 256 threads = 4 waves as 2x2; each wave owns a (BM/2)x(BN/2) sub-tile = TMxTN MFMA tiles.
//   * global -> LDS by global_load_lds_dwordx4 (no VGPR round trip).  One wave-instruction
//     writes 1 KiB = 8 rows x 128 B, lane-linear; the bank-conflict swizzle is therefore
//     applied on the SOURCE address and on the fragment read (rule 21):
//         16-byte chunk c of row r lives at chunk  c ^ ((r >> 1) & 7).
//     With 128-B rows two rows share one 256-B bank row; (r>>1)&7 makes every ds_read_b128
//     16-lane group hit 16 distinct 16-B slots (conflict-free).
//   * double-buffered: tile t+1 is in flight while tile t is multiplied; one barrier per K-tile.
//   * epilogue through LDS (fp32 tile) so that global stores / residual loads are full
//     16-byte, row-contiguous accesses; bias / GELU / residual / position-embedding are fused.
//   * blockIdx -> tile map is XCD-aware (bijective remap, §5 T1): the 8 XCDs each get a
//     contiguous run of tiles, so tiles sharing an A row-panel share an L2.
#pragma once
#include "common.h"

#ifndef MOCR_GEMM_MFMA16
#define MOCR_GEMM_MFMA16 1      // 1: v_mfma_f32_16x16x32_bf16 in gemm_kernel's bf16 path (+2..6 % at M = 806,912, r01); 0: 32x32x16
#endif

enum GemmEpilogue {
    EPI_SLAB = 0,        // fp32 partial sums of K-slice blockIdx.z -> out[z][m][n]   (split-K, no bias)
    EPI_BIAS = 1,        // out T   = acc + bias
    EPI_BIAS_GELU = 2,   // out T   = gelu(acc + bias)
    EPI_BIAS_RESID = 3,  // out f32 = (acc + bias) + resid        (out may alias resid)
    EPI_PATCH = 4,       // patch-embed: out f32[b*197+1+p] = (acc + bias) + pos[1+p]
    EPI_BIAS_F32 = 5,    // out f32 = acc + bias
    EPI_ARGMAX = 6       // LM head: per (row, N-tile) the fp32 max of acc + bias and its column (lowest column wins ties):
                         //   out f32 [M][ntn] values, cand_idx int [M][ntn] columns - the logits never go to memory
};

struct GemmParams {
    const void* A;
    const void* W;
    const float* bias;
    void* out;
    const float* resid;
    const float* pos;      // EPI_PATCH: [tokens][N] position embeddings
    int* cand_idx;         // EPI_ARGMAX: [M][ntn] winning columns
    int M, N;              // logical output size (rows >= M are computed but not stored)
    int lda, ldw, ldo;     // leading dimensions in elements
    int k_per_split;       // elements of K handled by one blockIdx.z
    long long slab_stride; // EPI_SLAB: elements between slabs
    int ntn;               // tiles along N
    int patches;           // EPI_PATCH: patches per image (196)
    long long a_yoff, w_yoff, o_yoff, b_yoff;  // element offsets per blockIdx.y (per-head batched GEMMs; EPI_BIAS only)
    int ablate;            // diagnostics only (MOCR_GEMM_ABLATE): 1 no MFMA, 2 no DMA after tile 0, 4 no epilogue
    int group_n;           // tile order: N-tiles per column group (0 = all: plain row-major tile order)
    int ntm;               // tiles along M
    int stagger;           // wide2: the first-round blocks start up to this many x 1024 cycles late (0 = together)
    int first_round;       // blocks that start at launch (one per CU)
    // LayerNorm folded into the encoder's persistent GEMMs (gemm_pers_kernel LNF, kernels_gemm_pers.h):
    float* ln_part;        // [M][4][2] per-row partial (sum, sum of squares) over each 256-column slice of the fp32 rows:
                           // written by the fp32-residual GEMM (slice = its N-tile), read by the GEMM that multiplies the rows
    const float* csum;     // [N] column sums of the folded weight W o gamma (as rounded to bf16)
    void* xb;              // fp32-residual GEMM: bf16 copy of the output rows, [M][ldo]
    float ln_eps;
};

// Linear tile id -> (tm, tn).  Tiles are ordered column-group by column-group: inside a group of
// `group_n` N-tiles the order is M-major / N-minor, so the blocks resident on one XCD at a time cover
// a few M-tiles x group_n N-tiles and the group's weight slice (group_n*BN rows of W) stays in that
// XCD's 4 MiB L2 while the A row-panels stream through it once per group.
__device__ __forceinline__ void gemm_tile_of(const GemmParams& p, int bid, int& tm, int& tn) {
    const int gn = p.group_n > 0 && p.group_n < p.ntn ? p.group_n : p.ntn;
    const int per_group = p.ntm * gn;
    int g = bid / per_group;
    const int ngroups = (p.ntn + gn - 1) / gn;
    if (g >= ngroups) g = ngroups - 1;
    const int r = bid - g * per_group;
    const int gw = min(gn, p.ntn - g * gn);
    tm = r / gw;
    tn = g * gn + r - tm * gw;
}

// Read the fp32 tile back from LDS row by row and apply the fused epilogue; every global access is
// a 16-byte, row-contiguous access.  NT = threads in the block.
// SWZ: the tile was written with its 16-column groups XOR-swizzled by (row>>2)&3 (the 16x16 MFMA's four
// lane groups sit 4 rows apart = the same bank; the swizzle spreads them over the banks).
template <typename T, int BM, int BN, int EPI, int NT, bool SWZ = false>
__device__ __forceinline__ void gemm_epilogue(const float* sC, const GemmParams& p, int m0, int n0, int tid, int z) {
    constexpr int TPR = BN / 4;      // threads per output row (4 columns each)
    constexpr int RPI = NT / TPR;    // rows per pass
    const int col = (tid % TPR) * 4;
    const int n = n0 + col;
    float bias4[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (EPI != EPI_SLAB) {
        const float4 bv = *reinterpret_cast<const float4*>(p.bias + n);
        bias4[0] = bv.x; bias4[1] = bv.y; bias4[2] = bv.z; bias4[3] = bv.w;
    }
#pragma unroll 4
    for (int it = 0; it < BM / RPI; ++it) {
        const int row = it * RPI + tid / TPR;
        const int m = m0 + row;
        if (m >= p.M) continue;
        const float4 cv = *reinterpret_cast<const float4*>(&sC[row * BN + (SWZ ? (col ^ (((row >> 2) & 3) << 4)) : col)]);
        float v[4] = {cv.x + bias4[0], cv.y + bias4[1], cv.z + bias4[2], cv.w + bias4[3]};
        if constexpr (EPI == EPI_SLAB) {
            float* o = reinterpret_cast<float*>(p.out) + (size_t)z * p.slab_stride + (size_t)m * p.ldo + n;
            *reinterpret_cast<float4*>(o) = cv;
        } else if constexpr (EPI == EPI_BIAS) {
            if constexpr (sizeof(T) == 2) {
                if (p.ablate & 8) {     // experiment: streaming (non-temporal) output stores
                    unsigned long long u = (unsigned long long)f2bf(v[0]) | ((unsigned long long)f2bf(v[1]) << 16) |
                                           ((unsigned long long)f2bf(v[2]) << 32) | ((unsigned long long)f2bf(v[3]) << 48);
                    __builtin_nontemporal_store(u, reinterpret_cast<unsigned long long*>(reinterpret_cast<T*>(p.out) + (size_t)m * p.ldo + n));
                    continue;
                }
            }
            elem<T>::st4(reinterpret_cast<T*>(p.out) + (size_t)m * p.ldo + n, v);
        } else if constexpr (EPI == EPI_BIAS_GELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_for<T>(v[e]);
            elem<T>::st4(reinterpret_cast<T*>(p.out) + (size_t)m * p.ldo + n, v);
        } else if constexpr (EPI == EPI_BIAS_RESID) {
            const float4 rv = *reinterpret_cast<const float4*>(p.resid + (size_t)m * p.ldo + n);
            float* o = reinterpret_cast<float*>(p.out) + (size_t)m * p.ldo + n;
            *reinterpret_cast<float4*>(o) = make_float4(v[0] + rv.x, v[1] + rv.y, v[2] + rv.z, v[3] + rv.w);
        } else if constexpr (EPI == EPI_PATCH) {
            const int b = m / p.patches, pi = m - b * p.patches;
            const float4 pv = *reinterpret_cast<const float4*>(p.pos + (size_t)(1 + pi) * p.N + n);
            float* o = reinterpret_cast<float*>(p.out) + ((size_t)b * (p.patches + 1) + 1 + pi) * p.ldo + n;
            *reinterpret_cast<float4*>(o) = make_float4(v[0] + pv.x, v[1] + pv.y, v[2] + pv.z, v[3] + pv.w);
        } else {  // EPI_BIAS_F32
            float* o = reinterpret_cast<float*>(p.out) + (size_t)m * p.ldo + n;
            *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

// EPI_ARGMAX: every output row of the tile is reduced to (max of acc + bias, its column; the lowest column wins ties).
// NT / BM threads share a row (adjacent lanes); a thread scans its BN / (NT/BM) physical columns of the LDS tile 4 at a
// time, starting at a row-dependent rotation so that the 16 lanes of a ds_read_b128 group hit 16 different 16-byte
// slots.  SWZ: the tile's 16-column groups are XORed with (row>>2)&3 (see gemm_epilogue): physical -> logical column.
template <int BM, int BN, int NT, bool SWZ>
__device__ __forceinline__ void gemm_epilogue_argmax(const float* sC, const GemmParams& p, int m0, int n0, int tid) {
    constexpr int TPRW = NT / BM, CPP = BN / TPRW;
    static_assert(TPRW == 2 || TPRW == 4, "two or four threads per row");
    const int row = tid / TPRW, part = tid % TPRW;
    const int m = m0 + row;
    const int sw = SWZ ? (((row >> 2) & 3) << 4) : 0;
    const int rot = 4 * (row % (CPP / 4));
    float best = -INFINITY;
    int bi = 0x7fffffff;
#pragma unroll
    for (int q = 0; q < CPP / 4; ++q) {
        const int pc = part * CPP + ((rot + 4 * q) % CPP);          // physical column of this float4
        const float4 cv = *reinterpret_cast<const float4*>(&sC[row * BN + pc]);
        const int n = n0 + (pc ^ sw);                                // logical (global) column of cv.x
        const float4 bv = *reinterpret_cast<const float4*>(p.bias + n);
        const float v[4] = {cv.x + bv.x, cv.y + bv.y, cv.z + bv.z, cv.w + bv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (v[e] > best || (v[e] == best && n + e < bi)) { best = v[e]; bi = n + e; }
    }
#pragma unroll
    for (int o = TPRW / 2; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (part == 0 && m < p.M) {
        const size_t c = (size_t)m * p.ntn + n0 / BN;
        reinterpret_cast<float*>(p.out)[c] = best;
        p.cand_idx[c] = bi;
    }
}

__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) void*)p;
}

// s_waitcnt vmcnt(N) needs an immediate: N = DMA instructions that may stay in flight
template <int N> __device__ __forceinline__ void wait_vmcnt() {
    static_assert(N == 0 || N == 4 || N == 6 || N == 8 || N == 12 || N == 16 || N == 20 || N == 22 || N == 24 || N == 63, "add the count");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if constexpr (N == 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    else if constexpr (N == 22) asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
    else if constexpr (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
}

// NST = depth of the LDS ring: K-tile t+NST-1 is issued while K-tile t is multiplied.  NST = 2 is
// the plain double buffer; the latency-bound skinny GEMMs of the decode step use NST = 4 (64x64
// tile: 4 x 16 KiB) so a DMA has three K-tiles of time to land.
template <typename T, int BM, int BN, int EPI, int NST = 2>
__global__ __launch_bounds__(256) void gemm_kernel(GemmParams p) {
    constexpr int TM = BM / 64, TN = BN / 64;          // 32x32 MFMA tiles per wave
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    constexpr int LPT = (BM / 8 + BN / 8) / 4;         // DMA instructions per wave per K-tile
    static_assert(BM * BN * 4 <= NST * STAGE, "fp32 epilogue tile must fit the staging buffers");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r32 = lane & 31, hh = lane >> 5;

    // XCD-aware bijective remap of the linear block id (8 XCDs, round-robin dispatch)
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    int tn, tm;
    gemm_tile_of(p, bid, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = blockIdx.z * p.k_per_split;
    const int nt = p.k_per_split / (128 / (int)sizeof(T));

    const char* Ab = (const char*)p.A + ((size_t)m0 * p.lda + kbeg + (size_t)blockIdx.y * p.a_yoff) * sizeof(T);
    const char* Wb = (const char*)p.W + ((size_t)n0 * p.ldw + kbeg + (size_t)blockIdx.y * p.w_yoff) * sizeof(T);
    const size_t a_row = (size_t)p.lda * sizeof(T), w_row = (size_t)p.ldw * sizeof(T);

    // per-lane part of the staging addresses: lane -> (row within 8-row piece, physical chunk)
    const int prow = lane >> 3, pchunk = lane & 7;

    auto stage = [&](int t, int buf) {
        char* sa = smem + buf * STAGE;
        char* sb = sa + A_BYTES;
#pragma unroll
        for (int pc = wave; pc < BM / 8; pc += 4) {
            const int row = pc * 8 + prow;
            const int c = pchunk ^ ((row >> 1) & 7);
            glds16(Ab + (size_t)row * a_row + (size_t)t * 128 + c * 16, sa + pc * 1024);
        }
#pragma unroll
        for (int pc = wave; pc < BN / 8; pc += 4) {
            const int row = pc * 8 + prow;
            const int c = pchunk ^ ((row >> 1) & 7);
            glds16(Wb + (size_t)row * w_row + (size_t)t * 128 + c * 16, sb + pc * 1024);
        }
    };

    f32x16 acc[TM][TN];
    f32x4 acc16[2 * TM][2 * TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
    for (int i = 0; i < 2 * TM; ++i)
#pragma unroll
        for (int j = 0; j < 2 * TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc16[i][j][r] = 0.f;

    int rowA[TM], rowB[TN], swA[TM], swB[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) { rowA[i] = wm * (BM / 2) + i * 32 + r32; swA[i] = (rowA[i] >> 1) & 7; }
#pragma unroll
    for (int j = 0; j < TN; ++j) { rowB[j] = wn * (BN / 2) + j * 32 + r32; swB[j] = (rowB[j] >> 1) & 7; }

#pragma unroll
    for (int i = 0; i < NST - 1; ++i)
        if (i < nt) stage(i, i);
    for (int t = 0; t < nt; ++t) {
        // tile t has landed: each wave waits for its own DMA, allowing the min(NST-2, nt-1-t) newer
        // tiles to stay in flight; the raw barrier (no vmcnt drain) covers the other waves' pieces
        // and tells that everyone has finished reading the slot tile t+NST-1 is about to overwrite
        const int newer = nt - 1 - t;
        if constexpr (NST == 2) {
            wait_vmcnt<0>();
        } else if constexpr (NST == 3) {
            if (newer >= 1) wait_vmcnt<LPT>(); else wait_vmcnt<0>();
        } else {
            static_assert(NST == 4, "ring depth");
            if (newer >= 2) wait_vmcnt<2 * LPT>(); else if (newer == 1) wait_vmcnt<LPT>(); else wait_vmcnt<0>();
        }
        // Also every LDS read this wave has issued must be BACK before the barrier: hipcc sinks the last MFMAs of the
        // previous K-tile (and the lgkmcnt wait for their fragments) below it, and behind the barrier the slot those
        // reads come from is handed to the DMA.  Under LDS contention (several blocks per CU) such a read has been seen
        // to lose that race (kernels_qqt.h, r01).
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + NST - 1 < nt) stage(t + NST - 1, (t + NST - 1) % NST);
        const char* sa = smem + (t % NST) * STAGE;
        const char* sb = sa + A_BYTES;
        if constexpr (sizeof(T) == 2 && MOCR_GEMM_MFMA16) {
            // 16x16x32 shape: the same LDS bytes per FLOP as 32x32x16, but the chip holds a higher clock
            // on it (MI355X_MICROARCH.md, DVFS give-back item 7)
            const int l15 = lane & 15, g4 = lane >> 4;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int c = 4 * s + g4;
                bf16x8 a[2 * TM], b[2 * TN];
#pragma unroll
                for (int i = 0; i < 2 * TM; ++i) {
                    const int row = wm * (BM / 2) + i * 16 + l15;
                    a[i] = *(const bf16x8*)(sa + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
                }
#pragma unroll
                for (int j = 0; j < 2 * TN; ++j) {
                    const int row = wn * (BN / 2) + j * 16 + l15;
                    b[j] = *(const bf16x8*)(sb + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
                }
#pragma unroll
                for (int i = 0; i < 2 * TM; ++i)
#pragma unroll
                    for (int j = 0; j < 2 * TN; ++j)
                        acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc16[i][j], 0, 0, 0);
            }
        } else if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int c = 2 * s + hh;
                bf16x8 a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = *(const bf16x8*)(sa + rowA[i] * 128 + ((c ^ swA[i]) << 4));
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = *(const bf16x8*)(sb + rowB[j] * 128 + ((c ^ swB[j]) << 4));
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int k = 2 * s + hh;            // A[i=lane&31][k=lane>>5] of a 32x32x2 step
                const int c = k >> 2, o = (k & 3) * 4;
                float a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = *(const float*)(sa + rowA[i] * 128 + ((c ^ swA[i]) << 4) + o);
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = *(const float*)(sb + rowB[j] * 128 + ((c ^ swB[j]) << 4) + o);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: accumulators -> LDS (fp32 [BM][BN]) -> fused epilogue -> 16-byte global stores
    __syncthreads();
    float* sC = reinterpret_cast<float*>(smem);
    constexpr bool SWZ = sizeof(T) == 2 && MOCR_GEMM_MFMA16;
    if constexpr (SWZ) {
#pragma unroll
        for (int i = 0; i < 2 * TM; ++i)
#pragma unroll
            for (int j = 0; j < 2 * TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // C/D map of the 16x16 MFMA: col = lane&15, row = 4*(lane>>4) + r
                    const int row = wm * (BM / 2) + i * 16 + 4 * (lane >> 4) + r;
                    const int col = wn * (BN / 2) + j * 16 + (lane & 15);
                    sC[row * BN + (col ^ (((row >> 2) & 3) << 4))] = acc16[i][j][r];
                }
    } else {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
                const int row = wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                const int col = wn * (BN / 2) + j * 32 + r32;
                sC[row * BN + col] = acc[i][j][r];
            }
    }
    __syncthreads();

    if constexpr (EPI == EPI_ARGMAX) {
        gemm_epilogue_argmax<BM, BN, 256, SWZ>(sC, p, m0, n0, tid);
    } else if constexpr (EPI == EPI_BIAS) {
        GemmParams q = p;
        q.out = reinterpret_cast<T*>(p.out) + (size_t)blockIdx.y * p.o_yoff;
        q.bias = p.bias + (size_t)blockIdx.y * p.b_yoff;
        gemm_epilogue<T, BM, BN, EPI, 256, SWZ>(sC, q, m0, n0, tid, blockIdx.z);
    } else {
        gemm_epilogue<T, BM, BN, EPI, 256, SWZ>(sC, p, m0, n0, tid, blockIdx.z);
    }
}

// ------------------------------------------------------------------------------------------------
// Fragment registers and LDS reads of the 256 x 256 encoder tile (gemm_pers_kernel, kernels_gemm_pers.h; also the r02
// kernel it replaced, kernels_gemm_lab.h): 8 waves as 2(M) x 4(N), a wave owns 128 x 64 = 8 x 4 tiles of
// v_mfma_f32_16x16x32_bf16 with SWAPPED operands (W fragment as A), one 32-deep k-step per K-tile: 12 ds_read_b128 for
// 32 MFMAs.  LDS image of a K-tile: 64-B rows, 16-B chunk c of row r at chunk c ^ 2*((r>>3)&1) (conflict-free for the
// four 16-lane groups of a ds_read_b128); one DMA piece = 1 KiB = 16 rows x 64 B, swizzle applied on the source address.
// ------------------------------------------------------------------------------------------------
struct WideFrags { bf16x8 fb[4]; bf16x8 fa[8]; };

// first six fragment reads of a K-tile (the four W fragments, then A fragments 0 and 1)
#define MOCR_W2_READ_HEAD(F, aA, aB)                                                                                    \
    asm volatile("ds_read_b128 %0, %7\n\tds_read_b128 %1, %7 offset:1024\n\tds_read_b128 %2, %7 offset:2048\n\t"        \
                 "ds_read_b128 %3, %7 offset:3072\n\tds_read_b128 %4, %6\n\tds_read_b128 %5, %6 offset:1024"             \
                 : "=&v"(F.fb[0]), "=&v"(F.fb[1]), "=&v"(F.fb[2]), "=&v"(F.fb[3]), "=&v"(F.fa[0]), "=&v"(F.fa[1])       \
                 : "v"(aA), "v"(aB)                                                                                      \
                 : "memory")
// the other six A fragments
#define MOCR_W2_READ_TAIL(F, aA)                                                                                        \
    asm volatile("ds_read_b128 %0, %6 offset:2048\n\tds_read_b128 %1, %6 offset:3072\n\tds_read_b128 %2, %6 offset:4096\n\t" \
                 "ds_read_b128 %3, %6 offset:5120\n\tds_read_b128 %4, %6 offset:6144\n\tds_read_b128 %5, %6 offset:7168" \
                 : "=&v"(F.fa[2]), "=&v"(F.fa[3]), "=&v"(F.fa[4]), "=&v"(F.fa[5]), "=&v"(F.fa[6]), "=&v"(F.fa[7])       \
                 : "v"(aA)                                                                                               \
                 : "memory")
// the same reads from the 64-deep image of gemm_pers_kernel's K64 form: 128-byte rows, fragments 2 KiB apart
#define MOCR_W2K_READ_HEAD(F, aA, aB)                                                                                   \
    asm volatile("ds_read_b128 %0, %7\n\tds_read_b128 %1, %7 offset:2048\n\tds_read_b128 %2, %7 offset:4096\n\t"        \
                 "ds_read_b128 %3, %7 offset:6144\n\tds_read_b128 %4, %6\n\tds_read_b128 %5, %6 offset:2048"             \
                 : "=&v"(F.fb[0]), "=&v"(F.fb[1]), "=&v"(F.fb[2]), "=&v"(F.fb[3]), "=&v"(F.fa[0]), "=&v"(F.fa[1])       \
                 : "v"(aA), "v"(aB)                                                                                      \
                 : "memory")
#define MOCR_W2K_READ_TAIL(F, aA)                                                                                       \
    asm volatile("ds_read_b128 %0, %6 offset:4096\n\tds_read_b128 %1, %6 offset:6144\n\tds_read_b128 %2, %6 offset:8192\n\t" \
                 "ds_read_b128 %3, %6 offset:10240\n\tds_read_b128 %4, %6 offset:12288\n\tds_read_b128 %5, %6 offset:14336" \
                 : "=&v"(F.fa[2]), "=&v"(F.fa[3]), "=&v"(F.fa[4]), "=&v"(F.fa[5]), "=&v"(F.fa[6]), "=&v"(F.fa[7])       \
                 : "v"(aA)                                                                                               \
                 : "memory")
// a half tile's wave (64 x 64, gemm_pers_kernel STRIP): A fragments 2 and 3 only
#define MOCR_W2_READ_TAIL2(F, aA)                                                                                       \
    asm volatile("ds_read_b128 %0, %2 offset:2048\n\tds_read_b128 %1, %2 offset:3072"                                   \
                 : "=&v"(F.fa[2]), "=&v"(F.fa[3])                                                                        \
                 : "v"(aA)                                                                                               \
                 : "memory")
#define MOCR_W2_GROUP(F, g)                                                                                             \
    _Pragma("unroll") for (int ii = 0; ii < 2; ++ii) _Pragma("unroll") for (int j = 0; j < 4; ++j)                        \
        acc[j][2 * (g) + ii] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F.fb[j], F.fa[2 * (g) + ii], acc[j][2 * (g) + ii], 0, 0, 0); \
    __builtin_amdgcn_sched_barrier(0)

