// A/B kernels of rounds 1-2, kept OUT of the product library: compiled only with -DMOCR_EXPERIMENTS
// (python manga-ocr_amd/build.py --experiments), which also turns the MOCR_* environment knobs on (engine.hip: env_int).
//   gemm256_kernel   (tile code 256):  256 x 128, BK = 64, 3-stage ring, LDS epilogue                       (r01)
//   gemm_wide_kernel (512 / 1024):     256 x 128 (two blocks per CU) / 256 x 256, BK = 32, register epilogue (r01)
//   gemm_wide2_kernel (2048):          256 x 256, 4-stage ring, fragments across the barrier, LDS epilogue   (r02)
// The product's encoder GEMM is gemm_pers_kernel (kernels_gemm_pers.h), which kept wide2's K loop.
#pragma once
#include "kernels_gemm.h"

// ------------------------------------------------------------------------------------------------
// Big-tile bf16 GEMM for the encoder (M = rows*197 is large): 256 x 128 output tile, BK = 64.
//   * 4 waves as 2(M) x 2(N), ONE wave per SIMD, each owning 128 x 64 = 4 x 2 MFMA 32x32 tiles
//     (128 accumulator registers).  Per 16-deep k-step a wave reads 4 A + 2 B fragments for 8
//     MFMAs: 0.75 ds_read_b128 per MFMA instead of 1.0 for the 64x64 wave tile of gemm_kernel,
//     which keeps the LDS read port (256 B/clk/CU) well below saturation.
//   * 3-stage LDS ring (3 x 48 KiB = 144 KiB, one block per CU), filled by global_load_lds;
//     K-tile t+2 is issued while K-tile t is multiplied, so a DMA has two full K-tiles (~2000
//     cycles) to land.  Each wave waits for its own pieces with a COUNTED s_waitcnt vmcnt(12)
//     (12 = DMA instructions a wave issues per K-tile: the newest tile stays in flight) and the
//     block meets at ONE raw s_barrier per K-tile (a __syncthreads() would drain vmcnt to 0).
//   * same source-side XOR swizzle, XCD-aware block remap and LDS-staged fused epilogue as
//     gemm_kernel (cdna_hip_programming.md §5: "Pipelining across barriers", rule 21, T1).
// ------------------------------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(256, 1) void gemm256_kernel(GemmParams p) {
    using T = bf16_t;
    constexpr int BM = 256, BN = 128, NST = 3;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;   // 48 KiB
    constexpr int LPT = (BM / 8 + BN / 8) / 4;                                         // 12 DMA / wave / K-tile
    static_assert(BM * BN * 4 <= NST * STAGE, "fp32 epilogue tile must fit the ring");
    static_assert(LPT == 12, "the counted wait below is written for 12");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r32 = lane & 31, hh = lane >> 5;
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    int tn, tm;
    gemm_tile_of(p, bid, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const int nt = p.k_per_split / 64;
    const char* Ab = (const char*)p.A + (size_t)m0 * p.lda * 2;
    const char* Wb = (const char*)p.W + (size_t)n0 * p.ldw * 2;
    const size_t a_row = (size_t)p.lda * 2, w_row = (size_t)p.ldw * 2;
    const int prow = lane >> 3, pchunk = lane & 7;

    auto stage = [&](int t, int buf) {
        char* sa = smem + buf * STAGE;
        char* sb = sa + A_BYTES;
#pragma unroll
        for (int i = 0; i < BM / 32; ++i) {
            const int pc = wave + 4 * i;
            const int row = pc * 8 + prow;
            const int c = pchunk ^ ((row >> 1) & 7);
            glds16(Ab + (size_t)row * a_row + (size_t)t * 128 + c * 16, sa + pc * 1024);
        }
#pragma unroll
        for (int i = 0; i < BN / 32; ++i) {
            const int pc = wave + 4 * i;
            const int row = pc * 8 + prow;
            const int c = pchunk ^ ((row >> 1) & 7);
            glds16(Wb + (size_t)row * w_row + (size_t)t * 128 + c * 16, sb + pc * 1024);
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int offA[4], offB[2], swA[4], swB[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int row = wm * 128 + i * 32 + r32; offA[i] = row * 128; swA[i] = (row >> 1) & 7; }
#pragma unroll
    for (int j = 0; j < 2; ++j) { const int row = wn * 64 + j * 32 + r32; offB[j] = A_BYTES + row * 128; swB[j] = (row >> 1) & 7; }

    stage(0, 0);
    if (nt > 1) stage(1, 1);
    for (int t = 0; t < nt; ++t) {
        // tile t landed: only the newest tile (if any) may still be in flight
        if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");     // lgkmcnt: see gemm_kernel
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // every wave is past its reads of tile t-1, so its ring slot can be refilled with tile t+2
        if (t + 2 < nt && !(p.ablate & 2)) stage(t + 2, (t + 2) % NST);
        const char* sbuf = smem + (t % NST) * STAGE;
        // software-pipelined fragments: k-step s+1 is read from LDS while k-step s is multiplied
        // (two register sets), so the wait in front of each MFMA group is a counted lgkmcnt
        bf16x8 fa[2][4], fb[2][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[0][i] = *(const bf16x8*)(sbuf + offA[i] + ((hh ^ swA[i]) << 4));
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[0][j] = *(const bf16x8*)(sbuf + offB[j] + ((hh ^ swB[j]) << 4));
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s < 3) {
                const int c = 2 * (s + 1) + hh;
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[(s + 1) & 1][i] = *(const bf16x8*)(sbuf + offA[i] + ((c ^ swA[i]) << 4));
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[(s + 1) & 1][j] = *(const bf16x8*)(sbuf + offB[j] + ((c ^ swB[j]) << 4));
            }
            __builtin_amdgcn_sched_barrier(0);     // keep the prefetch ahead of this step's MFMAs
            if (!(p.ablate & 1)) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s & 1][i], fb[s & 1][j], acc[i][j], 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(fa[s & 1][i]));
#pragma unroll
                for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(fb[s & 1][j]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();
    float* sC = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                const int col = wn * 64 + j * 32 + r32;
                sC[row * BN + col] = acc[i][j][r];
            }
    __syncthreads();
    if (!(p.ablate & 4)) gemm_epilogue<T, BM, BN, EPI, 256>(sC, p, m0, n0, tid, 0);
}

// ------------------------------------------------------------------------------------------------
// "Wide" bf16 GEMM for the encoder: 256 x 128 output tile with 32-deep K-tiles, TWO blocks per CU.
//   The 128x128 kernel above is bound by the L2 -> LDS intake (~11.5 TB/s chip-wide with the MFMAs
//   running, r01): this tile moves 0.75x the bytes per FLOP, and with 64-byte K-tile rows a 3-stage
//   ring is only 72 KiB, so two blocks still share a CU - one block's epilogue (VALU, stores) runs
//   under the other block's MFMAs.
//   * 4 waves as 2(M) x 2(N); a wave owns 128 x 64 = 8 x 4 tiles of v_mfma_f32_16x16x32_bf16 (128
//     accumulator registers), one 32-deep k-step per K-tile: 12 ds_read_b128 for 32 MFMAs.
//   * operands are SWAPPED (W fragment as A, activation fragment as B), so a lane ends up with FOUR
//     CONSECUTIVE output columns of one output row: the fused epilogue stores 16 bytes per lane straight
//     from the registers (bf16 outputs exchange halves with v_permlane16_swap first) - no LDS round trip,
//     no barrier, and the stores are in flight while the next tile starts.
//   * LDS image: 64-B rows, 16-B chunk c of row r at chunk c ^ 2*((r>>3)&1): with that the four
//     16-lane groups of a ds_read_b128 (lanes {0-3,12-15,20-27}, ...) each hit 16 different 16-B slots.
//     One DMA piece = 1 KiB = 16 rows x 64 B; the swizzle is applied on the source address.
// ------------------------------------------------------------------------------------------------
// The fused register epilogue of the wide kernels: bias / GELU / fp32 residual, 16-byte stores straight from the
// accumulators (acc[n-tile j][m-tile i]: lane holds C[m = 16i + l15][n = 16j + 4*g4 + r] of its wave's 128 x 64 part).
template <int EPI>
__device__ __forceinline__ void gemm_wide_epilogue(const GemmParams& p, f32x4 (&acc)[4][8], float (&bias)[4][4], int em0, int en0, int nb,
                                                   int wm, int wn, int l15, int g4, bool guard) {
    using T = bf16_t;
    (void)sizeof(T);
    {
        {
            // residual rows: inline-asm loads, software-pipelined one row group ahead of the stores (no LDS-DMA is in
            // flight any more: only ordinary loads and stores, which retire in issue order - the same assumption the
            // compiler's own waits make).  A load issued AFTER the previous group's stores could only be waited for
            // together with those stores' acknowledgements; requested before them, its wait lets the stores fly.
            // Rows >= M (guard) read row M-1 instead and are not stored.
            f32x4 rv[2][4];
#define MOCR_LOAD_RESID(buf, i_)                                                                                   \
    {                                                                                                              \
        int m_ = em0 + wm * 128 + 16 * (i_) + l15;                                                                 \
        if (guard && m_ >= p.M) m_ = p.M - 1;                                                                      \
        const float* rrow_ = p.resid + (size_t)m_ * p.ldo + nb;                                                    \
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rv[buf][0]) : "v"(rrow_) : "memory");               \
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rv[buf][1]) : "v"(rrow_ + 16) : "memory");          \
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rv[buf][2]) : "v"(rrow_ + 32) : "memory");          \
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rv[buf][3]) : "v"(rrow_ + 48) : "memory");          \
    }
            if constexpr (EPI == EPI_BIAS_RESID) MOCR_LOAD_RESID(0, 0)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int m = em0 + wm * 128 + 16 * i + l15;
                const bool ok = !guard || m < p.M;
                if constexpr (EPI == EPI_BIAS_RESID) {
                    float* orow = reinterpret_cast<float*>(p.out) + (size_t)m * p.ldo + nb;
                    if (i + 1 < 8) MOCR_LOAD_RESID((i + 1) & 1, i + 1)
                    // younger than this group's loads: the previous group's 4 stores and the next group's 4 loads
#define MOCR_RV_TIE "+v"(rv[i & 1][0]), "+v"(rv[i & 1][1]), "+v"(rv[i & 1][2]), "+v"(rv[i & 1][3])
                    // (no run-time branch between an asm load and the wait its registers are tied to: a register copy at
                    // a branch merge would read the register before the load has landed)
                    if (guard) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (i == 0 || i == 7) asm volatile("s_waitcnt vmcnt(4)" : MOCR_RV_TIE);      // resolved by the unroll
                    else asm volatile("s_waitcnt vmcnt(8)" : MOCR_RV_TIE);
#undef MOCR_RV_TIE
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (ok)
                            *reinterpret_cast<float4*>(orow + 16 * j) =
                                make_float4(acc[j][i][0] + bias[j][0] + rv[i & 1][j][0], acc[j][i][1] + bias[j][1] + rv[i & 1][j][1],
                                            acc[j][i][2] + bias[j][2] + rv[i & 1][j][2], acc[j][i][3] + bias[j][3] + rv[i & 1][j][3]);
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);      // stores of group i stay in front of the loads of group i+2
                } else {
                    T* orow = reinterpret_cast<T*>(p.out) + (size_t)m * p.ldo + en0 + wn * 64;
#pragma unroll
                    for (int jp = 0; jp < 2; ++jp) {
                        // this lane's 4 columns of the n-tiles 2jp and 2jp+1, as packed bf16 pairs
                        unsigned lo[2], hi[2];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            float v[4];
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                v[r] = acc[2 * jp + h][i][r] + bias[2 * jp + h][r];
                                if constexpr (EPI == EPI_BIAS_GELU) v[r] = gelu_fast(v[r]);
                            }
                            const unsigned w0 = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
                            const unsigned w1 = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
                            if (h == 0) { lo[0] = w0; lo[1] = w1; } else { hi[0] = w0; hi[1] = w1; }
                        }
                        // v_permlane16_swap exchanges (odd 16-lane rows of the first operand) with (even rows of the
                        // second).  Afterwards a lane holds 8 consecutive columns = 16 bytes:
                        //   even g4: n = 32jp + 4*g4 .. +7        (own lo, then the odd neighbour's lo)
                        //   odd  g4: n = 32jp + 16 + 4*(g4-1) .. +7   (the even neighbour's hi, then own hi)
                        unsigned x0 = lo[0], x1 = lo[1], y0 = hi[0], y1 = hi[1];
                        {
                            auto s0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
                            auto s1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
                            x0 = s0[0]; y0 = s0[1]; x1 = s1[0]; y1 = s1[1];
                        }
                        const int ncol = 32 * jp + ((g4 & 1) ? 16 + 4 * (g4 - 1) : 4 * g4);
                        if (ok) *reinterpret_cast<uint4*>(orow + ncol) = make_uint4(x0, x1, y0, y1);
                    }
                }
            }
        }
    }
#undef MOCR_LOAD_RESID
}

// WN = waves along N: 2 -> 256 x 128 tile, 256 threads, two blocks per CU;
//                     4 -> 256 x 256 tile, 512 threads (8 waves = 2 per SIMD), one block per CU: 2/3 of the L2 -> LDS
//                          bytes per FLOP of the 256 x 128 tile (the L2 itself, ~16 TB/s, is what bounds these GEMMs)
// One tile per block; XCD x (= blockIdx & 7) owns a contiguous chunk of the tile list.  (A persistent variant that
// requested the next tile's first K-tiles before the epilogue's stores, so that the stores drained behind the next
// tile's MFMAs, gained 2 % on the 256 x 256 tile and lost 12 % on 256 x 128 - these GEMMs are bound by the L2's
// bandwidth, writes included, not by the store drain at block end - and its waits had to count stores as younger
// instructions than the in-flight LDS-DMA, which is not safe on this part (see kernels_qqt.h).  Removed.)
template <int EPI, int WN>
__global__ __launch_bounds__(128 * WN, WN == 2 ? 2 : 1) void gemm_wide_kernel(GemmParams p) {
    constexpr int BM = 256, BN = 64 * WN, NST = 3, NW = 2 * WN;
    constexpr int PA = 16 / NW, PW = (BN / 16) / NW, LPT = PA + PW;   // DMA pieces per wave per K-tile: A rows, W rows
    constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64, STAGE = A_BYTES + B_BYTES;
    static_assert(EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_RESID, "epilogues of the encoder layers");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l15 = lane & 15, g4 = lane >> 4;
    // this block's tiles: chunk of XCD (blockIdx & 7), positions (blockIdx >> 3) + k * (gridDim >> 3)
    const int ntiles = p.ntm * p.ntn;
    int tile, tile_end;
    {
        const int xcd = blockIdx.x & 7, q = ntiles >> 3, r = ntiles & 7;
        const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        tile = start + (blockIdx.x >> 3);
        tile_end = start + (xcd < r ? q + 1 : q);
    }
    if (tile >= tile_end) return;
    const int nt = p.k_per_split / 32;
    const size_t a_row = (size_t)p.lda * 2, w_row = (size_t)p.ldw * 2;
    const bool guard = (p.M & (BM - 1)) != 0;     // rows >= M exist in the last M-tile: predicated stores, drained epilogue

    // DMA: piece pc = wave + NW*i; A pieces cover 16 rows each, then the W pieces.  Lane -> row lane>>2 of the
    // piece, physical chunk lane&3, which holds logical chunk (lane&3) ^ 2*((row>>3)&1), row>>3 = lane>>5
    const int drow = lane >> 2, dchunk = (lane & 3) ^ (((lane >> 5) & 1) << 1);
    const size_t a_lane = (size_t)drow * a_row + dchunk * 16, w_lane = (size_t)drow * w_row + dchunk * 16;
    int m0, n0;
    const char* a_base;       // A + m0 rows, this lane's row/chunk
    const char* w_base;
    auto set_tile = [&](int tl) {
        int tm, tn;
        gemm_tile_of(p, tl, tm, tn);
        m0 = tm * BM; n0 = tn * BN;
        a_base = (const char*)p.A + (size_t)m0 * a_row + a_lane;
        w_base = (const char*)p.W + (size_t)n0 * w_row + w_lane;
    };
    auto stage = [&](const char* ab, const char* wb, int t, int buf) {
        char* sa = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < PA; ++i) glds16(ab + (size_t)((wave + NW * i) * 16) * a_row + (size_t)t * 64, sa + (wave + NW * i) * 1024);
#pragma unroll
        for (int i = 0; i < PW; ++i) glds16(wb + (size_t)((wave + NW * i) * 16) * w_row + (size_t)t * 64, sa + A_BYTES + (wave + NW * i) * 1024);
    };

    // fragment read offset of this lane inside a 16-row group: row l15, logical chunk g4
    const int frag_off = l15 * 64 + ((g4 ^ (((l15 >> 3) & 1) << 1)) << 4);
    const int offA = (wm * 128) * 64 + frag_off, offB = A_BYTES + (wn * 64) * 64 + frag_off;

    set_tile(tile);
    stage(a_base, w_base, 0, 0);
    if (nt > 1) stage(a_base, w_base, 1, 1);

    {
        f32x4 acc[4][8];      // [n-tile j][m-tile i]: lane holds C[m = 16i + l15][n = 16j + 4*g4 + r]
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[j][i][r] = 0.f;

        for (int t = 0; t < nt; ++t) {
            const int slot = t % NST;
            // K-tile t has landed (this wave's pieces); the next K-tile's DMA (if any) stays in flight
            if (t + 1 < nt) wait_vmcnt<LPT>(); else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();                                            // ... everyone's; and slot (t+2)%3 is free
            asm volatile("" ::: "memory");
            if (t + 2 < nt && !(p.ablate & 2)) {
                stage(a_base, w_base, t + 2, (t + 2) % NST);
            }
            const char* sbuf = smem + slot * STAGE;
            // All twelve fragment reads are issued at once and the MFMAs wait only for what they consume (LDS
            // returns in order: counted lgkmcnt).  Left to itself hipcc reads two fragments, drains lgkmcnt,
            // runs 8 MFMAs, and exposes the LDS latency four times per K-tile.  The waits are tied to the
            // registers they cover ("+v"), so no MFMA can be scheduled above its wait.
            bf16x8 fa[8], fb[4];
            {
                const unsigned aA = lds_addr_of(sbuf + offA), aB = lds_addr_of(sbuf + offB);
                asm volatile(
                    "ds_read_b128 %0, %13\n\tds_read_b128 %1, %13 offset:1024\n\tds_read_b128 %2, %13 offset:2048\n\t"
                    "ds_read_b128 %3, %13 offset:3072\n\t"
                    "ds_read_b128 %4, %12\n\tds_read_b128 %5, %12 offset:1024\n\tds_read_b128 %6, %12 offset:2048\n\t"
                    "ds_read_b128 %7, %12 offset:3072\n\tds_read_b128 %8, %12 offset:4096\n\tds_read_b128 %9, %12 offset:5120\n\t"
                    "ds_read_b128 %10, %12 offset:6144\n\tds_read_b128 %11, %12 offset:7168"
                    : "=&v"(fb[0]), "=&v"(fb[1]), "=&v"(fb[2]), "=&v"(fb[3]), "=&v"(fa[0]), "=&v"(fa[1]), "=&v"(fa[2]), "=&v"(fa[3]),
                      "=&v"(fa[4]), "=&v"(fa[5]), "=&v"(fa[6]), "=&v"(fa[7])
                    : "v"(aA), "v"(aB)
                    : "memory");
            }
            if (p.ablate & 1) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); continue; }   // diagnostics: no MFMAs
            asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]), "+v"(fb[3]), "+v"(fa[0]), "+v"(fa[1]));
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (g == 1) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(fa[2]), "+v"(fa[3]));
                if (g == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fa[4]), "+v"(fa[5]));
                if (g == 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[6]), "+v"(fa[7]));
#pragma unroll
                for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[j][2 * g + ii] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[2 * g + ii], acc[j][2 * g + ii], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);      // the next wait stays behind this group's MFMAs
            }
        }

        // bias of this lane's 16 columns
        const int em0 = m0, en0 = n0;
        const int nb = en0 + wn * 64 + 4 * g4;
        float bias[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 bv = *reinterpret_cast<const float4*>(p.bias + nb + 16 * j);
            bias[j][0] = bv.x; bias[j][1] = bv.y; bias[j][2] = bv.z; bias[j][3] = bv.w;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(bias[j][r]));      // landed before the asm loads below are counted
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);

        if (!(p.ablate & 4)) gemm_wide_epilogue<EPI>(p, acc, bias, em0, en0, nb, wm, wn, l15, g4, guard);
    }
}


// ------------------------------------------------------------------------------------------------
// gemm_wide2_kernel: the 256 x 256 tile of gemm_wide_kernel<EPI, 4> with a deeper pipeline (r02).
//   What the ablations of the r01 kernel showed at M = 50,432 (QKV, us): full 217 = loop alone 152 + DMA +20..30 +
//   epilogue +45; the DMA stream alone moves 11.6 TB/s with at most two 32 KiB K-tiles in flight per CU, and every
//   K-tile restarts behind its barrier with twelve fragment reads whose latency nothing covers (both waves of a SIMD
//   stand at the same barrier).  Here:
//   * FOUR ring stages (128 KiB): K-tile t+3 is requested while K-tile t is multiplied - up to three tiles (96 KiB)
//     in flight per CU;
//   * the K-tile's barrier sits in the MIDDLE of its 32 MFMAs: after groups 0-1 a wave has all of tile t's fragments
//     in registers (lgkmcnt(0)), waits for its own DMA pieces of tile t+1 (counted vmcnt), meets the block, and
//     requests the first six fragments of tile t+1 (W x 4, A x 2) into the OTHER register set before it issues
//     groups 2-3 - so the reads' latency runs under 16 MFMAs, and the next K-tile starts with its operands in hand.
//     Two register sets, the loop unrolled by two (static names: no copies, no runtime-indexed arrays).
//   * slot reuse: the DMA of tile t+3 goes to the slot of tile t-1, and is requested after barrier(t-1), which every
//     wave passed with lgkmcnt(0) - no read of that slot can still be in flight (cf. the race of r01, gemm_kernel).
// ------------------------------------------------------------------------------------------------
// Epilogue of gemm_wide2_kernel through LDS (the ring is idle once the K loop has ended: 128 KiB = exactly one bf16
// C tile, or half an fp32 one).  The register epilogue above stores 16 bytes per lane at a ROW stride - one store
// instruction touches 16 rows x 64 B - and that path moves only ~7-14 B/clk/CU (r02: QKV at M = 50,432 spends 68 of its
// 220 us there, FC1 146 of 365, O-proj 78 of 152).  Here the tile is transposed through LDS so that every global
// access is a whole 512-byte row segment (a wave instruction = 2 rows x 512 B = 8 full cache lines), the residual
// rows are read the same way, and all 16 residual loads of a pass are in flight before the first one is needed.
//   LDS image: [256 rows][512 B], 16-byte chunk c of row r at chunk c ^ (r & 15).
template <int EPI>
__device__ __forceinline__ void gemm_wide2_epilogue_lds(const GemmParams& p, f32x4 (&acc)[4][8], float (&bias)[4][4], char* smem, int m0, int n0,
                                                        int wave, int lane, bool guard) {
    const int wm = wave >> 2, wn = wave & 3, l15 = lane & 15, g4 = lane >> 4;
    const int rsel = lane >> 5, lc = lane & 31;           // read-back: 2 rows per wave instruction, 32 chunks per row
    __syncthreads();                                       // every wave is out of the K loop; no DMA is in flight
    if constexpr (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = wm * 128 + 16 * i + l15;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = acc[j][i][r] + bias[j][r];
                    if constexpr (EPI == EPI_BIAS_GELU) v[r] = gelu_fast(v[r]);
                }
                uint2 u;
                u.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
                u.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
                const int chunk = wn * 8 + 2 * j + (g4 >> 1);
                *reinterpret_cast<uint2*>(smem + row * 512 + ((chunk ^ (row & 15)) << 4) + (g4 & 1) * 8) = u;
            }
        }
        __syncthreads();
        bf16_t* const obase = reinterpret_cast<bf16_t*>(p.out) + (size_t)m0 * p.ldo + n0 + 8 * lc;
#pragma unroll 4
        for (int it = 0; it < 16; ++it) {
            const int row = 32 * wave + 2 * it + rsel;
            const uint4 v = *reinterpret_cast<const uint4*>(smem + row * 512 + ((lc ^ (row & 15)) << 4));
            // non-temporal: 232 / 310 MB of QKV / FC1 output per layer at batch 256 pass through once; written with the
            // default policy they evict the weight slices and activations the other CUs are re-reading (r02: QKV 219 -> 201 us,
            // FC1 370 -> 353 us, and FC2 - which reads that output - 335 -> 327 us)
            if (!guard || m0 + row < p.M) st16_nt(obase + (size_t)row * p.ldo, v);
        }
    } else {
        static_assert(EPI == EPI_BIAS_RESID, "fp32 residual epilogue");
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            // this pass's residual rows: 16 loads of 16 B per lane, eight in flight at a time (requested before the LDS
            // round trip; row it + 8 is requested as soon as row it has been consumed)
            uint4 rv[8];
            const float* const rbase = p.resid + (size_t)m0 * p.ldo + n0 + 128 * h + 4 * lc;
            auto resid_row = [&](int it) {
                int row = 32 * wave + 2 * it + rsel;
                if (guard && m0 + row >= p.M) row = p.M - 1 - m0;
                return *reinterpret_cast<const uint4*>(rbase + (size_t)row * p.ldo);
            };
#pragma unroll
            for (int it = 0; it < 8; ++it) rv[it] = resid_row(it);
            if (h == 1) __syncthreads();                   // pass 0's rows have been read back by everyone
            if ((wn >> 1) == h) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int row = wm * 128 + 16 * i + l15;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int chunk = (wn & 1) * 16 + 4 * j + g4;
                        *reinterpret_cast<float4*>(smem + row * 512 + ((chunk ^ (row & 15)) << 4)) =
                            make_float4(acc[j][i][0] + bias[j][0], acc[j][i][1] + bias[j][1], acc[j][i][2] + bias[j][2], acc[j][i][3] + bias[j][3]);
                    }
                }
            }
            __syncthreads();
            float* const obase = reinterpret_cast<float*>(p.out) + (size_t)m0 * p.ldo + n0 + 128 * h + 4 * lc;
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int row = 32 * wave + 2 * it + rsel;
                const float4 v = *reinterpret_cast<const float4*>(smem + row * 512 + ((lc ^ (row & 15)) << 4));
                const uint4 ru = rv[it & 7];
                if (it + 8 < 16) rv[it & 7] = resid_row(it + 8);
                const float4 r = make_float4(__uint_as_float(ru.x), __uint_as_float(ru.y), __uint_as_float(ru.z), __uint_as_float(ru.w));
                if (!guard || m0 + row < p.M)
                    *reinterpret_cast<float4*>(obase + (size_t)row * p.ldo) = make_float4(v.x + r.x, v.y + r.y, v.z + r.z, v.w + r.w);
            }
        }
    }
}

template <int EPI>
__global__ __launch_bounds__(512, 1) void gemm_wide2_kernel(GemmParams p) {
    constexpr int BM = 256, BN = 256, NW = 8, WN = 4;      // four ring stages of 32 KiB
    constexpr int PA = 16 / NW, PW = (BN / 16) / NW, LPT = PA + PW;   // 2 + 2 DMA pieces per wave per K-tile
    constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64, STAGE = A_BYTES + B_BYTES;   // 32 KiB
    static_assert(EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_RESID, "epilogues of the encoder layers");
    static_assert(LPT == 4, "the counted waits below are written for 4 pieces per wave per K-tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l15 = lane & 15, g4 = lane >> 4;
    const int ntiles = p.ntm * p.ntn;
    int tile, tile_end;
    {
        const int xcd = blockIdx.x & 7, q = ntiles >> 3, r = ntiles & 7;
        const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        tile = start + (blockIdx.x >> 3);
        tile_end = start + (xcd < r ? q + 1 : q);
    }
    if (tile >= tile_end) return;
    // Experiment knob (MOCR_GEMM_STAGGER, default 0): start the first-round blocks at spread phases of one tile time
    // so that the CUs do not all store their output tiles at the same moment.  Measured r02 (M = 50,432): no gain at
    // any phase spread - an output tile's store drain costs the same whether or not the other CUs are storing.
    if (p.stagger > 0 && (int)blockIdx.x < p.first_round) {
        const int units = (int)((((blockIdx.x >> 3) * 37u + (blockIdx.x & 7u) * 5u) & 31u) * (unsigned)p.stagger) >> 5;
        for (int i = 0; i < units; ++i) __builtin_amdgcn_s_sleep(16);
    }
    const int nt = p.k_per_split / 32;            // even, >= 4 (checked on the host)
    const size_t a_row = (size_t)p.lda * 2, w_row = (size_t)p.ldw * 2;
    const bool guard = (p.M & (BM - 1)) != 0;

    const int drow = lane >> 2, dchunk = (lane & 3) ^ (((lane >> 5) & 1) << 1);
    const size_t a_lane = (size_t)drow * a_row + dchunk * 16, w_lane = (size_t)drow * w_row + dchunk * 16;
    int tm, tn;
    gemm_tile_of(p, tile, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const char* const a_base = (const char*)p.A + (size_t)m0 * a_row + a_lane;
    const char* const w_base = (const char*)p.W + (size_t)n0 * w_row + w_lane;
    auto stage = [&](int t, int buf) {
        char* sa = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < PA; ++i) glds16(a_base + (size_t)((wave + NW * i) * 16) * a_row + (size_t)t * 64, sa + (wave + NW * i) * 1024);
#pragma unroll
        for (int i = 0; i < PW; ++i) glds16(w_base + (size_t)((wave + NW * i) * 16) * w_row + (size_t)t * 64, sa + A_BYTES + (wave + NW * i) * 1024);
    };
    const int frag_off = l15 * 64 + ((g4 ^ (((l15 >> 3) & 1) << 1)) << 4);
    const unsigned offA = lds_addr_of(smem) + (wm * 128) * 64 + frag_off, offB = lds_addr_of(smem) + A_BYTES + (wn * 64) * 64 + frag_off;

    f32x4 acc[4][8];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[j][i][r] = 0.f;

    stage(0, 0);
    stage(1, 1);
    stage(2, 2);
    wait_vmcnt<8>();                        // tile 0 (this wave's pieces); tiles 1 and 2 stay in flight
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    WideFrags P, Q;
    MOCR_W2_READ_HEAD(P, offA, offB);

    // one K-tile: CUR holds its first six fragments (requested during the previous K-tile), NXT receives those of the next
#define MOCR_W2_KTILE(CUR, NXT, t_)                                                                                    \
    {                                                                                                                  \
        const int kt = (t_);                                                                                            \
        const unsigned so = (unsigned)((kt & 3) * STAGE), sn = (unsigned)(((kt + 1) & 3) * STAGE);                       \
        MOCR_W2_READ_TAIL(CUR, offA + so);                                                                             \
        if (kt + 3 < nt && !(p.ablate & 2)) stage(kt + 3, (kt + 3) & 3);                                                  \
        if (!(p.ablate & 1)) {                                                                                         \
            asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(CUR.fb[0]), "+v"(CUR.fb[1]), "+v"(CUR.fb[2]), "+v"(CUR.fb[3]),  \
                         "+v"(CUR.fa[0]), "+v"(CUR.fa[1]));                                                            \
            MOCR_W2_GROUP(CUR, 0);                                                                                     \
            asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(CUR.fa[2]), "+v"(CUR.fa[3]));                                   \
            MOCR_W2_GROUP(CUR, 1);                                                                                     \
        }                                                                                                              \
        /* every fragment of this K-tile is in registers: its slot may be refilled behind the next barrier */         \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(CUR.fa[4]), "+v"(CUR.fa[5]), "+v"(CUR.fa[6]), "+v"(CUR.fa[7]));     \
        {                                                                                                              \
            /* K-tile kt+1 landed (own pieces); the tiles requested after it (kt+2, kt+3) stay in flight.  No run-time   \
               branch may sit between an asm load and the wait its registers are tied to (a register copy at the      \
               merge would read them too early), so the last K-tile also meets the barrier and reads six fragments    \
               of a slot nobody uses; they are waited for behind the loop. */                                          \
            const int newer = nt - 2 - kt;                                                                              \
            if (newer >= 2) wait_vmcnt<8>(); else if (newer == 1) wait_vmcnt<4>(); else wait_vmcnt<0>();               \
            __builtin_amdgcn_s_barrier();                                                                              \
            asm volatile("" ::: "memory");                                                                             \
            MOCR_W2_READ_HEAD(NXT, offA + sn, offB + sn);                                                              \
        }                                                                                                              \
        if (!(p.ablate & 1)) {                                                                                         \
            MOCR_W2_GROUP(CUR, 2);                                                                                     \
            MOCR_W2_GROUP(CUR, 3);                                                                                     \
        }                                                                                                              \
    }
    for (int t = 0; t < nt; t += 2) {
        MOCR_W2_KTILE(P, Q, t)
        MOCR_W2_KTILE(Q, P, t + 1)
    }
#undef MOCR_W2_KTILE
    // the reads requested behind the last barrier: landed before their registers are used for anything else
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(P.fb[0]), "+v"(P.fb[1]), "+v"(P.fb[2]), "+v"(P.fb[3]), "+v"(P.fa[0]), "+v"(P.fa[1]));

    // bias of this lane's 16 columns
    const int em0 = m0, en0 = n0;
    const int nb = en0 + wn * 64 + 4 * g4;
    float bias[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float4 bv = *reinterpret_cast<const float4*>(p.bias + nb + 16 * j);
        bias[j][0] = bv.x; bias[j][1] = bv.y; bias[j][2] = bv.z; bias[j][3] = bv.w;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(bias[j][r]));
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (p.ablate & 4) return;
    if (p.ablate & 16) gemm_wide_epilogue<EPI>(p, acc, bias, em0, en0, nb, wm, wn, l15, g4, guard);     // r01's register epilogue (A/B)
    else gemm_wide2_epilogue_lds<EPI>(p, acc, bias, smem, (p.ablate & 64) ? 0 : (p.ablate & 32) ? (em0 & 2047) : em0, en0, wave, lane, guard);
}
