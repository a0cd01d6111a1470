// Fused query path of the latent decode attention (bf16, fat batches):
//
//     q_h  = x . Wq_h^T + bq_h                 [rows, 64]   per head h      (TF/models/bert/modeling_bert.py:151-166 / 218-230)
//     Qt_h = bf16(q_h) . (Wk_h^T / 8)          [rows, 768]                  (the absorbed query, see kernels_latent.h)
//
// As two launches (gemm_dec_q: 16 us, gemm_dec_qt: 31 us at 4096 rows, four times per decode step) the second
// GEMM has a single 64-deep K-tile per block: all prologue and epilogue, and its 75 MB of output are written at
// 2.5 TB/s.  Here one block owns (128 rows, one head): phase 1 is an ordinary K = 768 GEMM for the 128 x 64
// query tile, which goes to LDS as bf16 (the same rounding the two-launch path applies when it stores q);
// phase 2 multiplies it by the six 128-row slices of Wk_h^T/8, double-buffered across the slices, and stores Qt
// straight from the registers.  MFMA shape, K order and roundings are those of gemm_kernel's
// bf16 path, so the result is bit-identical to the two-launch path.
//
// Both products use SWAPPED operands (weight fragment as A): a lane then holds four consecutive output columns
// of one row - an 8-byte LDS store in phase 1, and after v_permlane16_swap a 16-byte global store in phase 2
// (see gemm_wide_kernel).  LDS images are those of gemm_kernel: 128-byte rows, 16-byte chunk c of row r at
// chunk c ^ ((r >> 1) & 7), swizzle applied on the DMA source address.
#pragma once
#include "common.h"

struct QqtParams {
    const bf16_t* x;      // [rows_pad][768] layer input (bf16)
    const bf16_t* wq;     // [768][768] query weight, row n = output feature (head h: rows 64h .. 64h+63)
    const float* bq;      // [768]
    const bf16_t* wkT;    // [768 n][768]: column block 64h .. 64h+63 of row n = (Wk_h^T / 8)[:, n]
    bf16_t* qt;           // [rows_pad][16][768]
};

// Ring depth of phase 1 (r02).  The first form double-buffered both phases behind full vmcnt(0) drains: every one of the
// 12 K-tiles and 6 Wk^T slices then costs a whole memory round trip (one tile in flight: 24 us per launch at 2560 rows
// whatever the rows, i.e. pure latency).  Now phase 1 keeps three K-tiles in flight (counted vmcnt, loads only), and
// phase 2 requests ALL six slices at once into the ring phase 1 has left (6 x 16 KiB = the 4 x 24 KiB ring), waits once
// and never again - its stores are fire-and-forget: 25.0 -> 20.6 us at 2560 rows (what is left is the 47 MB of Qt the
// launch writes).  -DQQT_NST=2 builds the first form for A/B runs.
// r04: BM = rows per block.  The launch is a serial chain per block (12 K-tiles, then six slices: ~15 us at 512 rows as at 2560
// rows, four times per decode step).  With 64-row blocks a K-tile is 16 KiB, the ring holds eight of them (seven in flight
// instead of three) and twice as many blocks cover the chip when the batch is not fat: the launch for batches of up to 1280 rows.
#ifndef QQT_NST
#define QQT_NST 4
#endif
template <int BM> struct QqtCfg {
    static constexpr int NST = BM == 128 ? QQT_NST : 8;                                   // ring slots of phase 1
    static constexpr int STAGE = (BM + 64) * 128;
    static constexpr int LDS = NST * STAGE + BM * 128;        // phase-1 ring (phase 2: the six Wk^T slices in it, 96 KiB) + the q tile
    static_assert(QQT_NST < 4 || NST * STAGE >= 6 * 128 * 128, "phase 2 keeps all six slices in the ring");
};
#define QQT_LDS (QqtCfg<128>::LDS)

template <int BM>
__global__ __launch_bounds__(256, 1) void dec_qqt_kernel(QqtParams p) {
    static_assert(BM == 128 || BM == 64, "rows per block");
    constexpr int D = 768, KT = D / 64;                 // 12 K-tiles of 64 in phase 1
    constexpr int NSTQ = QqtCfg<BM>::NST, RT = BM / 32; // RT: 16-row tiles per wave (a wave owns BM / 2 rows)
    constexpr int A_BYTES = BM * 128, STAGE = QqtCfg<BM>::STAGE;
    constexpr int XP = BM / 32;                         // x pieces per wave and K-tile (BM / 8 pieces of 8 rows over four waves)
    constexpr int PER_TILE = XP + 2;                    // DMA instructions per wave and K-tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const sQ = smem + NSTQ * STAGE;               // [BM rows][64 dims] bf16, swizzled 128-B rows
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, g4 = lane >> 4;
    const int m0 = blockIdx.x * BM, h = blockIdx.y;

    // DMA pieces: 1 KiB = 8 rows x 128 B; lane -> row lane>>3, physical chunk lane&7 = logical chunk ^ ((row>>1)&7)
    const int prow = lane >> 3, pchunk = lane & 7;
    const char* const xb = reinterpret_cast<const char*>(p.x + (size_t)m0 * D);
    const char* const wqb = reinterpret_cast<const char*>(p.wq + (size_t)(h * 64) * D);
    auto stage1 = [&](int t, int buf) {
        char* sa = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < XP; ++i) {                  // BM / 8 pieces of x
            const int pc = wave + 4 * i, row = pc * 8 + prow, c = pchunk ^ ((row >> 1) & 7);
            glds16(xb + (size_t)row * (D * 2) + t * 128 + c * 16, sa + pc * 1024);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {                   // 8 pieces of Wq_h
            const int pc = wave + 4 * i, row = pc * 8 + prow, c = pchunk ^ ((row >> 1) & 7);
            glds16(wqb + (size_t)row * (D * 2) + t * 128 + c * 16, sa + A_BYTES + pc * 1024);
        }
    };

    // ---------------- phase 1: q tile.  Wave (wm, wn): rows (BM / 2) wm .. , query dims 32 wn .. +31
    f32x4 qa[2][RT];     // [dim tile j][row tile i]: lane holds q[m = 16i + l15][d = 16j + 4 g4 + r]
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) qa[j][i][r] = 0.f;
#pragma unroll
    for (int i = 0; i < NSTQ - 1; ++i) stage1(i, i);
    for (int t = 0; t < KT; ++t) {
        // K-tile t has landed: this wave issued 6 DMA instructions per tile, and only loads are in its queue here, so the
        // count of the younger tiles' instructions may stay in flight.  lgkmcnt(0) too: hipcc sinks the last MFMAs of
        // the previous K-tile (and the wait for their fragments) below this barrier, and behind it the buffer those
        // fragment reads come from is handed to the DMA
        {
            const int newer = KT - 1 - t < NSTQ - 2 ? KT - 1 - t : NSTQ - 2;      // younger K-tiles that may stay in flight
            if constexpr (BM == 128) {       // 6 instructions per tile
                if (newer >= 2) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
                else if (newer == 1) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            } else {                         // 4 instructions per tile, up to six younger tiles
                static_assert(PER_TILE == 4 || BM == 128, "the counts below");
                switch (newer) {
                    case 0: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;
                    case 1: asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory"); break;
                    case 2: asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory"); break;
                    case 3: asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory"); break;
                    case 4: asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory"); break;
                    case 5: asm volatile("s_waitcnt vmcnt(20) lgkmcnt(0)" ::: "memory"); break;
                    default: asm volatile("s_waitcnt vmcnt(24) lgkmcnt(0)" ::: "memory"); break;
                }
            }
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + NSTQ - 1 < KT) stage1(t + NSTQ - 1, (t + NSTQ - 1) % NSTQ);
        const char* sa = smem + (t % NSTQ) * STAGE;
        const char* sb = sa + A_BYTES;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int c = 4 * s + g4;
            bf16x8 fx[RT], fw[2];
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                const int row = wm * (BM / 2) + i * 16 + l15;
                fx[i] = *(const bf16x8*)(sa + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = wn * 32 + j * 16 + l15;
                fw[j] = *(const bf16x8*)(sb + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    qa[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[j], fx[i], qa[j][i], 0, 0, 0);
        }
    }
    // Wk^T slices of phase 2: LDS-DMA, double-buffered.  The wait for a slice is a FULL drain (vmcnt(0)), previous
    // slice's global stores included: a counted s_waitcnt vmcnt(N) that lets YOUNGER STORES fly does not prove that
    // an OLDER LDS-DMA has landed on this part (measured: with the weights cold in L2, vmcnt(8) in front of the 8
    // younger stores let the MFMAs read a slice that was still arriving, about once in 40 launches; staging the
    // slices through registers instead - ordinary loads do retire in order with stores - was correct but no faster
    // than the two-launch path).
    const char* const wkb = reinterpret_cast<const char*>(p.wkT + h * 64);
    auto stage2 = [&](int nt2, int buf) {
        char* sw = smem + buf * (128 * 128);
#pragma unroll
        for (int i = 0; i < 4; ++i) {                   // 16 pieces: rows 128 nt2 .. +127 of Wk^T, this head's 128-byte column block
            const int pc = wave + 4 * i, row = pc * 8 + prow, c = pchunk ^ ((row >> 1) & 7);
            glds16(wkb + (size_t)(nt2 * 128 + row) * (D * 2) + c * 16, sw + pc * 1024);
        }
    };
    // q + bias -> bf16 -> sQ (8 bytes per lane and tile)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int d = wn * 32 + 16 * j + 4 * g4;
        const float4 bv = *reinterpret_cast<const float4*>(p.bq + h * 64 + d);
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const int row = wm * (BM / 2) + 16 * i + l15;
            const unsigned w0 = (unsigned)f2bf(qa[j][i][0] + bv.x) | ((unsigned)f2bf(qa[j][i][1] + bv.y) << 16);
            const unsigned w1 = (unsigned)f2bf(qa[j][i][2] + bv.z) | ((unsigned)f2bf(qa[j][i][3] + bv.w) << 16);
            *reinterpret_cast<uint2*>(sQ + row * 128 + (((d >> 3) ^ ((row >> 1) & 7)) << 4) + (d & 7) * 2) = make_uint2(w0, w1);
        }
    }
    __syncthreads();                                    // every wave is past its last phase-1 fragment reads: the ring is free
#if QQT_NST >= 4
#pragma unroll
    for (int k = 0; k < 6; ++k) stage2(k, k);           // all six slices at once: 96 KiB = the whole ring
#else
    stage2(0, 0);
#endif

    // ---------------- phase 2: Qt tile = q tile . (Wk_h^T/8) slices.  Wave (wm, wn): rows (BM / 2) wm .. , columns 64 wn .. +63 of a slice
    bf16x8 fq[2][RT];    // q fragments of this wave's rows, both 32-deep k-steps: the same for all six slices
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const int row = wm * (BM / 2) + i * 16 + l15, c = 4 * s + g4;
            fq[s][i] = *(const bf16x8*)(sQ + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
        }
    bf16_t* const orow0 = p.qt + ((size_t)(m0 + wm * (BM / 2) + l15) * 16 + h) * D + wn * 64;
#if QQT_NST >= 4
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");          // the six slices landed (nothing else is in flight)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#endif
    for (int nt2 = 0; nt2 < 6; ++nt2) {
#if QQT_NST >= 4
        const char* sw = smem + nt2 * (128 * 128);
#else
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // slice nt2 landed (and the previous slice's stores acknowledged)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (nt2 + 1 < 6) stage2(nt2 + 1, (nt2 + 1) & 1);      // its buffer was last read one slice ago
        const char* sw = smem + (nt2 & 1) * (128 * 128);
#endif
        f32x4 acc[4][RT];    // [column tile j][row tile i]
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[j][i][r] = 0.f;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 fw[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = wn * 64 + j * 16 + l15, c = 4 * s + g4;
                fw[j] = *(const bf16x8*)(sw + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[j], fq[s][i], acc[j][i], 0, 0, 0);
        }
        // stores: 2 RT per lane (RT row tiles x 2 column-tile pairs), 16 bytes each
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            bf16_t* orow = orow0 + (size_t)(16 * i) * 16 * D + nt2 * 128;
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                unsigned x0 = (unsigned)f2bf(acc[2 * jp][i][0]) | ((unsigned)f2bf(acc[2 * jp][i][1]) << 16);
                unsigned x1 = (unsigned)f2bf(acc[2 * jp][i][2]) | ((unsigned)f2bf(acc[2 * jp][i][3]) << 16);
                unsigned y0 = (unsigned)f2bf(acc[2 * jp + 1][i][0]) | ((unsigned)f2bf(acc[2 * jp + 1][i][1]) << 16);
                unsigned y1 = (unsigned)f2bf(acc[2 * jp + 1][i][2]) | ((unsigned)f2bf(acc[2 * jp + 1][i][3]) << 16);
                {   // even 16-lane rows keep column tile 2jp and take the odd neighbour's half of it; odd rows keep 2jp+1
                    auto s0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
                    auto s1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
                    x0 = s0[0]; y0 = s0[1]; x1 = s1[0]; y1 = s1[1];
                }
                const int ncol = 32 * jp + ((g4 & 1) ? 16 + 4 * (g4 - 1) : 4 * g4);
                *reinterpret_cast<uint4*>(orow + ncol) = make_uint4(x0, x1, y0, y1);
            }
        }
    }
}
