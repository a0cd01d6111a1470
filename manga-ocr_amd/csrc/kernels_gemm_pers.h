// gemm_pers_kernel: the encoder's layer GEMMs as ONE persistent block per CU (r03).
//
// What r02's measurements said about gemm_wide2_kernel at M = 50,432 (one 256 x 256 tile per block, one block per CU):
//   * the K loop alone runs at 1.3 PF, the whole kernel at 0.43 (O-proj) .. 0.95 PF (QKV): every tile pays a block
//     turnover (launch, three K-tiles of cold DMA latency) and an epilogue nothing overlaps;
//   * tools/probe/store_probe.hip (r03): a CU writes a 128 KiB output tile in 1.1 us when it stores alone (50 B/clk) but in
//     4.3 us when all 256 CUs store at the same moment (7.8 TB/s chip-wide) - and the one-tile-per-block grid makes them
//     all reach their epilogue together, round after round; stores issued by four of the eight waves drain behind the
//     other waves' MFMAs at no cost (768 MFMAs per wave: 25.4k -> 25.9k cycles).
// Here a block walks its whole list of tiles:
//   * the LDS ring never drains: the K-tile requested at iteration g is K-tile g + 3 of the block's whole sequence, so
//     the first three K-tiles of tile i + 1 arrive while tile i is multiplied and stored - no cold start per tile, and
//     the blocks drift apart instead of meeting at every epilogue;
//   * the global stores of the epilogue are issued by waves 4-7 only (store_probe: four waves' stores drain behind the
//     MFMAs of all eight), waves 0-3 never have a store in their queue.  Every wave requests its share of the LDS-DMA;
//     the K loop's counted waits count LOADS only (loads return in issue order; stores are not ordered with older
//     LDS-DMA on this part), so a store wave's first waits of a tile can at worst hold until its stores
//     have drained - they cannot end early;
//   * the epilogue goes through the 64 KiB of LDS that are free at that moment - the slot of the K-tile just consumed
//     plus the 32 KiB the ring leaves over - in passes of 128 bf16 rows (2 passes) or 64 fp32 rows (4 passes): all eight
//     waves write their accumulators (bias / GELU / bf16 rounding applied), the store waves read whole rows back and
//     write 512-byte / 1-KiB row segments.
// K loop, fragment prefetch across the barrier, swizzles: exactly gemm_wide2_kernel's (kernels_gemm.h).
#pragma once
#include "kernels_gemm.h"

// cross-lane helpers of the LayerNorm statistics (gfx950: v_permlane32_swap / v_permlane16_swap, DPP)
typedef unsigned mocr_u32x2 __attribute__((ext_vector_type(2)));
// lanes 0-31: a[l] + a[l + 32]; lanes 32-63: b[l - 32] + b[l]
__device__ __forceinline__ float swap32_sum(float a, float b) {
    const mocr_u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}
// 16-lane rows 0 and 2: a[l] + a[l + 16]; rows 1 and 3: b[l - 16] + b[l]
__device__ __forceinline__ float swap16_sum(float a, float b) {
    const mocr_u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}
template <int CTRL> __device__ __forceinline__ float dpp_f32(float v) {
    return __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), CTRL, 0xf, 0xf, false));
}

// K64 (r03, second session): with the one-barrier-per-two-K-tiles schedule (PAIR) the LDS image of a pair of K-tiles is
// ONE 64-deep tile of 128-byte rows: a DMA request then covers 8 rows x 128 B - eight whole cache lines - instead of 16
// rows x 64 B - sixteen half lines.  tools/probe/dma_probe.hip: the LDS-DMA stream of this kernel's operand pattern (A
// panels shared by 8-32 blocks of an XCD, one W slice for all) runs at 55-61 GB/s per CU in 64-byte rows and at 94-117
// GB/s per CU in 128-byte rows; the K loops ran at the former rate.  -DMOCR_PERS_K64=0 builds the 32-deep image for A/B.
#ifndef MOCR_PERS_K64
#define MOCR_PERS_K64 1
#endif

constexpr int PERS_LDS = 160 * 1024;      // four 32 KiB ring slots + 32 KiB that only the epilogue uses

// SPLIT_DMA: the LDS-DMA of a K-tile is requested by waves 0-3 alone (8 pieces each; those waves then never have a
// store in their queue) instead of by all eight waves (4 pieces each).
// (r03, measured and dropped: the requests as `buffer_load_dwordx4 ... offen lds` - descriptor and offsets in SGPRs, no
// VALU address arithmetic per request: QKV without epilogue 147 -> 174 us, the other three GEMMs unchanged.)
// PAIR: ONE barrier per two K-tiles.  K-tile 2p runs barrier-free on fragments and data the barrier of K-tile 2p - 1
// already proved landed; the barrier in the middle of K-tile 2p + 1 proves K-tiles 2p + 2 and 2p + 3 and hands the slots of
// 2p and 2p + 1 back (refilled right behind it and at the head of the next K-tile).  Two K-tiles in flight instead of
// three.
// STRIP (with PAIR, EPI_BIAS_RESID: the N = 768 GEMMs): the block owns a STRIP of rows of ONE 256-column slice instead of
// whole 256-row tiles of a shared list - see the schedule below.
// LNF: the encoder's LayerNorms folded into the GEMMs on both sides of them (r03; the LayerNorm launches were 7.5 % of the
// encoder at batch 256 for no FLOP).  LN(x) W^T + b = rstd (x (W o gamma)^T - mean colsum(W o gamma)) + (b + W beta):
//   * EPI_BIAS_RESID (the GEMM that WRITES the residual stream x): its store waves hold whole 256-column row segments of
//     the new x - they also write them as bf16 (p.xb: the A operand of the next GEMM, x itself, not LN(x)) and their
//     (sum, sum of squares) to p.ln_part[row][N-tile];
//   * EPI_BIAS / EPI_BIAS_GELU (the GEMM that multiplies LN(x)): A = bf16 x, W = bf16(W o gamma), p.bias = b + W beta,
//     p.csum = column sums of the folded weight; the epilogue finishes each row's statistics from the partials (one-pass
//     variance in fp32: fine while mean^2 is not orders of magnitude above the variance) and applies the identity above.
template <int EPI, bool SPLIT_DMA, bool PAIR = false, bool STRIP = false, bool LNF = false>
__global__ __launch_bounds__(512, 1) void gemm_pers_kernel(GemmParams p) {
    static_assert(!STRIP || SPLIT_DMA, "the strip schedule is built on the split-role kernel");
    constexpr int BM = 256, BN = 256, WN = 4;
    constexpr bool K64 = PAIR && (MOCR_PERS_K64 != 0);
    constexpr int DW = SPLIT_DMA ? 4 : 8;             // waves that request LDS-DMA
    // 32-deep image: slot = [A 256 x 64 B][W 256 x 64 B] per K-tile, four slots.  K64: two 64-KiB slots per PAIR of K-tiles,
    // [A 256 x 128 B][W 256 x 128 B]; a "K-tile" g is then k-step g & 1 of 64-deep tile g >> 1, and the request batch g is
    // the A rows (g even) or the W rows (g odd) of that tile.
    constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64, STAGE = A_BYTES + B_BYTES;   // 32 KiB per 32-deep K-tile
    constexpr int RING = 4 * STAGE;                                                  // 128 KiB; + 32 KiB spare = 160 KiB
    static_assert(EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_RESID, "epilogues of the encoder layers");
    extern __shared__ __attribute__((aligned(16))) char smem[];

#ifdef MOCR_EXPERIMENTS
    const int ablate = p.ablate, stagger = p.stagger;      // diagnostics (MOCR_GEMM_ABLATE / MOCR_GEMM_STAGGER): experiments build only
#else
    constexpr int ablate = 0, stagger = 0;
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l15 = lane & 15, g4 = lane >> 4;
    const bool dma_wave = wave < 4;                 // waves 0-3: every global_load_lds; waves 4-7: every global store
    const int sw = wave - 4;                        // store wave index 0..3 (valid when !dma_wave)

    // this block's tiles: XCD (blockIdx & 7) owns a contiguous chunk of the tile list; its blocks take the chunk's tiles
    // round-robin, so the tiles in flight on one XCD at any time are neighbours (shared A row-panels / W slices in L2)
    //
    // STRIP: at M = 50,432 the N = 768 GEMMs have 591 tiles for 256 CUs - three rounds for 2.31 rounds of work, and every CU
    // reaches its HBM-bound fp32 epilogue (256 KiB read + 256 KiB written) in the same microseconds.  Instead the rows are
    // cut into `nstrips` strips of whole 16-row units (85 strips of 592 / 608 rows there), a block owns one strip of one
    // 256-column slice and walks it as 256-row tiles plus ONE half tile (128 x 256: a wave owns 64 x 64, half the MFMAs, A
    // fragments and A pieces of a K-tile) for a remainder of up to 128 rows.  The half tile comes first, in the middle or
    // last by strip number, so that the blocks of the chip reach their epilogues at different times.  The `ntn` blocks of
    // a strip are neighbours in one XCD's block order (they read the same A row-panel at the same time).
    const int ntiles = p.ntm * p.ntn;
    int tile, tile_end, tstride;
    int s_rs = 0, s_re = 0, s_n0 = 0, s_rem = 0, s_hpos = -1;      // STRIP: rows [s_rs, s_re), column, remainder rows, half tile's position
    if constexpr (!STRIP) {
        tstride = gridDim.x >> 3;
        const int xcd = blockIdx.x & 7, q = ntiles >> 3, r = ntiles & 7;
        const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        tile = start + (blockIdx.x >> 3);
        tile_end = start + (xcd < r ? q + 1 : q);
    } else {
        const int slots = gridDim.x >> 3, xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
        const int spx = slots / p.ntn, used = spx * p.ntn;          // whole strips inside one XCD's blocks
        int strip, col;
        if (k < used) { strip = xcd * spx + k / p.ntn; col = k - (k / p.ntn) * p.ntn; }
        else { const int j = (k - used) * 8 + xcd; strip = 8 * spx + j / p.ntn; col = j - (j / p.ntn) * p.ntn; }   // the XCDs' left-over blocks
        const int nstrips = 8 * spx + (8 * (slots - used)) / p.ntn;
        tile = 0; tile_end = 0; tstride = 1;
        if (strip < nstrips) {
            const int U = (p.M + 15) >> 4, base = U / nstrips, extra = U - base * nstrips;
            const int u0 = strip * base + min(strip, extra), nu = base + (strip < extra ? 1 : 0);
            s_rs = u0 * 16; s_re = min(p.M, (u0 + nu) * 16);
            if (s_re > s_rs) {
                const int R = s_re - s_rs;
                s_rem = R & 255;
                tile_end = (R >> 8) + (s_rem ? 1 : 0);
                if (s_rem && s_rem <= 128) {
#ifdef MOCR_EXPERIMENTS
                    const int hmode = p.first_round;      // MOCR_GEMM_HPOS: 0 by thirds (last / first / middle), 1 by halves (last / first), 2 all last, 3 all first
#else
                    constexpr int hmode = 0;
#endif
                    const int v = hmode == 0 ? strip % 3 : hmode == 1 ? strip % 2 : hmode == 2 ? 0 : 1;
                    s_hpos = v == 0 ? tile_end - 1 : v == 1 ? 0 : tile_end >> 1;
                }
                s_n0 = col * BN;
            }
        }
    }
    if (tile >= tile_end) return;
    // Experiment knob (MOCR_GEMM_STAGGER, x 1024 cycles): the four CUs that are neighbours in an XCD's block order start a
    // quarter, a half, three quarters of that time apart, so that the chip's blocks do not all reach their epilogues at once
    if (stagger > 0) {
        const int units = (int)((blockIdx.x >> 3) & 3) * stagger / 4;
        for (int i = 0; i < units; ++i) __builtin_amdgcn_s_sleep(16);
    }
#ifdef MOCR_EXPERIMENTS
    // diagnostics (MOCR_GEMM_ABLATE & 32768 / 65536): static priority for the requesting waves / for the store waves
    if ((ablate & 32768) && dma_wave) __builtin_amdgcn_s_setprio(1);
    if ((ablate & 65536) && !dma_wave) __builtin_amdgcn_s_setprio(1);
#endif
    // diagnostics (MOCR_GEMM_ABLATE & 131072 / 262144): the requesting waves / the store waves skip half of their MFMAs (wrong
    // products; timing only) - is the K loop bound by the requesting waves' chain of MFMAs + requests?
    const bool skip23 = ((ablate & 131072) && dma_wave) || ((ablate & 262144) && !dma_wave);
    const int nt = p.k_per_split / 32;            // even, >= 4 (checked on the host)
    const size_t a_row = (size_t)p.lda * 2, w_row = (size_t)p.ldw * 2;
    const bool guard = STRIP || (p.M & (BM - 1)) != 0;
    // tile t of this block (a global tile index; STRIP: the t-th tile of the strip): first row of the 256-row window,
    // first column, the rows [lo, hi) it owns, half tile or not
    auto tile_desc = [&](int t, int& m0, int& n0, int& lo, int& hi, bool& half) {
        if constexpr (!STRIP) {
            int tm, tn;
            gemm_tile_of(p, t, tm, tn);
            m0 = tm * BM; n0 = tn * BN; lo = m0; hi = min(p.M, m0 + BM); half = false;
        } else {
            half = t == s_hpos;
            lo = s_rs + 256 * t - ((s_hpos >= 0 && t > s_hpos) ? 256 - s_rem : 0);
            hi = min(s_re, lo + (half ? s_rem : 256));
            // the 256-row (half tile: 128-row) window ends with the tile's last row where the tile is short: no row behind
            // the strip is ever read (rows of the window above `lo` are another tile's: multiplied, never stored)
            m0 = max(0, min(lo, hi - (half ? 128 : 256)));
            n0 = s_n0;
        }
    };

    // ---- DMA: piece = 16 rows x 64 B = 1 KiB; wave w requests pieces w, w+8 of the A tile and of the W tile: 4 per
    // K-tile (r03 measurement: with the eight requests of a K-tile on four waves those waves' issue time - ~35 visible
    // cycles per request - was the K loop's critical path: +18..30 % over the loop without DMA).
    // Lane -> row lane>>2 of the piece, logical chunk (lane&3) ^ 2*((row>>3)&1).
    // K64: piece = 8 rows x 128 B; wave w requests pieces w, w + 4, ... (eight of one operand per batch); lane -> row lane>>3 of
    // the piece, logical 16-byte chunk (lane & 7) ^ ((row >> 1) & 7) - the same for all of a wave's pieces (8 (w + 4 i) rows)
    const int drow = K64 ? lane >> 3 : lane >> 2;
    const int dchunk = K64 ? (lane & 7) ^ ((4 * wave + (lane >> 4)) & 7) : (lane & 3) ^ (((lane >> 5) & 1) << 1);
    const size_t a_lane = (size_t)(wave * (K64 ? 8 : 16) + drow) * a_row + dchunk * 16, w_lane = (size_t)(wave * (K64 ? 8 : 16) + drow) * w_row + dchunk * 16;
    constexpr int NPIECE = SPLIT_DMA ? 4 : 2, PSTEP = SPLIT_DMA ? 4 : 8, LPT = 2 * NPIECE;      // per wave and K-tile: A pieces, W pieces; requests
    const bool issues_dma = !SPLIT_DMA || dma_wave;
    int pf_tile = tile, pf_kt = 0;               // the next K-tile to request: (pf_tile, pf_kt), global index pf_g
    int pf_g = 0;
    const char* pf_a = nullptr;
    const char* pf_w = nullptr;
    bool pf_half = false;                        // STRIP: the tile being requested is a half tile (A rows 128..255 are not requested)
    auto pf_set = [&]() {
        int m0_, n0_, lo_, hi_;
        bool half_;
        tile_desc(pf_tile, m0_, n0_, lo_, hi_, half_);
        pf_half = half_;
        if (ablate & 16) { m0_ &= 3 * BM; n0_ = 0; }       // diagnostics: every tile reads the same few (L2-resident) operand panels
        pf_a = (const char*)p.A + (size_t)m0_ * a_row + a_lane;
        pf_w = (const char*)p.W + (size_t)n0_ * w_row + w_lane;
    };
    pf_set();
    auto stage_next = [&]() {                    // no-op once the block's last K-tile has been requested
        if (pf_tile >= tile_end) return;
        if (!issues_dma) { ++pf_g; if (++pf_kt == nt) { pf_kt = 0; pf_tile += tstride; } return; }
        if ((ablate & 2) && pf_g >= 3) { ++pf_g; if (++pf_kt == nt) { pf_kt = 0; pf_tile += tstride; } return; }      // diagnostics: K loop without DMA
        if constexpr (K64) {
            // batch pf_g: eight pieces of A (even) or of W (odd) of 64-deep tile pf_g >> 1, into its slot's A / W half
            char* const dst = smem + ((pf_g >> 1) & 1) * (2 * STAGE) + (pf_g & 1) * (2 * A_BYTES) + wave * 1024;
            const bool w_batch = pf_g & 1;
            const char* const src = (w_batch ? pf_w : pf_a) + (size_t)(pf_kt >> 1) * 128;
            const size_t rstep = (size_t)(8 * DW) * (w_batch ? w_row : a_row);      // pieces w + DW i: 8 DW rows apart
#pragma unroll
            for (int i = 0; i < 32 / DW; ++i) {
                if (STRIP && pf_half && !w_batch && i >= 16 / DW) continue;       // a half tile has no A rows 128..255
                glds16(src + rstep * i, dst + i * (DW * 1024));
            }
            ++pf_g;
            if (++pf_kt == nt) {
                pf_kt = 0;
                pf_tile += tstride;
                if (pf_tile < tile_end) pf_set();
            }
            return;
        }
        char* sa = smem + (pf_g & 3) * STAGE + wave * 1024;
        const char* ga = pf_a + (size_t)pf_kt * 64;
        const char* gw = pf_w + (size_t)pf_kt * 64;
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) {
            // a half tile has no A rows 128..255: not requested (PAIR) / rows 0..127 requested again (the counted waits of
            // the !PAIR loop need every K-tile's requests alike) - never a row behind the tile's last
            const bool upper = STRIP && pf_half && i >= NPIECE / 2;
            if (PAIR && upper) continue;
            glds16(ga + (size_t)(16 * PSTEP * (upper ? i - NPIECE / 2 : i)) * a_row, sa + i * PSTEP * 1024);
        }
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) glds16(gw + (size_t)(16 * PSTEP * i) * w_row, sa + A_BYTES + i * PSTEP * 1024);
        ++pf_g;
        if (++pf_kt == nt) {
            pf_kt = 0;
            pf_tile += tstride;
            if (pf_tile < tile_end) pf_set();
        }
    };

    // K64: row r of the 64-deep image at r * 128, logical chunk c (k-step s: c = 4 s + g4) at chunk c ^ ((r >> 1) & 7): the
    // address of k-step 1 is the address of k-step 0 with bit 6 flipped; fragments 16 rows = 2 KiB apart
    const int frag_off = K64 ? l15 * 128 + ((g4 ^ ((l15 >> 1) & 7)) << 4) : l15 * 64 + ((g4 ^ (((l15 >> 3) & 1) << 1)) << 4);
    const unsigned offA = lds_addr_of(smem) + (wm * 128) * (K64 ? 128 : 64) + frag_off;
    const unsigned offB = lds_addr_of(smem) + (K64 ? 2 * A_BYTES : A_BYTES) + (wn * 64) * (K64 ? 128 : 64) + frag_off;

    // EPI_BIAS_RESID: the accumulators START as the tile's residual rows (out = resid + A.W^T + bias): the residual is
    // read in the accumulator layout - lane: 16 bytes of row 16i + l15, columns 16j + 4 g4 - for tile i + 1 while tile i's
    // epilogue runs, m-tile by m-tile as the epilogue hands the registers back, so no load sits in front of a store of the
    // epilogue and the read hides behind the other passes (r03: read by the store waves inside the epilogue, eight rows at
    // a time, it cost 27 us per tile).  fp32 sums: the products are added onto the residual instead of the other way
    // round - a reordering of one fp32 addition per 32 products, ~1e-7 relative.
    f32x4 acc[4][8];
    auto load_resid = [&](int tl, int i) {       // m-tile i of tile tl -> acc[.][i]
        int m0_, n0_, lo_, hi_;
        bool half_;
        tile_desc(tl, m0_, n0_, lo_, hi_, half_);
        if (STRIP && half_ && i >= 4) return;    // a half tile's wave owns m-tiles 0..3: rows 64 wm + 16 i
        int m = m0_ + wm * (STRIP && half_ ? 64 : 128) + 16 * i + l15;
        if (guard && m >= p.M) m = p.M - 1;      // rows behind M: any finite values (never stored; nor are rows outside [lo, hi))
        const float* src = p.resid + (size_t)m * p.ldo + n0_ + wn * 64 + 4 * g4;
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j][i] = *reinterpret_cast<const f32x4*>(src + 16 * j);
    };
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if constexpr (EPI == EPI_BIAS_RESID) load_resid(tile, i);
        else {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }

    int g = 0;                                   // global index of the K-tile being multiplied
    stage_next(); stage_next(); stage_next();
    if (issues_dma) {
        if constexpr (STRIP && PAIR) wait_vmcnt<0>(); // (a half tile's K-tiles have fewer requests: no count)
        else if constexpr (PAIR) wait_vmcnt<LPT>();   // K-tiles 0 and 1 (own pieces) landed; K-tile 2 stays in flight
        else wait_vmcnt<2 * LPT>();                   // K-tile 0 (own pieces); K-tiles 1 and 2 stay in flight
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // STRIP: a half tile's wave reads its A fragments from rows 64 wm .. of the K-tile instead of 128 wm ..: `hadj` bytes
    // lower (hadj: the tile being multiplied; hadj_nx: the block's next tile)
    const unsigned hadj_unit = STRIP ? (unsigned)(wm * 64 * (K64 ? 128 : 64)) : 0u;
    unsigned hadj = 0, hadj_nx = 0;
    if constexpr (STRIP) {
        if (tile == s_hpos) hadj = hadj_unit;
    }
    WideFrags P, Q;
    if constexpr (K64) { MOCR_W2K_READ_HEAD(P, offA - hadj, offB); }
    else { MOCR_W2_READ_HEAD(P, offA - hadj, offB); }

    // one K-tile: CUR holds its first six fragments (requested during the previous K-tile), NXT receives those of the next
#define MOCR_PERS_KTILE(CUR, NXT)                                                                                      \
    {                                                                                                                  \
        const unsigned so = (unsigned)((g & 3) * STAGE), sn = (unsigned)(((g + 1) & 3) * STAGE);                        \
        MOCR_W2_READ_TAIL(CUR, offA + so - hadj);                                                                           \
        stage_next();                                     /* K-tile g + 3 into the slot of K-tile g - 1 */              \
        asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(CUR.fb[0]), "+v"(CUR.fb[1]), "+v"(CUR.fb[2]), "+v"(CUR.fb[3]),      \
                     "+v"(CUR.fa[0]), "+v"(CUR.fa[1]));                                                                \
        MOCR_W2_GROUP(CUR, 0);                                                                                         \
        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(CUR.fa[2]), "+v"(CUR.fa[3]));                                       \
        MOCR_W2_GROUP(CUR, 1);                                                                                         \
        /* every fragment of this K-tile is in registers: its slot may be refilled behind the next barrier */         \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(CUR.fa[4]), "+v"(CUR.fa[5]), "+v"(CUR.fa[6]), "+v"(CUR.fa[7]));     \
        {                                                                                                              \
            /* K-tile g + 1 landed (own pieces); the K-tiles requested after it stay in flight.  N counts LOADS only:  \
               loads (DMA or ordinary) return in issue order, so with at most N operations outstanding the pieces of  \
               K-tile g + 1 cannot be among them - whatever the stores of a store wave's last epilogue are doing (they \
               are OLDER than the newest DMA: the wait may hold until they have drained, it can never end early). */   \
            const int ahead = pf_g - g - 2;                                                                            \
            if (issues_dma) {                                                                                          \
                if (ahead >= 2) wait_vmcnt<2 * LPT>(); else if (ahead == 1) wait_vmcnt<LPT>(); else wait_vmcnt<0>();   \
            }                                                                                                          \
        }                                                                                                              \
        __builtin_amdgcn_s_barrier();                                                                                  \
        asm volatile("" ::: "memory");                                                                                 \
        MOCR_W2_READ_HEAD(NXT, offA + sn - hadj_k, offB + sn);      /* hadj_k: of the tile K-tile g + 1 belongs to */   \
        if (!(STRIP && half) && !skip23) {                           /* a half tile's wave owns 64 rows: m-tiles 0..3 */           \
            MOCR_W2_GROUP(CUR, 2);                                                                                     \
            MOCR_W2_GROUP(CUR, 3);                                                                                     \
        }                                                                                                              \
        ++g;                                                                                                           \
    }

    // PAIR, first K-tile of a pair: no wait for DMA, no barrier
#define MOCR_PERS_KTILE_A(CUR, NXT)                                                                                    \
    {                                                                                                                  \
        const unsigned so = (unsigned)((g & 3) * STAGE), sn = (unsigned)(((g + 1) & 3) * STAGE);                        \
        MOCR_W2_READ_TAIL(CUR, offA + so - hadj);                                                                             \
        stage_next();                                     /* K-tile g + 3 into the slot of K-tile g - 1 */              \
        asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(CUR.fb[0]), "+v"(CUR.fb[1]), "+v"(CUR.fb[2]), "+v"(CUR.fb[3]),      \
                     "+v"(CUR.fa[0]), "+v"(CUR.fa[1]));                                                                \
        MOCR_W2_GROUP(CUR, 0);                                                                                         \
        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(CUR.fa[2]), "+v"(CUR.fa[3]));                                       \
        MOCR_W2_GROUP(CUR, 1);                                                                                         \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(CUR.fa[4]), "+v"(CUR.fa[5]), "+v"(CUR.fa[6]), "+v"(CUR.fa[7]));     \
        MOCR_W2_READ_HEAD(NXT, offA + sn - hadj, offB + sn);     /* K-tile g + 1 (same tile): proved landed by the last barrier */ \
        if (!(STRIP && half) && !skip23) {                           /* a half tile's wave owns 64 rows: m-tiles 0..3 */           \
            MOCR_W2_GROUP(CUR, 2);                                                                                     \
            MOCR_W2_GROUP(CUR, 3);                                                                                     \
        }                                                                                                              \
        ++g;                                                                                                           \
    }
    // PAIR, second K-tile of a pair: everything requested so far has landed (K-tiles g + 1, g + 2), barrier, refill
#define MOCR_PERS_KTILE_B(CUR, NXT)                                                                                    \
    {                                                                                                                  \
        const unsigned so = (unsigned)((g & 3) * STAGE), sn = (unsigned)(((g + 1) & 3) * STAGE);                        \
        MOCR_W2_READ_TAIL(CUR, offA + so - hadj);                                                                             \
        asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(CUR.fb[0]), "+v"(CUR.fb[1]), "+v"(CUR.fb[2]), "+v"(CUR.fb[3]),      \
                     "+v"(CUR.fa[0]), "+v"(CUR.fa[1]));                                                                \
        MOCR_W2_GROUP(CUR, 0);                                                                                         \
        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(CUR.fa[2]), "+v"(CUR.fa[3]));                                       \
        MOCR_W2_GROUP(CUR, 1);                                                                                         \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(CUR.fa[4]), "+v"(CUR.fa[5]), "+v"(CUR.fa[6]), "+v"(CUR.fa[7]));     \
        if (issues_dma) wait_vmcnt<0>();                  /* a DMA wave's queue holds nothing but these requests */     \
        __builtin_amdgcn_s_barrier();                                                                                  \
        asm volatile("" ::: "memory");                                                                                 \
        stage_next();                                     /* K-tile g + 3 into the slot of K-tile g - 1 */              \
        MOCR_W2_READ_HEAD(NXT, offA + sn - hadj_h, offB + sn);      /* K-tile g + 1: the NEXT tile's when this is the tile's last */ \
        if (!(STRIP && half) && !skip23) {                                                                                        \
            MOCR_W2_GROUP(CUR, 2);                                                                                     \
            MOCR_W2_GROUP(CUR, 3);                                                                                     \
        }                                                                                                              \
        ++g;                                                                                                           \
    }

    // diagnostics (experiments build, MOCR_GEMM_ABLATE & 8192): cycles per wave between the pair barriers - [0] multiply +
    // request (barrier exit .. ready for the DMA wait), [1] the DMA wait, [2] the barrier - summed over the block's K loops
    // and written to p.pos [block][wave][4] (s_memtime is a scalar-memory read: stamped only where lgkmcnt is 0 anyway)
#ifdef MOCR_EXPERIMENTS
    unsigned long long stamp_prev = 0, stamp_acc0 = 0, stamp_acc1 = 0, stamp_acc2 = 0;
    const bool stamping = (ablate & 8192) != 0;
#define MOCR_STAMP(k)                                                                                                  \
    if (stamping) {                                                                                                    \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();                                                  \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                             \
        if ((k) == 0) { if (stamp_prev) stamp_acc0 += now_ - stamp_prev; }                                             \
        else if ((k) == 1) stamp_acc1 += now_ - stamp_prev;                                                            \
        else stamp_acc2 += now_ - stamp_prev;                                                                          \
        stamp_prev = now_;                                                                                             \
    }
#else
#define MOCR_STAMP(k)
#endif
    // K64 forms of the two: k-step 0 (g even) and k-step 1 (g odd) of the 64-deep tile g >> 1 in slot (g >> 1) & 1
#define MOCR_PERS_KTILE_A64(CUR, NXT)                                                                                  \
    {                                                                                                                  \
        const unsigned st = (unsigned)(((g >> 1) & 1) * (2 * STAGE));                                                   \
        MOCR_W2K_READ_TAIL(CUR, offA + st - hadj);                                                                     \
        stage_next();                                     /* batch g + 3: the W rows of tile (g >> 1) + 1 */            \
        asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(CUR.fb[0]), "+v"(CUR.fb[1]), "+v"(CUR.fb[2]), "+v"(CUR.fb[3]),      \
                     "+v"(CUR.fa[0]), "+v"(CUR.fa[1]));                                                                \
        MOCR_W2_GROUP(CUR, 0);                                                                                         \
        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(CUR.fa[2]), "+v"(CUR.fa[3]));                                       \
        MOCR_W2_GROUP(CUR, 1);                                                                                         \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(CUR.fa[4]), "+v"(CUR.fa[5]), "+v"(CUR.fa[6]), "+v"(CUR.fa[7]));     \
        MOCR_W2K_READ_HEAD(NXT, (offA + st - hadj) ^ 64u, (offB + st) ^ 64u);      /* k-step 1 of the same tile */       \
        if (!(STRIP && half) && !skip23) {                                                                                        \
            MOCR_W2_GROUP(CUR, 2);                                                                                     \
            MOCR_W2_GROUP(CUR, 3);                                                                                     \
        }                                                                                                              \
        ++g;                                                                                                           \
    }
#define MOCR_PERS_KTILE_B64(CUR, NXT)                                                                                  \
    {                                                                                                                  \
        const unsigned st = (unsigned)(((g >> 1) & 1) * (2 * STAGE)), sn = (unsigned)((((g + 1) >> 1) & 1) * (2 * STAGE)); \
        MOCR_W2K_READ_TAIL(CUR, (offA + st - hadj) ^ 64u);                                                             \
        asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(CUR.fb[0]), "+v"(CUR.fb[1]), "+v"(CUR.fb[2]), "+v"(CUR.fb[3]),      \
                     "+v"(CUR.fa[0]), "+v"(CUR.fa[1]));                                                                \
        MOCR_W2_GROUP(CUR, 0);                                                                                         \
        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(CUR.fa[2]), "+v"(CUR.fa[3]));                                       \
        MOCR_W2_GROUP(CUR, 1);                                                                                         \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(CUR.fa[4]), "+v"(CUR.fa[5]), "+v"(CUR.fa[6]), "+v"(CUR.fa[7]));     \
        MOCR_STAMP(0);                                                                                                 \
        if (issues_dma) wait_vmcnt<0>();                  /* tile (g >> 1) + 1 has landed (own pieces) */               \
        MOCR_STAMP(1);                                                                                                 \
        __builtin_amdgcn_s_barrier();                     /* ... everyone's; this tile's slot is free */                \
        asm volatile("" ::: "memory");                                                                                 \
        MOCR_STAMP(2);                                                                                                 \
        stage_next();                                     /* batch g + 3: the A rows of tile (g >> 1) + 2, into this tile's slot */ \
        MOCR_W2K_READ_HEAD(NXT, offA + sn - hadj_h, offB + sn);                                                        \
        if (!(STRIP && half) && !skip23) {                                                                                        \
            MOCR_W2_GROUP(CUR, 2);                                                                                     \
            MOCR_W2_GROUP(CUR, 3);                                                                                     \
        }                                                                                                              \
        ++g;                                                                                                           \
    }

    for (; tile < tile_end; tile += tstride) {
        int m0, n0, row_lo, row_hi;
        bool half;
        tile_desc(tile, m0, n0, row_lo, row_hi, half);
        [[maybe_unused]] const unsigned own_lo = (unsigned)(row_lo - m0), own_n = (unsigned)(row_hi - row_lo);      // STRIP: the tile's rows inside the window
        if constexpr (STRIP) {
            hadj = half ? hadj_unit : 0u;
            hadj_nx = (tile + tstride == s_hpos) ? hadj_unit : 0u;
        }
        // bias of this lane's 16 columns: requested here, used in the epilogue (a load in front of the K loop instead of
        // a round trip in the epilogue)
        // (EPI_BIAS_RESID: every register counts there - the bias is added by the store waves instead, to whole rows: a
        // lane then needs the four values of its own columns only)
        float bias[4][4];
        float4 bias_row = make_float4(0.f, 0.f, 0.f, 0.f);
        // LNF (the GEMM behind a LayerNorm): the row statistics and the column constants are requested HERE by the store
        // waves alone - lane l of store wave sw: row 64 sw + l of the tile and column 64 sw + l - and handed to every wave
        // through LDS in the epilogue.  A DMA wave must not load anything in its epilogue: the load queues behind the next
        // tile's K-tiles and its wait holds until they have landed (r03: +2 us per tile).
        [[maybe_unused]] f32x4 st_a, st_b;
        [[maybe_unused]] float cs_l = 0.f, bias_l = 0.f;
        if constexpr (EPI == EPI_BIAS_RESID) {
            if (!dma_wave) bias_row = *reinterpret_cast<const float4*>(p.bias + n0 + 4 * lane);
        } else if constexpr (LNF) {
            if (!dma_wave) {
                int m = m0 + 64 * sw + lane;
                if (guard && m >= p.M) m = p.M - 1;
                // inline asm: hipcc waits for an ordinary load in a kernel with LDS-DMA in flight at once - vmcnt(0) right behind
                // the request, a memory round trip in front of every tile's first barrier (r03: QKV +25 us); the wait is in
                // ln_handover, where the values are used
                const float* sp = p.ln_part + (size_t)m * 8;
                const float* cp = p.csum + n0 + 64 * sw + lane;
                const float* bp = p.bias + n0 + 64 * sw + lane;
                asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:16\n\t"
                             "global_load_dword %2, %5, off\n\tglobal_load_dword %3, %6, off"
                             : "=&v"(st_a), "=&v"(st_b), "=&v"(cs_l), "=&v"(bias_l)
                             : "v"(sp), "v"(cp), "v"(bp)
                             : "memory");
            }
        } else {
            const float* bsrc = p.bias + n0 + wn * 64 + 4 * g4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 bv = *reinterpret_cast<const float4*>(bsrc + 16 * j);
                bias[j][0] = bv.x; bias[j][1] = bv.y; bias[j][2] = bv.z; bias[j][3] = bv.w;
            }
        }
        // LNF, the GEMM behind a LayerNorm: the store waves put what they requested above into the spare 32 KiB (free between
        // epilogues) in front of the tile's LAST pair of K-tiles - [256] (mean, rstd) of the tile's rows, then [256] (column
        // sum, bias) of its columns; the barriers of that pair publish it, the epilogue only reads
        auto ln_handover = [&]() {
            if constexpr (LNF && EPI != EPI_BIAS_RESID) {
                if (!dma_wave) {
                    // the loads requested at the head of the tile have landed long ago; the wait also names their registers,
                    // so that none of the arithmetic can move above it
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(st_a), "+v"(st_b), "+v"(cs_l), "+v"(bias_l));      // (a store wave: stores + these loads)
                    float* const sst = reinterpret_cast<float*>(smem + RING);
                    const float inv_k = 1.0f / (float)p.k_per_split;
                    const float mean = ((st_a.x + st_a.z) + (st_b.x + st_b.z)) * inv_k;
                    const float var = fmaxf(((st_a.y + st_a.w) + (st_b.y + st_b.w)) * inv_k - mean * mean, 0.f);
                    *reinterpret_cast<float2*>(sst + 2 * (64 * sw + lane)) = make_float2(mean, 1.0f / sqrtf(var + p.ln_eps));
                    *reinterpret_cast<float2*>(sst + 512 + 2 * (64 * sw + lane)) = make_float2(cs_l, bias_l);
                }
            }
        };
        if constexpr (STRIP) {
            for (int t = 0; t < nt; t += 2) {
                if (t + 2 >= nt) ln_handover();
                const unsigned hadj_h = t + 2 < nt ? hadj : hadj_nx;      // the tile of the K-tile behind this pair
                if constexpr (K64) {
                    MOCR_PERS_KTILE_A64(P, Q)
                    MOCR_PERS_KTILE_B64(Q, P)
                } else if constexpr (PAIR) {
                    MOCR_PERS_KTILE_A(P, Q)
                    MOCR_PERS_KTILE_B(Q, P)
                } else {
                    { const unsigned hadj_k = hadj; MOCR_PERS_KTILE(P, Q) }
                    { const unsigned hadj_k = hadj_h; MOCR_PERS_KTILE(Q, P) }
                }
            }
        } else if constexpr (PAIR) {
            constexpr unsigned hadj_h = 0;
            for (int t = 0; t < nt; t += 2) {
                if (t + 2 >= nt) ln_handover();
                if constexpr (K64) {
                    MOCR_PERS_KTILE_A64(P, Q)
                    MOCR_PERS_KTILE_B64(Q, P)
                } else {
                    MOCR_PERS_KTILE_A(P, Q)
                    MOCR_PERS_KTILE_B(Q, P)
                }
            }
        } else {
            constexpr unsigned hadj_k = 0;
            for (int t = 0; t < nt; t += 2) {
                if (t + 2 >= nt) ln_handover();
                MOCR_PERS_KTILE(P, Q)
                MOCR_PERS_KTILE(Q, P)
            }
        }
        // ------------------------------------------------------------------------------------------ epilogue of `tile`
        // LDS free right now: the slot of the K-tile just multiplied, (g - 1) & 3 (every wave passed that K-tile's barrier
        // with all its fragments in registers), and the spare 32 KiB.  K-tiles g, g + 1, g + 2 of the NEXT tile sit in the
        // other three slots (landed / in flight); P holds the first six fragments of K-tile g already.
        const bool give_up_p = LNF || STRIP || (ablate & 4096);      // (4096: diagnostics - the plain kernel pays the re-request too)
        if (give_up_p) {
            // The LayerNorm-folding epilogues need ~30 registers more than the plain ones (the strip schedule a few), and a spill in this kernel is a
            // scratch LOAD in a DMA wave's queue: its wait drains the ring.  The first six fragments of the next K-tile (24
            // registers, requested by the last K-tile) are given up here and requested again behind the epilogue.
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(P.fb[0]), "+v"(P.fb[1]), "+v"(P.fb[2]), "+v"(P.fb[3]), "+v"(P.fa[0]), "+v"(P.fa[1]));
        }
        if (ablate & 4) {                      // diagnostics: no epilogue
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 8; ++i) { asm volatile("" :: "v"(acc[j][i])); acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            if constexpr (LNF || STRIP) {
                if constexpr (K64) { const unsigned sg = (unsigned)(((g >> 1) & 1) * (2 * STAGE)); MOCR_W2K_READ_HEAD(P, offA + sg - hadj_nx, offB + sg); }
                else { const unsigned sg = (unsigned)((g & 3) * STAGE); MOCR_W2_READ_HEAD(P, offA + sg - hadj_nx, offB + sg); }
            }
            continue;
        }
        // (the row stride is laundered per tile: hipcc otherwise hoists every row's `row * ldo` out of the tile loop - 17
        // loop-invariant 64-bit offsets it then has to spill)
        int ldo = p.ldo;
        asm volatile("" : "+s"(ldo));
        char* const piece0 = smem + RING;
        // (K64: the W half of the last 64-deep tile's slot - its A half already receives the tile after next)
        char* const piece1 = K64 ? smem + (((g - 1) >> 1) & 1) * (2 * STAGE) + 2 * A_BYTES : smem + ((g - 1) & 3) * STAGE;
        if constexpr (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) {
            [[maybe_unused]] float cs[4][4], mu[8], rs[8];
            if constexpr (LNF) {
                // handed over by the store waves in front of the last pair of K-tiles (ln_handover)
                float* const sst = reinterpret_cast<float*>(piece0);
                if (ablate & 1024) {                       // diagnostics: no hand-over reads, no barrier
#pragma unroll
                    for (int i = 0; i < 8; ++i) { mu[i] = 0.25f; rs[i] = 1.5f; }
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) { cs[j][r] = 0.5f; bias[j][r] = 0.125f; }
                } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float2 mr = *reinterpret_cast<const float2*>(sst + 2 * (wm * (STRIP && half ? 64 : 128) + 16 * i + l15));
                    mu[i] = mr.x; rs[i] = mr.y;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 c0 = *reinterpret_cast<const float4*>(sst + 512 + 2 * (wn * 64 + 16 * j + 4 * g4));
                    const float4 c1 = *reinterpret_cast<const float4*>(sst + 512 + 2 * (wn * 64 + 16 * j + 4 * g4) + 4);
                    cs[j][0] = c0.x; bias[j][0] = c0.y; cs[j][1] = c0.z; bias[j][1] = c0.w;
                    cs[j][2] = c1.x; bias[j][2] = c1.y; cs[j][3] = c1.z; bias[j][3] = c1.w;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                 // read: the staging passes may write the spare area
                asm volatile("" ::: "memory");
                }
            }
            // two passes of 128 rows x 512 B: pass h takes rows 64h .. 64h+63 of each wave's 128 (m-tiles 4h .. 4h+3);
            // staging row sr = 64 wm + (row within the 64): rows 0-63 in piece0, 64-127 in piece1;
            // 16-byte chunk c of staging row sr at chunk c ^ (sr & 15)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (STRIP && h == 1 && half) continue;      // a half tile's wave holds m-tiles 0..3 only (window rows 64 wm + 16 i)
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) {
                    const int i = 4 * h + ii;
                    const int srl = 16 * ii + l15;                      // row within this wave's piece
                    char* const base = (wm ? piece1 : piece0) + srl * 512;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float v[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if constexpr (LNF) v[r] = (ablate & 2048) ? acc[j][i][r] + bias[j][r] : fmaf(rs[i], fmaf(-mu[i], cs[j][r], acc[j][i][r]), bias[j][r]);
                            else v[r] = acc[j][i][r] + bias[j][r];
                            if constexpr (EPI == EPI_BIAS_GELU) { if (!(ablate & 32)) v[r] = gelu_fast(v[r]); }
                            acc[j][i][r] = 0.f;
                        }
                        uint2 u;
                        u.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
                        u.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
                        const int chunk = wn * 8 + 2 * j + (g4 >> 1);
                        *reinterpret_cast<uint2*>(base + ((chunk ^ (srl & 15)) << 4) + (g4 & 1) * 8) = u;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // staged (a __syncthreads() would drain the DMA waves' ring)
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (!dma_wave) {
                    // store wave sw reads staging rows 32 sw .. 32 sw + 31, two rows (2 x 512 B) per wave instruction
                    const int rsel = lane >> 5, lc = lane & 31;
                    bf16_t* const obase = reinterpret_cast<bf16_t*>(p.out) + (size_t)m0 * ldo + n0 + 8 * lc;
                    // two halves of eight instructions (32 registers of staged data at a time); the barrier that hands the
                    // staging area back sits behind the LAST read, in front of the second half's stores
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        uint4 v[8];
#pragma unroll
                        for (int it = 0; it < 8; ++it) {
                            const int sr = 32 * sw + 16 * hf + 2 * it + rsel;
                            const char* src = (sr < 64 ? piece0 : piece1) + (sr & 63) * 512;
                            v[it] = *reinterpret_cast<const uint4*>(src + ((lc ^ (sr & 15)) << 4));
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        if (hf == 1) {
                            __builtin_amdgcn_s_barrier();                // read back: the staging area may be rewritten
                            asm volatile("" ::: "memory");
                        }
#pragma unroll
                        for (int it = 0; it < 8; ++it) {
                            const int sr = 32 * sw + 16 * hf + 2 * it + rsel;
                            const int row = (sr >> 6) * (STRIP && half ? 64 : 128) + 64 * h + (sr & 63);
                            const bool mine = STRIP ? (unsigned)row - own_lo < own_n : (!guard || m0 + row < p.M);
                            // non-temporal: the QKV / FC1 output passes through once (r02: +4 % on the encoder)
                            if (mine && !(ablate & 8)) st16_nt(obase + (size_t)row * ldo, v[it]);
                        }
                    }
                } else {
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                }
            }
        } else {
            // fp32 (the residual is already in the sums): four passes of 64 rows x 1 KiB: pass q takes m-tiles 2q, 2q+1 (32
            // rows) of each wave; staging row sr = 32 wm + (row within the 32): rows 0-31 in piece0, 32-63 in piece1;
            // 16-byte chunk c (0..63) of staging row sr at chunk c ^ (sr & 15)
            // (laundered like `ldo`: the next tile's residual addresses are not to be computed above the K loop)
            int nxt = tile + tstride;
            asm volatile("" : "+s"(nxt));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (STRIP && q >= 2 && half) {        // a half tile's wave holds m-tiles 0..3 only (window rows 64 wm + 16 i)
                    if (nxt < tile_end) { load_resid(nxt, 2 * q); load_resid(nxt, 2 * q + 1); }
                    continue;
                }
#pragma unroll
                for (int ii = 0; ii < 2; ++ii) {
                    const int i = 2 * q + ii;
                    const int srl = 16 * ii + l15;
                    char* const base = (wm ? piece1 : piece0) + srl * 1024;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int chunk = wn * 16 + 4 * j + g4;
                        *reinterpret_cast<f32x4*>(base + ((chunk ^ (srl & 15)) << 4)) = acc[j][i];
                    }
                }
                // these two m-tiles' registers are free: the next tile's residual rows go there (requested BEFORE this
                // pass's stores; used by the first MFMAs of the next tile)
                if (nxt < tile_end) { load_resid(nxt, 2 * q); load_resid(nxt, 2 * q + 1); }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (!dma_wave) {
                    // store wave sw: staging rows 16 sw .. 16 sw + 15, one 1-KiB row per wave instruction, in two halves of
                    // eight; the barrier that hands the staging area back sits behind the LAST read
                    float* const obase = reinterpret_cast<float*>(p.out) + (size_t)m0 * ldo + n0 + 4 * lane;
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        float4 v[8];
#pragma unroll
                        for (int it = 0; it < 8; ++it) {
                            const int sr = 16 * sw + 8 * hf + it;
                            const char* src = (sr < 32 ? piece0 : piece1) + (sr & 31) * 1024;
                            v[it] = *reinterpret_cast<const float4*>(src + ((lane ^ (sr & 15)) << 4));
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        if (hf == 1) {
                            __builtin_amdgcn_s_barrier();                // read back: the staging area may be rewritten
                            asm volatile("" ::: "memory");
                        }
                        const int rstep = STRIP && half ? 64 : 128;
                        [[maybe_unused]] float a1[8], a2[8];
#pragma unroll
                        for (int it = 0; it < 8; ++it) {
                            const int sr = 16 * sw + 8 * hf + it;
                            const int row = (sr >> 5) * rstep + 32 * q + (sr & 31);
                            const bool mine = STRIP ? (unsigned)row - own_lo < own_n : (!guard || m0 + row < p.M);
                            const float4 o = make_float4(v[it].x + bias_row.x, v[it].y + bias_row.y, v[it].z + bias_row.z, v[it].w + bias_row.w);
                            if (mine && !(ablate & 8)) {
                                // (non-temporal here measured r03: no difference - 12.74-12.81 against 12.82-13.01 ms for the encoder)
                                *reinterpret_cast<float4*>(obase + (size_t)row * ldo) = o;
                                // (r03, measured and dropped: the copy as 16-byte stores, two rows per instruction after a lane-pair
                                // exchange - O-proj + 29-37 us instead of + 37-41 for the emit in isolation: the copy is bytes, not
                                // instructions)
                                if (LNF && !(ablate & 64)) {
                                    uint2 u;
                                    u.x = pack_bf16x2(o.x, o.y); u.y = pack_bf16x2(o.z, o.w);
                                    *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.xb) + (size_t)(m0 + row) * ldo + n0 + 4 * lane) = u;
                                }
                            }
                            if constexpr (LNF) {
                                a1[it] = (o.x + o.y) + (o.z + o.w);
                                a2[it] = fmaf(o.x, o.x, o.y * o.y) + fmaf(o.z, o.z, o.w * o.w);
                            }
                        }
                        if (LNF && !(ablate & 128)) {
                            // eight rows' sums over the wave: halves of the wave take halves of the rows (v_permlane32_swap,
                            // v_permlane16_swap, row_ror:8), then three plain DPP steps; lanes 8 k hold row 4 b5 + 2 b4 + b3 of their
                            // lane number
                            const bool h5 = lane & 32, h4 = lane & 16, h3 = lane & 8;
                            float b1[4], b2[4], c1[2], c2[2];
#pragma unroll
                            for (int r = 0; r < 4; ++r) { b1[r] = swap32_sum(a1[r], a1[r + 4]); b2[r] = swap32_sum(a2[r], a2[r + 4]); }
#pragma unroll
                            for (int r = 0; r < 2; ++r) { c1[r] = swap16_sum(b1[r], b1[r + 2]); c2[r] = swap16_sum(b2[r], b2[r + 2]); }
                            float d1 = (h3 ? c1[1] : c1[0]) + dpp_f32<0x128>(h3 ? c1[0] : c1[1]);      // row_ror:8 = lane ^ 8
                            float d2 = (h3 ? c2[1] : c2[0]) + dpp_f32<0x128>(h3 ? c2[0] : c2[1]);
                            d1 += dpp_f32<0xB1>(d1); d2 += dpp_f32<0xB1>(d2);                           // quad_perm [1,0,3,2]
                            d1 += dpp_f32<0x4E>(d1); d2 += dpp_f32<0x4E>(d2);                           // quad_perm [2,3,0,1]
                            d1 += dpp_f32<0x141>(d1); d2 += dpp_f32<0x141>(d2);                         // row_half_mirror
                            const int it_l = (h5 ? 4 : 0) + (h4 ? 2 : 0) + (h3 ? 1 : 0);
                            const int sr_l = 16 * sw + 8 * hf + it_l;
                            const int row_l = (sr_l >> 5) * rstep + 32 * q + (sr_l & 31);
                            const bool mine_l = STRIP ? (unsigned)row_l - own_lo < own_n : (!guard || m0 + row_l < p.M);
                            if ((lane & 7) == 0 && mine_l && !(ablate & 8))
                                *reinterpret_cast<float2*>(p.ln_part + (size_t)(m0 + row_l) * 8 + (n0 >> 8) * 2) = make_float2(d1, d2);
                        }
                    }
                } else {
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                }
            }
        }
        if (give_up_p) {
            if constexpr (K64) { const unsigned sg = (unsigned)(((g >> 1) & 1) * (2 * STAGE)); MOCR_W2K_READ_HEAD(P, offA + sg - hadj_nx, offB + sg); }
            else { const unsigned sg = (unsigned)((g & 3) * STAGE); MOCR_W2_READ_HEAD(P, offA + sg - hadj_nx, offB + sg); }
        }
    }
#undef MOCR_PERS_KTILE
#undef MOCR_PERS_KTILE_A
#undef MOCR_PERS_KTILE_B
#undef MOCR_PERS_KTILE_A64
#undef MOCR_PERS_KTILE_B64
#ifdef MOCR_EXPERIMENTS
    if (stamping && lane == 0 && p.pos) {
        unsigned long long* d = reinterpret_cast<unsigned long long*>(const_cast<float*>(p.pos)) + ((size_t)blockIdx.x * 8 + wave) * 4;
        d[0] = stamp_acc0; d[1] = stamp_acc1; d[2] = stamp_acc2; d[3] = (unsigned long long)g;
    }
#endif
#undef MOCR_STAMP
    // the reads requested behind the last barrier: landed before their registers are used for anything else
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(P.fb[0]), "+v"(P.fb[1]), "+v"(P.fb[2]), "+v"(P.fb[3]), "+v"(P.fa[0]), "+v"(P.fa[1]));
}
