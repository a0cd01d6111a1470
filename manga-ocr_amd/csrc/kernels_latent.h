// Latent ("absorbed") single-query attention for the decode step, bf16, on the matrix cores.
//
// For one head h with K = X Wk_h^T + bk, V = X Wv_h^T + bv (X = the 768-wide rows the keys/values
// are projected from):
//     score(key)   = q_h . K[key] = (Wk_h^T q_h) . X[key]  +  q_h . bk        (2nd term: same for every
//                                                                              key, cancels in softmax)
//     ctx_h        = sum_key p[key] V[key] = Wv_h (sum_key p[key] X[key]) + bv   (sum p = 1)
// So with qt_h = Wk_h^T q_h (768 wide, scale 1/8 folded in) the step only has to stream X ONCE for
// all 12 heads and for keys and values together: 1,536 B per key instead of 2 x 12 x 128 = 3,072 B.
// For the cross-attention X is the encoder output itself (the cross-K/V GEMM and its 1.2 MB/crop
// buffer disappear); for the self-attention X is the layer's input row, cached per token.
//   (exact in real arithmetic; TF/models/bert/modeling_bert.py:139-279 is the projected form)
//
// One block (4 waves) per sequence.  Heads are the MFMA M dimension (12 padded to 16):
//     S[16 x 32 keys]  = Qt[16 x 768] . Xtile^T          v_mfma_f32_16x16x32_bf16, K split over the waves
//     C[16 x 768]     += P[16 x 32]  . Xtile[32 x 768]   each wave owns 192 of the 768 columns
// with an online softmax across key tiles of 32.  X tiles (48 KiB) arrive by global_load_lds into
// a 3-deep LDS ring (two tiles in flight, counted vmcnt); 16-byte chunks are XOR-swizzled with
// the key index (on the DMA source address and on every read) so the ds_read_b128 row reads of the
// S product are bank-conflict-free; the P.X product reads the same image column-wise with
// ds_read_b64_tr_b16 (2-way conflicts, irrelevant next to the 48 KiB/tile HBM stream).
#pragma once
#include "common.h"
#include "kernels_attn.h"

#define LAT_D 768
#define LAT_TK 32
#define LAT_TILE_BYTES (LAT_TK * LAT_D * 2)
#define LAT_NST 3
#define LAT_OUT_HS (LAT_D * 2 + 16)   // head-row stride of the output staging image
#define LAT_LDS (LAT_NST * LAT_TILE_BYTES + 4 * 16 * 32 * 4 + 4 * 16 * 32 * 2)     // TK = 32; LatCfg<TK>::LDS in general

typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

struct LatentParams {
    const bf16_t* qt;           // [rows][16][768]  (heads 12..15 zero)
    const bf16_t* x;            // keys: row-major [.., 768]
    bf16_t* out;                // [rows][16][768]  sum_key p * X[key]   (heads 12..15 not written)
    long long x_batch_stride;   // elements between two sequences' first key
    const int* rowmap;          // null: sequence r reads slot r of x; else (compacted batches, r04) slot rowmap[r]
    const int* step;            // self: context length = step[0] + 1 for EVERY row (a batch decodes in lockstep); null: fixed_len
    int fixed_len;
    int heads;                  // 12
    int rows;                   // sequences
    unsigned long long* dbg;    // diagnostics: per-segment cycle sums of block 0 / wave 0 (s_memtime), 8 slots; null normally
    int ablate;                 // diagnostics (MOCR_LAT_ABLATE): 1 no S MFMAs, 2 no softmax reductions, 4 no P.X, 8 no DMA after tile 1
};

// byte offset of logical 16-byte chunk c of key-row r inside a tile image
__device__ __forceinline__ int lat_off(int r, int c) { return r * (LAT_D * 2) + (((c & ~15) | ((c ^ r) & 15)) << 4); }

// all-reduce over the 16 lanes of a DPP row (= one head group) with row_ror rotations: VALU only,
// no LDS round trip (a __shfl_xor is a ds_bpermute: ~100+ cycles of latency per level)
template <int N> __device__ __forceinline__ float dpp_ror(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + N, 0xf, 0xf, false));
}
// max(v, v rotated by N inside its 16-lane row) as ONE instruction; the s_nop covers the
// "VALU write -> DPP read" hazard (2 wait states), which hipcc does not pad inside asm (§5.7 item 2)
#define DPP_MAX_STEP(v, N) asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 row_ror:" #N " row_mask:0xf bank_mask:0xf" : "=v"(v) : "v"(v))
__device__ __forceinline__ float row16_max(float v) {
#ifdef MOCR_ASM_DPP_MAX
    DPP_MAX_STEP(v, 8); DPP_MAX_STEP(v, 4); DPP_MAX_STEP(v, 2); DPP_MAX_STEP(v, 1);
#else
    v = fmaxf(v, dpp_ror<8>(v)); v = fmaxf(v, dpp_ror<4>(v)); v = fmaxf(v, dpp_ror<2>(v)); v = fmaxf(v, dpp_ror<1>(v));
#endif
    return v;
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_ror<8>(v); v += dpp_ror<4>(v); v += dpp_ror<2>(v); v += dpp_ror<1>(v);
    return v;
}

// (lds_addr: kernels_attn.h)
// Six transposed 4x16 block reads + their wait in ONE asm statement (the compiler neither counts
// nor pads inline-asm LDS reads: cdna_hip_programming.md §5.7 form (i)).  hipcc puts a
// s_waitcnt vmcnt(0) in front of the ds_read_tr builtin while LDS-DMA is in flight (it cannot
// tell the read from the DMA's target slot), which would drain the ring once per tile.
__device__ __forceinline__ void tr_read6(uint2* o, const unsigned* a) {
    asm volatile(
        "ds_read_b64_tr_b16 %0, %6\n\tds_read_b64_tr_b16 %1, %7\n\tds_read_b64_tr_b16 %2, %8\n\t"
        "ds_read_b64_tr_b16 %3, %9\n\tds_read_b64_tr_b16 %4, %10\n\tds_read_b64_tr_b16 %5, %11\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5])
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5])
        : "memory");
}
// twelve row reads (the whole S operand of a tile: 2 sub-tiles x 6 k-steps) in flight at once
__device__ __forceinline__ void lds_read12_b128(uint4* o, const unsigned* a) {
    asm volatile(
        "ds_read_b128 %0, %12\n\tds_read_b128 %1, %13\n\tds_read_b128 %2, %14\n\tds_read_b128 %3, %15\n\t"
        "ds_read_b128 %4, %16\n\tds_read_b128 %5, %17\n\tds_read_b128 %6, %18\n\tds_read_b128 %7, %19\n\t"
        "ds_read_b128 %8, %20\n\tds_read_b128 %9, %21\n\tds_read_b128 %10, %22\n\tds_read_b128 %11, %23\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]),
          "=&v"(o[8]), "=&v"(o[9]), "=&v"(o[10]), "=&v"(o[11])
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]), "v"(a[9]),
          "v"(a[10]), "v"(a[11])
        : "memory");
}
__device__ __forceinline__ void lds_read6_b128(uint4* o, const unsigned* a) {
    asm volatile(
        "ds_read_b128 %0, %6\n\tds_read_b128 %1, %7\n\tds_read_b128 %2, %8\n\tds_read_b128 %3, %9\n\t"
        "ds_read_b128 %4, %10\n\tds_read_b128 %5, %11\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5])
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5])
        : "memory");
}
__device__ __forceinline__ void tr_read12(uint2* o, const unsigned* a) {
    asm volatile(
        "ds_read_b64_tr_b16 %0, %12\n\tds_read_b64_tr_b16 %1, %13\n\tds_read_b64_tr_b16 %2, %14\n\t"
        "ds_read_b64_tr_b16 %3, %15\n\tds_read_b64_tr_b16 %4, %16\n\tds_read_b64_tr_b16 %5, %17\n\t"
        "ds_read_b64_tr_b16 %6, %18\n\tds_read_b64_tr_b16 %7, %19\n\tds_read_b64_tr_b16 %8, %20\n\t"
        "ds_read_b64_tr_b16 %9, %21\n\tds_read_b64_tr_b16 %10, %22\n\tds_read_b64_tr_b16 %11, %23\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]),
          "=&v"(o[8]), "=&v"(o[9]), "=&v"(o[10]), "=&v"(o[11])
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]), "v"(a[9]),
          "v"(a[10]), "v"(a[11])
        : "memory");
}
__device__ __forceinline__ void lds_read2_b128(uint4* o, unsigned a0, unsigned a1) {
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(o[0]), "=&v"(o[1]) : "v"(a0), "v"(a1) : "memory");
}
__device__ __forceinline__ void lds_read_b128_b64(uint4& o0, uint2& o1, unsigned a0, unsigned a1) {
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b64 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(o0), "=&v"(o1) : "v"(a0), "v"(a1) : "memory");
}
__device__ __forceinline__ void lds_read3_b128(uint4* o, unsigned a0, unsigned a1, unsigned a2) {
    asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %4\n\tds_read_b128 %2, %5\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]) : "v"(a0), "v"(a1), "v"(a2) : "memory");
}

// Persistent: block i handles sequences i, i + gridDim.x, ...; the tile stream (and the DMA ring)
// runs across sequence boundaries, so only the very first tile of a block waits for HBM.
//
// vmcnt bookkeeping.  LOADS (LDS-DMA and ordinary global loads) come back in issue order, so "load X has
// landed" is s_waitcnt vmcnt(number of LOADS this wave issued after X).  STORES must not be among the instructions
// such a wait lets fly: measured on this part (kernels_qqt.h), vmcnt(N) with N younger stores in flight does not
// prove that an older LDS-DMA has landed - the stores' acknowledgements can overtake it - and counting them out
// instead makes every wait behind a store wait for its acknowledgement (4 us per row under load).  So the waves
// are specialised: waves 0-2 request ALL the tile DMA (16 of a tile's 48 pieces each) and never store; wave 3
// requests no DMA, never waits on a tile (the block's barrier covers it) and stores every finished row.  A wave
// that waits on DMA thus has only loads in its queue.  The wave keeps their count at run time: `issued` (up to
// 16 global_load_lds per tile, 6 inline-asm loads per Qt prefetch), and every ring slot remembers the count at
// which its tile was requested.  The Qt loads are inline asm on purpose: an ordinary global load
// beside LDS-DMA makes hipcc put s_waitcnt vmcnt(0) in front of its first use, which would drain the
// ring once per tile (cdna_hip_programming.md §5, "Three .s-level traps" (b)).  Because the compiler
// believes an asm load's destination is valid at once, the prefetched Qt lives in registers local to
// the row's loop body and is copied only behind a counted wait that proves it landed - no register
// that a load is still filling is ever carried around a loop back-edge (§5.7 item 1).
__device__ __forceinline__ void asm_load_q(bf16x8* q, const bf16_t* qrow) {
#pragma unroll
    for (int s = 0; s < 6; ++s)
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(q[s]) : "v"(qrow + 32 * s) : "memory");
}
// wait until at most `newer` of this wave's VMEM instructions are outstanding (exact: rounding down
// would make the wave wait for DMA of the NEXT tile; vmcnt is a 6-bit immediate, hence the switch)
#define LAT_VM_CASE(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
__device__ __forceinline__ void wait_vm_newer(int newer) {
    switch (newer) {
        LAT_VM_CASE(1) LAT_VM_CASE(2) LAT_VM_CASE(3) LAT_VM_CASE(4) LAT_VM_CASE(5) LAT_VM_CASE(6) LAT_VM_CASE(7)
        LAT_VM_CASE(8) LAT_VM_CASE(9) LAT_VM_CASE(10) LAT_VM_CASE(11) LAT_VM_CASE(12) LAT_VM_CASE(13) LAT_VM_CASE(14)
        LAT_VM_CASE(15) LAT_VM_CASE(16) LAT_VM_CASE(17) LAT_VM_CASE(18) LAT_VM_CASE(19) LAT_VM_CASE(20) LAT_VM_CASE(21)
        LAT_VM_CASE(22) LAT_VM_CASE(23) LAT_VM_CASE(24) LAT_VM_CASE(25) LAT_VM_CASE(26) LAT_VM_CASE(27) LAT_VM_CASE(28)
        LAT_VM_CASE(29) LAT_VM_CASE(30) LAT_VM_CASE(31) LAT_VM_CASE(32) LAT_VM_CASE(33) LAT_VM_CASE(34) LAT_VM_CASE(35)
        LAT_VM_CASE(36) LAT_VM_CASE(37) LAT_VM_CASE(38) LAT_VM_CASE(39) LAT_VM_CASE(40) LAT_VM_CASE(41) LAT_VM_CASE(42)
        LAT_VM_CASE(43) LAT_VM_CASE(44) LAT_VM_CASE(45) LAT_VM_CASE(46) LAT_VM_CASE(47) LAT_VM_CASE(48) LAT_VM_CASE(49)
        LAT_VM_CASE(50) LAT_VM_CASE(51) LAT_VM_CASE(52) LAT_VM_CASE(53) LAT_VM_CASE(54) LAT_VM_CASE(55) LAT_VM_CASE(56)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;   // 0, or anything unexpected: drain
    }
}
#undef LAT_VM_CASE

// One int through the SCALAR cache (the address is wave-uniform).  Written as asm because hipcc turns `rowmap[row]` into a
// vector load followed by s_waitcnt vmcnt(0) + v_readfirstlane - a drain of the DMA ring at every row (the pointer is not
// provably invariant, so it will not use s_load by itself).
__device__ __forceinline__ int lat_sload(const int* p) {
    int v;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}

// request one tile: DMA wave w (0..2) takes the pieces pc = w + 3i (1 KiB = 64 consecutive 16-byte chunks of the
// tile image) with pc < np.  A full tile is np = 48 pieces (16 DMA instructions per DMA wave); a sequence's last tile
// asks only for the pieces that hold its valid keys (src_off: this lane's 16 source offsets).  The branch is
// wave-uniform; the caller adds lat_pieces_of(np, w) to its load count.
template <int I0, int I1, int NPW>
__device__ __forceinline__ void lat_stage(const char* src, char* dst, const unsigned (&src_off)[NPW], int w, int np) {
#pragma unroll
    for (int i = I0; i < I1; ++i)
        if (w + 3 * i < np) LAT_GLDS(src + src_off[i], dst + (w + 3 * i) * 1024);
}
__device__ __forceinline__ int lat_pieces_of(int np, int w) { return (np - w + 2) / 3; }   // #i in 0..NPW-1 with w+3i < np (np <= 3 NPW)

// SELF only names the instantiation (self: context = step[0] + 1 keys of the per-token cache; cross: the fixed 197
// encoder rows), so that profiles list the two launches of a decode layer as two kernels.
//
// TK = keys per tile (r04).  TK = 32: one block per CU, 48-KiB tiles, 156 KiB of LDS (r01-r03).  TK = 16: 24-KiB tiles, 78
// KiB of LDS and at most 256 registers, so that TWO blocks share a CU: tools/probe/stream_probe.hip (r03) showed that the
// memory system delivers this byte stream at 0.88-0.91 of the HBM peak while the one-block kernel takes 0.62 - its
// per-tile chain (wait, barrier, score MFMAs, exchange through LDS, barrier, softmax, barrier, P.X) runs on ONE wave per
// SIMD with nothing to fill its gaps.  With two co-resident blocks the second block's waves are that filler: each block
// keeps its own ring, its own barriers and its own request bookkeeping (nothing is shared, nothing new to count), and
// the hardware interleaves the two chains.  A 16-key tile is one 16-key score sub-tile and, for P.X, one
// v_mfma_f32_16x16x16_bf16 per column tile (K = 16 keys: one transposed block read each); the 197 cross keys are 13
// tiles with 11 keys of padding work instead of 7 tiles with 27.
template <int TK> struct LatCfg {
    static_assert(TK == 16 || TK == 32, "key tile");
    static constexpr int TILE_BYTES = TK * LAT_D * 2;
    static constexpr int NPW = TK / 2;               // 1-KiB pieces a DMA wave requests per full tile
    static constexpr int NP = 3 * NPW;               // pieces of a full tile
    static constexpr int J = TK / 16;                // 16-key score sub-tiles
    static constexpr int LDS = LAT_NST * TILE_BYTES + 4 * 16 * TK * 4 + 4 * 16 * TK * 2;
    static constexpr int BLOCKS_PER_CU = TK == 16 ? 2 : 1;
    // the tile's request in three parts (behind the top, the score and the probability barrier)
    static constexpr int I1 = TK == 32 ? 6 : 3, I2 = TK == 32 ? 11 : 6;
};
typedef short lat_s16x4 __attribute__((ext_vector_type(4)));

template <bool SELF, int TK = 32>
__global__ __launch_bounds__(256, LatCfg<TK>::BLOCKS_PER_CU) void latent_attn_kernel(LatentParams p) {
    using C = LatCfg<TK>;
    constexpr int TILE_BYTES = C::TILE_BYTES, NPW = C::NPW, NP = C::NP, J = C::J;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // plain locals: the lambdas below must not take the address of the kernel-argument struct (that
    // would push it to scratch, and scratch traffic is VMEM traffic the bookkeeping does not count)
    const bf16_t* const P_qt = p.qt;
    const bf16_t* const P_x = p.x;
    bf16_t* const P_out = p.out;
    const long long P_xstride = p.x_batch_stride;
    const int* const P_rowmap = p.rowmap;
    const int P_rows = p.rows, P_heads = p.heads;
    // Diagnostic cycle stamps (s_memtime) are compiled in only with -DMOCR_LAT_STAMPS: a runtime
    // "if (dbg)" branch right behind an MFMA chain jumps over the compiler's hazard padding
    // (MFMA write -> v_accvgpr_read) and silently corrupts the last accumulator.
#ifdef MOCR_LAT_STAMPS
    unsigned long long* const P_dbg = p.dbg;
    unsigned long long tsum[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
#define STAMP(k)                                                                                   \
    {                                                                                              \
        unsigned long long tn;                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tn)::"memory");                    \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        tsum[k] += tn - tprev;                                                                     \
        tprev = tn;                                                                                \
    }
#else
#define STAMP(k)
#endif
    float* sS = reinterpret_cast<float*>(smem + LAT_NST * TILE_BYTES);             // [4][16][TK] partial scores
    bf16_t* sP = reinterpret_cast<bf16_t*>(sS + 4 * 16 * TK);                      // [16][TK] probabilities (a quarter of its area)
    float* sAl = reinterpret_cast<float*>(sP + 16 * TK);                           // [16] per-head rescale factors, then [16] row sums
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int nblk = gridDim.x;
    // one context length for the whole launch, read ONCE before any DMA is in flight (an ordinary load
    // later would make the compiler drain the ring)
    const int L = p.step ? p.step[0] + 1 : p.fixed_len /* before any lambda */;
    const int ntile = (L + TK - 1) / TK;

    // DMA source offsets of this lane for the NPW pieces a wave copies per tile: piece pc covers the
    // linear 16-byte chunks 64*pc .. 64*pc+63 of the tile image
    unsigned src_off[NPW];
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        const int gch = 64 * ((wave < 3 ? wave : 0) + 3 * i) + lane;      // 0 .. 96 TK - 1 (wave 3 requests nothing)
        const int r = gch / 96, cp = gch - r * 96;       // key row, physical chunk
        const int c = (cp & ~15) | ((cp ^ r) & 15);      // logical chunk stored there
        src_off[i] = (unsigned)(r * (LAT_D * 2) + c * 16);
    }
    // byte offsets (inside a tile image) of this lane's transposed block reads, one per column tile.
    // TK = 32 (K = 32 MFMA: keys 8g .. 8g+7 per lane group): the block of keys 8g .. 8g+3; the block of keys 8g+4 .. 8g+7 is
    // 4 rows further with chunk bit 2 flipped by the swizzle: (off + 4 * 1536) ^ 64.  TK = 16 (K = 16 MFMA): keys 4g .. 4g+3.
    unsigned tr_off[12];
    {
        const int q4 = l15 >> 2, p4 = l15 & 3, r0 = (TK == 32 ? 8 : 4) * g + q4;
#pragma unroll
        for (int dt = 0; dt < 12; ++dt) {
            const int c = ((192 * wave + 16 * dt) >> 3) + (p4 >> 1);
            tr_off[dt] = lat_off(r0, c) + 8 * (p4 & 1);
        }
    }
    // byte offsets (inside a tile image) of this lane's row reads of the S product: [sub-tile j][k-step s]
    unsigned s_off[6 * J];
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
        for (int s = 0; s < 6; ++s) s_off[6 * j + s] = lat_off(16 * j + l15, 24 * wave + 4 * s + g);
    const unsigned smem_base = lds_addr(smem);
    // A operand of the S product: Qt[head = lane&15][192*wave + 32*s + 8*g .. +7]
    // (the MFMA rows 12..15 are padding: their lanes re-read heads 0..3 - same cache lines, no extra
    // traffic - and their results are never stored)
    const bf16_t* const q_lane = P_qt + (size_t)(l15 < P_heads ? l15 : l15 - P_heads) * LAT_D + 192 * wave + 8 * g;
#define Q_PTR(row) (q_lane + (size_t)(row) * 16 * LAT_D)
    // keys of sequence `row`: its own slot of x, or (r04, compacted batches) the slot its rowmap entry names
#define X_ROW(row) (P_x + (size_t)(P_rowmap ? lat_sload(P_rowmap + (row)) : (row)) * P_xstride)

    int cr = blockIdx.x;                 // sequence being consumed
    if (cr >= P_rows) return;
    const int cnt = ntile;
    bf16x8 qf[6];                        // Qt of the current row
    asm_load_q(qf, Q_PTR(cr));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // nothing else in flight yet: landed before the loop
#pragma unroll
    for (int s = 0; s < 6; ++s) asm volatile("" : "+v"(qf[s]));
    int issued = 0;                      // LOADS (DMA, Qt prefetch) issued by this wave since then; stores are not counted
    int mk0 = 0, mk1 = 0, mk2 = 0;       // `issued` right after the tile of ring slot 0/1/2 was requested
    // issue side of the ring: next (sequence, tile) to request and the slot it goes to
    int ir = cr, it = 0, islot = 0;
    const bf16_t* ix = X_ROW(cr);        // key rows of sequence `ir` (a wave-uniform, i.e. scalar, rowmap load: no VMEM)
#define ISSUE_ADVANCE_ROW()                                                                                       \
    do {                                                                                                          \
        ir += nblk; it = 0;                                                                                       \
        if (ir < P_rows) ix = X_ROW(ir);                                                                          \
    } while (0)
#define ISSUE_NEXT()                                                                                              \
    do {                                                                                                          \
        if (ir < P_rows) {                                                                                        \
            const int np_ = it == cnt - 1 ? np_last : NP;                                                         \
            if (wave < 3) {                                                                                       \
                lat_stage<0, NPW, NPW>(reinterpret_cast<const char*>(ix) + (size_t)it * TILE_BYTES,                \
                                       smem + islot * TILE_BYTES, src_off, wave, np_);                            \
                issued += lat_pieces_of(np_, wave);                                                               \
            }                                                                                                     \
            if (islot == 0) mk0 = issued; else if (islot == 1) mk1 = issued; else mk2 = issued;                   \
            islot = islot + 1 == LAT_NST ? 0 : islot + 1;                                                         \
            if (++it == cnt) ISSUE_ADVANCE_ROW();                                                                 \
        }                                                                                                         \
    } while (0)
    // The same request spread over the three phases of an iteration (r02): a DMA wave's 16 pieces cost it 60-185 cycles
    // of issue each, and issued as one burst behind the top barrier they delayed the wave's own score phase - and with
    // it the block's next barrier.  The slot being refilled is read by nobody during the whole iteration, so pieces
    // 0-5 go out behind the top barrier, 6-10 behind the score barrier (the softmax phase is short and latency-bound)
    // and 11-15 behind the probability barrier (TK = 16: 0-2, 3-5, 6-7); the bookkeeping (load count, slot mark, ring
    // position) is done once, with the last part.  Cross launch at 2560 rows: 193.6 -> 182.7 us.  (Moving ALL the DMA
    // issue to two extra loader waves - a six-wave block - was also built and measured: 183.9 us, i.e. nothing more;
    // removed.)
    const char* is_src = nullptr;
    char* is_dst = nullptr;
    int is_np = 0;
#define ISSUE_BEGIN()                                                                                             \
    do {                                                                                                          \
        is_np = 0;                                                                                                \
        if (ir < P_rows) {                                                                                        \
            is_np = it == cnt - 1 ? np_last : NP;                                                                 \
            is_src = reinterpret_cast<const char*>(ix) + (size_t)it * TILE_BYTES;                                  \
            is_dst = smem + islot * TILE_BYTES;                                                                   \
        }                                                                                                         \
    } while (0)
#define ISSUE_PART(I0, I1)                                                                                        \
    do {                                                                                                          \
        if (is_np > 0 && wave < 3) lat_stage<I0, I1, NPW>(is_src, is_dst, src_off, wave, is_np);                  \
    } while (0)
#define ISSUE_END()                                                                                               \
    do {                                                                                                          \
        if (is_np > 0) {                                                                                          \
            if (wave < 3) issued += lat_pieces_of(is_np, wave);                                                   \
            if (islot == 0) mk0 = issued; else if (islot == 1) mk1 = issued; else mk2 = issued;                   \
            islot = islot + 1 == LAT_NST ? 0 : islot + 1;                                                         \
            if (++it == cnt) ISSUE_ADVANCE_ROW();                                                                 \
        }                                                                                                         \
    } while (0)
    // pieces of a sequence's last tile that hold valid keys (rows are 96 chunks, pieces 64 chunks)
    const int np_last = (96 * (L - (cnt - 1) * TK) + 63) >> 6;
    // The key rows such a trimmed request leaves alone keep whatever the slot held before: older X rows
    // (finite; their probabilities are exactly 0) - or, the first time round, whatever the previous kernel
    // left in LDS, possibly NaN patterns, and 0 x NaN would poison P.X.  So the ring is cleared once.
    if (np_last < NP) {
#pragma unroll 4
        for (int i = tid; i < LAT_NST * TILE_BYTES / 16; i += 256)
            *reinterpret_cast<uint4*>(smem + (size_t)i * 16) = make_uint4(0, 0, 0, 0);
        __syncthreads();
    }
    ISSUE_NEXT();
    ISSUE_NEXT();
    int slot = 0;

    while (cr < P_rows) {
        // Qt of the next row (L2-resident, 6 loads per lane): requested now, consumed at the row's end
        bf16x8 qn[6];
        {
            const int nx = cr + nblk;
            asm_load_q(qn, Q_PTR(nx < P_rows ? nx : cr));
        }
        issued += 6;
        const int mkq = issued;
        f32x4 cacc[12];
#pragma unroll
        for (int dt = 0; dt < 12; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) cacc[dt][r] = 0.f;
        float m_run = -INFINITY, l_run = 0.f;       // this wave's heads {4g + wave}

        for (int t = 0; t < cnt; ++t) {
            STAMP(5)   // everything since the last stamp of the previous tile (loop overhead, row end)
            // the tile of `slot` has landed: allow exactly the instructions issued after its request
            if (wave < 3) wait_vm_newer(issued - (slot == 0 ? mk0 : slot == 1 ? mk1 : mk2));
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            ISSUE_BEGIN();                        // refills the slot every wave has finished reading, in three parts
            ISSUE_PART(0, C::I1);
            const char* xt = smem + slot * TILE_BYTES;
            slot = slot + 1 == LAT_NST ? 0 : slot + 1;
            STAMP(0)   // wait + barrier + DMA issue
#ifndef MOCR_LAT_NOCOMPUTE   // timing experiment only: DMA ring without any consumer work

            // ---- partial scores over this wave's 192 dims: S[head][key], J 16-key sub-tiles.  All the
            // operand reads are issued before the first MFMA (left to itself hipcc serialises
            // read -> wait -> MFMA twelve times through one register quad)
            const unsigned xt_a = smem_base + (unsigned)(xt - smem);
            f32x4 sacc[J];
            {
                unsigned sa[6 * J];
                uint4 xs[6 * J];
#pragma unroll
                for (int k = 0; k < 6 * J; ++k) sa[k] = xt_a + s_off[k];
                if constexpr (J == 2) lds_read12_b128(xs, sa);
                else lds_read6_b128(xs, sa);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < J; ++j) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) sacc[j][r] = 0.f;
#pragma unroll
                    for (int s = 0; s < 6; ++s) {
                        union { uint4 u; bf16x8 v; } cv;
                        cv.u = xs[6 * j + s];
                        sacc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[s], cv.v, sacc[j], 0, 0, 0);
                    }
                }
            }
            STAMP(1)   // S-phase reads + MFMAs
            // C/D map of the 16x16 MFMA: col = lane&15 (key), row = 4*(lane>>4) + reg (head)
#pragma unroll
            for (int j = 0; j < J; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) sS[(wave * 16 + 4 * g + r) * TK + 16 * j + l15] = sacc[j][r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            ISSUE_PART(C::I1, C::I2);
            // ---- online softmax, split over the waves: wave w owns accumulator register r = w, i.e. heads
            // {4g + w}; a head's TK keys sit on the 16 lanes of its group x J sub-tiles
            float v[J];
            {
                const int o = (4 * g + wave) * TK + l15;
#pragma unroll
                for (int j = 0; j < J; ++j)
                    v[j] = (sS[o + 16 * j] + sS[16 * TK + o + 16 * j]) + (sS[32 * TK + o + 16 * j] + sS[48 * TK + o + 16 * j]);
            }
            STAMP(2)   // partial-score exchange through LDS (write, barrier, read)
#pragma unroll
            for (int j = 0; j < J; ++j)
                if (t * TK + 16 * j + l15 >= L) v[j] = -INFINITY;
            {
                float mx = v[0];
                if constexpr (J == 2) mx = v[0] > v[1] ? v[0] : v[1];
                mx = row16_max(mx);
                const float mn = mx > m_run ? mx : m_run;          // finite: every tile has a valid key
                const float al = __expf(m_run - mn);
                float pe[J], psum = 0.f;
#pragma unroll
                for (int j = 0; j < J; ++j) { pe[j] = __expf(v[j] - mn); psum += pe[j]; }
                l_run = l_run * al + row16_sum(psum);
                m_run = mn;
                bf16_t* pw = sP + (4 * g + wave) * TK + l15;        // shared P[head][key]
#pragma unroll
                for (int j = 0; j < J; ++j) pw[16 * j] = f2bf(pe[j]);
                if (l15 == 0) sAl[4 * g + wave] = al;               // shared alpha[head]
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            ISSUE_PART(C::I2, NPW);
            ISSUE_END();
            // alpha[4g..4g+3] and the A operand P[head = lane&15][keys of this lane group] through inline asm: a plain
            // LDS load here gets a compiler s_waitcnt vmcnt(0) in front (DMA-alias conservatism)
            uint4 ap[2];
            if constexpr (TK == 32) lds_read2_b128(ap, lds_addr(sAl) + 16 * g, lds_addr(sP) + (l15 * 32 + 8 * g) * 2);
            else {
                uint2 p2;
                lds_read_b128_b64(ap[0], p2, lds_addr(sAl) + 16 * g, lds_addr(sP) + (l15 * 16 + 4 * g) * 2);
                ap[1] = make_uint4(p2.x, p2.y, 0u, 0u);
            }
            __builtin_amdgcn_sched_barrier(0);
            // rescale the accumulators only when some head's running max moved (wave-uniform test)
            {
                const float a0 = __uint_as_float(ap[0].x), a1 = __uint_as_float(ap[0].y), a2 = __uint_as_float(ap[0].z),
                            a3 = __uint_as_float(ap[0].w);
                if (__any((a0 != 1.0f) | (a1 != 1.0f) | (a2 != 1.0f) | (a3 != 1.0f))) {
#pragma unroll
                    for (int dt = 0; dt < 12; ++dt) {
                        cacc[dt][0] *= a0; cacc[dt][1] *= a1; cacc[dt][2] *= a2; cacc[dt][3] *= a3;
                    }
                }
            }
            STAMP(3)   // softmax + rescale
            // ---- C[head][d] += P[head][key] X[key][d] over this wave's 192 columns
            if constexpr (TK == 32) {
                union { uint4 u; bf16x8 v; } pcv;
                pcv.u = ap[1];
                const bf16x8 pf = pcv.v;
                // B operand: X[key = 8*g + jj][d0 + (lane&15)], read transposed: lane 4q+p of a 16-lane group
                // supplies the address of row q, columns 4p..4p+3 of a 4 x 16 block and receives column
                // (lane&15) of the 4 rows.  Two read groups of six column tiles (12 block reads) each.
#pragma unroll
                for (int grp = 0; grp < 2; ++grp) {
                    unsigned ad[12];
                    uint2 xr[12];
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        const unsigned lo = tr_off[6 * grp + k];
                        ad[2 * k] = xt_a + lo;
                        ad[2 * k + 1] = xt_a + ((lo + 4 * LAT_D * 2) ^ 64u);
                    }
                    tr_read12(xr, ad);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        union { uint4 u; bf16x8 v; } cv;
                        cv.u = make_uint4(xr[2 * k].x, xr[2 * k].y, xr[2 * k + 1].x, xr[2 * k + 1].y);
                        cacc[6 * grp + k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, cv.v, cacc[6 * grp + k], 0, 0, 0);
                    }
                }
            } else {
                // K = 16: A = P[head = lane&15][key = 4g + jj], B = X[key = 4g + jj][d0 + (lane&15)] - one transposed
                // block read per column tile, all twelve in flight at once
                union { uint2 u; lat_s16x4 v; } pcv;
                pcv.u = make_uint2(ap[1].x, ap[1].y);
                const lat_s16x4 pf = pcv.v;
                unsigned ad[12];
                uint2 xr[12];
#pragma unroll
                for (int k = 0; k < 12; ++k) ad[k] = xt_a + tr_off[k];
                tr_read12(xr, ad);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 12; ++k) {
                    union { uint2 u; lat_s16x4 v; } cv;
                    cv.u = xr[k];
                    cacc[k] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pf, cv.v, cacc[k], 0, 0, 0);
                }
            }
            STAMP(4)   // P read + transposed reads + P.X MFMAs
#else
            (void)xt;
            ISSUE_PART(C::I1, C::I2);      // the bare ring: the whole request, no consumer work
            ISSUE_PART(C::I2, NPW);
            ISSUE_END();
#endif
        }
        // ---- finish the row: normalise, stage through LDS (12 heads x 768 bf16 = 18 KiB in the ring slot the last tile
        // freed), store with 18 full 16-byte accesses per lane of wave 3
#ifndef MOCR_LAT_NOEPI       // timing experiment only
        STAMP(5)   // loop exit
        if (l15 == 0) sAl[16 + 4 * g + wave] = l_run;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        STAMP(6)   // row end: first barrier
        float inv[4];
        {
            uint4 l4;     // inline asm for the same reason as the alpha/P read above
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(l4) : "v"(lds_addr(sAl) + 64 + 16 * g) : "memory");
            __builtin_amdgcn_sched_barrier(0);
            inv[0] = 1.0f / __uint_as_float(l4.x); inv[1] = 1.0f / __uint_as_float(l4.y);
            inv[2] = 1.0f / __uint_as_float(l4.z); inv[3] = 1.0f / __uint_as_float(l4.w);
        }
        // The ring slot of the tile just consumed is free until the next tile's top-of-loop barrier (its refill
        // is requested after that barrier), so all 12 heads x 768 bf16 are staged there in ONE pass: head rows
        // LAT_OUT_HS bytes apart (1536 + 16: the three lane groups of a wave then write different banks).
        static_assert(12 * LAT_OUT_HS <= TILE_BYTES, "the finished row is staged in one ring slot");
        char* const stg = smem + islot * TILE_BYTES;
        if (g < 3) {                                // lane group 3 holds the padding heads 12..15
            char* const wb = stg + (4 * g) * LAT_OUT_HS + (192 * wave + l15) * 2;
#pragma unroll
            for (int dt = 0; dt < 12; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    *reinterpret_cast<bf16_t*>(wb + r * LAT_OUT_HS + dt * 32) = f2bf(cacc[dt][r] * inv[r]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        STAMP(7)   // row end: normalise + stage + second barrier
        {
            // 1152 chunks of 16 B = 12 heads x 1536 B, contiguous in the output: wave 3 moves them all, lane l the
            // chunks l + 64k, k = 0..17, six at a time.  Its queue holds only its own Qt prefetch and stores: the
            // vmcnt(0) in front proves the prefetch (and acknowledges the previous row's stores, a row old by now).
            if (wave == 3) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                char* ob = reinterpret_cast<char*>(P_out + (size_t)cr * 16 * LAT_D);
                const unsigned sa = lds_addr(stg);
#pragma unroll
                for (int b3 = 0; b3 < 3; ++b3) {
                    uint4 v[6];
                    unsigned ad[6];
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        const int idx = lane + 64 * (6 * b3 + k), hh = idx / 96;
                        ad[k] = sa + (unsigned)(hh * LAT_OUT_HS + (idx - 96 * hh) * 16);
                    }
                    lds_read3_b128(v, ad[0], ad[1], ad[2]);
                    lds_read3_b128(v + 3, ad[3], ad[4], ad[5]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < 6; ++k) *reinterpret_cast<uint4*>(ob + (lane + 64 * (6 * b3 + k)) * 16) = v[k];
                }
            }
        }
        STAMP(8)   // row end: the stores (wave 3)
#endif
        // ---- the next row's Qt: prove the prefetch landed, then (and only then) copy it.  With >= 3 tiles
        // the wait for tile 2 (requested after the prefetch) already proved it; shorter rows wait here
        // (at most 2 tiles = 2 NPW younger loads).
        if (cnt < 3 && wave < 3) wait_vm_newer(issued - mkq);
#pragma unroll
        for (int s = 0; s < 6; ++s) asm volatile("" : "+v"(qn[s]));     // no copy may move above the wait
#pragma unroll
        for (int s = 0; s < 6; ++s) qf[s] = qn[s];
        cr += nblk;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef MOCR_LAT_STAMPS
    if (P_dbg && blockIdx.x == 0 && lane == 0)
        for (int k = 0; k < 10; ++k) P_dbg[10 * wave + k] = tsum[k];
#endif
#undef STAMP
#undef ISSUE_BEGIN
#undef ISSUE_PART
#undef ISSUE_END
#undef ISSUE_NEXT
#undef ISSUE_ADVANCE_ROW
#undef Q_PTR
#undef X_ROW
}
