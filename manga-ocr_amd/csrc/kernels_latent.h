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

#define LAT_D 768
#define LAT_TK 32
#define LAT_TILE_BYTES (LAT_TK * LAT_D * 2)
#define LAT_NST 3
#define LAT_LDS (LAT_NST * LAT_TILE_BYTES + 4 * 16 * 32 * 4 + 4 * 16 * 32 * 2)

typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

struct LatentParams {
    const bf16_t* qt;           // [rows][16][768]  (heads 12..15 zero)
    const bf16_t* x;            // keys: row-major [.., 768]
    bf16_t* out;                // [rows][16][768]  sum_key p * X[key]   (heads 12..15 not written)
    long long x_batch_stride;   // elements between two sequences' first key
    const int* step;            // self: context length = step[0] + 1 for EVERY row (a batch decodes in lockstep); null: fixed_len
    int fixed_len;
    int heads;                  // 12
    int rows;                   // sequences
    int ablate;                 // diagnostics (MOCR_LAT_ABLATE): 1 no S MFMAs, 2 no softmax reductions, 4 no P.X, 8 no DMA after tile 1
};

// byte offset of logical 16-byte chunk c of key-row r inside a tile image
__device__ __forceinline__ int lat_off(int r, int c) { return r * (LAT_D * 2) + (((c & ~15) | ((c ^ r) & 15)) << 4); }

// all-reduce over the 16 lanes of a DPP row (= one head group) with row_ror rotations: VALU only,
// no LDS round trip (a __shfl_xor is a ds_bpermute: ~100+ cycles of latency per level)
template <int N> __device__ __forceinline__ float dpp_ror(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + N, 0xf, 0xf, false));
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_ror<8>(v)); v = fmaxf(v, dpp_ror<4>(v)); v = fmaxf(v, dpp_ror<2>(v)); v = fmaxf(v, dpp_ror<1>(v));
    return v;
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_ror<8>(v); v += dpp_ror<4>(v); v += dpp_ror<2>(v); v += dpp_ror<1>(v);
    return v;
}

// LDS byte address (what ds_* instructions take) of a pointer into the dynamic LDS array
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const char*)p;
}
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
__device__ __forceinline__ void lds_read3_b128(uint4* o, unsigned a0, unsigned a1, unsigned a2) {
    asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %4\n\tds_read_b128 %2, %5\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]) : "v"(a0), "v"(a1), "v"(a2) : "memory");
}

// Persistent: block i handles sequences i, i + gridDim.x, ...; the tile stream (and the DMA ring)
// runs across sequence boundaries, so only the very first tile of a block waits for HBM.
//
// vmcnt bookkeeping.  CDNA counts loads, stores and LDS-DMA together, in issue order, so "X has
// landed" is s_waitcnt vmcnt(number of VMEM instructions this wave issued after X).  The wave keeps
// that number at run time: `issued` counts its VMEM instructions (12 global_load_lds per tile, 6
// inline-asm loads per Qt prefetch, 6 global stores per finished row - the row is staged through
// LDS so that every thread stores exactly 6 x 16 B), and every ring slot remembers the count at
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
// wait until at most `newer` (a multiple of 6) of this wave's VMEM instructions are outstanding
__device__ __forceinline__ void wait_vm_newer(int newer) {
    switch (newer) {
        case 36: asm volatile("s_waitcnt vmcnt(36)" ::: "memory"); break;
        case 30: asm volatile("s_waitcnt vmcnt(30)" ::: "memory"); break;
        case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
        case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;   // 0, or anything unexpected: drain
    }
}

// request one 48 KiB tile: 12 DMA instructions per wave (src_off: this lane's 12 source offsets)
__device__ __forceinline__ void lat_stage(const char* src, char* dst, const int (&src_off)[12], int wave) {
#pragma unroll
    for (int i = 0; i < 12; ++i) glds16(src + src_off[i], dst + (wave + 4 * i) * 1024);
}

__global__ __launch_bounds__(256, 1) void latent_attn_kernel(LatentParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // plain locals: the lambdas below must not take the address of the kernel-argument struct (that
    // would push it to scratch, and scratch traffic is VMEM traffic the bookkeeping does not count)
    const bf16_t* const P_qt = p.qt;
    const bf16_t* const P_x = p.x;
    bf16_t* const P_out = p.out;
    const long long P_xstride = p.x_batch_stride;
    const int P_rows = p.rows, P_heads = p.heads, P_ablate = p.ablate;
    (void)P_heads;
    float* sS = reinterpret_cast<float*>(smem + LAT_NST * LAT_TILE_BYTES);        // [4][16][32] partial scores
    bf16_t* sP = reinterpret_cast<bf16_t*>(sS + 4 * 16 * 32);                      // [4][16][32] probabilities
    bf16_t* sO = reinterpret_cast<bf16_t*>(sS);                                    // [8 heads][768] output staging (12 KiB)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int nblk = gridDim.x;
    // one context length for the whole launch, read ONCE before any DMA is in flight (an ordinary load
    // later would make the compiler drain the ring)
    const int L = p.step ? p.step[0] + 1 : p.fixed_len /* before any lambda */;
    const int ntile = (L + LAT_TK - 1) / LAT_TK;

    // DMA source offsets of this lane for the 12 pieces a wave copies per tile: piece pc covers the
    // linear 16-byte chunks 64*pc .. 64*pc+63 of the tile image
    int src_off[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        const int gch = 64 * (wave + 4 * i) + lane;      // 0 .. 3071
        const int r = gch / 96, cp = gch - r * 96;       // key row, physical chunk
        const int c = (cp & ~15) | ((cp ^ r) & 15);      // logical chunk stored there
        src_off[i] = r * (LAT_D * 2) + c * 16;
    }
    // byte offsets (inside a tile image) of this lane's transposed block reads for keys 8g .. 8g+3, one
    // per column tile; the block of keys 8g+4 .. 8g+7 is 4 rows further with chunk bit 2 flipped by the
    // swizzle: (off + 4 * 1536) ^ 64
    unsigned tr_off[12];
    {
        const int q4 = l15 >> 2, p4 = l15 & 3, r0 = 8 * g + q4;
#pragma unroll
        for (int dt = 0; dt < 12; ++dt) {
            const int c = ((192 * wave + 16 * dt) >> 3) + (p4 >> 1);
            tr_off[dt] = lat_off(r0, c) + 8 * (p4 & 1);
        }
    }
    const unsigned smem_base = lds_addr(smem);
    // A operand of the S product: Qt[head = lane&15][192*wave + 32*s + 8*g .. +7]
    const bf16_t* const q_lane = P_qt + (size_t)l15 * LAT_D + 192 * wave + 8 * g;
#define Q_PTR(row) (q_lane + (size_t)(row) * 16 * LAT_D)

    int cr = blockIdx.x;                 // sequence being consumed
    if (cr >= P_rows) return;
    const int cnt = ntile;
    bf16x8 qf[6];                        // Qt of the current row
    asm_load_q(qf, Q_PTR(cr));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // nothing else in flight yet: landed before the loop
#pragma unroll
    for (int s = 0; s < 6; ++s) asm volatile("" : "+v"(qf[s]));
    int issued = 0;                      // VMEM instructions issued by this wave since then
    int mk0 = 0, mk1 = 0, mk2 = 0;       // `issued` right after the tile of ring slot 0/1/2 was requested
    // issue side of the ring: next (sequence, tile) to request and the slot it goes to
    int ir = cr, it = 0, islot = 0;
#define ISSUE_NEXT()                                                                                              \
    do {                                                                                                          \
        if (ir < P_rows) {                                                                                        \
            lat_stage(reinterpret_cast<const char*>(P_x + (size_t)ir * P_xstride) + (size_t)it * LAT_TILE_BYTES,   \
                      smem + islot * LAT_TILE_BYTES, src_off, wave);                                              \
            issued += 12;                                                                                         \
            if (islot == 0) mk0 = issued; else if (islot == 1) mk1 = issued; else mk2 = issued;                   \
            islot = islot + 1 == LAT_NST ? 0 : islot + 1;                                                         \
            if (++it == cnt) { ir += nblk; it = 0; }                                                              \
        }                                                                                                         \
    } while (0)
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
        float m_run[4], l_run[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { m_run[r] = -INFINITY; l_run[r] = 0.f; }

        for (int t = 0; t < cnt; ++t) {
            // the tile of `slot` has landed: allow exactly the instructions issued after its request
            wait_vm_newer(issued - (slot == 0 ? mk0 : slot == 1 ? mk1 : mk2));
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (!(P_ablate & 8)) ISSUE_NEXT();    // refills the slot every wave has finished reading
            const char* xt = smem + slot * LAT_TILE_BYTES;
            slot = slot + 1 == LAT_NST ? 0 : slot + 1;

            // ---- partial scores over this wave's 192 dims: S[head][key], two 16-key sub-tiles
            f32x4 sacc[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int r = 0; r < 4; ++r) sacc[j][r] = 0.f;
                const int key = 16 * j + l15;
                if (!(P_ablate & 1))
#pragma unroll
                    for (int s = 0; s < 6; ++s) {
                        const int c = 24 * wave + 4 * s + g;
                        const bf16x8 xf = *reinterpret_cast<const bf16x8*>(xt + lat_off(key, c));
                        sacc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[s], xf, sacc[j], 0, 0, 0);
                    }
            }
            // C/D map of the 16x16 MFMA: col = lane&15 (key), row = 4*(lane>>4) + reg (head)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) sS[(wave * 16 + 4 * g + r) * 32 + 16 * j + l15] = sacc[j][r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            float sv[2][4];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = (4 * g + r) * 32 + 16 * j + l15;
                    sv[j][r] = (sS[o] + sS[512 + o]) + (sS[1024 + o] + sS[1536 + o]);
                }
            // ---- online softmax per head (4 heads per lane; a head's 32 keys sit on 16 lanes x 2)
            const bool ok0 = t * LAT_TK + l15 < L, ok1 = t * LAT_TK + 16 + l15 < L;
            float alpha[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v0 = ok0 ? sv[0][r] : -INFINITY, v1 = ok1 ? sv[1][r] : -INFINITY;
                float mx = fmaxf(v0, v1);
                if (!(P_ablate & 2)) mx = row16_max(mx);
                const float mn = fmaxf(m_run[r], mx);            // finite: every tile has a valid key
                alpha[r] = __expf(m_run[r] - mn);
                const float p0 = __expf(v0 - mn), p1 = __expf(v1 - mn);
                float ps = p0 + p1;
                if (!(P_ablate & 2)) ps = row16_sum(ps);
                l_run[r] = l_run[r] * alpha[r] + ps;
                m_run[r] = mn;
                bf16_t* pw = sP + (wave * 16 + 4 * g + r) * 32 + l15;
                pw[0] = f2bf(p0);
                pw[16] = f2bf(p1);
            }
            // rescale the accumulators only when some head's running max moved (wave-uniform test)
            if (__any((alpha[0] != 1.0f) | (alpha[1] != 1.0f) | (alpha[2] != 1.0f) | (alpha[3] != 1.0f))) {
#pragma unroll
                for (int dt = 0; dt < 12; ++dt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) cacc[dt][r] *= alpha[r];
            }
            // sP is private to the wave: LDS is in order per wave, only the counter must be waited for
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // ---- C[head][d] += P[head][key] X[key][d] over this wave's 192 columns
            // A operand: P[head = lane&15][key = 8*g + jj]
            const bf16x8 pf = *reinterpret_cast<const bf16x8*>(sP + (wave * 16 + l15) * 32 + 8 * g);
            if (!(P_ablate & 4)) {
                // B operand: X[key = 8*g + jj][d0 + (lane&15)], read transposed: lane 4q+p of a 16-lane group
                // supplies the address of row q, columns 4p..4p+3 of a 4 x 16 block and receives column
                // (lane&15) of the 4 rows.  Two read groups of six column tiles each.
                const unsigned xt_a = smem_base + (unsigned)(xt - smem);
#pragma unroll
                for (int grp = 0; grp < 4; ++grp) {
                    unsigned ad[6];
                    uint2 xr[6];
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const unsigned lo = tr_off[3 * grp + k];
                        ad[2 * k] = xt_a + lo;
                        ad[2 * k + 1] = xt_a + ((lo + 4 * LAT_D * 2) ^ 64u);
                    }
                    tr_read6(xr, ad);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        union { uint4 u; bf16x8 v; } cv;
                        cv.u = make_uint4(xr[2 * k].x, xr[2 * k].y, xr[2 * k + 1].x, xr[2 * k + 1].y);
                        cacc[3 * grp + k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, cv.v, cacc[3 * grp + k], 0, 0, 0);
                    }
                }
            }
        }
        // ---- finish the row: normalise, stage through LDS (two halves of 8 heads x 768 bf16 = 12 KiB,
        // the score/probability scratch), store with 3 + 3 full 16-byte accesses per thread
        float inv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) inv[r] = 1.0f / l_run[r];
        char* ob = reinterpret_cast<char*>(P_out + (size_t)cr * 16 * LAT_D);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();           // scratch free: every wave is past its last reads
            asm volatile("" ::: "memory");
            if ((g >> 1) == half) {                 // lane groups g = 2*half, 2*half+1 hold heads 8*half .. 8*half+7
#pragma unroll
                for (int dt = 0; dt < 12; ++dt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        sO[(4 * (g & 1) + r) * LAT_D + 192 * wave + 16 * dt + l15] = f2bf(cacc[dt][r] * inv[r]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            {
                // 768 chunks of 16 B = 8 heads x 768 x 2 B: thread tid moves chunks tid, tid+256, tid+512
                uint4 v[3];
                const unsigned so_a = lds_addr(sO) + tid * 16;
                lds_read3_b128(v, so_a, so_a + 4096, so_a + 8192);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    *reinterpret_cast<uint4*>(ob + half * (8 * LAT_D * 2) + (tid + 256 * k) * 16) = v[k];
            }
        }
        issued += 6;                                  // the 3 + 3 stores above
        // ---- the next row's Qt: prove the prefetch landed, then (and only then) copy it.  With >= 3 tiles
        // the wait for tile 2 (requested after the prefetch) already proved it; shorter rows wait here
        // (at most 2 tiles + 6 stores = 30 younger instructions).
        if (cnt < 3) wait_vm_newer(issued - mkq);
#pragma unroll
        for (int s = 0; s < 6; ++s) asm volatile("" : "+v"(qn[s]));     // no copy may move above the wait
#pragma unroll
        for (int s = 0; s < 6; ++s) qf[s] = qn[s];
        cr += nblk;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef ISSUE_NEXT
#undef Q_PTR
}
