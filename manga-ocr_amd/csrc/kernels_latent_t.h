// latent_attnT_kernel (r04): the latent decode attention with the score tile TRANSPOSED - two barriers per tile instead of three.
//
// latent_attn_kernel (kernels_latent.h) computes S[head][key] split over the waves' quarters of the 768 dims, exchanges the
// partial sums through LDS (barrier), runs the softmax split over the waves by head, and exchanges the probabilities through
// LDS again (barrier) because P.X needs P[head][key] of ALL heads in every wave.  In-kernel stamps of the 16-key form (r04,
// cycles per tile and wave): wait + request 1,120, score MFMAs 320, exchange 680, softmax 1,210, P.X 470 - the two
// exchanges and their barriers are half of the chain.  Here the score product has its operands swapped:
//     S^T[key 16][head 16] (partial, this wave's 192 dims) = Xtile . Qt^T       A = X rows (ds_read_b128), B = Qt fragments
// whose C/D layout - lane (head = lane & 15, g = lane >> 4) holds keys 4g .. 4g+3 - IS the A-operand layout of P[head][key]
// for v_mfma_f32_16x16x16_bf16.  So after ONE exchange (the partial sums, 16 bytes per lane and wave) every wave adds up the
// whole score tile, runs the softmax of all 12 heads itself (4 values per lane, two cross-row exchanges by v_permlane32_swap
// / v_permlane16_swap; redundant across the waves and identical) and feeds the probabilities straight from its registers into
// P.X: no probability exchange, no third barrier, no alpha hand-over through LDS (the accumulators have heads on their rows,
// 4g + r: the rescale factors and the final 1 / sum come by ds_bpermute, and the rescale is skipped while no head's
// running max moved).
// (Also built and measured in r04: EVERY wave computing the whole score tile - 24 MFMAs instead of 6, Qt in 96 registers, one
// barrier per tile: 13-15 % SLOWER than the split form at every length; removed.)
// The DMA ring (three 24-KiB slots, two tiles in flight, requests spread over the iteration), the wave roles (waves 0-2
// request, wave 3 stores), the counted waits, the trimmed last tile and the staged row end are latent_attn_kernel<.., 16>'s;
// by default THREE blocks share a CU (two-slot rings: 52 KiB of LDS each, 152 registers; NST = 3: 76 KiB, two blocks).
#pragma once
#include "kernels_latent.h"

#define LATT_TK 16
#define LATT_TILE_BYTES (LATT_TK * LAT_D * 2)
#define LATT_LDS_OF(NST) ((NST) * LATT_TILE_BYTES + 4 * 16 * 16 * 4)      // ring + [4 waves][16 heads][16 keys] partial scores
#define LATT_LDS LATT_LDS_OF(3)

// the four waves' partial scores of this lane's (head, 4 keys): 1 KiB apart
__device__ __forceinline__ void latT_read_partials(uint4* o, unsigned a) {
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]) : "v"(a) : "memory");
}
__device__ __forceinline__ void tr_read6(uint2* o, const unsigned* a);     // kernels_latent.h

// max / sum over the four lanes l15, l15 + 16, l15 + 32, l15 + 48 (every lane ends with the total)
__device__ __forceinline__ float latT_xg_max(float v) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(r.x), __uint_as_float(r.y));
    r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r.x), __uint_as_float(r.y));
}
__device__ __forceinline__ float latT_xg_sum(float v) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r.x) + __uint_as_float(r.y);
    r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}
// the value lane `src` holds
__device__ __forceinline__ float latT_from_lane(float v, int src) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src << 2, __float_as_int(v)));
}
// request the pieces w + 3 i, i in [I0, I1), of one tile (pieces behind np hold no valid key: not requested)
template <int I0, int I1>
__device__ __forceinline__ void latT_stage(const char* src, char* dst, unsigned src_base, int w, int np) {
    asm volatile("" : "+v"(src_base));      // (the per-piece offsets are two VALU each: not to be computed once and kept - spilled - instead)
#pragma unroll
    for (int i = I0; i < I1; ++i)
        if (w + 3 * i < np) LAT_GLDS(src + ((src_base ^ (unsigned)(32 * i)) + (unsigned)(2 * i * LAT_D * 2)), dst + (w + 3 * i) * 1024);
}

// NST = ring slots: 3 (two tiles in flight, 76 KiB: two blocks per CU) or 2 (one tile in flight, 52 KiB: THREE blocks per CU -
// the kernel needs 152 registers)
template <bool SELF, int NST = 3>
__global__ __launch_bounds__(256, NST == 2 ? 3 : 2) void latent_attnT_kernel(LatentParams p) {
    static_assert(NST == 2 || NST == 3, "ring slots");
    constexpr int TK = LATT_TK, TILE_BYTES = LATT_TILE_BYTES, NPW = TK / 2, NP = 3 * NPW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const bf16_t* const P_qt = p.qt;
    const bf16_t* const P_x = p.x;
    bf16_t* const P_out = p.out;
    const long long P_xstride = p.x_batch_stride;
    const int* const P_rowmap = p.rowmap;
    const int P_rows = p.rows, P_heads = p.heads;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: the wave roles and piece tests are SALU / branches
    const int l15 = lane & 15, g = lane >> 4;
    const int nblk = gridDim.x;
    // one context length for the whole launch, read ONCE before any DMA is in flight
    const int L = p.step ? p.step[0] + 1 : p.fixed_len;
    const int cnt = (L + TK - 1) / TK;

    // DMA source offsets (as latent_attn_kernel: piece pc = the linear 16-byte chunks 64 pc .. 64 pc + 63 of the tile image;
    // DMA wave w requests the pieces w + 3 i).  A row is 96 chunks, so piece w + 3 i covers the same chunk pattern as piece w,
    // two key rows per i further down: with r = 2 i + b the swizzle term (cp ^ r) & 15 is ((cp ^ b) & 15) ^ 2 i, i.e.
    //     offset_i = (offset_0 ^ 32 i) + 2 i * 1536
    // - ONE register instead of eight (this kernel holds 96 registers of Qt; a spill here is a scratch load with a
    // vmcnt(0) wait in front of every request).
    unsigned src_base;
    {
        const int gch = 64 * (wave < 3 ? wave : 0) + lane;        // < 192: key row b = 0 or 1
        const int b = gch / 96, cp = gch - b * 96;
        src_base = (unsigned)(b * (LAT_D * 2) + (((cp & ~15) | ((cp ^ b) & 15)) << 4));
    }
    // A operand of S^T over this wave's 192 dims: X[key = l15][192 wave + 32 s + 8 g .. + 7] = chunk 24 wave + 4 s + g of key row l15
    unsigned s_off[6];
#pragma unroll
    for (int s = 0; s < 6; ++s) s_off[s] = lat_off(l15, 24 * wave + 4 * s + g);
    // B operand of P.X: X[key = 4 g + jj][d0 + l15], read transposed (lane 4 q + p of a 16-lane group supplies the address of
    // row q, columns 4 p .. 4 p + 3): one block read per 16-column tile of this wave's 192 columns
    // Column tile e = 12 wave + dt (0 .. 47) is chunks 2 e + hp: chunk group e >> 3, nibble (2 (e & 7) + hp) ^ r0, so
    //     offset(e) = (offset(0) ^ 32 (e & 7)) + 256 (e >> 3)
    // with wave-uniform (scalar) constants: one register instead of twelve.
    unsigned tr_base;
    {
        const int q4 = l15 >> 2, p4 = l15 & 3, r0 = 4 * g + q4;
        tr_base = (unsigned)(lat_off(r0, p4 >> 1) + 8 * (p4 & 1));
    }
    const int e0 = 12 * wave;
    const unsigned smem_base = lds_addr(smem);
    // B operand of S^T: Qt[head = l15][192 wave + 32 s + 8 g .. + 7] (the MFMA columns 12 .. 15 are padding: their lanes re-read
    // heads 0 .. 3 and what they compute is never used)
    const bf16_t* const q_lane = P_qt + (size_t)(l15 < P_heads ? l15 : l15 - P_heads) * LAT_D + 192 * wave + 8 * g;
    // partial scores: [wave][head][key 16] fp32; this lane's 16 bytes (head l15, keys 4 g .. 4 g + 3) of wave w at w * 1024 + sS_lane
    const unsigned sS_lane = smem_base + NST * TILE_BYTES + (unsigned)(l15 * 64 + g * 16);
#define Q_PTR(row) (q_lane + (size_t)(row) * 16 * LAT_D)
#define X_ROW(row) (P_x + (size_t)(P_rowmap ? lat_sload(P_rowmap + (row)) : (row)) * P_xstride)

    int cr = blockIdx.x;
    if (cr >= P_rows) return;
    bf16x8 qf[6];                        // Qt of the current row
    asm_load_q(qf, Q_PTR(cr));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // nothing else in flight yet
#pragma unroll
    for (int s = 0; s < 6; ++s) asm volatile("" : "+v"(qf[s]));
    int issued = 0;                      // LOADS (DMA, Qt) issued by this wave; stores are not counted (wave 3 issues no DMA)
    int mk0 = 0, mk1 = 0, mk2 = 0;
    int ir = cr, it = 0, islot = 0;
    const bf16_t* ix = X_ROW(cr);
    const int np_last = (96 * (L - (cnt - 1) * TK) + 63) >> 6;
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
                latT_stage<0, NPW>(reinterpret_cast<const char*>(ix) + (size_t)it * TILE_BYTES,                    \
                                   smem + islot * TILE_BYTES, src_base, wave, np_);                               \
                issued += lat_pieces_of(np_, wave);                                                               \
            }                                                                                                     \
            if (islot == 0) mk0 = issued; else if (islot == 1) mk1 = issued; else mk2 = issued;                   \
            islot = islot + 1 == NST ? 0 : islot + 1;                                                         \
            if (++it == cnt) ISSUE_ADVANCE_ROW();                                                                 \
        }                                                                                                         \
    } while (0)
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
        if (is_np > 0 && wave < 3) latT_stage<I0, I1>(is_src, is_dst, src_base, wave, is_np);                     \
    } while (0)
#define ISSUE_END()                                                                                               \
    do {                                                                                                          \
        if (is_np > 0) {                                                                                          \
            if (wave < 3) issued += lat_pieces_of(is_np, wave);                                                   \
            if (islot == 0) mk0 = issued; else if (islot == 1) mk1 = issued; else mk2 = issued;                   \
            islot = islot + 1 == NST ? 0 : islot + 1;                                                         \
            if (++it == cnt) ISSUE_ADVANCE_ROW();                                                                 \
        }                                                                                                         \
    } while (0)
    // a trimmed last tile leaves the key rows behind the context untouched: whatever the slot held before (finite X rows,
    // probability exactly 0) - or, the first time round, what the previous kernel left in LDS: cleared once
    if (np_last < NP) {
#pragma unroll 4
        for (int i = tid; i < NST * TILE_BYTES / 16; i += 256)
            *reinterpret_cast<uint4*>(smem + (size_t)i * 16) = make_uint4(0, 0, 0, 0);
        __syncthreads();
    }
    ISSUE_NEXT();
    if constexpr (NST == 3) ISSUE_NEXT();
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
        float m_run = -INFINITY, l_run = 0.f;       // head l15 (the four lanes of a head - and all four waves - hold the same values)

        for (int t = 0; t < cnt; ++t) {
            // the tile of `slot` has landed (own pieces: counted wait; everyone's: the barrier), and every wave is through
            // with the tile before it: that one's slot is refilled behind the barrier
            if (wave < 3) wait_vm_newer(issued - (slot == 0 ? mk0 : slot == 1 ? mk1 : mk2));
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            ISSUE_BEGIN();
            ISSUE_PART(0, 3);
            const unsigned xt_a = smem_base + (unsigned)(slot * TILE_BYTES);
            slot = slot + 1 == NST ? 0 : slot + 1;

            // ---- partial S^T[key][head] over this wave's 192 dims
            f32x4 sacc = {0.f, 0.f, 0.f, 0.f};
            {
                unsigned sa[6];
                uint4 xs[6];
#pragma unroll
                for (int k = 0; k < 6; ++k) sa[k] = xt_a + s_off[k];
                lds_read6_b128(xs, sa);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    union { uint4 u; bf16x8 v; } cv;
                    cv.u = xs[k];
                    sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cv.v, qf[k], sacc, 0, 0, 0);
                }
            }
            // the one exchange of the tile: 16 bytes per lane
            // (a plain store: the compiler pads the MFMA -> LDS-store hazard on sacc, which it does not do inside asm)
            *reinterpret_cast<f32x4*>(smem + NST * TILE_BYTES + wave * 1024 + l15 * 64 + g * 16) = sacc;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            ISSUE_PART(3, 6);
            // ---- online softmax of head l15 over this lane's keys 4 g + r and its three partner lanes
            float pe[4];
            float al;
            {
                uint4 ps[4];
                latT_read_partials(ps, sS_lane);
                __builtin_amdgcn_sched_barrier(0);
                float sv[4];
                sv[0] = (__uint_as_float(ps[0].x) + __uint_as_float(ps[1].x)) + (__uint_as_float(ps[2].x) + __uint_as_float(ps[3].x));
                sv[1] = (__uint_as_float(ps[0].y) + __uint_as_float(ps[1].y)) + (__uint_as_float(ps[2].y) + __uint_as_float(ps[3].y));
                sv[2] = (__uint_as_float(ps[0].z) + __uint_as_float(ps[1].z)) + (__uint_as_float(ps[2].z) + __uint_as_float(ps[3].z));
                sv[3] = (__uint_as_float(ps[0].w) + __uint_as_float(ps[1].w)) + (__uint_as_float(ps[2].w) + __uint_as_float(ps[3].w));
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (t * TK + 4 * g + r >= L) sv[r] = -INFINITY;
                float mx = fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3]));
                mx = latT_xg_max(mx);
                const float mn = mx > m_run ? mx : m_run;          // finite: every tile has a valid key
                al = __expf(m_run - mn);
#pragma unroll
                for (int r = 0; r < 4; ++r) pe[r] = __expf(sv[r] - mn);
                l_run = l_run * al + latT_xg_sum((pe[0] + pe[1]) + (pe[2] + pe[3]));
                m_run = mn;
            }
            ISSUE_PART(6, NPW);
            ISSUE_END();
            // rescale the accumulators only when some head's running max moved (wave-uniform test); the factor of head
            // 4 g + r comes from the lane that owns that head
            if (__any(al != 1.0f)) {
                const float a0 = latT_from_lane(al, 4 * g), a1 = latT_from_lane(al, 4 * g + 1), a2 = latT_from_lane(al, 4 * g + 2),
                            a3 = latT_from_lane(al, 4 * g + 3);
#pragma unroll
                for (int dt = 0; dt < 12; ++dt) {
                    cacc[dt][0] *= a0; cacc[dt][1] *= a1; cacc[dt][2] *= a2; cacc[dt][3] *= a3;
                }
            }
            // ---- C[head][d] += P[head][key] X[key][d] over this wave's 192 columns, K = 16 keys: A = P[head = l15][key = 4 g + r]
            // is what the softmax left in this lane
            {
                union { uint2 u; lat_s16x4 v; } pcv;
                pcv.u = make_uint2(pack_bf16x2(pe[0], pe[1]), pack_bf16x2(pe[2], pe[3]));
                const lat_s16x4 pf = pcv.v;
                unsigned tb = tr_base;
                asm volatile("" : "+v"(tb));        // (the twelve addresses are not to be computed once and kept)
#pragma unroll
                for (int grp = 0; grp < 2; ++grp) {
                    unsigned ad[6];
                    uint2 xr[6];
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        const int e = e0 + 6 * grp + k;
                        ad[k] = (tb ^ (unsigned)(32 * (e & 7))) + (xt_a + (unsigned)(256 * (e >> 3)));
                    }
                    tr_read6(xr, ad);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        union { uint2 u; lat_s16x4 v; } cv;
                        cv.u = xr[k];
                        cacc[6 * grp + k] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pf, cv.v, cacc[6 * grp + k], 0, 0, 0);
                    }
                }
            }
        }
        // ---- finish the row: normalise, stage through the ring slot the last tile freed, store (wave 3)
        // (barrier: every wave is through with the last tile's P.X reads before anyone writes into its slot)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // (the row end's addresses are derived from a laundered lane number: hipcc otherwise computes all ~25 of them in front of
        // the row loop and keeps them live - i.e. spills them - through the tile loop)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int l15e = ln & 15, ge = ln >> 4;
        float inv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) inv[r] = 1.0f / latT_from_lane(l_run, 4 * ge + r);
        static_assert(12 * LAT_OUT_HS <= TILE_BYTES, "the finished row is staged in one ring slot");
        char* const stg = smem + islot * TILE_BYTES;
        if (ge < 3) {                               // lane group 3 holds the padding heads 12..15
            char* const wb = stg + (4 * ge) * LAT_OUT_HS + (192 * wave + l15e) * 2;
#pragma unroll
            for (int dt = 0; dt < 12; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    *reinterpret_cast<bf16_t*>(wb + r * LAT_OUT_HS + dt * 32) = f2bf(cacc[dt][r] * inv[r]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (wave == 3) {
            // its queue holds only its own Qt prefetch and stores: vmcnt(0) proves the prefetch (and acknowledges the previous
            // row's stores, a row old by now)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            char* ob = reinterpret_cast<char*>(P_out + (size_t)cr * 16 * LAT_D);
            const unsigned sa = lds_addr(stg);
#pragma unroll
            for (int b3 = 0; b3 < 3; ++b3) {
                uint4 v[6];
                unsigned ad[6];
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const int idx = ln + 64 * (6 * b3 + k), hh = idx / 96;
                    ad[k] = sa + (unsigned)(hh * LAT_OUT_HS + (idx - 96 * hh) * 16);
                }
                lds_read3_b128(v, ad[0], ad[1], ad[2]);
                lds_read3_b128(v + 3, ad[3], ad[4], ad[5]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 6; ++k) *reinterpret_cast<uint4*>(ob + (ln + 64 * (6 * b3 + k)) * 16) = v[k];
            }
        }
        // ---- the next row's Qt: prove the prefetch landed, then (and only then) copy it.  With >= 3 tiles the wait for tile
        // 2 (requested after the prefetch) already proved it; shorter rows wait here
        if (cnt < 3 && wave < 3) wait_vm_newer(issued - mkq);
#pragma unroll
        for (int s = 0; s < 6; ++s) asm volatile("" : "+v"(qn[s]));     // no copy may move above the wait
#pragma unroll
        for (int s = 0; s < 6; ++s) qf[s] = qn[s];
        cr += nblk;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef ISSUE_BEGIN
#undef ISSUE_PART
#undef ISSUE_END
#undef ISSUE_NEXT
#undef ISSUE_ADVANCE_ROW
#undef Q_PTR
#undef X_ROW
}
