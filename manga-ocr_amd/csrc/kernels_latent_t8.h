// latent_attnT8_kernel (r04): the fp8 (OCP e4m3) latent decode attention in the transposed-score-tile form of
// kernels_latent_t.h - the opt-in "fp8 MFMA attention" of BASELINE configs[4] / MOCR_FLAG_FP8_ATTENTION.
//
// What carries over from latent_attn_fp8_kernel (kernels_latent8.h, r02), unchanged in meaning: the key/value rows are e4m3
// bytes (768 B per key, one static scale sx per source); both products run on v_mfma_f32_16x16x32_fp8_fp8; the absorbed query
// is quantised in the kernel, once per row, per head (scale = the head's amax over its 768 dims / 448); probabilities are
// quantised as e4m3(256 p) and the row sum is taken over the QUANTISED weights; the softmax is fp32; a finished row is staged
// OUTSIDE the ring (a ring slot must never hold anything but e4m3 bytes: the key rows a trimmed last tile leaves untouched
// are multiplied by probability 0, and a stale 0x7F / 0xFF byte is an e4m3 NaN).
// What changes (kernels_latent_t.h has the reasoning and the measurements of the bf16 form):
//   * one 32-key tile (24 KiB) per iteration on a TWO-slot ring, 67.5 KiB of LDS: TWO blocks per CU instead of one block that
//     consumes two tiles per iteration on a five-slot ring;
//   * S^T[key][head] = Xtile . Qt8^T (operands swapped): after ONE exchange of the partial sums (this wave's 192 of the 768
//     bytes; 2 x 16 bytes per lane) every wave adds up the whole score tile and runs the softmax of all 12 heads itself; the
//     C/D layout of the two 16-key sub-tiles - lane (head = lane & 15, g = lane >> 4) holds keys 4g .. 4g+3 and 16+4g .. 16+4g+3 -
//     is, byte for byte after the e4m3 rounding, the A operand of P8[head][32 k-slots] for the P.X product: the probabilities
//     never go through LDS.  The B operand follows the same key order: the transposed block read of lane group g
//     (ds_read_b64_tr_b8: lane 2q + p addresses key row q' of the group's eight) is given rows 4g .. 4g+3, 16+4g .. 16+4g+3;
//   * two barriers per tile instead of three (per 64 keys) - and a second block on the CU to fill them.
#pragma once
#include "kernels_latent8.h"
#include "kernels_latent_t.h"

#define LATT8_TK 32
#define LATT8_TILE_BYTES (LATT8_TK * LAT_D)          // 24 KiB
#define LATT8_NST 2
#define LATT8_SCRATCH (19 * 1024)                    // partial scores [4 waves][2 sub-tiles][16 heads][16 B x 4] = 8 KiB; at a row's end the staged row (18,624 B)
#define LATT8_LDS (LATT8_NST * LATT8_TILE_BYTES + LATT8_SCRATCH + 512)

__device__ __forceinline__ void lds_read8_b128_2k(uint4* o, unsigned a) {      // [j][w]: sub-tile j at + 1024, wave w at + 2048
    asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:2048\n\tds_read_b128 %2, %8 offset:4096\n\tds_read_b128 %3, %8 offset:6144\n\t"
                 "ds_read_b128 %4, %8 offset:1024\n\tds_read_b128 %5, %8 offset:3072\n\tds_read_b128 %6, %8 offset:5120\n\tds_read_b128 %7, %8 offset:7168\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]) : "v"(a) : "memory");
}

template <bool SELF>
__global__ __launch_bounds__(256, 2) void latent_attnT8_kernel(Latent8Params p) {
    constexpr int TK = LATT8_TK, TILE_BYTES = LATT8_TILE_BYTES, NST = LATT8_NST, NPW = 8, NP = 24;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const bf16_t* const P_qt = p.qt;
    const uint8_t* const P_x = p.x8;
    bf16_t* const P_out = p.out;
    const long long P_xstride = p.x_batch_stride;
    const int* const P_rowmap = p.rowmap;
    const int P_rows = p.rows, P_heads = p.heads;
    const float P_sx = p.sx;
    char* const scratch = smem + NST * TILE_BYTES;
    float* const sAm = reinterpret_cast<float*>(scratch + LATT8_SCRATCH);          // [16 heads][4 waves] query amax
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int nblk = gridDim.x;
    const int L = p.step ? p.step[0] + 1 : p.fixed_len;
    const int cnt = (L + TK - 1) / TK;

    // DMA: piece pc = wave + 3 i covers the 16-byte slots 64 pc .. 64 pc + 63 of the tile image (48 slots per key row)
    unsigned src_off[NPW];
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        const int q = 64 * ((wave < 3 ? wave : 0) + 3 * i) + lane;       // 0 .. 1535
        const int r = q / 48, sp = q - r * 48;
        const int lc = (2 * sp) ^ ((r & 15) << 1);                       // logical 8-byte chunk stored at physical chunk 2 sp
        src_off[i] = (unsigned)(r * LAT_D + lc * 8);
    }
    // A operand of S^T, sub-tile j, k-step s: X8[key = 16 j + l15][bytes 192 wave + 32 s + 8 g .. + 7]
    unsigned s_off[12];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int s = 0; s < 6; ++s) s_off[6 * j + s] = lat8_off(16 * j + l15, 24 * wave + 4 * s + g);
    // B operand of P.X: transposed block reads; lane 2 q + p of a 16-lane group addresses the group's key row q (of eight), byte
    // columns 8 p .. 8 p + 7 of the 16-column tile: rows 4 g + q (q < 4) and 16 + 4 g + (q - 4) - the k-slot order of the
    // probabilities as the softmax leaves them in this lane group
    unsigned tr_off[12];
    {
        const int q8 = l15 >> 1, p2 = l15 & 1, r0 = q8 < 4 ? 4 * g + q8 : 16 + 4 * g + (q8 - 4);
#pragma unroll
        for (int dt = 0; dt < 12; ++dt) tr_off[dt] = lat8_off(r0, 24 * wave + 2 * dt + p2);
    }
    const unsigned smem_base = lds_addr(smem);
    const unsigned sS_lane = smem_base + NST * TILE_BYTES + (unsigned)(l15 * 64 + g * 16);      // + 2048 w + 1024 j
    const bf16_t* const q_lane = P_qt + (size_t)(l15 < P_heads ? l15 : l15 - P_heads) * LAT_D + 192 * wave + 8 * g;
#define Q_PTR(row) (q_lane + (size_t)(row) * 16 * LAT_D)
#define X8_ROW(row) (reinterpret_cast<const char*>(P_x) + (size_t)(P_rowmap ? lat_sload(P_rowmap + (row)) : (row)) * P_xstride)

    int cr = blockIdx.x;
    if (cr >= P_rows) return;
    bf16x8 qb[6];                        // bf16 absorbed query of the current row (this lane's 48 values of head l15)
    asm_load_q(qb, Q_PTR(cr));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int s = 0; s < 6; ++s) asm volatile("" : "+v"(qb[s]));
    int issued = 0;
    int mk0 = 0, mk1 = 0;
    int ir = cr, it = 0, islot = 0;
    const char* ix = X8_ROW(cr);
    const int np_last = (48 * (L - (cnt - 1) * TK) + 63) >> 6;      // pieces of a sequence's last tile that hold valid keys
#define ISSUE_ADVANCE_ROW()                                                                                       \
    do {                                                                                                          \
        ir += nblk; it = 0;                                                                                       \
        if (ir < P_rows) ix = X8_ROW(ir);                                                                         \
    } while (0)
#define ISSUE_NEXT8()                                                                                             \
    do {                                                                                                          \
        if (ir < P_rows) {                                                                                        \
            const int np_ = it == cnt - 1 ? np_last : NP;                                                         \
            if (wave < 3) {                                                                                       \
                lat8_stage(ix + (size_t)it * TILE_BYTES, smem + islot * TILE_BYTES, src_off, wave, np_);          \
                issued += lat_pieces_of(np_, wave);                                                               \
            }                                                                                                     \
            if (islot == 0) mk0 = issued; else mk1 = issued;                                                      \
            islot ^= 1;                                                                                           \
            if (++it == cnt) ISSUE_ADVANCE_ROW();                                                                 \
        }                                                                                                         \
    } while (0)
    const char* is_src = nullptr;
    char* is_dst = nullptr;
    int is_np = 0;
#define ISSUE_BEGIN8()                                                                                            \
    do {                                                                                                          \
        is_np = 0;                                                                                                \
        if (ir < P_rows) {                                                                                        \
            is_np = it == cnt - 1 ? np_last : NP;                                                                 \
            is_src = ix + (size_t)it * TILE_BYTES;                                                                \
            is_dst = smem + islot * TILE_BYTES;                                                                   \
        }                                                                                                         \
    } while (0)
#define ISSUE_PART8(I0, I1)                                                                                       \
    do {                                                                                                          \
        if (is_np > 0 && wave < 3) lat8_stage<I0, I1>(is_src, is_dst, src_off, wave, is_np);                      \
    } while (0)
#define ISSUE_END8()                                                                                              \
    do {                                                                                                          \
        if (is_np > 0) {                                                                                          \
            if (wave < 3) issued += lat_pieces_of(is_np, wave);                                                   \
            if (islot == 0) mk0 = issued; else mk1 = issued;                                                      \
            islot ^= 1;                                                                                           \
            if (++it == cnt) ISSUE_ADVANCE_ROW();                                                                 \
        }                                                                                                         \
    } while (0)
    // a trimmed last tile leaves key rows of its slot untouched: they must hold finite e4m3 bytes (0 x NaN = NaN)
    if (np_last < NP) {
#pragma unroll 4
        for (int i = tid; i < NST * TILE_BYTES / 16; i += 256)
            *reinterpret_cast<uint4*>(smem + (size_t)i * 16) = make_uint4(0, 0, 0, 0);
        __syncthreads();
    }
    ISSUE_NEXT8();
    int slot = 0;

    while (cr < P_rows) {
        bf16x8 qn[6];
        {
            const int nx = cr + nblk;
            asm_load_q(qn, Q_PTR(nx < P_rows ? nx : cr));
        }
        issued += 6;
        const int mkq = issued;
        // ---- quantise this row's absorbed query: head l15's amax over the 768 dims (4 lane groups x 4 waves), q8 = e4m3(q 448 / amax)
        unsigned long long q8[6];
        float sc;                             // the score of (head l15, key) is S8 * sc
        {
            float am = 0.f;
#pragma unroll
            for (int s = 0; s < 6; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) am = fmaxf(am, fabsf((float)qb[s][j]));
            am = latT_xg_max(am);
            if (g == 0) sAm[l15 * 4 + wave] = am;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            uint4 a4;
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(a4) : "v"(lds_addr(sAm) + 16 * l15) : "memory");
            __builtin_amdgcn_sched_barrier(0);
            const float mine = fmaxf(fmaxf(fmaxf(__uint_as_float(a4.x), __uint_as_float(a4.y)), fmaxf(__uint_as_float(a4.z), __uint_as_float(a4.w))), 1e-30f);
            sc = mine * (1.0f / 448.0f) * P_sx;
            const float inv = 448.0f / mine;
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                const unsigned lo = pack4_fp8((float)qb[s][0] * inv, (float)qb[s][1] * inv, (float)qb[s][2] * inv, (float)qb[s][3] * inv);
                const unsigned hi = pack4_fp8((float)qb[s][4] * inv, (float)qb[s][5] * inv, (float)qb[s][6] * inv, (float)qb[s][7] * inv);
                q8[s] = ((unsigned long long)hi << 32) | lo;
            }
        }
        f32x4 cacc[12];
#pragma unroll
        for (int dt = 0; dt < 12; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) cacc[dt][r] = 0.f;
        float m_run = -INFINITY, l_run = 0.f;       // head l15 (identical in the four lanes of a head and in all four waves)

        for (int t = 0; t < cnt; ++t) {
            if (wave < 3) wait_vm_newer(issued - (slot == 0 ? mk0 : mk1));
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            ISSUE_BEGIN8();                         // the previous tile's slot: every wave is through with it
            ISSUE_PART8(0, 3);
            const unsigned xt_a = smem_base + (unsigned)(slot * TILE_BYTES);
            slot ^= 1;
            // ---- partial S^T[key][head] over this wave's 192 bytes: two 16-key sub-tiles, two interleaved chains
            f32x4 sacc[2];
            {
                unsigned sa[12];
                unsigned long long xa[12];
#pragma unroll
                for (int k = 0; k < 12; ++k) sa[k] = xt_a + s_off[k];
                lds_read12_b64(xa, sa);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sacc[j][r] = 0.f;
#pragma unroll
                for (int s = 0; s < 6; ++s) {
                    sacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8((long)xa[s], (long)q8[s], sacc[0], 0, 0, 0);
                    sacc[1] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8((long)xa[6 + s], (long)q8[s], sacc[1], 0, 0, 0);
                }
            }
            // the one exchange of the tile (plain stores: the compiler pads the MFMA -> LDS-store hazard)
            {
                char* const w0 = scratch + wave * 2048 + l15 * 64 + g * 16;
                *reinterpret_cast<f32x4*>(w0) = sacc[0];
                *reinterpret_cast<f32x4*>(w0 + 1024) = sacc[1];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            ISSUE_PART8(3, 6);
            // ---- online softmax of head l15 over this lane's keys 4 g + r, 16 + 4 g + r and its three partner lanes
            unsigned long long pa;          // A operand of P.X: e4m3(256 p) of this lane's eight keys, in k-slot order
            float al;
            {
                uint4 ps[8];
                lds_read8_b128_2k(ps, sS_lane);
                __builtin_amdgcn_sched_barrier(0);
                float v[8];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    v[4 * j + 0] = ((__uint_as_float(ps[4 * j].x) + __uint_as_float(ps[4 * j + 1].x)) + (__uint_as_float(ps[4 * j + 2].x) + __uint_as_float(ps[4 * j + 3].x))) * sc;
                    v[4 * j + 1] = ((__uint_as_float(ps[4 * j].y) + __uint_as_float(ps[4 * j + 1].y)) + (__uint_as_float(ps[4 * j + 2].y) + __uint_as_float(ps[4 * j + 3].y))) * sc;
                    v[4 * j + 2] = ((__uint_as_float(ps[4 * j].z) + __uint_as_float(ps[4 * j + 1].z)) + (__uint_as_float(ps[4 * j + 2].z) + __uint_as_float(ps[4 * j + 3].z))) * sc;
                    v[4 * j + 3] = ((__uint_as_float(ps[4 * j].w) + __uint_as_float(ps[4 * j + 1].w)) + (__uint_as_float(ps[4 * j + 2].w) + __uint_as_float(ps[4 * j + 3].w))) * sc;
                }
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (t * TK + 16 * j + 4 * g + r >= L) v[4 * j + r] = -INFINITY;
                float mx = fmaxf(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])), fmaxf(fmaxf(v[4], v[5]), fmaxf(v[6], v[7])));
                mx = latT_xg_max(mx);
                const float mn = mx > m_run ? mx : m_run;          // finite: every tile has a valid key
                al = __expf(m_run - mn);
                float pe[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) pe[k] = __expf(v[k] - mn) * LAT8_PSCALE;
                const unsigned lo = pack4_fp8(pe[0], pe[1], pe[2], pe[3]), hi = pack4_fp8(pe[4], pe[5], pe[6], pe[7]);
                // the row sum over the QUANTISED weights: the weights that multiply X then sum to exactly 1 after the division by l
                float pq = ((__builtin_amdgcn_cvt_f32_fp8((int)lo, 0) + __builtin_amdgcn_cvt_f32_fp8((int)lo, 1)) +
                            (__builtin_amdgcn_cvt_f32_fp8((int)lo, 2) + __builtin_amdgcn_cvt_f32_fp8((int)lo, 3))) +
                           ((__builtin_amdgcn_cvt_f32_fp8((int)hi, 0) + __builtin_amdgcn_cvt_f32_fp8((int)hi, 1)) +
                            (__builtin_amdgcn_cvt_f32_fp8((int)hi, 2) + __builtin_amdgcn_cvt_f32_fp8((int)hi, 3)));
                l_run = l_run * al + latT_xg_sum(pq) * (1.0f / LAT8_PSCALE);
                m_run = mn;
                pa = ((unsigned long long)hi << 32) | lo;
            }
            ISSUE_PART8(6, NPW);
            ISSUE_END8();
            if (__any(al != 1.0f)) {
                const float a0 = latT_from_lane(al, 4 * g), a1 = latT_from_lane(al, 4 * g + 1), a2 = latT_from_lane(al, 4 * g + 2),
                            a3 = latT_from_lane(al, 4 * g + 3);
#pragma unroll
                for (int dt = 0; dt < 12; ++dt) {
                    cacc[dt][0] *= a0; cacc[dt][1] *= a1; cacc[dt][2] *= a2; cacc[dt][3] *= a3;
                }
            }
            // ---- C[head][d] += P8[head][key] X8[key][d] over this wave's 192 columns: one transposed read per 16-column tile
            {
                unsigned ad[12];
                unsigned long long xr[12];
#pragma unroll
                for (int k = 0; k < 12; ++k) ad[k] = xt_a + tr_off[k];
                tr8_read12(xr, ad);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 12; ++k) cacc[k] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8((long)pa, (long)xr[k], cacc[k], 0, 0, 0);
            }
        }
        // ---- finish the row: normalise (1 / l, the probability scale 256 and the key scale sx), stage in the scratch, store
        // (barrier: every wave has read its partial scores - the staging area overlies them)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        int ln = lane;
        asm volatile("" : "+v"(ln));          // (row-end addresses from a laundered lane number: not hoisted in front of the row loop)
        const int l15e = ln & 15, ge = ln >> 4;
        float inv[4];
        {
            const float k = P_sx * (1.0f / LAT8_PSCALE);
#pragma unroll
            for (int r = 0; r < 4; ++r) inv[r] = k / latT_from_lane(l_run, 4 * ge + r);
        }
        static_assert(12 * LAT_OUT_HS <= LATT8_SCRATCH, "the finished row is staged in the scratch");
        if (ge < 3) {
            char* const wb = scratch + (4 * ge) * LAT_OUT_HS + (192 * wave + l15e) * 2;
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
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // its queue: its own query prefetch and stores
            char* ob = reinterpret_cast<char*>(P_out + (size_t)cr * 16 * LAT_D);
            const unsigned sa = lds_addr(scratch);
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
        // the staged row must be out of the scratch before the next row's amax exchange / partial scores go in: the next row's
        // quantisation barrier (all four waves, wave 3 behind its reads) orders that
        // ---- the next row's query: prove the prefetch landed, then copy
        if (cnt < 3 && wave < 3) wait_vm_newer(issued - mkq);
#pragma unroll
        for (int s = 0; s < 6; ++s) asm volatile("" : "+v"(qn[s]));
#pragma unroll
        for (int s = 0; s < 6; ++s) qb[s] = qn[s];
        cr += nblk;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef ISSUE_NEXT8
#undef ISSUE_BEGIN8
#undef ISSUE_PART8
#undef ISSUE_END8
#undef ISSUE_ADVANCE_ROW
#undef Q_PTR
#undef X8_ROW
}
