// fp8 (OCP e4m3) variant of the latent decode attention (kernels_latent.h) - the opt-in "fp8 MFMA attention" of
// BASELINE configs[4] / MOCR_FLAG_FP8_ATTENTION.  Same algorithm, same block structure; what changes:
//
//   * the key/value source rows are stored as e4m3 bytes: 768 B per key instead of 1,536 B - the dominant HBM stream
//     of a fat decode batch halves.  ONE static scale per source (sx): the rows are LayerNorm outputs, whose
//     elements are bounded by max|gamma| * sqrt(767) + max|beta| whatever the input, so x8 = e4m3(x / sx) with
//     sx = bound / 448 can never overflow, and e4m3 is a floating-point format: the relative precision (3 mantissa
//     bits) is the same over the ~14 binades above 2^-6, so a loose bound costs nothing but subnormal range.
//   * S = Qt . X^T and C += P . X run on v_mfma_f32_16x16x32_fp8_fp8 (the bf16 shape's lane map and rate: lane l holds
//     row/column l & 15, k = 8 (l >> 4) + byte - probed on the device, tools/probe/tr_b8_probe.hip).  The absorbed
//     query is quantised IN the kernel once per row (per-head scale from the head's amax: one cross-wave max);
//     probabilities are quantised as e4m3(256 p) (p in [0, 1]: 256 p stays below 448 and is a normal number down
//     to p = 6e-5); the softmax itself is fp32.
//   * tile = 32 keys x 768 B = 24 KiB, FIVE ring stages (120 KiB); an iteration consumes TWO tiles (64 keys), so the
//     barriers and LDS round trips between the phases are paid once per 64 keys.  The finished
//     row is staged outside the ring: unlike in the bf16 kernel a ring slot must never hold anything but e4m3
//     bytes, because the key rows a trimmed last tile leaves untouched are multiplied (by probability 0) and a stale
//     0x7F / 0xFF byte is an e4m3 NaN.
//     LDS image: 8-byte chunk c of key row r at chunk c ^ 2 (r & 15) - conflict-free both for the ds_read_b64 row
//     reads of the S product and for the ds_read_b64_tr_b8 transposed reads of the P.X product (one read per
//     16-column tile covers all 32 keys: lane 2q + p of a 16-lane group addresses key row 8g + q, columns 8p..8p+7,
//     and receives column (lane & 15) of the group's 8 key rows).
//
// Accuracy is NOT that of the bf16 path (the parity default): tests/test_gpu_fp8_attention.py reports the
// teacher-forced logit error and the id-match rate against the oracle next to the bf16 engine's.
#pragma once
#include "common.h"
#include "kernels_latent.h"

#define LAT8_TILE_BYTES (LAT_TK * LAT_D)      // 24 KiB
#define LAT8_NST 5
#define LAT8_PIECES (LAT8_TILE_BYTES / 1024)  // 24 DMA pieces of 1 KiB per tile
#define LAT8_STG (12 * LAT_OUT_HS)            // output staging: 12 heads x (768 bf16 + 16 B) = 18,624 B, shared with the
                                              // score / probability scratch (16 KiB + 1 KiB), which is idle at a row's end
#define LAT8_LDS (LAT8_NST * LAT8_TILE_BYTES + 19 * 1024 + 512)
#define LAT8_PSCALE 256.0f

struct Latent8Params {
    const bf16_t* qt;           // [rows][16][768] bf16 absorbed queries (quantised per row and head in the kernel)
    const uint8_t* x8;          // keys: e4m3 rows of 768 B
    bf16_t* out;                // [rows][16][768] bf16  sum_key p * x[key]
    long long x_batch_stride;   // bytes between two sequences' first key
    const int* rowmap;          // null: sequence r reads slot r of x8; else (compacted batches) slot rowmap[r]
    const int* step;            // self: context length = step[0] + 1; null: fixed_len
    int fixed_len;
    int heads;
    int rows;
    float sx;                   // x = x8 * sx
};

// byte offset of logical 8-byte chunk c (0..95) of key row r inside a tile image
__device__ __forceinline__ int lat8_off(int r, int c) { return r * LAT_D + ((c ^ ((r & 15) << 1)) << 3); }

template <int I0 = 0, int I1 = 8>
__device__ __forceinline__ void lat8_stage(const char* src, char* dst, const unsigned (&src_off)[8], int w, int np) {
#pragma unroll
    for (int i = I0; i < I1; ++i)
        if (w + 3 * i < np) LAT_GLDS(src + src_off[i], dst + (w + 3 * i) * 1024);
}

__device__ __forceinline__ void lds_read12_b64(unsigned long long* o, const unsigned* a) {
    asm volatile(
        "ds_read_b64 %0, %12\n\tds_read_b64 %1, %13\n\tds_read_b64 %2, %14\n\tds_read_b64 %3, %15\n\t"
        "ds_read_b64 %4, %16\n\tds_read_b64 %5, %17\n\tds_read_b64 %6, %18\n\tds_read_b64 %7, %19\n\t"
        "ds_read_b64 %8, %20\n\tds_read_b64 %9, %21\n\tds_read_b64 %10, %22\n\tds_read_b64 %11, %23\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]),
          "=&v"(o[8]), "=&v"(o[9]), "=&v"(o[10]), "=&v"(o[11])
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]), "v"(a[9]),
          "v"(a[10]), "v"(a[11])
        : "memory");
}
__device__ __forceinline__ void tr8_read12(unsigned long long* o, const unsigned* a) {
    asm volatile(
        "ds_read_b64_tr_b8 %0, %12\n\tds_read_b64_tr_b8 %1, %13\n\tds_read_b64_tr_b8 %2, %14\n\t"
        "ds_read_b64_tr_b8 %3, %15\n\tds_read_b64_tr_b8 %4, %16\n\tds_read_b64_tr_b8 %5, %17\n\t"
        "ds_read_b64_tr_b8 %6, %18\n\tds_read_b64_tr_b8 %7, %19\n\tds_read_b64_tr_b8 %8, %20\n\t"
        "ds_read_b64_tr_b8 %9, %21\n\tds_read_b64_tr_b8 %10, %22\n\tds_read_b64_tr_b8 %11, %23\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]),
          "=&v"(o[8]), "=&v"(o[9]), "=&v"(o[10]), "=&v"(o[11])
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]), "v"(a[9]),
          "v"(a[10]), "v"(a[11])
        : "memory");
}

// Same wave roles and vmcnt bookkeeping as latent_attn_kernel (see there): waves 0-2 request all tile DMA and never
// store, wave 3 stores every finished row and requests none; `issued` counts this wave's LOADS, every ring slot
// remembers the count at its request.
template <bool SELF>
__global__ __launch_bounds__(256, 1) void latent_attn_fp8_kernel(Latent8Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const bf16_t* const P_qt = p.qt;
    const uint8_t* const P_x = p.x8;
    bf16_t* const P_out = p.out;
    const long long P_xstride = p.x_batch_stride;
    const int* const P_rowmap = p.rowmap;
#define X8_SLOT(row) ((size_t)(P_rowmap ? lat_sload(P_rowmap + (row)) : (row)))      // wave-uniform: a scalar load (asm: kernels_latent.h), no VMEM
    const int P_rows = p.rows, P_heads = p.heads;
    const float P_sx = p.sx;
    float* sS = reinterpret_cast<float*>(smem + LAT8_NST * LAT8_TILE_BYTES);       // [4][16][64] partial scores (16 KiB)
    uint8_t* sP = reinterpret_cast<uint8_t*>(sS + 4 * 16 * 64);                     // [16][64] e4m3 probabilities (1 KiB)
    char* const stg = reinterpret_cast<char*>(sS);                                  // finished row: 12 heads x LAT_OUT_HS (over sS / sP)
    float* sAl = reinterpret_cast<float*>(reinterpret_cast<char*>(sS) + 19 * 1024); // [16] alpha, [16] row sums, [4][16] query amax
    static_assert(LAT8_STG <= 19 * 1024, "staging fits the scratch");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int nblk = gridDim.x;
    const int L = p.step ? p.step[0] + 1 : p.fixed_len;
    const int ntile = (L + LAT_TK - 1) / LAT_TK;

    // DMA: piece pc = (wave) + 3 i covers the 16-byte slots 64 pc .. 64 pc + 63 of the tile image (48 slots per key row)
    unsigned src_off[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int q = 64 * ((wave < 3 ? wave : 0) + 3 * i) + lane;       // 0 .. 1535
        const int r = q / 48, sp = q - r * 48;
        const int lc = (2 * sp) ^ ((r & 15) << 1);                       // logical 8-byte chunk stored at physical chunk 2 sp
        src_off[i] = (unsigned)(r * LAT_D + lc * 8);
    }
    // transposed block reads of the P.X product: d-tile dt of this wave's 192 columns; lane 2q + p -> key row 8g + q
    unsigned tr_off[12];
    {
        const int q8 = l15 >> 1, p2 = l15 & 1, r0 = 8 * g + q8;
#pragma unroll
        for (int dt = 0; dt < 12; ++dt) tr_off[dt] = lat8_off(r0, 24 * wave + 2 * dt + p2);
    }
    // row reads of the S product: [sub-tile j][k-step s]: key 16 j + l15, bytes 192 wave + 32 s + 8 g ..
    unsigned s_off[12];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int s = 0; s < 6; ++s) s_off[6 * j + s] = lat8_off(16 * j + l15, 24 * wave + 4 * s + g);
    const unsigned smem_base = lds_addr(smem);
    const bf16_t* const q_lane = P_qt + (size_t)(l15 < P_heads ? l15 : l15 - P_heads) * LAT_D + 192 * wave + 8 * g;
#define Q_PTR(row) (q_lane + (size_t)(row) * 16 * LAT_D)

    int cr = blockIdx.x;
    if (cr >= P_rows) return;
    const int cnt = ntile;
    bf16x8 qb[6];                        // bf16 absorbed query of the current row (this lane's 48 values of head l15)
    asm_load_q(qb, Q_PTR(cr));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int s = 0; s < 6; ++s) asm volatile("" : "+v"(qb[s]));
    int issued = 0;
    int mk0 = 0, mk1 = 0, mk2 = 0, mk3 = 0, mk4 = 0;
    int ir = cr, it = 0, islot = 0;
    const int np_last = (48 * (L - (cnt - 1) * LAT_TK) + 63) >> 6;      // pieces of a sequence's last tile that hold valid keys
#define ISSUE_NEXT8()                                                                                              \
    do {                                                                                                          \
        if (ir < P_rows) {                                                                                        \
            const int np_ = it == cnt - 1 ? np_last : LAT8_PIECES;                                                \
            if (wave < 3) {                                                                                       \
                lat8_stage(reinterpret_cast<const char*>(P_x) + X8_SLOT(ir) * P_xstride + (size_t)it * LAT8_TILE_BYTES, \
                           smem + islot * LAT8_TILE_BYTES, src_off, wave, np_);                                   \
                issued += lat_pieces_of(np_, wave);                                                               \
            }                                                                                                     \
            switch (islot) { case 0: mk0 = issued; break; case 1: mk1 = issued; break; case 2: mk2 = issued; break;  \
                             case 3: mk3 = issued; break; default: mk4 = issued; }                                \
            islot = islot + 1 == LAT8_NST ? 0 : islot + 1;                                                        \
            if (++it == cnt) { ir += nblk; it = 0; }                                                              \
        }                                                                                                         \
    } while (0)
    // the second request of an iteration goes out in two halves, behind the score and the probability barrier (see
    // latent_attn_kernel: the DMA issue of a burst delays the issuing wave's own phase)
    const char* is_src = nullptr;
    char* is_dst = nullptr;
    int is_np = 0;
#define ISSUE_BEGIN8()                                                                                            \
    do {                                                                                                          \
        is_np = 0;                                                                                                \
        if (ir < P_rows) {                                                                                        \
            is_np = it == cnt - 1 ? np_last : LAT8_PIECES;                                                        \
            is_src = reinterpret_cast<const char*>(P_x) + X8_SLOT(ir) * P_xstride + (size_t)it * LAT8_TILE_BYTES;   \
            is_dst = smem + islot * LAT8_TILE_BYTES;                                                              \
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
            switch (islot) { case 0: mk0 = issued; break; case 1: mk1 = issued; break; case 2: mk2 = issued; break;  \
                             case 3: mk3 = issued; break; default: mk4 = issued; }                                \
            islot = islot + 1 == LAT8_NST ? 0 : islot + 1;                                                        \
            if (++it == cnt) { ir += nblk; it = 0; }                                                              \
        }                                                                                                         \
    } while (0)
    // a trimmed last tile leaves key rows of its slot untouched: they must hold finite e4m3 bytes (0 x NaN = NaN)
    // (and a row's odd last tile reads - with probability 0 - the ring's next slot, which at the start of a block may
    // never have been written)
    if (np_last < LAT8_PIECES || (cnt & 1)) {
#pragma unroll 4
        for (int i = tid; i < LAT8_NST * LAT8_TILE_BYTES / 16; i += 256)
            *reinterpret_cast<uint4*>(smem + (size_t)i * 16) = make_uint4(0, 0, 0, 0);
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < LAT8_NST - 1; ++k) ISSUE_NEXT8();
    int slot = 0;
    bool prev_pair = false;              // the previous iteration consumed two ring slots (so two are free to refill)

    while (cr < P_rows) {
        bf16x8 qn[6];
        {
            const int nx = cr + nblk;
            asm_load_q(qn, Q_PTR(nx < P_rows ? nx : cr));
        }
        issued += 6;
        const int mkq = issued;
        // ---- quantise this row's absorbed query: per-head amax over the 768 dims (4 lane groups x 4 waves), e4m3(q / qs)
        float qs_head[4];                 // scale of heads 4g' + r as the softmax phase needs them: heads {4 g + wave}... see below
        unsigned long long q8[6];
        {
            float am = 0.f;
#pragma unroll
            for (int s = 0; s < 6; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) am = fmaxf(am, fabsf((float)qb[s][j]));
            am = fmaxf(am, __shfl_xor(am, 16, 64));
            am = fmaxf(am, __shfl_xor(am, 32, 64));
            if (g == 0) sAl[32 + wave * 16 + l15] = am;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            uint4 a4[4];     // amax[wave'][4 g .. 4 g + 3] for the four waves (inline asm: see latent_attn_kernel on LDS reads beside DMA)
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:64\n\tds_read_b128 %2, %4 offset:128\n\tds_read_b128 %3, %4 offset:192\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(a4[0]), "=&v"(a4[1]), "=&v"(a4[2]), "=&v"(a4[3]) : "v"(lds_addr(sAl) + 128 + 16 * g) : "memory");
            __builtin_amdgcn_sched_barrier(0);
            float hm[4];
            hm[0] = fmaxf(fmaxf(__uint_as_float(a4[0].x), __uint_as_float(a4[1].x)), fmaxf(__uint_as_float(a4[2].x), __uint_as_float(a4[3].x)));
            hm[1] = fmaxf(fmaxf(__uint_as_float(a4[0].y), __uint_as_float(a4[1].y)), fmaxf(__uint_as_float(a4[2].y), __uint_as_float(a4[3].y)));
            hm[2] = fmaxf(fmaxf(__uint_as_float(a4[0].z), __uint_as_float(a4[1].z)), fmaxf(__uint_as_float(a4[2].z), __uint_as_float(a4[3].z)));
            hm[3] = fmaxf(fmaxf(__uint_as_float(a4[0].w), __uint_as_float(a4[1].w)), fmaxf(__uint_as_float(a4[2].w), __uint_as_float(a4[3].w)));
#pragma unroll
            for (int r = 0; r < 4; ++r) qs_head[r] = fmaxf(hm[r], 1e-30f) * (1.0f / 448.0f);     // scale of head 4 g + r
            // this lane's own head is l15: its amax sits with lane group l15 >> 2, register l15 & 3
            float mine = hm[0];
            {
                const int src = ((l15 >> 2) << 4) | l15;      // a lane of group l15 >> 2 (any l15 there)
                const float h0 = __shfl(hm[0], src, 64), h1 = __shfl(hm[1], src, 64), h2 = __shfl(hm[2], src, 64), h3 = __shfl(hm[3], src, 64);
                const int rr = l15 & 3;
                mine = rr == 0 ? h0 : rr == 1 ? h1 : rr == 2 ? h2 : h3;
            }
            const float inv = 448.0f / fmaxf(mine, 1e-30f);
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                const unsigned lo = pack4_fp8((float)qb[s][0] * inv, (float)qb[s][1] * inv, (float)qb[s][2] * inv, (float)qb[s][3] * inv);
                const unsigned hi = pack4_fp8((float)qb[s][4] * inv, (float)qb[s][5] * inv, (float)qb[s][6] * inv, (float)qb[s][7] * inv);
                q8[s] = ((unsigned long long)hi << 32) | lo;
            }
        }
        // the score of (head h, key) is S8 * qs[h] * sx: this wave's softmax heads are {4 g + wave}
        const float sc = (wave == 0 ? qs_head[0] : wave == 1 ? qs_head[1] : wave == 2 ? qs_head[2] : qs_head[3]) * P_sx;
        f32x4 cacc[12];
#pragma unroll
        for (int dt = 0; dt < 12; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) cacc[dt][r] = 0.f;
        float m_run = -INFINITY, l_run = 0.f;

        // Two key tiles (64 keys) per iteration: the three block-wide barriers and the LDS round trips between the phases
        // (partial scores, probabilities) are paid once per 64 keys.  With 32 keys per iteration the kernel was bound
        // by exactly that chain (r02: 2.25 us per 24 KiB tile = 3.05 TB/s, only 1.2x the bf16 kernel's rate for half
        // the bytes; 1.72 us per tile now).  Also built and measured: the software-pipelined single-tile loop (P.X of
        // tile t beside the scores of tile t+1, two barriers per tile): 2.15 us per tile - slower than pairing; the two
        // cannot be combined in a five-slot ring (four tiles resident + two in flight).  A row's odd last tile runs with the ring's next slot as its partner: those keys lie beyond L, get
        // probability 0, and the slot holds e4m3 bytes of SOME tile (finite), so it is only read, not consumed.
        for (int t = 0; t < cnt; t += 2) {
            const bool pair = t + 1 < cnt;
            const int slot2 = slot + 1 == LAT8_NST ? 0 : slot + 1;
            if (wave < 3) {
                const int ws = pair ? slot2 : slot;        // the younger of the tiles this iteration consumes
                const int mk = ws == 0 ? mk0 : ws == 1 ? mk1 : ws == 2 ? mk2 : ws == 3 ? mk3 : mk4;
                wait_vm_newer(issued - mk);
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            ISSUE_NEXT8();                        // refill what the previous iteration consumed: every wave is past its reads
            const bool second = t == 0 ? prev_pair : true;
            is_np = 0;
            if (second) ISSUE_BEGIN8();
            const unsigned xt_a = smem_base + (unsigned)(slot * LAT8_TILE_BYTES), xt_b = smem_base + (unsigned)(slot2 * LAT8_TILE_BYTES);
            slot = pair ? (slot2 + 1 == LAT8_NST ? 0 : slot2 + 1) : slot2;
            // ---- partial scores over this wave's 192 dims: four 16-key sub-tiles
            f32x4 sacc[4];
            {
                unsigned sa[12];
                unsigned long long xa[12], xb[12];
#pragma unroll
                for (int k = 0; k < 12; ++k) sa[k] = xt_a + s_off[k];
                lds_read12_b64(xa, sa);
#pragma unroll
                for (int k = 0; k < 12; ++k) sa[k] = xt_b + s_off[k];
                lds_read12_b64(xb, sa);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sacc[j][r] = 0.f;
#pragma unroll
                for (int s = 0; s < 6; ++s) {       // four independent accumulation chains interleaved
                    sacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8((long)q8[s], (long)xa[s], sacc[0], 0, 0, 0);
                    sacc[1] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8((long)q8[s], (long)xa[6 + s], sacc[1], 0, 0, 0);
                    sacc[2] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8((long)q8[s], (long)xb[s], sacc[2], 0, 0, 0);
                    sacc[3] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8((long)q8[s], (long)xb[6 + s], sacc[3], 0, 0, 0);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) sS[(wave * 16 + 4 * g + r) * 64 + 16 * j + l15] = sacc[j][r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            ISSUE_PART8(0, 4);
            float v[4];
            {
                const int o = (4 * g + wave) * 64 + l15;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[j] = ((sS[o + 16 * j] + sS[1024 + o + 16 * j]) + (sS[2048 + o + 16 * j] + sS[3072 + o + 16 * j])) * sc;
                    if (t * LAT_TK + 16 * j + l15 >= L) v[j] = -INFINITY;
                }
            }
            {
                float mx = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
                mx = row16_max(mx);
                const float mn = mx > m_run ? mx : m_run;          // finite: the first tile of the pair has a valid key
                const float al = __expf(m_run - mn);
                const float p0 = __expf(v[0] - mn), p1 = __expf(v[1] - mn), p2 = __expf(v[2] - mn), p3 = __expf(v[3] - mn);
                // e4m3(256 p): one byte each (v_cvt_pk_fp8_f32 rounds to nearest even).  The row sum is taken over the
                // QUANTISED weights, so the weights that multiply X still sum to exactly 1 after the division by l.
                const unsigned pk = pack4_fp8(p0 * LAT8_PSCALE, p1 * LAT8_PSCALE, p2 * LAT8_PSCALE, p3 * LAT8_PSCALE);
                const float pq = ((__builtin_amdgcn_cvt_f32_fp8((int)pk, 0) + __builtin_amdgcn_cvt_f32_fp8((int)pk, 1)) +
                                  (__builtin_amdgcn_cvt_f32_fp8((int)pk, 2) + __builtin_amdgcn_cvt_f32_fp8((int)pk, 3))) * (1.0f / LAT8_PSCALE);
                l_run = l_run * al + row16_sum(pq);
                m_run = mn;
                uint8_t* pw = sP + (4 * g + wave) * 64 + l15;
                pw[0] = (uint8_t)(pk & 0xff);
                pw[16] = (uint8_t)((pk >> 8) & 0xff);
                pw[32] = (uint8_t)((pk >> 16) & 0xff);
                pw[48] = (uint8_t)(pk >> 24);
                if (l15 == 0) sAl[4 * g + wave] = al;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            ISSUE_PART8(4, 8);
            ISSUE_END8();
            uint4 al4;
            unsigned long long pa, pb;       // A operands: P8[head = lane & 15][key = 8 g .. 8 g + 7] of the two tiles
            asm volatile("ds_read_b128 %0, %3\n\tds_read_b64 %1, %4\n\tds_read_b64 %2, %4 offset:32\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(al4), "=&v"(pa), "=&v"(pb) : "v"(lds_addr(sAl) + 16 * g), "v"(lds_addr(sP) + l15 * 64 + 8 * g) : "memory");
            __builtin_amdgcn_sched_barrier(0);
            {
                const float a0 = __uint_as_float(al4.x), a1 = __uint_as_float(al4.y), a2 = __uint_as_float(al4.z), a3 = __uint_as_float(al4.w);
                if (__any((a0 != 1.0f) | (a1 != 1.0f) | (a2 != 1.0f) | (a3 != 1.0f))) {
#pragma unroll
                    for (int dt = 0; dt < 12; ++dt) {
                        cacc[dt][0] *= a0; cacc[dt][1] *= a1; cacc[dt][2] *= a2; cacc[dt][3] *= a3;
                    }
                }
            }
            // ---- C[head][d] += P8[head][key] X8[key][d] over this wave's 192 columns: one transposed read per d-tile and key tile
            {
                unsigned ad[12];
                unsigned long long xr[12], xq[12];
#pragma unroll
                for (int k = 0; k < 12; ++k) ad[k] = xt_a + tr_off[k];
                tr8_read12(xr, ad);
#pragma unroll
                for (int k = 0; k < 12; ++k) ad[k] = xt_b + tr_off[k];
                tr8_read12(xq, ad);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 12; ++k) cacc[k] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8((long)pa, (long)xr[k], cacc[k], 0, 0, 0);
#pragma unroll
                for (int k = 0; k < 12; ++k) cacc[k] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8((long)pb, (long)xq[k], cacc[k], 0, 0, 0);
            }
            prev_pair = pair;
        }
        // ---- finish the row: normalise (1 / l, the probability scale 256 and the key scale sx), stage, store
        if (l15 == 0) sAl[16 + 4 * g + wave] = l_run;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        float inv[4];
        {
            uint4 l4;
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(l4) : "v"(lds_addr(sAl) + 64 + 16 * g) : "memory");
            __builtin_amdgcn_sched_barrier(0);
            const float k = P_sx * (1.0f / LAT8_PSCALE);
            inv[0] = k / __uint_as_float(l4.x); inv[1] = k / __uint_as_float(l4.y);
            inv[2] = k / __uint_as_float(l4.z); inv[3] = k / __uint_as_float(l4.w);
        }
        // (staging area of its own; wave 3 has read the previous row out of it before it reaches this row's barriers)
        if (g < 3) {
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
        // ---- the next row's query: prove the prefetch landed, then copy.  With >= LAT8_NST tiles per row a tile wait
        // behind the prefetch already proved it; shorter rows wait here.
        if (cnt < LAT8_NST && wave < 3) wait_vm_newer(issued - mkq);
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
#undef Q_PTR
#undef X8_SLOT
}

// bf16 rows -> e4m3 rows with one static scale (x8 = e4m3(x * inv_sx)): the encoder output of a batch, once per batch.
__global__ __launch_bounds__(256) void quant_rows_fp8_kernel(const bf16_t* __restrict__ x, uint8_t* __restrict__ x8, long long n16, float inv_sx) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;          // 16 elements per thread
    if (i >= n16) return;
    float v[16];
    elem<bf16_t>::ld8(x + i * 16, v);
    elem<bf16_t>::ld8(x + i * 16 + 8, v + 8);
    uint4 o;
    o.x = pack4_fp8(v[0] * inv_sx, v[1] * inv_sx, v[2] * inv_sx, v[3] * inv_sx);
    o.y = pack4_fp8(v[4] * inv_sx, v[5] * inv_sx, v[6] * inv_sx, v[7] * inv_sx);
    o.z = pack4_fp8(v[8] * inv_sx, v[9] * inv_sx, v[10] * inv_sx, v[11] * inv_sx);
    o.w = pack4_fp8(v[12] * inv_sx, v[13] * inv_sx, v[14] * inv_sx, v[15] * inv_sx);
    *reinterpret_cast<uint4*>(x8 + i * 16) = o;
}
