// Row-wise kernels of the greedy decode step (BERT decoder with post-LN residual blocks,
// TF/models/bert/modeling_bert.py:282-293, 340-351, 466-496; greedy loop TF/generation/utils.py:
// 2783-2973).  The skinny projections (M = batch) run as split-K GEMMs that leave fp32 partial
// slabs; every consumer below sums the slabs in a fixed order (deterministic, unlike float
// atomics), adds the bias and fuses the element-wise work that follows.
#pragma once
#include "common.h"

// out = LayerNorm( [gelu]( sum_s slab_s + bias ) [+ resid] ) ; writes fp32 (next residual) and T
// (next GEMM operand).  One block of D/4 threads per row, one float4 per thread; the slab loads
// are issued four at a time (independent accumulators) so the row costs ~nslab/4 memory round
// trips instead of nslab.
template <typename T, int D, bool GELU>
__global__ __launch_bounds__(D / 4) void dec_add_ln_kernel(const float* __restrict__ slabs, int nslab, long long slab_stride,
                                                           const float* __restrict__ bias, const float* __restrict__ resid,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* __restrict__ out_f32, T* __restrict__ out_t, int rows, float eps,
                                                           T* __restrict__ cache = nullptr, long long cache_batch_stride = 0,
                                                           const int* __restrict__ step = nullptr, uint8_t* __restrict__ cache8 = nullptr,
                                                           float inv_sx8 = 0.f, const int* __restrict__ rowmap = nullptr) {
    constexpr int NW = D / 256;                      // waves per block
    __shared__ float s_red[2][NW];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = tid * 4;
    const float* sp = slabs + (size_t)row * D + c;
    const float4 bv = *reinterpret_cast<const float4*>(bias + c);
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (resid) r = *reinterpret_cast<const float4*>(resid + (size_t)row * D + c);
    // eight slab loads in flight at a time; indices past nslab are clamped (a valid, cached
    // address) and weighted 0 - a per-load branch would serialise the loads
    float v[4] = {bv.x, bv.y, bv.z, bv.w};
    for (int s0 = 0; s0 < nslab; s0 += 8) {
        float4 x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = *reinterpret_cast<const float4*>(sp + (size_t)min(s0 + u, nslab - 1) * slab_stride);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float w = (s0 + u < nslab) ? 1.f : 0.f;
            v[0] += w * x[u].x; v[1] += w * x[u].y; v[2] += w * x[u].z; v[3] += w * x[u].w;
        }
    }
    if (GELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_for<T>(v[e]);
    }
    v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
    float sm = wave_sum((v[0] + v[1]) + (v[2] + v[3]));
    if (lane == 0) s_red[0][wave] = sm;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) tot += s_red[0][w];
    const float mean = tot * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float d = v[e] - mean; q += d * d; }
    q = wave_sum(q);
    if (lane == 0) s_red[1][wave] = q;
    __syncthreads();
    tot = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) tot += s_red[1][w];
    const float rstd = 1.0f / sqrtf(tot * (1.0f / D) + eps);
    const float4 g = *reinterpret_cast<const float4*>(gamma + c);
    const float4 b = *reinterpret_cast<const float4*>(beta + c);
    float o[4] = {(v[0] - mean) * rstd * g.x + b.x, (v[1] - mean) * rstd * g.y + b.y, (v[2] - mean) * rstd * g.z + b.z,
                  (v[3] - mean) * rstd * g.w + b.w};
    if (out_f32) *reinterpret_cast<float4*>(out_f32 + (size_t)row * D + c) = make_float4(o[0], o[1], o[2], o[3]);
    elem<T>::st4(out_t + (size_t)row * D + c, o);
    // latent attention: this row is also the key/value source of position step[row] for the next layer
    const int crow = rowmap ? rowmap[row] : row;     // the per-row caches stay where the row started (compacted batches)
    if (cache) elem<T>::st4(cache + (size_t)crow * cache_batch_stride + (size_t)step[row] * D + c, o);
    // fp8 attention mode: the same row as e4m3 bytes with the cache's static scale (kernels_latent8.h)
    if (cache8)
        *reinterpret_cast<unsigned*>(cache8 + (size_t)crow * cache_batch_stride + (size_t)step[row] * D + c) =
            pack4_fp8(o[0] * inv_sx8, o[1] * inv_sx8, o[2] * inv_sx8, o[3] * inv_sx8);
}

// out T = gelu( sum_s slab_s + bias )   (the decoder FFN's intermediate activation)
template <typename T>
__global__ void dec_bias_gelu_kernel(const float* __restrict__ slabs, int nslab, long long slab_stride,
                                     const float* __restrict__ bias, T* __restrict__ out, int rows, int N) {
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= (long long)rows * N) return;
    const int c = (int)(i % N);
    float4 a0 = *reinterpret_cast<const float4*>(bias + c);
    float4 a1 = make_float4(0.f, 0.f, 0.f, 0.f);
    int s = 0;
    for (; s + 2 <= nslab; s += 2) {
        const float4 x0 = *reinterpret_cast<const float4*>(slabs + (size_t)s * slab_stride + i);
        const float4 x1 = *reinterpret_cast<const float4*>(slabs + (size_t)(s + 1) * slab_stride + i);
        a0.x += x0.x; a0.y += x0.y; a0.z += x0.z; a0.w += x0.w;
        a1.x += x1.x; a1.y += x1.y; a1.z += x1.z; a1.w += x1.w;
    }
    if (s < nslab) {
        const float4 x0 = *reinterpret_cast<const float4*>(slabs + (size_t)s * slab_stride + i);
        a0.x += x0.x; a0.y += x0.y; a0.z += x0.z; a0.w += x0.w;
    }
    float o[4] = {gelu_for<T>(a0.x + a1.x), gelu_for<T>(a0.y + a1.y), gelu_for<T>(a0.z + a1.z), gelu_for<T>(a0.w + a1.w)};
    elem<T>::st4(out + i, o);
}

// Per-sequence decode state, all device resident so a decode step is a fixed launch sequence
// (replayable as a HIP graph): no host value changes from step to step.
struct DecState {
    int* ids;          // [B][max_len] int32 output rows (ids[b][0] = start_id)
    int* step;         // [B] position of the token being consumed
    int* finished;     // [B]
    int* len;          // [B] tokens in the row incl. start and EOS
    int* n_unfinished; // [1]
    const int* forced; // [B][forced_T] teacher-forced inputs (test hook) or null
    int forced_T;
    float* logits_out; // [B][forced_T][V] (test hook) or null
    int ids_ld;        // row stride of ids
    int max_len;       // generate(max_length): a row is finished when it holds max_len tokens
    int start_id, eos_id, pad_id;
    int n_real;        // rows >= n_real are padding (the batch is rounded up to a graph-friendly row count): born finished
    // Decode SLOT -> ROW of the batch (r04).  ids / finished / len and the per-row key/value caches are indexed by row and
    // never move; the activations of a step (x, slabs, ...) and `step` are indexed by slot.  Identity from the start token
    // on; compact_rows (engine.hip) rewrites it so that the unfinished rows occupy the first slots and the decode steps
    // can run on fewer of them - a finished row stops costing, as in the reference, where every crop is its own
    // generate() call (TF/generation/utils.py:2929-2937, src/core/workers.py:213-225).
    int* rowmap;
};

// End of a decode step, one block per sequence:
//   logits = sum_s slab_s + bias  -> fp32 argmax (lowest index wins ties, like torch.argmax)
//   finished rows emit pad_id (utils.py:2929); EOS or reaching max_len finishes a row (:2936)
//   then the NEXT step's input embedding: LN(word[tok] + type[0] + pos[t+1])  (modeling_bert.py:98-106)
// FIRST = true: no logits yet; emits the start token's embedding at position 0.
template <typename T, int D, bool FIRST>
__global__ __launch_bounds__(256) void dec_token_kernel(const float* __restrict__ slabs, int nslab, long long slab_stride,
                                                        const float* __restrict__ vbias, int V,
                                                        DecState st,
                                                        const float* __restrict__ word, const float* __restrict__ type0,
                                                        const float* __restrict__ pos, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ x_f32,
                                                        T* __restrict__ x_t, float eps, T* __restrict__ cache = nullptr,
                                                        long long cache_batch_stride = 0,
                                                        const float* __restrict__ cand_val = nullptr,
                                                        const int* __restrict__ cand_idx = nullptr, int ncand = 0,
                                                        uint8_t* __restrict__ cache8 = nullptr, float inv_sx8 = 0.f) {
    __shared__ float s_val[4];
    __shared__ int s_idx[4];
    __shared__ float s_red[4];
    __shared__ int s_tok, s_pos;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = FIRST ? b : st.rowmap[b];       // slot b decodes row `row` of the batch
    if (FIRST) {
        if (tid == 0) {
            st.rowmap[b] = b;
            st.ids[(size_t)b * st.ids_ld] = st.start_id;
            st.step[b] = 0; st.finished[b] = b >= st.n_real ? 1 : 0; st.len[b] = st.max_len;
            s_tok = st.start_id; s_pos = 0;
            if (b == 0) *st.n_unfinished = st.n_real;
        }
    } else {
        const int t = st.step[b];
        float best = -INFINITY;
        int bi = 0x7fffffff;
        if (cand_val) {
            // the LM-head GEMM already reduced every 64/128-column tile (EPI_ARGMAX): ncand candidates per row
            for (int c = tid; c < ncand; c += 256) {
                const float cv = cand_val[(size_t)b * ncand + c];
                const int ci = cand_idx[(size_t)b * ncand + c];
                if (cv > best || (cv == best && ci < bi)) { best = cv; bi = ci; }
            }
        } else {
        constexpr int NC = 6;                            // vocab = NC * 1024 columns (6144)
        float4 a[NC];
#pragma unroll
        for (int j = 0; j < NC; ++j) a[j] = *reinterpret_cast<const float4*>(vbias + tid * 4 + j * 1024);
        for (int s = 0; s < nslab; ++s) {
            const float* sp = slabs + (size_t)s * slab_stride + (size_t)b * V + tid * 4;
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                const float4 x = *reinterpret_cast<const float4*>(sp + j * 1024);
                a[j].x += x.x; a[j].y += x.y; a[j].z += x.z; a[j].w += x.w;
            }
        }
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const int c = tid * 4 + j * 1024;
            if (st.logits_out)
                *reinterpret_cast<float4*>(st.logits_out + ((size_t)b * st.forced_T + t) * V + c) = a[j];
            if (a[j].x > best) { best = a[j].x; bi = c; }
            if (a[j].y > best) { best = a[j].y; bi = c + 1; }
            if (a[j].z > best) { best = a[j].z; bi = c + 2; }
            if (a[j].w > best) { best = a[j].w; bi = c + 3; }
        }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if (lane == 0) { s_val[wave] = best; s_idx[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < 4; ++w)
                if (s_val[w] > best || (s_val[w] == best && s_idx[w] < bi)) { best = s_val[w]; bi = s_idx[w]; }
            if ((unsigned)bi >= (unsigned)V) bi = 0;      // all-NaN logits win no comparison: stay inside the embedding table
            int fin = st.finished[row];
            int tok = fin ? st.pad_id : bi;
            if (st.forced) tok = (t + 1 < st.forced_T) ? st.forced[(size_t)b * st.forced_T + t + 1] : st.pad_id;
            if (t + 1 < st.ids_ld) st.ids[(size_t)row * st.ids_ld + t + 1] = tok;
            if (!fin && !st.forced && (tok == st.eos_id || t + 2 >= st.max_len)) {
                st.finished[row] = 1;
                st.len[row] = t + 2;
                atomicSub(st.n_unfinished, 1);
            }
            st.step[b] = t + 1;
            s_tok = tok; s_pos = t + 1;
        }
    }
    __syncthreads();
    const int tok = s_tok, ps = s_pos;
    // embedding + LayerNorm of the next input, block-wide over D = 768 (3 per thread)
    constexpr int PER = D / 256;
    float v[PER];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int d = tid + i * 256;
        float e = word[(size_t)tok * D + d] + type0[d];
        e = e + pos[(size_t)ps * D + d];
        v[i] = e; s += e;
    }
    s = wave_sum(s);
    if (lane == 0) s_red[wave] = s;
    __syncthreads();
    const float mean = (s_red[0] + s_red[1] + s_red[2] + s_red[3]) * (1.0f / D);
    __syncthreads();
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) { const float d = v[i] - mean; q += d * d; }
    q = wave_sum(q);
    if (lane == 0) s_red[wave] = q;
    __syncthreads();
    const float rstd = 1.0f / sqrtf((s_red[0] + s_red[1] + s_red[2] + s_red[3]) * (1.0f / D) + eps);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int d = tid + i * 256;
        const float o = (v[i] - mean) * rstd * gamma[d] + beta[d];
        x_f32[(size_t)b * D + d] = o;
        elem<T>::st(x_t + (size_t)b * D + d, o);
        if (cache) elem<T>::st(cache + (size_t)row * cache_batch_stride + (size_t)ps * D + d, o);   // layer-0 key/value source
        if (cache8) cache8[(size_t)row * cache_batch_stride + (size_t)ps * D + d] = (uint8_t)(pack4_fp8(o * inv_sx8, 0.f, 0.f, 0.f) & 0xff);
    }
}

// ------------------------------------------------------------------------------------------------
// Row compaction (r04): finished rows stop costing.
// In the reference every crop is its own generate() call and stops at its own EOS (TF/generation/utils.py:2929-2937,
// called per crop at src/ui/main_window.py:9801).  A merged batch decodes in lockstep; between two chunks of steps the
// scheduler (engine.hip: compact_rows) moves the unfinished rows to the first slots and lets the next chunk run on fewer
// slots.  Only the step's input rows (x_f32 / x_t) move: ids, finished, len and every key/value cache are indexed by ROW
// through DecState::rowmap and stay where they are.
//
// compact_plan_kernel, ONE block: stable partition of the np slots by "row unfinished" -> new_map[s'] = row of new slot
// s', src_slot[s'] = the slot it comes from.  The finished rows fill the slots behind the live ones (any slot the next
// chunks still run - the row count is rounded up to a graph-friendly grid - then holds a finished row: it emits pad_id
// into that row's tail, which is what an uncompacted batch does to it).
__global__ __launch_bounds__(1024) void compact_plan_kernel(const int* __restrict__ rowmap, const int* __restrict__ finished, int np,
                                                            int* __restrict__ new_map, int* __restrict__ src_slot) {
    __shared__ int s_cnt[1024];
    const int tid = threadIdx.x;
    const int per = (np + 1023) / 1024, s0 = tid * per, s1 = min(np, s0 + per);
    int live = 0;
    for (int s = s0; s < s1; ++s) live += finished[rowmap[s]] ? 0 : 1;
    s_cnt[tid] = live;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {          // inclusive scan
        const int v = tid >= off ? s_cnt[tid - off] : 0;
        __syncthreads();
        s_cnt[tid] += v;
        __syncthreads();
    }
    const int n_live = s_cnt[1023];
    int pl = s_cnt[tid] - live;                          // live slots in front of s0
    for (int s = s0; s < s1; ++s) {
        const int row = rowmap[s];
        const bool lv = !finished[row];
        const int dst = lv ? pl : n_live + (s - pl);
        new_map[dst] = row; src_slot[dst] = s;
        pl += lv ? 1 : 0;
    }
}

// phase 0: tmp[s'] = x[src_slot[s']] for the np_new slots that stay; phase 1: x[s'] = tmp[s'], rowmap = new_map (np_old entries).
// One block of D/4 threads per slot.
template <typename T, int D>
__global__ __launch_bounds__(D / 4) void compact_move_kernel(int phase, const int* __restrict__ new_map, const int* __restrict__ src_slot,
                                                              float* __restrict__ x_f32, T* __restrict__ x_t, float* __restrict__ tmp_f32,
                                                              T* __restrict__ tmp_t, int* __restrict__ rowmap, int np_new, int np_old) {
    const int s = blockIdx.x, c = threadIdx.x * 4;
    if (phase == 0) {
        const int src = src_slot[s];
        *reinterpret_cast<float4*>(tmp_f32 + (size_t)s * D + c) = *reinterpret_cast<const float4*>(x_f32 + (size_t)src * D + c);
        float v[4];
        elem<T>::ld4(x_t + (size_t)src * D + c, v);
        elem<T>::st4(tmp_t + (size_t)s * D + c, v);
    } else {
        *reinterpret_cast<float4*>(x_f32 + (size_t)s * D + c) = *reinterpret_cast<const float4*>(tmp_f32 + (size_t)s * D + c);
        float v[4];
        elem<T>::ld4(tmp_t + (size_t)s * D + c, v);
        elem<T>::st4(x_t + (size_t)s * D + c, v);
        for (int i = s * (D / 4) + threadIdx.x; i < np_old; i += gridDim.x * (D / 4)) rowmap[i] = new_map[i];
    }
}
