// libmocr_hip.so - engine and C ABI (include/mocr.h) of the MI355X-native Manga-OCR recogniser.
//
// Data flow of one batch of n crops (M = n * 197 encoder rows), all buffers resident in HBM:
//
//   gray u8 [n,224,224] --patchify--> Ape T[n*196,256] --GEMM+bias+pos--> X f32 [M,768]   (CLS rows apart)
//   12 x { LN(X)->Xn T ; QKV GEMM ->QKV T[M,2304] ; attention -> CTX T[M,768] ; O GEMM + resid -> X ;
//          LN(X)->Xn ; FC1 GEMM+GELU -> Hb T[M,3072] ; FC2 GEMM + resid -> X }
//   LN(X) -> ENC T[M,768] ; cross-K/V GEMM (both decoder layers at once) -> CKV T[M, 2*2*768]
//   greedy loop, <= max_len-1 steps over n rows: see decode_step().
//
// The residual stream X is fp32 in both modes; T is the storage type of GEMM operands
// (bf16 or fp32).  Decode-step projections (M = n) are split-K GEMMs writing fp32 partial slabs.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <exception>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

#include "../../include/mocr.h"
#include "common.h"
#include "kernels_attn.h"
#include "kernels_decode.h"
#include "kernels_gemm.h"
#include "kernels_gemm_pers.h"
#ifdef MOCR_EXPERIMENTS
#include "kernels_gemm_lab.h"      // A/B kernels of earlier rounds: not in the product library
#endif
#include "kernels_latent.h"
#include "kernels_latent8.h"
#include "kernels_latent_t.h"
#include "kernels_latent_t8.h"
#include "kernels_smallm.h"
#include "preprocess.h"
#include "prep_pipeline.h"
#include "kernels_qqt.h"
#include "kernels_misc.h"

namespace {

struct HipError { hipError_t code; const char* what; int line; };
#define HIPCHECK(x)                                             \
    do {                                                        \
        hipError_t err__ = (x);                                 \
        if (err__ != hipSuccess) throw HipError{err__, #x, __LINE__}; \
    } while (0)
struct ArgError { std::string msg; int code; };

static inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

static inline uint16_t host_f2bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

static inline float host_bf2f(uint16_t h) {
    const uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

struct EncLayerW {
    void *wqkv, *wo, *w1, *w2;
    float *bqkv, *bo, *b1, *b2, *ln1g, *ln1b, *ln2g, *ln2b;
    // bf16 engines, LayerNorm folded into the GEMM behind it (gemm_pers_kernel LNF): W o gamma, its column sums (of the
    // bf16-rounded values), b + W beta
    void *wqkv_f = nullptr, *w1_f = nullptr;
    float *sqkv = nullptr, *s1 = nullptr, *bqkv_f = nullptr, *b1_f = nullptr;
};
struct DecLayerW {
    void *wqkv, *wo, *wqc, *woc, *w1, *w2;
    void *wkT_s = nullptr, *wkT_c = nullptr;   // latent attention: (Wk^T)/8 of the self / cross attention, [768 in][768 out]
    float *bqkv, *bo, *bqc, *boc, *b1, *b2, *ln1g, *ln1b, *ln2g, *ln2b, *ln3g, *ln3b;
};
struct Weights {
    void* wpe = nullptr; float *bpe = nullptr, *cls = nullptr, *pos_enc = nullptr;
    std::vector<EncLayerW> enc;
    float *lnfg = nullptr, *lnfb = nullptr;
    void* wckv = nullptr; float* bckv = nullptr;
    float *word = nullptr, *posd = nullptr, *type0 = nullptr, *embg = nullptr, *embb = nullptr;
    std::vector<DecLayerW> dec;
    void* wt = nullptr; float *bt = nullptr, *lntg = nullptr, *lntb = nullptr;
    void* wv = nullptr; float* bv = nullptr;
    float* lut = nullptr;
    float* zero_bias = nullptr;
    // fp8 attention (MOCR_FLAG_FP8_ATTENTION): static e4m3 scales of the key/value sources (x = x8 * sx)
    float sx_enc = 1.f;                 // encoder output = LayerNorm_f
    std::vector<float> sx_self;         // per decoder layer: its input rows (embedding LayerNorm / previous layer's LayerNorm 3)
};

struct ProfRec { int kid; hipEvent_t e0, e1; double flops, bytes; };

}  // namespace

// Everything one in-flight batch needs: a HIP stream and its own workspace.  The engine keeps
// `lanes` of them (the counterpart of the reference's pool of QueueProcessorWorker threads,
// src/ui/main_window.py:4286-4327): independent batches overlap on the GPU, which is what fills
// the chip during the latency-bound decode steps.  Host code runs single-threaded under the
// engine mutex and "binds" one lane at a time: mocr_engine derives from LaneCtx, and bind()
// copies the lane's pointers into that base, so the launch code simply says e->X, e->stream.
struct LaneCtx {
    int lane_id = 0;
    hipStream_t stream = nullptr;
    uint8_t *d_in = nullptr, *d_rgb = nullptr;
    float* X = nullptr;
    void *Xn = nullptr, *QKV = nullptr, *CTX = nullptr, *Hb = nullptr, *ENC = nullptr, *CKV = nullptr;
    float* ln_part = nullptr;                  // [Mp][4][2] row statistics partials of X (LayerNorm folded into the encoder GEMMs)
    void *kcache = nullptr, *vcache = nullptr;     // [dec_layers][Bp][H][max_len][64]
    float* slabs = nullptr; long long slab_cap = 0; // floats
    float* cand_val = nullptr; int* cand_idx = nullptr;   // [Bp][vocab/64] per-tile argmax candidates of the LM head
    float *x_f32 = nullptr, *a_f32 = nullptr, *c_f32 = nullptr;
    float* ln_stats = nullptr;                      // small-batch path: (mean, rstd) per row of the three pre-LayerNorm sums, [3][Bp][2]
    void *x_t = nullptr, *a_t = nullptr, *c_t = nullptr, *ctx_t = nullptr, *h_t = nullptr, *z_t = nullptr;
    int *ids = nullptr, *step = nullptr, *finished = nullptr, *len = nullptr, *n_unf = nullptr;
    int* forced = nullptr; float* logits_dbg = nullptr; size_t forced_cap = 0, logits_cap = 0;
    int* h_pinned = nullptr;                        // [4] pinned: early-exit flags
    // latent attention (bf16): q [Bp,768], Qt / Et [Bp,16,768], per-layer input rows [layers][Bp][max_len][768]
    void *q_t = nullptr, *qt = nullptr, *et = nullptr, *xcache = nullptr;
    int* rowmap = nullptr;                          // [Bp] decode slot -> row of the batch (identity until the batch is compacted)
    int* rowmap_tmp = nullptr;                      // [2][Bp] compaction scratch: the new map, and every new slot's old slot
    // fp8 attention: e4m3 copies of the encoder output [Mp][768] and of the per-layer input rows [layers][Bp][max_len][768]
    uint8_t *enc8 = nullptr, *x8cache = nullptr;
};

// One recognise request of <= max_batch crops.
struct Job {
    hipEvent_t wait_ev = nullptr;   // device planes that a preparation stream is still writing: the lane's stream waits for this first
    const uint8_t* src = nullptr;   // host images (src_host) or device luminance planes
    bool src_host = false;
    int channels = 1;
    int64_t row_stride = 0, image_stride = 0;
    int n = 0, max_len = 0;
    int32_t* out_ids = nullptr;     // host (out_host) or device
    int32_t* out_len = nullptr;
    bool out_host = false;
};

struct Lane {
    LaneCtx ctx;
    bool active = false;
    std::vector<Job> jobs;          // requests merged into this lane's current batch, in row order
    int n = 0, max_len = 0;         // rows of the merged batch, its generate(max_length)
    int np = 0;                     // slots the decode steps run on: n rounded up (graph_rows), the extra ones are born finished; shrinks when the batch is compacted
    int np0 = 0;                    // np at the start of the batch = its kernel regime
    int t = 0, steps = 0, chunk = 0;
    bool finishing = false;         // a flag of this batch has reported a finished row: rows are leaving, chunks get shorter
    bool flag_pending[2] = {false, false};
    hipEvent_t flag_ev[2] = {nullptr, nullptr};
};

struct mocr_engine : LaneCtx {
    mocr_config cfg{};
    std::string err;
    std::mutex mu;
    std::mutex err_mu;              // guards err: mocr_last_error may be called while another thread fails
    std::atomic<bool> poisoned{false};   // a HIP call failed: HIP errors are sticky, so every later call is refused (read without the mutex)
    bool committed = false;
    int gen_max_len = 0;            // generate(max_length) of the device-buffer submissions (mocr_set_generate_max_length)
    bool fp8attn = false;           // latent attention on e4m3 key/value rows + fp8 MFMA (MOCR_FLAG_FP8_ATTENTION, opt-in)
    bool latent = false;            // bf16 engines: latent (absorbed) decode attention ...
    int classic_rows = 0;           // ... for batches of more than this many rows; smaller ones use the classic kernels
    // LayerNorm folding (bf16, persistent encoder GEMMs) is gated by a measurement on THIS checkpoint (calibrate_ln_fold):
    // the fold feeds bf16(x) - the raw residual stream - into the matrix cores where the launches feed bf16(LN(x)); per row
    // the input-rounding noise of the two forms is in the ratio sqrt(sum (x g rstd)^2 / sum (LN(x))^2), ~1 on a
    // near-normalised stream and |mean| / spread on a stream with a DC offset.
    struct FoldCalib { std::vector<std::vector<float>> g, b; size_t idx = 0; double worst = 0.0; std::vector<float> hx; };
    FoldCalib* calib = nullptr;     // non-null only inside calibrate_ln_fold
    bool fold_ok = true;            // the measured ratio allows the fold (<= FOLD_RATIO_MAX on every LayerNorm input)
    float fold_ratio = 0.f;         // the worst ratio measured (0: not measured)
    long long n_slot_steps = 0;     // decode slots x steps enqueued so far (mocr_decode_slot_steps): what the steps cost, in rows
    long long n_compactions = 0;    // batches whose rows were compacted, counted per compaction (mocr_compaction_count)
    int lat_tk = 18;                // bf16 latent attention kernel: 18 = latent_attnT_kernel, three blocks per CU (default); 32 = latent_attn_kernel on 32-key tiles; experiments: 17, 16
    int Bc = 0;                     // rows the classic K/V buffers are sized for
    // Kernel regime of the batch being decoded: the row count the batch STARTED with (0: the launch's own row count).
    // Every choice a decode step makes by row count - attention path, GEMM tile, split-K slabs, fused query kernel,
    // cache policy - is made by rrows(n), so a batch whose rows were compacted (r04: fewer slots per step as rows finish)
    // keeps the summation order it started with and a row's ids do not depend on when its neighbours finished.
    int regime = 0;
    int rrows(int n) const { return regime > 0 ? regime : n; }
    bool use_latent(int n) const { return latent && n > classic_rows; }
    int smallm_rows = 0;            // bf16: batches of up to this many rows take the one-launch-per-projection path (kernels_smallm.h)
    bool use_smallm(int n) const { return n <= smallm_rows && !use_latent(n); }
    std::map<std::string, std::vector<float>> host_w;
    std::map<std::string, std::vector<int64_t>> host_shape;
    std::vector<void*> allocs;
    Weights w;
    // geometry
    int S = 0, G = 0, D = 0, H = 0, F = 0, V = 0, Bp = 0, Mp = 0, NCKV = 0;
    int num_cus = 256;           // compute units, rounded down to a multiple of 8 (persistent grids: equal share per XCD)
    size_t esz = 2;
    std::vector<Lane> lanes;
    std::vector<Job> pending;
    // decode-step HIP graphs, keyed by (lane, rows, max_len, steps per graph)
    std::map<std::tuple<int, int, int, int, int>, hipGraphExec_t> graphs;      // + the regime
    void bind(int i) { static_cast<LaneCtx&>(*this) = lanes[i].ctx; }
    void unbind(int i) { lanes[i].ctx = static_cast<LaneCtx&>(*this); }
    // device preprocessing (preprocess.h): resample tables per input size, grow-only scratch
    std::map<int, ResampleTable> rs_tables;
    struct Scratch { void* p = nullptr; size_t cap = 0; };
    Scratch rs_src, rs_tmp, rs_desc, rs_coef, rs_bounds, rs_gray;
    // PINNED host staging of the packed pixel rows (grow-only): the H2D copy is one DMA at link speed.  Two buffers: the
    // host packs chunk k + 1 into one while the copy engine reads chunk k from the other (prepare_and_decode).
    Scratch rs_pin[2];
    hipStream_t prep_stream = nullptr;      // pack -> H2D -> resize of the host entry points: never a lane's stream
    void* grow_pinned(int slot, size_t bytes) {
        Scratch& sp = rs_pin[slot & 1];
        if (bytes > sp.cap) {
            if (sp.p) HIPCHECK(hipHostFree(sp.p));
            sp.p = nullptr; sp.cap = 0;
            const size_t cap = std::max<size_t>(bytes + bytes / 2, 1 << 20);
            HIPCHECK(hipHostMalloc(&sp.p, cap, hipHostMallocDefault));
            sp.cap = cap;
        }
        return sp.p;
    }
    void* grow(Scratch& s, size_t bytes) {
        if (bytes > s.cap) {
            if (s.p) HIPCHECK(hipFree(s.p));
            s.p = nullptr; s.cap = 0;
            const size_t cap = std::max<size_t>(bytes + bytes / 2, 4096);
            HIPCHECK(hipMalloc(&s.p, cap));
            s.cap = cap;
        }
        return s.p;
    }
    // profiling
    bool prof_on = false;
    std::vector<std::string> knames;
    std::vector<ProfRec> recs;
    std::vector<hipEvent_t> ev_pool;
    std::vector<mocr_kernel_stat> stats;

    template <typename X_> X_* dalloc(size_t count) {
        void* p = nullptr;
        HIPCHECK(hipMalloc(&p, std::max<size_t>(count * sizeof(X_), 256)));
        HIPCHECK(hipMemset(p, 0, std::max<size_t>(count * sizeof(X_), 256)));
        // the fill runs on the null stream, which the lanes' non-blocking streams do not wait for: a buffer
        // allocated lazily (test hooks) could otherwise be zeroed AFTER its first asynchronous upload
        HIPCHECK(hipDeviceSynchronize());
        allocs.push_back(p);
        return reinterpret_cast<X_*>(p);
    }
    int kid(const char* name) {
        for (size_t i = 0; i < knames.size(); ++i)
            if (knames[i] == name) return (int)i;
        knames.push_back(name);
        mocr_kernel_stat s{};
        snprintf(s.name, sizeof(s.name), "%s", name);
        stats.push_back(s);
        return (int)knames.size() - 1;
    }
    hipEvent_t get_event() {
        if (!ev_pool.empty()) { hipEvent_t e = ev_pool.back(); ev_pool.pop_back(); return e; }
        hipEvent_t e; HIPCHECK(hipEventCreate(&e)); return e;
    }
    void prof_begin(const char* name, double flops, double bytes) {
        if (!prof_on) return;
        ProfRec r{kid(name), get_event(), get_event(), flops, bytes};
        HIPCHECK(hipEventRecord(r.e0, stream));
        recs.push_back(r);
    }
    void prof_end() {
        if (!prof_on) return;
        HIPCHECK(hipEventRecord(recs.back().e1, stream));
    }
    void prof_collect() {
        if (recs.empty()) return;
        HIPCHECK(hipDeviceSynchronize());
        for (auto& r : recs) {
            float ms = 0.f;
            HIPCHECK(hipEventElapsedTime(&ms, r.e0, r.e1));
            auto& s = stats[r.kid];
            s.launches += 1; s.total_ms += ms; s.flops += r.flops; s.bytes += r.bytes;
            ev_pool.push_back(r.e0); ev_pool.push_back(r.e1);
        }
        recs.clear();
    }
};

namespace {

struct ProfScope {
    mocr_engine* e;
    ProfScope(mocr_engine* e_, const char* name, double flops, double bytes) : e(e_) { e->prof_begin(name, flops, bytes); }
    ~ProfScope() { e->prof_end(); }
};

// Tuning knobs: the MOCR_* environment overrides exist in the experiments build only (-DMOCR_EXPERIMENTS,
// `python manga-ocr_amd/build.py --experiments`, used by tools/); the product library runs the measured defaults and
// reads no environment variable in its launch code.
#ifdef MOCR_EXPERIMENTS
static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}
#else
static constexpr int env_int(const char*, int dflt) { return dflt; }
#endif

template <typename K> void set_max_lds(K kernel, int bytes) {
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
}

// ---------------------------------------------------------------------------------------- GEMM
template <int BM> constexpr int gemm_ring() { return 2; }   // LDS ring depth per tile size

template <typename T, int BM, int BN, int EPI>
void launch_gemm_t(mocr_engine* e, const GemmParams& p0, int split, int ybatch) {
    GemmParams p = p0;
    p.ntn = p.N / BN;
    const int ntm = (p.M + BM - 1) / BM;
    p.ntm = ntm;
    constexpr int NST = gemm_ring<BM>();
    constexpr int lds = NST * (BM + BN) * 128;
    dim3 grid(ntm * p.ntn, ybatch, split);
    // Grids of at most one block per CU walk their K-tiles alone on a CU, one memory round trip per K-tile on the two-slot
    // ring.  A four-slot ring (three K-tiles in flight, 128 KiB of LDS for the 128 x 128 tile) changed nothing for the encoder
    // of a few crops (r02: QKV of one crop 16.1 vs 15.9 us - its cost is the DMA issue), but the decode step's split-K
    // projections of a batch decoding ALONE are such grids too (240 blocks x 6-24 K-tiles at 2560 rows): r04,
    // tools/r04_deep_ab.sh, isolated batches, two / four slots: 128 rows 68.0 / 66.2 ms, 256 rows 90.6 / 88.6, 320 rows 108.5 /
    // 104.7, 512 rows 128.5 / 122.4, 1024 rows 196.0 / 193.7 - and the headline's two lanes x 2560 rows 7.02-7.11 / 6.90-6.92 k
    // crops/s: two 64-KiB blocks of two lanes share a CU, a 128-KiB block does not.  So: four slots for batches below the
    // rows from which a queue is split over two lanes (1280).  Same K order, same sums: bit-identical either way.
    static const int deep_rows = env_int("MOCR_GEMM_DEEP_ROWS", 1280);
    const int ktiles = p.k_per_split / (128 / (int)sizeof(T));
    static const int deep_mult64 = env_int("MOCR_GEMM_DEEP_MULT64", 1);      // (experiments: 64 x 64 tiles, two 64-KiB rings fit a CU)
    // (the ring depth does not touch the arithmetic, so a compacted batch's tail chooses it by the rows it has LEFT, as
    // launch_qqt its rows per block: r04, tools/r04_neutral_ab.sh, mixed-lengths leg 11.93 -> 12.11 k crops/s, ids bit-identical;
    // MOCR_NEUTRAL_BY_ROWS=0: by the batch's regime)
    static const int neutral_by_rows = env_int("MOCR_NEUTRAL_BY_ROWS", 2);
    if ((neutral_by_rows ? p.M : e->rrows(p.M)) < deep_rows && ktiles >= 4 && (long long)grid.x * grid.y * grid.z <= (long long)e->num_cus * (BM == 64 ? deep_mult64 : 1)) {
        hipLaunchKernelGGL((gemm_kernel<T, BM, BN, EPI, 4>), grid, dim3(256), 4 * (BM + BN) * 128, e->stream, p);
    } else {
        hipLaunchKernelGGL((gemm_kernel<T, BM, BN, EPI, NST>), grid, dim3(256), lds, e->stream, p);
    }
    HIPCHECK(hipGetLastError());
}

template <typename T, int BM, int BN>
void launch_gemm_epi(mocr_engine* e, const GemmParams& p, int epi, int split, int ybatch = 1) {
    switch (epi) {
        case EPI_SLAB: launch_gemm_t<T, BM, BN, EPI_SLAB>(e, p, split, ybatch); break;
        case EPI_BIAS: launch_gemm_t<T, BM, BN, EPI_BIAS>(e, p, split, ybatch); break;
        case EPI_BIAS_GELU: launch_gemm_t<T, BM, BN, EPI_BIAS_GELU>(e, p, split, ybatch); break;
        case EPI_BIAS_RESID: launch_gemm_t<T, BM, BN, EPI_BIAS_RESID>(e, p, split, ybatch); break;
        case EPI_PATCH: launch_gemm_t<T, BM, BN, EPI_PATCH>(e, p, split, ybatch); break;
        case EPI_BIAS_F32: launch_gemm_t<T, BM, BN, EPI_BIAS_F32>(e, p, split, ybatch); break;
        case EPI_ARGMAX: launch_gemm_t<T, BM, BN, EPI_ARGMAX>(e, p, split, ybatch); break;
        default: throw ArgError{"unknown GEMM epilogue", MOCR_ERR_ARG};
    }
}

#ifdef MOCR_EXPERIMENTS
template <int EPI>
void launch_gemm256_t(mocr_engine* e, const GemmParams& p0) {
    GemmParams p = p0;
    p.ntn = p.N / 128;
    const int ntm = (p.M + 255) / 256;
    p.ntm = ntm;
    hipLaunchKernelGGL((gemm256_kernel<EPI>), dim3(ntm * p.ntn), dim3(256), 3 * (256 + 128) * 128, e->stream, p);
    HIPCHECK(hipGetLastError());
}

void launch_gemm256(mocr_engine* e, const GemmParams& p, int epi) {
    switch (epi) {
        case EPI_BIAS: launch_gemm256_t<EPI_BIAS>(e, p); break;
        case EPI_BIAS_GELU: launch_gemm256_t<EPI_BIAS_GELU>(e, p); break;
        case EPI_BIAS_RESID: launch_gemm256_t<EPI_BIAS_RESID>(e, p); break;
        case EPI_PATCH: launch_gemm256_t<EPI_PATCH>(e, p); break;
        case EPI_BIAS_F32: launch_gemm256_t<EPI_BIAS_F32>(e, p); break;
        default: throw ArgError{"gemm256: unsupported epilogue", MOCR_ERR_ARG};
    }
}

template <int EPI, int WN>
void launch_gemm_wide_t(mocr_engine* e, const GemmParams& p0) {
    GemmParams p = p0;
    p.ntn = p.N / (64 * WN);
    const int ntm = (p.M + 255) / 256;
    p.ntm = ntm;
    // persistent: one (256x256) or two (256x128) blocks per CU, a multiple of 8 so that every XCD gets the same count
    const int ntiles = ntm * p.ntn;
    // one tile per block; the grid is rounded up to 8 so that every XCD gets the same count (an empty block returns at once)
    const int grid = (ntiles + 7) / 8 * 8;
    hipLaunchKernelGGL((gemm_wide_kernel<EPI, WN>), dim3(grid), dim3(128 * WN), 3 * (256 + 64 * WN) * 64, e->stream, p);
    HIPCHECK(hipGetLastError());
}

template <int EPI>
void launch_gemm_wide2_t(mocr_engine* e, const GemmParams& p0) {
    GemmParams p = p0;
    p.ntn = p.N / 256;
    p.ntm = (p.M + 255) / 256;
    const int grid = (p.ntm * p.ntn + 7) / 8 * 8;
    static const int stagger_env = env_int("MOCR_GEMM_STAGGER", -1);
    p.stagger = stagger_env > 0 ? stagger_env : 0;      // measured r02: no gain (the store drain is not what a phase shift hides), default off
    p.first_round = e->num_cus;
    hipLaunchKernelGGL((gemm_wide2_kernel<EPI>), dim3(grid), dim3(512), 4 * (256 + 256) * 64, e->stream, p);
    HIPCHECK(hipGetLastError());
}

#endif  // MOCR_EXPERIMENTS

// The persistent kernel (kernels_gemm_pers.h): one block per CU walks its share of the 256 x 256 tiles.  `blocks` = 0:
// one block per CU (a multiple of 8, at most one per tile); a test may ask for fewer blocks (longer tile sequences).
// The strip schedule of the persistent kernel (kernels_gemm_pers.h, STRIP): rows per strip for a grid of `grid` blocks,
// 0 when the grid cannot hold a strip.  Host mirror of the kernel's arithmetic.
static int pers_strip_rows(int M, int ntn, int grid) {
    const int slots = grid >> 3, spx = slots / ntn;
    const int nstrips = 8 * spx + (8 * (slots - spx * ntn)) / ntn;
    if (nstrips <= 0) return 0;
    const int U = (M + 15) >> 4;
    return ((U + nstrips - 1) / nstrips) * 16;
}

// Rounds of 256 x 256 tiles the slowest block walks: tile list (XCD chunks dealt round-robin) vs strips (a half tile costs
// ~0.6 of a tile: one wave per SIMD multiplies).
static bool pers_strip_wins(int M, int ntn, int grid) {
    const int ntiles = ((M + 255) / 256) * ntn;
    if (ntiles < 2 * grid) return false;
    const int chunk = (ntiles + 7) / 8, per_block = (chunk + (grid >> 3) - 1) / (grid >> 3);
    const int R = pers_strip_rows(M, ntn, grid);
    if (!R) return false;
    const int rem = R & 255;
    // (a half tile: 0.5-0.75 of a tile by its K; FC1 at batch 256 - 9 tiles + a half tile against 10 - measured 274-279 ->
    // 265-271 us with the 64-deep image)
    const double strip_cost = (R >> 8) + (rem == 0 ? 0.0 : rem <= 128 ? 0.6 : 1.0);
    return strip_cost <= 0.965 * per_block;
}

template <int EPI, bool SPLIT_DMA, bool PAIR = false, bool STRIP = false, bool LNF = false>
void launch_gemm_pers_t(mocr_engine* e, const GemmParams& p0, int blocks) {
    GemmParams p = p0;
    p.ntn = p.N / 256;
    p.ntm = (p.M + 255) / 256;
    const int ntiles = p.ntm * p.ntn;
    int grid = std::min(blocks > 0 ? blocks : e->num_cus, (ntiles + 7) / 8 * 8);
    grid = std::max(8, grid / 8 * 8);
    static const int stagger_env = env_int("MOCR_GEMM_STAGGER", 0);
    p.stagger = stagger_env;
    static const int hpos_env = env_int("MOCR_GEMM_HPOS", 0);
    p.first_round = hpos_env;
    hipLaunchKernelGGL((gemm_pers_kernel<EPI, SPLIT_DMA, PAIR, STRIP, LNF>), dim3(grid), dim3(512), PERS_LDS, e->stream, p);
    HIPCHECK(hipGetLastError());
}

// strip: 0 tile list, 1 strips, -1 whichever walks fewer rounds (EPI_BIAS_RESID only; the other epilogues keep the list).
// p.ln_part set: the LayerNorm-folding forms of the kernel (kernels_gemm_pers.h, LNF).
template <bool SPLIT_DMA, bool PAIR = false>
void launch_gemm_pers(mocr_engine* e, const GemmParams& p, int epi, int blocks, int strip = 0) {
    if (p.k_per_split % 64 || p.k_per_split < 128) throw ArgError{"persistent gemm: K must be a multiple of 64, >= 128", MOCR_ERR_ARG};
    const bool lnf = p.ln_part != nullptr;
    if (lnf && !(SPLIT_DMA && (epi == EPI_BIAS_RESID ? (PAIR && p.N <= 1024 && p.xb) : p.csum != nullptr)))
        throw ArgError{"persistent gemm: LayerNorm folding needs the product kernel forms and their operands", MOCR_ERR_ARG};
    if constexpr (SPLIT_DMA) {
        bool strips = false;
        if (strip) {
            const int ntn = p.N / 256, ntiles = ((p.M + 255) / 256) * ntn;
            const int grid = std::max(8, std::min(blocks > 0 ? blocks : e->num_cus, (ntiles + 7) / 8 * 8) / 8 * 8);
            strips = strip > 0 ? pers_strip_rows(p.M, ntn, grid) > 0 : pers_strip_wins(p.M, ntn, grid);
        }
        if constexpr (PAIR) {
            if (epi == EPI_BIAS_RESID) {
                if (strips && lnf) { launch_gemm_pers_t<EPI_BIAS_RESID, true, true, true, true>(e, p, blocks); return; }
                if (strips) { launch_gemm_pers_t<EPI_BIAS_RESID, true, true, true>(e, p, blocks); return; }
                if (lnf) { launch_gemm_pers_t<EPI_BIAS_RESID, true, true, false, true>(e, p, blocks); return; }
            }
            // the bf16 epilogues on the one-barrier-per-two-K-tiles loop (K64: 64-deep LDS image, whole-line DMA requests)
            if (epi == EPI_BIAS_GELU) {
                if (strips && lnf) { launch_gemm_pers_t<EPI_BIAS_GELU, true, true, true, true>(e, p, blocks); return; }
                if (strips) { launch_gemm_pers_t<EPI_BIAS_GELU, true, true, true>(e, p, blocks); return; }
                if (lnf) { launch_gemm_pers_t<EPI_BIAS_GELU, true, true, false, true>(e, p, blocks); return; }
            }
            if (epi == EPI_BIAS) {
                if (strips && lnf) { launch_gemm_pers_t<EPI_BIAS, true, true, true, true>(e, p, blocks); return; }
                if (strips) { launch_gemm_pers_t<EPI_BIAS, true, true, true>(e, p, blocks); return; }
                if (lnf) { launch_gemm_pers_t<EPI_BIAS, true, true, false, true>(e, p, blocks); return; }
            }
        } else {
            if (epi == EPI_BIAS_RESID && strips) { launch_gemm_pers_t<EPI_BIAS_RESID, true, false, true>(e, p, blocks); return; }
            if (epi == EPI_BIAS_GELU) {
                if (strips && lnf) { launch_gemm_pers_t<EPI_BIAS_GELU, true, false, true, true>(e, p, blocks); return; }
                if (strips) { launch_gemm_pers_t<EPI_BIAS_GELU, true, false, true>(e, p, blocks); return; }
                if (lnf) { launch_gemm_pers_t<EPI_BIAS_GELU, true, false, false, true>(e, p, blocks); return; }
            }
            if (epi == EPI_BIAS) {
                if (strips && lnf) { launch_gemm_pers_t<EPI_BIAS, true, false, true, true>(e, p, blocks); return; }
                if (strips) { launch_gemm_pers_t<EPI_BIAS, true, false, true>(e, p, blocks); return; }
                if (lnf) { launch_gemm_pers_t<EPI_BIAS, true, false, false, true>(e, p, blocks); return; }
            }
        }
        if (lnf) throw ArgError{"persistent gemm: no LayerNorm-folding form of this kernel variant", MOCR_ERR_ARG};
    }
    switch (epi) {
        case EPI_BIAS: launch_gemm_pers_t<EPI_BIAS, SPLIT_DMA, PAIR>(e, p, blocks); break;
        case EPI_BIAS_GELU: launch_gemm_pers_t<EPI_BIAS_GELU, SPLIT_DMA, PAIR>(e, p, blocks); break;
        case EPI_BIAS_RESID: launch_gemm_pers_t<EPI_BIAS_RESID, SPLIT_DMA, PAIR>(e, p, blocks); break;
        default: throw ArgError{"persistent gemm: unsupported epilogue", MOCR_ERR_ARG};
    }
}

#ifdef MOCR_EXPERIMENTS
void launch_gemm_wide2(mocr_engine* e, const GemmParams& p, int epi) {
    if (p.k_per_split % 64 || p.k_per_split < 128) throw ArgError{"wide2 gemm: K must be a multiple of 64, >= 128", MOCR_ERR_ARG};
    switch (epi) {
        case EPI_BIAS: launch_gemm_wide2_t<EPI_BIAS>(e, p); break;
        case EPI_BIAS_GELU: launch_gemm_wide2_t<EPI_BIAS_GELU>(e, p); break;
        case EPI_BIAS_RESID: launch_gemm_wide2_t<EPI_BIAS_RESID>(e, p); break;
        default: throw ArgError{"wide2 gemm: unsupported epilogue", MOCR_ERR_ARG};
    }
}

template <int WN>
void launch_gemm_wide(mocr_engine* e, const GemmParams& p, int epi) {
    switch (epi) {
        case EPI_BIAS: launch_gemm_wide_t<EPI_BIAS, WN>(e, p); break;
        case EPI_BIAS_GELU: launch_gemm_wide_t<EPI_BIAS_GELU, WN>(e, p); break;
        case EPI_BIAS_RESID: launch_gemm_wide_t<EPI_BIAS_RESID, WN>(e, p); break;
        default: throw ArgError{"wide gemm: unsupported epilogue", MOCR_ERR_ARG};
    }
}

#endif  // MOCR_EXPERIMENTS

// A [M,K] (lda), W [N,K] (ldw=K), out (ldo).  tile: 64 / 128 (gemm_kernel), 4096 (the persistent 256 x 256 encoder kernel;
// 4097: the same on 8 blocks, a test hook); experiments build only: 256, 512, 1024, 2048, 4098 (kernels_gemm_lab.h).
// split > 1 only with EPI_SLAB.
struct HeadBatch { int heads = 1; long long a_yoff = 0, w_yoff = 0, o_yoff = 0, b_yoff = 0; int ldw = 0; };
// LayerNorm folded into the persistent encoder GEMMs (tile code 4096; kernels_gemm_pers.h LNF).  EPI_BIAS_RESID: `part` and
// `xb` are written; EPI_BIAS / EPI_BIAS_GELU: `part` and `csum` are read (W and bias are the folded ones).
struct LnFold { float* part = nullptr; const float* csum = nullptr; void* xb = nullptr; };

template <typename T>
void gemm(mocr_engine* e, const char* name, const void* A, int lda, const void* W, const float* bias, void* out, int ldo,
          const float* resid, int M, int N, int K, int epi, int tile, int split, long long slab_stride = 0,
          const float* pos = nullptr, int patches = 0, const HeadBatch* hb = nullptr, int group_n = 0, int* cand_idx = nullptr,
          const LnFold* lnf = nullptr) {
    const int kt = 128 / (int)sizeof(T);
    if (N % (tile >= 1024 ? 256 : tile >= 256 ? 128 : std::max(tile, 1)) || K % (kt * split) || (split > 1 && epi != EPI_SLAB) ||
        (tile >= 256 && (sizeof(T) != 2 || split != 1)))
        throw ArgError{std::string("gemm shape not tileable: ") + name, MOCR_ERR_ARG};
    GemmParams p{};
    p.A = A; p.W = W; p.bias = bias; p.out = out; p.resid = resid; p.pos = pos; p.cand_idx = cand_idx;
    p.M = M; p.N = N; p.lda = lda; p.ldw = K; p.ldo = ldo;
    int ybatch = 1;
    if (hb) {
        if (epi != EPI_BIAS || tile >= 256) throw ArgError{"per-head batched GEMM needs EPI_BIAS on the 64/128 kernel", MOCR_ERR_ARG};
        ybatch = hb->heads; p.a_yoff = hb->a_yoff; p.w_yoff = hb->w_yoff; p.o_yoff = hb->o_yoff; p.b_yoff = hb->b_yoff;
        if (hb->ldw) p.ldw = hb->ldw;
    }
    p.k_per_split = K / split; p.slab_stride = slab_stride; p.patches = patches;
    if (lnf) {
        if (tile < 4096 || tile > 4100 || sizeof(T) != 2) throw ArgError{"LayerNorm folding: persistent bf16 GEMMs only", MOCR_ERR_ARG};
        p.ln_part = lnf->part; p.csum = lnf->csum; p.xb = lnf->xb; p.ln_eps = e->cfg.ln_eps;
    }
    static const int ablate = env_int("MOCR_GEMM_ABLATE", 0);
    p.ablate = ablate;
#ifdef MOCR_EXPERIMENTS
    static unsigned long long* stamp_buf = nullptr;      // diagnostics: MOCR_GEMM_ABLATE & 8192 (kernels_gemm_pers.h, MOCR_STAMP)
    if ((ablate & 8192) && tile >= 4096 && epi != EPI_PATCH) {
        if (!stamp_buf) HIPCHECK(hipMalloc(&stamp_buf, 1024 * 8 * 4 * 8));
        HIPCHECK(hipMemsetAsync(stamp_buf, 0, 1024 * 8 * 4 * 8, e->stream));
        p.pos = reinterpret_cast<const float*>(stamp_buf);
    }
#endif
    static const int group_env = env_int("MOCR_GEMM_GROUPN", -1);
    p.group_n = group_env >= 0 ? group_env : group_n;
    const double out_b = (epi == EPI_BIAS || epi == EPI_BIAS_GELU) ? sizeof(T) : 4.0;
    const double bytes = ((double)M * K + (double)N * K) * sizeof(T) + (double)M * N * out_b * (epi == EPI_SLAB ? split : 1) +
                         (epi == EPI_BIAS_RESID ? (double)M * N * 4 : 0) + (lnf && epi == EPI_BIAS_RESID ? (double)M * N * 2 : 0);
    ProfScope ps(e, name, 2.0 * M * N * K * ybatch, bytes * ybatch);
    // one barrier per two K-tiles for the fp32-residual GEMMs (r03, M = 50,432: O-proj 137 -> 129 us, FC2 305 -> 302; the
    // bf16-output GEMMs lose with it: QKV 175 -> 208 us)
    // 4099 / 4100: the strip schedule forced (whole grid / 8 blocks) - test hooks like 4097
    if (tile == 4096 || tile == 4097 || tile == 4099 || tile == 4100) {
        const int blocks = (tile == 4097 || tile == 4100) ? 8 : 0;     // 4097: test hook, 8 blocks walk all the tiles
        static const int strip_env = env_int("MOCR_GEMM_STRIP", -1);     // -1: strips where they walk fewer rounds (r03, M = 50,432: O-proj 133 -> 113 us, FC2 303 -> 275)
        const int strip = tile >= 4099 ? 1 : tile == 4097 ? 0 : strip_env;
        // the bf16-epilogue GEMMs: 1 = the one-barrier-per-two-K-tiles loop with the 64-deep LDS image (K64), 0 = one barrier per
        // 32-deep K-tile, three K-tiles in flight (r03 first session's choice, when both loops fed on half-line requests)
#ifdef MOCR_EXPERIMENTS
        static const int pair_bf16 = env_int("MOCR_GEMM_PAIR_BF16", 1);
        if (epi != EPI_BIAS_RESID && !pair_bf16) launch_gemm_pers<true, false>(e, p, epi, blocks, strip);
        else
#endif
        launch_gemm_pers<true, true>(e, p, epi, blocks, strip);
    }
#ifdef MOCR_EXPERIMENTS
    else if (tile == 4098) launch_gemm_pers<false>(e, p, epi, 0);   // experiment: every wave requests LDS-DMA
    else if (tile == 4101) launch_gemm_pers<true, true>(e, p, epi, 0);    // experiment: one barrier per two K-tiles
    else if (tile == 4102) launch_gemm_pers<true, true>(e, p, epi, 8);
    else if (tile == 4105) launch_gemm_pers<false, true>(e, p, epi, 0);        // experiment: the pair loop with every wave requesting LDS-DMA
    else if (tile == 4106) launch_gemm_pers<false, true>(e, p, epi, 8);
    else if (tile == 4103) launch_gemm_pers<true, false>(e, p, epi, 0, 1);      // experiment: strips on the one-barrier-per-K-tile loop
    else if (tile == 4104) launch_gemm_pers<true, false>(e, p, epi, 8, 1);
    else if (tile == 2048) launch_gemm_wide2(e, p, epi);
    else if (tile == 1024) launch_gemm_wide<4>(e, p, epi);
    else if (tile == 512) launch_gemm_wide<2>(e, p, epi);
    else if (tile == 256) launch_gemm256(e, p, epi);
#else
    else if (tile == 256 || tile == 512 || tile == 1024 || tile == 2048 || tile == 4098)
        throw ArgError{"this GEMM tile code is an A/B kernel of the experiments build (build.py --experiments)", MOCR_ERR_UNSUPPORTED};
#endif
    else if (tile == 128) launch_gemm_epi<T, 128, 128>(e, p, epi, split, ybatch);
    else if (tile == 64) launch_gemm_epi<T, 64, 64>(e, p, epi, split, ybatch);
    else throw ArgError{"gemm tile must be 64, 128 or 4096", MOCR_ERR_ARG};
#ifdef MOCR_EXPERIMENTS
    if ((ablate & 8192) && tile >= 4096 && stamp_buf) {
        HIPCHECK(hipStreamSynchronize(e->stream));
        std::vector<unsigned long long> h(256 * 8 * 4);
        HIPCHECK(hipMemcpy(h.data(), stamp_buf, h.size() * 8, hipMemcpyDeviceToHost));
        double s0[2] = {0, 0}, s1[2] = {0, 0}, s2[2] = {0, 0}, pairs = 0;
        int nb = 0;
        for (int b = 0; b < 256; ++b) {
            if (!h[(b * 8) * 4 + 3]) continue;
            ++nb;
            pairs += (double)h[(b * 8) * 4 + 3] / 2;
            for (int w = 0; w < 8; ++w) { s0[w >> 2] += h[(b * 8 + w) * 4]; s1[w >> 2] += h[(b * 8 + w) * 4 + 1]; s2[w >> 2] += h[(b * 8 + w) * 4 + 2]; }
        }
        if (nb) {
            const double n = pairs * 4;      // wave-pairs per role
            fprintf(stderr, "[stamps] %s: %d blocks, %.0f K-tile pairs each; cycles per pair  DMA waves: work %.0f, DMA wait %.0f, barrier %.0f | store waves: work %.0f, -, barrier %.0f\n",
                    name, nb, pairs / nb, s0[0] / n, s1[0] / n, s2[0] / n, s0[1] / n, (s1[1] + s2[1]) / n);
        }
    }
#endif
}

// ---------------------------------------------------------------------------------------- encoder
template <typename T>
void layernorm(mocr_engine* e, const float* x, const float* g, const float* b, void* out, int M) {
    ProfScope ps(e, "layernorm", 0, (double)M * e->D * (4 + sizeof(T)));
    hipLaunchKernelGGL((layernorm_kernel<T, 768>), dim3((M + 3) / 4), dim3(256), 0, e->stream, x, g, b,
                       reinterpret_cast<T*>(out), M, e->cfg.ln_eps);
    HIPCHECK(hipGetLastError());
}

template <typename T>
void enc_attention(mocr_engine* e, const void* qkv, void* ctx, int n, int impl) {
    const int S = e->S, H = e->H;
    const double flops = 4.0 * n * H * (double)S * S * 64;
    const double bytes = (double)n * S * e->D * 4 * sizeof(T);
    if (impl == 1 && sizeof(T) == 2) {
        ProfScope ps(e, "enc_attn_mfma", flops, bytes);
        // a few crops: two / four blocks per (image, head) share its thirteen 16-query units while n * H blocks would leave
        // most of the chip idle (three blocks fit a CU)
        const long long blocks = (long long)n * H;
        const int ysplit = blocks * 4 <= 2LL * e->num_cus ? 4 : blocks * 2 <= 2LL * e->num_cus ? 2 : 1;
        static const int ablate2_env = env_int("MOCR_ENC_ATTN_ABLATE", 0);
        hipLaunchKernelGGL(enc_attn2_kernel, dim3(n * H, ysplit), dim3(256), EA2_LDS, e->stream,
                           reinterpret_cast<const bf16_t*>(qkv), reinterpret_cast<bf16_t*>(ctx), H, 3 * e->D, e->D, ablate2_env);
#ifdef MOCR_EXPERIMENTS
    } else if (impl == 2 && sizeof(T) == 2) {          // r01-r02 kernel (K / V staged through registers), A/B only
        ProfScope ps(e, "enc_attn_mfma_r02", flops, bytes);
        constexpr int lds = ENC_SP * 128 + 64 * ENC_VT_LD * 2;
        static const int qsplit_env = env_int("MOCR_ENC_ATTN_QSPLIT", 1);
        const int ysplit = (qsplit_env && n * H * 2 <= e->num_cus) ? 2 : 1;
        static const int ablate_env = env_int("MOCR_ENC_ATTN_ABLATE", 0);
        hipLaunchKernelGGL(enc_attn_mfma_kernel, dim3(n * H, ysplit), dim3(256), lds, e->stream,
                           reinterpret_cast<const bf16_t*>(qkv), reinterpret_cast<bf16_t*>(ctx), H, 3 * e->D, e->D, ablate_env);
#endif
    } else if (impl == 1 && sizeof(T) == 4) {          // r04: the parity mode on the f32-input matrix cores
        ProfScope ps(e, "enc_attn_f32_mfma", flops, bytes);
        hipLaunchKernelGGL(enc_attn_f32_kernel, dim3(n * H), dim3(256), EAF_LDS, e->stream, reinterpret_cast<const float*>(qkv),
                           reinterpret_cast<float*>(ctx), H, 3 * e->D, e->D);
    } else {
        ProfScope ps(e, "enc_attn_simple", flops, bytes);
        constexpr int lds = (200 * 65 + 200 * 64 + 4 * 64 + 4 * 256) * 4;
        hipLaunchKernelGGL((enc_attn_simple_kernel<T>), dim3(n * H), dim3(256), lds, e->stream,
                           reinterpret_cast<const T*>(qkv), reinterpret_cast<T*>(ctx), S, H, 3 * e->D, e->D, 0.125f);
    }
    HIPCHECK(hipGetLastError());
}

// calibrate_ln_fold's observer: X [M, 768] is the input of the LayerNorm about to run.  Per row, the rounding noise the GEMM
// behind it sees when it is fed bf16(x) (the fold) over the noise when it is fed bf16(LN(x)) (the launch):
//     sqrt( sum_k (x_k g_k rstd)^2 / sum_k ((x_k - mean) rstd g_k + b_k)^2 ),    rms over the rows; the worst LayerNorm counts.
static void calib_observe(mocr_engine* e, int M) {
    auto& c = *e->calib;
    if (c.idx >= c.g.size()) return;                 // (the encoder's final LayerNorm is never folded)
    const int D = e->D;
    c.hx.resize((size_t)M * D);
    HIPCHECK(hipMemcpyAsync(c.hx.data(), e->X, (size_t)M * D * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    HIPCHECK(hipStreamSynchronize(e->stream));
    const std::vector<float>&g = c.g[c.idx], &b = c.b[c.idx];
    double num = 0.0, den = 0.0;
    for (int m = 0; m < M; ++m) {
        const float* x = c.hx.data() + (size_t)m * D;
        double s = 0.0, q = 0.0;
        for (int k = 0; k < D; ++k) s += x[k];
        const double mean = s / D;
        for (int k = 0; k < D; ++k) { const double d = x[k] - mean; q += d * d; }
        const double rstd = 1.0 / std::sqrt(q / D + (double)e->cfg.ln_eps);
        for (int k = 0; k < D; ++k) {
            const double a = (double)x[k] * g[k] * rstd, y = ((double)x[k] - mean) * rstd * g[k] + b[k];
            num += a * a; den += y * y;
        }
    }
    c.worst = std::max(c.worst, std::sqrt(num / std::max(den, 1e-300)));
    c.idx += 1;
}

template <typename T>
void run_encoder(mocr_engine* e, const uint8_t* d_gray, int n) {
    const int D = e->D, F = e->F, S = e->S, M = n * S, P = e->cfg.patch_size, IMG = e->cfg.image_size;
    const int NP = e->G * e->G, MPATCH = n * NP;
    auto& w = e->w;
    {
        const long long total = (long long)n * IMG * e->G;
        ProfScope ps(e, "patchify", 0, (double)n * IMG * IMG * (1 + sizeof(T)));
        hipLaunchKernelGGL((patchify_kernel<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, e->stream, d_gray,
                           w.lut, reinterpret_cast<T*>(e->Hb), n, IMG, P);
        HIPCHECK(hipGetLastError());
    }
    {
        ProfScope ps(e, "cls_rows", 0, (double)n * D * 4);
        hipLaunchKernelGGL(cls_rows_kernel, dim3((n * D + 255) / 256), dim3(256), 0, e->stream, w.cls, w.pos_enc, e->X, n, S, D);
        HIPCHECK(hipGetLastError());
    }
    // MOCR_ENC_TILE forces one tile code for the layer GEMMs (experiments); the patch embedding has its own epilogue
    // and stays on the 128x128 kernel unless a tile code that supports it (64/128/256) is forced
    static const int enc_tile_env = env_int("MOCR_ENC_TILE", 0);
    const int ET = (enc_tile_env && enc_tile_env <= 256) ? enc_tile_env
                   : ((long long)((MPATCH + 127) / 128) * (e->D / 128) * 5 <= 4LL * e->num_cus ? 64 : 128);
    // The 128x128 kernel walks QKV / FC1 in column groups of 9 / 12 N-tiles (a 1.7 / 2.3 MB weight slice stays in
    // the XCD's L2): +4 % / +2 % at M = 806,912 (r01).
    // Layer GEMMs: the 256x256 "wide" kernel (tile code 1024) once it fills the chip about three times over
    // (+12..15 % at M = 806,912, +10..20 % at M = 100,864; even at 12,608 rows), else the 128x128 kernel
    // ... and 64 x 64 tiles while the 128 x 128 grid would leave CUs idle (up to ~0.8 blocks per CU): a block's K-tile costs
    // ~1 us whatever the tile (8 DMA instructions per wave to issue against 32 MFMAs), so a few crops are done sooner as
    // four times as many blocks of a quarter of the work.  r02, encoder of 1 / 4 / 8 crops: 1.41 / 1.49 / 1.54 -> 0.95 /
    // 1.05 / 1.39 ms (QKV of one crop 16.0 -> 10.4 us, O-proj 16.7 -> 9.6, FC1 17.5 -> 11.2, FC2 40.1 -> 21.3)
    auto small_tile = [&](int N) {
        const long long blocks128 = (long long)((M + 127) / 128) * (N / 128);
        return blocks128 * 5 <= 4LL * e->num_cus ? 64 : 128;
    };
    auto layer_tile = [&](int N) {
        if (enc_tile_env) return enc_tile_env;
        const long long tiles = (long long)((M + 255) / 256) * (N / 256);
        // r03: the persistent kernel (tile code 4096, kernels_gemm_pers.h) instead of one 256 x 256 tile per block (2048)
        static const int big_env = env_int("MOCR_ENC_BIG_TILE", 4096);
        // from two tiles per CU (r02 asked for three rounds of one-tile blocks; a persistent block has no turnover to
        // amortise, and at batch 256 the N = 768 GEMMs have 591 tiles)
        // (r03, 64-deep image: from 1.7 tiles per CU - QKV of a 64-crop batch, 441 tiles: 64-68 -> 54 us; MOCR_ENC_BIG_ROUNDS x 10)
        static const int big_rounds10 = env_int("MOCR_ENC_BIG_ROUNDS10", 17);
        // ... and the N = 768 GEMMs also where ONE round of tiles covers half the chip or more (56-111 crops: at 64 crops - 147
        // tiles - FC2 111 -> 96 us and, all four layer GEMMs then being persistent, the LayerNorm launches go: encoder 4.20 ->
        // 3.76 ms; at 128 crops - 297 tiles, two rounds for 1.16 - and at 32 - 75 tiles - the 128 x 128 kernel stays ahead)
        static const int one_round = env_int("MOCR_ENC_ONE_ROUND", 1);
        const bool big = tiles * 10 >= (long long)big_rounds10 * e->num_cus ||
                         (one_round && N <= 1024 && tiles * 2 >= e->num_cus && tiles <= e->num_cus);
        return (sizeof(T) == 2 && big) ? big_env : small_tile(N);
    };
    // per-GEMM overrides for experiments: MOCR_ENC_TILE_QKV / _O / _FC1 / _FC2 (tile codes as in gemm())
    static const int tq_env = env_int("MOCR_ENC_TILE_QKV", 0), to_env = env_int("MOCR_ENC_TILE_O", 0),
                     t1_env = env_int("MOCR_ENC_TILE_FC1", 0), t2_env = env_int("MOCR_ENC_TILE_FC2", 0);
    const int ETQ = tq_env ? tq_env : layer_tile(3 * D), ETO = to_env ? to_env : layer_tile(D), ET1 = t1_env ? t1_env : layer_tile(F);
    const int ET2 = t2_env ? t2_env : ETO;
    // tile order of QKV / FC1: column groups whose weight slices stay in an XCD's 4 MiB L2 while the A row-panels stream through
    // once per group.  128 x 128 tiles: groups of 9 / 12 N-tiles (r01).  Persistent 256 x 256 tiles: groups of 6 (r04: QKV 6 + 3,
    // FC1 6 + 6: the whole 3.4 / 4.5 MiB weight does not fit beside the A panels, and without groups every XCD re-fetches it
    // per round of row-panels - at batch 256 QKV 210 -> 200 us, FC1 337 -> 331 us, one box; tools/r04_groupn_ab.sh)
    const int gq = ETQ >= 4096 ? 6 : 9, g1 = ET1 >= 4096 ? 6 : 12;
    gemm<T>(e, "gemm_patch_embed", e->Hb, P * P, w.wpe, w.bpe, e->X, D, nullptr, MPATCH, D, P * P, EPI_PATCH, ET, 1, 0,
            w.pos_enc, NP);
    const int impl = (e->cfg.flags & MOCR_FLAG_SIMPLE_ATTENTION) ? 0 : 1;
    // A few crops (the 64 x 64 grid of the two N = 768 GEMMs is at most one block per CU: up to 5 crops): O-proj and FC2
    // are split over K into fp32 slabs - for one crop 12 / 48 K-tiles walked alone by 48 blocks become 4 / 6 K-tiles on
    // 144 / 384 blocks - and the LayerNorm launch that follows anyway finishes them (bias + slabs + residual, in place,
    // fixed order; layernorm_slab_kernel).  r02, encoder of 1 / 2 / 4 crops: 1.07 / 1.07 / 1.12 -> 0.88 / 0.90 / 1.01 ms
    // (one crop: FC2 28.3 -> 9.3 us, O-proj 11.8 -> 7.7, the LayerNorm behind them 6.6 -> 10.3)
    static const int enc_split_env = env_int("MOCR_ENC_SPLITK", 1);
    const long long blocks64 = (long long)((M + 63) / 64) * (D / 64);
    int split_o = 1, split_2 = 1;
    static const int enc_split_blocks = env_int("MOCR_ENC_SPLITK_BLOCKS", 0);      // largest unsplit 64 x 64 grid that is split (0: one block per CU)
    if (enc_split_env && !enc_tile_env && !e->calib && ETO == 64 && blocks64 <= (enc_split_blocks ? enc_split_blocks : e->num_cus)) {
        split_o = 3; split_2 = 8;
        while (split_2 > 1 && (long long)M * D * split_2 > e->slab_cap) split_2 >>= 1;
        if ((long long)M * D * split_o > e->slab_cap) split_o = 1;
        if (split_2 < 2) split_2 = 1;
    }
    // bf16, all four layer GEMMs on the persistent kernel: the 24 LayerNorms between them are FOLDED into those GEMMs
    // (kernels_gemm_pers.h LNF; r03: 25 launches of 38-40 us were 7.5 % of the encoder at batch 256).  Xn then holds x itself
    // as bf16; only the final LayerNorm (the encoder's output) is a launch.
    bool fold = false;
    if constexpr (sizeof(T) == 2) {
        static const int fold_env = env_int("MOCR_ENC_LN_FOLD", 1);
        fold = fold_env && !(e->cfg.flags & MOCR_FLAG_NO_LN_FOLD) && ETQ == 4096 && ETO == 4096 && ET1 == 4096 && ET2 == 4096 &&
               D == 768 && w.enc[0].wqkv_f && !e->calib && (e->fold_ok || (e->cfg.flags & MOCR_FLAG_FORCE_LN_FOLD));
    }
    const float* pend_bias = nullptr;      // bias of a split GEMM whose slabs the next LayerNorm has to add to X
    int pend_slabs = 0;
    auto norm = [&](const float* g, const float* b, void* out) {
        if (e->calib && !pend_slabs) calib_observe(e, M);
        if (!pend_slabs) { layernorm<T>(e, e->X, g, b, out, M); return; }
        ProfScope ps(e, "layernorm_slab", 0, (double)M * D * (4.0 * (pend_slabs + 2) + sizeof(T)));
        hipLaunchKernelGGL((layernorm_slab_kernel<T, 768>), dim3((M + 3) / 4), dim3(256), 0, e->stream, e->X, e->slabs, pend_slabs,
                           (long long)M * D, pend_bias, g, b, reinterpret_cast<T*>(out), M, e->cfg.ln_eps);
        HIPCHECK(hipGetLastError());
        pend_slabs = 0;
    };
    if (fold) {
        {
            ProfScope ps(e, "ln_prep", 0, (double)M * D * 6);
            hipLaunchKernelGGL((ln_prep_kernel<768>), dim3((M + 3) / 4), dim3(256), 0, e->stream, e->X, reinterpret_cast<bf16_t*>(e->Xn),
                               e->ln_part, M);
            HIPCHECK(hipGetLastError());
        }
        for (int l = 0; l < e->cfg.enc_layers; ++l) {
            const EncLayerW& L = w.enc[l];
            LnFold use_q{e->ln_part, L.sqkv, nullptr}, use_1{e->ln_part, L.s1, nullptr}, emit{e->ln_part, nullptr, e->Xn};
            gemm<T>(e, "gemm_enc_qkv", e->Xn, D, L.wqkv_f, L.bqkv_f, e->QKV, 3 * D, nullptr, M, 3 * D, D, EPI_BIAS, ETQ, 1, 0, nullptr, 0, nullptr,
                    gq, nullptr, &use_q);
            enc_attention<T>(e, e->QKV, e->CTX, n, impl);
            gemm<T>(e, "gemm_enc_oproj", e->CTX, D, L.wo, L.bo, e->X, D, e->X, M, D, D, EPI_BIAS_RESID, ETO, 1, 0, nullptr, 0, nullptr, 0,
                    nullptr, &emit);
            gemm<T>(e, "gemm_enc_fc1", e->Xn, D, L.w1_f, L.b1_f, e->Hb, F, nullptr, M, F, D, EPI_BIAS_GELU, ET1, 1, 0, nullptr, 0, nullptr, g1,
                    nullptr, &use_1);
            gemm<T>(e, "gemm_enc_fc2", e->Hb, F, L.w2, L.b2, e->X, D, e->X, M, D, F, EPI_BIAS_RESID, ET2, 1, 0, nullptr, 0, nullptr, 0,
                    nullptr, &emit);
        }
        norm(w.lnfg, w.lnfb, e->ENC);
        return;
    }
    for (int l = 0; l < e->cfg.enc_layers; ++l) {
        const EncLayerW& L = w.enc[l];
        norm(L.ln1g, L.ln1b, e->Xn);
        gemm<T>(e, "gemm_enc_qkv", e->Xn, D, L.wqkv, L.bqkv, e->QKV, 3 * D, nullptr, M, 3 * D, D, EPI_BIAS, ETQ, 1, 0, nullptr, 0, nullptr, gq);
        enc_attention<T>(e, e->QKV, e->CTX, n, impl);
        if (split_o > 1) {
            gemm<T>(e, "gemm_enc_oproj", e->CTX, D, L.wo, nullptr, e->slabs, D, nullptr, M, D, D, EPI_SLAB, 64, split_o, (long long)M * D);
            pend_bias = L.bo; pend_slabs = split_o;
        } else {
            gemm<T>(e, "gemm_enc_oproj", e->CTX, D, L.wo, L.bo, e->X, D, e->X, M, D, D, EPI_BIAS_RESID, ETO, 1);
        }
        norm(L.ln2g, L.ln2b, e->Xn);
        gemm<T>(e, "gemm_enc_fc1", e->Xn, D, L.w1, L.b1, e->Hb, F, nullptr, M, F, D, EPI_BIAS_GELU, ET1, 1, 0, nullptr, 0, nullptr, g1);
        if (split_2 > 1) {
            gemm<T>(e, "gemm_enc_fc2", e->Hb, F, L.w2, nullptr, e->slabs, D, nullptr, M, D, F, EPI_SLAB, 64, split_2, (long long)M * D);
            pend_bias = L.b2; pend_slabs = split_2;
        } else {
            gemm<T>(e, "gemm_enc_fc2", e->Hb, F, L.w2, L.b2, e->X, D, e->X, M, D, F, EPI_BIAS_RESID, ET2, 1);
        }
    }
    norm(w.lnfg, w.lnfb, e->ENC);
}

// ---------------------------------------------------------------------------------------- decoder
static int dec_tile(int rows) {
    static const int forced = env_int("MOCR_DEC_TILE", 0);
    // 128 x 128 tiles from 1024 rows (r02: from 512 - at 256 rows the 64 x 64 tiles need a third of the split-K slabs, 113 -> 101
    // ms for an isolated 256-crop batch.  r04, tools/r04_dectile_ab.sh, isolated batches, 64 / 128 tiles: 512 rows 128.0 / 134.3
    // ms, 640 rows 143.4 / 147.4, 768 rows 157.5 / 162.7, 1024 rows 199.8 / 195.4, 1280 rows 229 / 224)
    static const int fat = env_int("MOCR_DEC_FAT_ROWS", 1024);
    if (forced) return forced;
    return rows >= fat ? 128 : 64;
}

// The tile a decode-step GEMM is LAUNCHED on: the tile size does not touch a sum (same K-tiles, same order; the split over K
// stays the regime's, pick_split), so a compacted batch takes the tile of the rows it has left (MOCR_NEUTRAL_BY_ROWS >= 2, the
// default; r04, tools/r04_neutral_ab.sh, mixed-lengths leg, by regime / ring depth + query blocks by rows / + tile by rows:
// 11.96 / 12.12 / 12.39 k crops/s, ids bit-identical to the uncompacted engine in all three).
static int dec_launch_tile(const mocr_engine* e, int rows) {
    static const int neutral_by_rows = env_int("MOCR_NEUTRAL_BY_ROWS", 2);
    return dec_tile(neutral_by_rows >= 2 ? rows : e->rrows(rows));
}

static int pick_split(int N, int K, int kt, int rows, long long slab_cap_per_row) {
    // Split K until ~`target` blocks cover the chip, bounded by the K-tiles and by the slab buffer.
    // Every extra slab is an fp32 [rows,N] write plus a read by the consumer, so fat batches
    // (many row tiles) split less.
    // measured (r01): 150 for 64..1024 rows; fat batches (>= 2048 rows) gain from one more halving of K
    // (FC2 at 4096 rows: 48 -> 33 us: 192 tiles -> 384 blocks).  r02: 200 instead of 300 there - the same split at 4096
    // rows, but the N = 768 projections of a 2560-row batch (120 tiles) stop at 2 slabs instead of 4: the add + LayerNorm
    // behind them is bound by its slab traffic (bench 6.30-6.36 k -> 6.44-6.45 k crops/s; targets 60 / 100 / 450 / 600: 6.30 /
    // 6.32 / 6.35 / 6.10 k)
    static const int target_env = env_int("MOCR_DEC_BLOCKS", 0);
    // (r04, with the four-slot rings of lone batches, tools/r04_decblocks_small_ab.sh, targets 150 / 100: 96 rows 58.4 / 56.4 ms,
    // 128 rows 66.3 / 63.0, 40 / 64 / 192 / 256 rows equal; 320 and 768 rows lose with 100)
    const int target = target_env ? target_env : (rows >= 2048 ? 200 : rows <= 128 ? 100 : 150);
    const int tile = dec_tile(rows);
    const int tiles = (N / tile) * ((rows + tile - 1) / tile);
    const int ktiles = K / kt;
    int split = 1;
    while (tiles * split < target && split * 2 <= ktiles && ktiles % (split * 2) == 0 && (long long)(split * 2) * N <= slab_cap_per_row)
        split *= 2;
    if (tiles * split * 3 <= target * 2 && ktiles % (split * 3) == 0 && (long long)(split * 3) * N <= slab_cap_per_row) split *= 3;
    return split;
}

template <typename T>
int dec_gemm(mocr_engine* e, const char* name, const void* A, int lda, const void* W, int N, int K, int rows) {
    const int kt = 128 / (int)sizeof(T);
    const int split = pick_split(N, K, kt, e->rrows(rows), e->slab_cap / e->Bp);
    gemm<T>(e, name, A, lda, W, nullptr, e->slabs, N, nullptr, rows, N, K, EPI_SLAB, dec_launch_tile(e, rows), split, (long long)e->Bp * N);
    return split;
}

template <typename T>
void dec_add_ln(mocr_engine* e, int nslab, int N, const float* bias, const float* resid, const float* g, const float* b,
                float* out_f32, void* out_t, int rows, bool gelu, int cache_layer = -1) {
    ProfScope ps(e, "dec_add_ln", 0, (double)rows * N * 4 * (nslab + 3));
    T* cache = nullptr;
    uint8_t* cache8 = nullptr;
    float inv8 = 0.f;
    const long long cstride = (long long)e->cfg.max_len * e->D;
    if (cache_layer >= 0) {
        if (e->fp8attn) {
            cache8 = e->x8cache + (size_t)cache_layer * e->Bp * cstride;
            inv8 = 1.0f / e->w.sx_self[cache_layer];
        } else {
            cache = reinterpret_cast<T*>(e->xcache) + (size_t)cache_layer * e->Bp * cstride;
        }
    }
    if (gelu)
        hipLaunchKernelGGL((dec_add_ln_kernel<T, 768, true>), dim3(rows), dim3(192), 0, e->stream, e->slabs, nslab,
                           (long long)e->Bp * N, bias, resid, g, b, out_f32, reinterpret_cast<T*>(out_t), rows, e->cfg.ln_eps,
                           cache, cstride, (const int*)e->step, cache8, inv8, (const int*)e->rowmap);
    else
        hipLaunchKernelGGL((dec_add_ln_kernel<T, 768, false>), dim3(rows), dim3(192), 0, e->stream, e->slabs, nslab,
                           (long long)e->Bp * N, bias, resid, g, b, out_f32, reinterpret_cast<T*>(out_t), rows, e->cfg.ln_eps,
                           cache, cstride, (const int*)e->step, cache8, inv8, (const int*)e->rowmap);
    HIPCHECK(hipGetLastError());
}

static DecState make_state(mocr_engine* e, int max_len, const int* forced, int forced_T, float* logits_out, int n_real) {
    DecState st{};
    st.n_real = n_real;
    st.ids = e->ids; st.step = e->step; st.finished = e->finished; st.len = e->len; st.n_unfinished = e->n_unf;
    st.forced = forced; st.forced_T = forced_T; st.logits_out = logits_out;
    st.ids_ld = e->cfg.max_len; st.max_len = max_len;
    st.start_id = e->cfg.start_id; st.eos_id = e->cfg.eos_id; st.pad_id = e->cfg.pad_id;
    st.rowmap = e->rowmap;
    return st;
}

template <typename T, bool FIRST>
void dec_token(mocr_engine* e, const DecState& st, int nslab, int n, int ncand = 0) {
    auto& w = e->w;
    ProfScope ps(e, FIRST ? "dec_token_first" : "dec_token", 0, FIRST ? 0.0 : (double)n * e->V * 4 * nslab);
    const bool lat = e->use_latent(e->rrows(n));
    hipLaunchKernelGGL((dec_token_kernel<T, 768, FIRST>), dim3(n), dim3(256), 0, e->stream, e->slabs, nslab,
                       (long long)e->Bp * e->V, w.bv, e->V, st, w.word, w.type0, w.posd, w.embg, w.embb, e->x_f32,
                       reinterpret_cast<T*>(e->x_t), e->cfg.ln_eps, (lat && !e->fp8attn) ? reinterpret_cast<T*>(e->xcache) : nullptr,
                       (long long)e->cfg.max_len * e->D, ncand ? e->cand_val : nullptr, ncand ? e->cand_idx : nullptr, ncand,
                       (lat && e->fp8attn) ? e->x8cache : nullptr, (lat && e->fp8attn) ? 1.0f / w.sx_self[0] : 0.f);
    HIPCHECK(hipGetLastError());
}

template <typename T, bool SELF>
void dec_attn(mocr_engine* e, int layer, int nslab, int n, const float* bias, int approx_len) {
    const int D = e->D, H = e->H;
    DecAttnParams p{};
    p.slabs = e->slabs; p.nslab = nslab;
    p.ldq = SELF ? 3 * D : D;
    p.slab_stride = (long long)e->Bp * p.ldq;
    p.bias = bias;
    if (SELF) {
        const size_t per_layer = (size_t)e->Bc * H * e->cfg.max_len * 64;
        p.kbase = reinterpret_cast<char*>(e->kcache) + (size_t)layer * per_layer * sizeof(T);
        p.vbase = reinterpret_cast<char*>(e->vcache) + (size_t)layer * per_layer * sizeof(T);
        p.kv_batch_stride = (long long)H * e->cfg.max_len * 64;
        p.kv_head_stride = (long long)e->cfg.max_len * 64;
        p.kv_row_stride = 64;
        p.step = e->step;
    } else {
        p.kbase = reinterpret_cast<char*>(e->CKV) + (size_t)(layer * 2 * D) * sizeof(T);
        p.vbase = reinterpret_cast<char*>(e->CKV) + (size_t)(layer * 2 * D + D) * sizeof(T);
        p.kv_batch_stride = (long long)e->S * e->NCKV;
        p.kv_head_stride = 64;
        p.kv_row_stride = e->NCKV;
        p.cross_len = e->S;
    }
    p.ctx = e->ctx_t; p.H = H; p.scale = 0.125f;
    p.rowmap = e->rowmap;
    // K/V of a 64-row batch (2 layers x 197 keys x 3,072 B = 77 MB + the self cache) live in the Infinity Cache between
    // steps; from about 128 rows they no longer do and the non-temporal policy wins (r02, isolated batch: 256 rows
    // 100.2 -> 91.2 ms, 128 rows 69.1 -> 67.5 ms, 64 rows 49.8 -> 50.8 ms)
    static const int nt_rows = env_int("MOCR_ATTN_NT_ROWS", 128);
    p.nt = e->rrows(n) >= nt_rows;
    ProfScope ps(e, SELF ? "dec_attn_self" : "dec_attn_cross", 4.0 * n * H * approx_len * 64,
                 2.0 * n * H * approx_len * 64 * sizeof(T));
    if (SELF) {
        // NG = 8-key groups a wave may own: pick the smallest variant that covers approx_len keys
        // (approx_len is an upper bound of every row's context length during this launch)
        const int need = ((approx_len + 3) / 4 + 7) / 8;
        if (need <= 3) hipLaunchKernelGGL((dec_attn_kernel<T, true, 3>), dim3(n * H), dim3(256), 0, e->stream, p);
        else if (need <= 5) hipLaunchKernelGGL((dec_attn_kernel<T, true, 5>), dim3(n * H), dim3(256), 0, e->stream, p);
        else if (need <= 8) hipLaunchKernelGGL((dec_attn_kernel<T, true, 8>), dim3(n * H), dim3(256), 0, e->stream, p);
        else if (need <= 10) hipLaunchKernelGGL((dec_attn_kernel<T, true, 10>), dim3(n * H), dim3(256), 0, e->stream, p);
        else throw ArgError{"max_len > 320 needs a larger NG", MOCR_ERR_UNSUPPORTED};
    } else {
        hipLaunchKernelGGL((dec_attn_kernel<T, false, 7>), dim3(n * H), dim3(256), 0, e->stream, p);
    }
    HIPCHECK(hipGetLastError());
}

// The bf16 latent attention launch (r04).  Default: latent_attnT_kernel<.., 2> - 16-key tiles, the score tile transposed so that
// the softmax's probabilities feed P.X from registers (two barriers per tile; kernels_latent_t.h), THREE persistent blocks per
// CU on two-slot rings (52 KiB of LDS, 152 registers).  MOCR_FLAG_LATENT_TILE32 = r03's kernel shape, the A/B partner:
// latent_attn_kernel on 32-key tiles, one block per CU, three barriers per tile (kernels_latent.h).  Experiments build,
// MOCR_LAT_TK = 17: the default kernel on three-slot rings, two blocks per CU (-2 % in the bench); 16: latent_attn_kernel on
// 16-key tiles, two blocks per CU (the first half of r04; equal to 17 within noise).
void launch_latent(mocr_engine* e, bool self, const LatentParams& p) {
    static const int lat_blocks = env_int("MOCR_LAT_BLOCKS", 0);      // persistent blocks (experiments; 0: one or two per CU by tile)
    const int per_cu = e->lat_tk == 32 ? 1 : e->lat_tk == 18 ? 3 : 2;
    const int grid = std::min(p.rows, lat_blocks > 0 ? lat_blocks : per_cu * e->num_cus);
    if (e->lat_tk == 18) {          // the default: three blocks per CU on two-slot rings
        if (self) hipLaunchKernelGGL((latent_attnT_kernel<true, 2>), dim3(grid), dim3(256), LATT_LDS_OF(2), e->stream, p);
        else hipLaunchKernelGGL((latent_attnT_kernel<false, 2>), dim3(grid), dim3(256), LATT_LDS_OF(2), e->stream, p);
#ifdef MOCR_EXPERIMENTS
    } else if (e->lat_tk == 17) {
        if (self) hipLaunchKernelGGL(latent_attnT_kernel<true>, dim3(grid), dim3(256), LATT_LDS, e->stream, p);
        else hipLaunchKernelGGL(latent_attnT_kernel<false>, dim3(grid), dim3(256), LATT_LDS, e->stream, p);
    } else if (e->lat_tk == 16) {
        if (self) hipLaunchKernelGGL((latent_attn_kernel<true, 16>), dim3(grid), dim3(256), LatCfg<16>::LDS, e->stream, p);
        else hipLaunchKernelGGL((latent_attn_kernel<false, 16>), dim3(grid), dim3(256), LatCfg<16>::LDS, e->stream, p);
#endif
    } else {
        if (self) hipLaunchKernelGGL((latent_attn_kernel<true, 32>), dim3(grid), dim3(256), LatCfg<32>::LDS, e->stream, p);
        else hipLaunchKernelGGL((latent_attn_kernel<false, 32>), dim3(grid), dim3(256), LatCfg<32>::LDS, e->stream, p);
    }
    HIPCHECK(hipGetLastError());
}

// The fp8 latent attention launch: latent_attnT8_kernel (r04: transposed score tile, one 32-key tile per iteration, two blocks per
// CU; kernels_latent_t8.h) or - MOCR_FLAG_LATENT_TILE32, the A/B partner - r02's latent_attn_fp8_kernel (two tiles per iteration
// on a five-slot ring, one block per CU).
void launch_latent8(mocr_engine* e, bool self, const Latent8Params& p) {
    static const int lat_blocks = env_int("MOCR_LAT_BLOCKS", 0);
    if (e->lat_tk == 32) {
        const int grid = std::min(p.rows, lat_blocks > 0 ? lat_blocks : e->num_cus);
        if (self) hipLaunchKernelGGL(latent_attn_fp8_kernel<true>, dim3(grid), dim3(256), LAT8_LDS, e->stream, p);
        else hipLaunchKernelGGL(latent_attn_fp8_kernel<false>, dim3(grid), dim3(256), LAT8_LDS, e->stream, p);
    } else {
        const int grid = std::min(p.rows, lat_blocks > 0 ? lat_blocks : 2 * e->num_cus);
        if (self) hipLaunchKernelGGL(latent_attnT8_kernel<true>, dim3(grid), dim3(256), LATT8_LDS, e->stream, p);
        else hipLaunchKernelGGL(latent_attnT8_kernel<false>, dim3(grid), dim3(256), LATT8_LDS, e->stream, p);
    }
    HIPCHECK(hipGetLastError());
}

// Latent attention of n rows: Qt [n,16,768] x keys (self: cached layer-input rows; cross: encoder
// output) -> Et [n,16,768].  bytes: the X rows streamed once (1,536 B per key).
void latent_attn(mocr_engine* e, bool self, int layer, int n, int approx_len) {
    if (e->fp8attn) {
        Latent8Params p{};
        p.qt = reinterpret_cast<const bf16_t*>(e->qt);
        p.out = reinterpret_cast<bf16_t*>(e->et);
        p.heads = e->H;
        p.rows = n;
        p.rowmap = e->rowmap;
        if (self) {
            p.x8 = e->x8cache + (size_t)layer * e->Bp * e->cfg.max_len * e->D;
            p.x_batch_stride = (long long)e->cfg.max_len * e->D;
            p.step = e->step;
            p.sx = e->w.sx_self[layer];
        } else {
            p.x8 = e->enc8;
            p.x_batch_stride = (long long)e->S * e->D;
            p.fixed_len = e->S;
            p.sx = e->w.sx_enc;
        }
        ProfScope ps(e, self ? "lat8_attn_self" : "lat8_attn_cross", 4.0 * n * 16 * approx_len * e->D,
                     (double)n * approx_len * e->D + 2.0 * n * e->H * e->D * 2);     // e4m3 keys + Qt in + Et out (12 heads, bf16)
        launch_latent8(e, self, p);
        return;
    }
    LatentParams p{};
    p.qt = reinterpret_cast<const bf16_t*>(e->qt);
    p.out = reinterpret_cast<bf16_t*>(e->et);
    p.heads = e->H;
    p.rows = n;
    p.rowmap = e->rowmap;
    if (self) {
        p.x = reinterpret_cast<const bf16_t*>(e->xcache) + (size_t)layer * e->Bp * e->cfg.max_len * e->D;
        p.x_batch_stride = (long long)e->cfg.max_len * e->D;
        p.step = e->step;
    } else {
        p.x = reinterpret_cast<const bf16_t*>(e->ENC);
        p.x_batch_stride = (long long)e->S * e->D;
        p.fixed_len = e->S;
    }
    ProfScope ps(e, self ? "lat_attn_self" : "lat_attn_cross", 4.0 * n * 16 * approx_len * e->D,
                 (double)n * approx_len * e->D * 2 + 2.0 * n * e->H * e->D * 2);     // keys + Qt in + Et out (12 heads)
    launch_latent(e, self, p);
}

// The fused query launch (kernels_qqt.h): blocks of 64 rows for batches of up to MOCR_QQT_BM64_ROWS rows (a 16-KiB K-tile, seven
// in flight, twice the blocks: r04, tools/r04_qqt_bm_ab.sh), of 128 rows above - there every CU has a block either way and the
// 128-row block reads each weight tile for twice the rows.  `regime_rows` = the row count the choice is made by.
void launch_qqt(mocr_engine* e, const QqtParams& q, int n, int regime_rows) {
    static const int bm64_rows = env_int("MOCR_QQT_BM64_ROWS", 1280);
    if (regime_rows <= bm64_rows) hipLaunchKernelGGL(dec_qqt_kernel<64>, dim3((n + 63) / 64, 12), dim3(256), QqtCfg<64>::LDS, e->stream, q);
    else hipLaunchKernelGGL(dec_qqt_kernel<128>, dim3((n + 127) / 128, 12), dim3(256), env_int("MOCR_QQT_LDS", QQT_LDS), e->stream, q);
    HIPCHECK(hipGetLastError());
}

// q -> Qt -> latent attention -> ctx: the attention block of the latent path up to (not including)
// the output projection.  wq/bq: query projection; wkT: (Wk^T)/8; wv/bv: value projection.
void latent_block(mocr_engine* e, bool self, int layer, int n, int t, const void* xin, const void* wq, const float* bq,
                  const void* wkT, const void* wv, const float* bv) {
    using T = bf16_t;
    const int D = e->D;
    // fat batches: q and Qt in one launch (kernels_qqt.h), 37 us instead of 16 + 31 at 4096 rows; bit-identical to the
    // two-launch path.  MOCR_DEC_QQT_ROWS = rows from which it is used (0 = never)
    // (r04, tools/r04_qqt_rows_ab.sh, isolated batch, two launches / fused: 512 rows 136.2 / 134.9 ms, 768 rows 165.4 / 163.5, below
    // 512 rows the two launches stay ahead: the switch moved from 1024 to 512 rows)
    // (r04, 64-row blocks, tools/r04_qqt_rows_ab2.sh, two launches / fused: 288 rows 103.6 / 100.4 ms, 320 rows 105.0 / 101.5, 384 rows
    // 107.7 / 104.6, 448 and 512 rows equal: every latent batch - 257 rows and up - takes the fused launch)
    static const int qqt_rows = env_int("MOCR_DEC_QQT_ROWS", 257);
    if (qqt_rows > 0 && e->rrows(n) >= qqt_rows && D == 768 && e->H == 12 && !(e->cfg.flags & MOCR_FLAG_NO_FUSED_QQT)) {
        QqtParams q{};
        q.x = reinterpret_cast<const bf16_t*>(xin); q.wq = reinterpret_cast<const bf16_t*>(wq); q.bq = bq;
        q.wkT = reinterpret_cast<const bf16_t*>(wkT); q.qt = reinterpret_cast<bf16_t*>(e->qt);
        {
            ProfScope ps(e, "dec_qqt", 4.0 * n * D * D, (double)n * D * 2 + 2.0 * D * D * 2 + (double)n * e->H * D * 2);
            static const int neutral_by_rows = env_int("MOCR_NEUTRAL_BY_ROWS", 2);
            launch_qqt(e, q, n, neutral_by_rows ? n : e->rrows(n));
        }
        latent_attn(e, self, layer, n, self ? t + 1 : e->S);
        HeadBatch hc2; hc2.heads = e->H; hc2.a_yoff = D; hc2.w_yoff = (long long)64 * D; hc2.o_yoff = 64; hc2.b_yoff = 64; hc2.ldw = D;
        gemm<T>(e, "gemm_dec_ctx", e->et, 16 * D, wv, bv, e->ctx_t, D, nullptr, n, 64, D, EPI_BIAS, 64, 1, 0, nullptr, 0, &hc2);
        return;
    }
    static const int qtile = env_int("MOCR_DEC_QTILE", 64), qttile_env = env_int("MOCR_DEC_QTTILE", 0);
    const int qttile = qttile_env ? qttile_env : (e->rrows(n) >= 1024 ? 128 : 64);      // Qt is output-write bound: fewer, fatter blocks
    gemm<T>(e, "gemm_dec_q", xin, D, wq, bq, e->q_t, D, nullptr, n, D, D, EPI_BIAS, qtile, 1);
    HeadBatch hq; hq.heads = e->H; hq.a_yoff = 64; hq.w_yoff = 64; hq.o_yoff = D; hq.b_yoff = 0; hq.ldw = D;
    gemm<T>(e, "gemm_dec_qt", e->q_t, D, wkT, e->w.zero_bias, e->qt, 16 * D, nullptr, n, D, 64, EPI_BIAS, qttile, 1, 0, nullptr, 0, &hq);
    latent_attn(e, self, layer, n, self ? t + 1 : e->S);
    HeadBatch hc; hc.heads = e->H; hc.a_yoff = D; hc.w_yoff = (long long)64 * D; hc.o_yoff = 64; hc.b_yoff = 64; hc.ldw = D;
    gemm<T>(e, "gemm_dec_ctx", e->et, 16 * D, wv, bv, e->ctx_t, D, nullptr, n, 64, D, EPI_BIAS, 64, 1, 0, nullptr, 0, &hc);
}

template <int PRO, int EPI>
void smallm_gemm(mocr_engine* e, const char* name, SmallMParams p) {
    p.eps = e->cfg.ln_eps;
    const int mt = (p.rows + 15) / 16;
    if (mt < 1 || mt > 2 || p.N % SM_NT || (p.K != 768 && p.K != 3072) || (PRO == SM_PRO_LN && p.K != 768))
        throw ArgError{"small-batch GEMM: unsupported shape", MOCR_ERR_UNSUPPORTED};
    ProfScope ps(e, name, 2.0 * p.rows * p.N * p.K, (double)p.N * p.K * 2);
    const dim3 grid(p.N / SM_NT), block(64 * SM_NW);
    if constexpr (PRO == SM_PRO_LN) {
        if (mt == 1) hipLaunchKernelGGL((smallm_gemm_kernel<PRO, EPI, 1, 3>), grid, block, SM_LDS(PRO, 1), e->stream, p);
        else hipLaunchKernelGGL((smallm_gemm_kernel<PRO, EPI, 2, 3>), grid, block, SM_LDS(PRO, 2), e->stream, p);
    } else {
        if (p.K == 768) {
            if (mt == 1) hipLaunchKernelGGL((smallm_gemm_kernel<PRO, EPI, 1, 3>), grid, block, SM_LDS(PRO, 1), e->stream, p);
            else hipLaunchKernelGGL((smallm_gemm_kernel<PRO, EPI, 2, 3>), grid, block, SM_LDS(PRO, 2), e->stream, p);
        } else {
            if (mt == 1) hipLaunchKernelGGL((smallm_gemm_kernel<PRO, EPI, 1, 12>), grid, block, SM_LDS(PRO, 1), e->stream, p);
            else hipLaunchKernelGGL((smallm_gemm_kernel<PRO, EPI, 2, 12>), grid, block, SM_LDS(PRO, 2), e->stream, p);
        }
    }
    HIPCHECK(hipGetLastError());
}

// One greedy step of a SMALL bf16 batch (<= 32 rows, classic attention): 19 launches instead of 28 (kernels_smallm.h).
// x_f32 / a_f32 / c_f32 hold PRE-LayerNorm sums here (s3 of the previous layer - or the embedding rows for layer 0 -,
// s1, s2); ln_stats[k] the (mean, rstd) of s(k+1), published by the first projection that normalises it.
void decode_step_smallm(mocr_engine* e, const DecState& st, int n, int t) {
    using T = bf16_t;
    const int D = e->D, F = e->F;
    auto& w = e->w;
    float* const st1 = e->ln_stats, * const st2 = e->ln_stats + 2 * e->Bp, * const st3 = e->ln_stats + 4 * e->Bp;
    auto W = [](const void* q) { return reinterpret_cast<const bf16_t*>(q); };
    for (int l = 0; l < e->cfg.dec_layers; ++l) {
        const DecLayerW& L = w.dec[l];
        const DecLayerW* P = l ? &w.dec[l - 1] : nullptr;        // the layer whose LayerNorm 3 produces this layer's input
        SmallMParams q{};
        q.rows = n; q.K = D; q.w = W(L.wqkv); q.N = 3 * D; q.out = e->slabs; q.ldo = 3 * D;
        if (!P) { q.a_bf16 = W(e->x_t); smallm_gemm<SM_PRO_PLAIN, SM_EPI_RAW>(e, "sm_qkv", q); }
        else { q.a_f32 = e->x_f32; q.ln_g = P->ln3g; q.ln_b = P->ln3b; q.stats_out = st3; smallm_gemm<SM_PRO_LN, SM_EPI_RAW>(e, "sm_qkv", q); }
        dec_attn<T, true>(e, l, 1, n, L.bqkv, t + 1);
        SmallMParams o{};
        o.rows = n; o.K = D; o.a_bf16 = W(e->ctx_t); o.w = W(L.wo); o.N = D; o.bias = L.bo; o.out = e->a_f32; o.ldo = D;
        o.resid = e->x_f32;
        if (P) { o.resid_stats = st3; o.resid_g = P->ln3g; o.resid_b = P->ln3b; }
        smallm_gemm<SM_PRO_PLAIN, SM_EPI_SUM>(e, "sm_proj", o);                       // s1 = ctx Wo^T + bo + layer input
        SmallMParams c{};
        c.rows = n; c.K = D; c.a_f32 = e->a_f32; c.ln_g = L.ln1g; c.ln_b = L.ln1b; c.stats_out = st1;
        c.w = W(L.wqc); c.N = D; c.out = e->slabs; c.ldo = D;
        smallm_gemm<SM_PRO_LN, SM_EPI_RAW>(e, "sm_qc", c);
        dec_attn<T, false>(e, l, 1, n, L.bqc, e->S);
        SmallMParams oc{};
        oc.rows = n; oc.K = D; oc.a_bf16 = W(e->ctx_t); oc.w = W(L.woc); oc.N = D; oc.bias = L.boc; oc.out = e->c_f32; oc.ldo = D;
        oc.resid = e->a_f32; oc.resid_stats = st1; oc.resid_g = L.ln1g; oc.resid_b = L.ln1b;
        smallm_gemm<SM_PRO_PLAIN, SM_EPI_SUM>(e, "sm_proj", oc);                      // s2 = ctx Woc^T + boc + LN1(s1)
        SmallMParams f1{};
        f1.rows = n; f1.K = D; f1.a_f32 = e->c_f32; f1.ln_g = L.ln2g; f1.ln_b = L.ln2b; f1.stats_out = st2;
        f1.w = W(L.w1); f1.N = F; f1.bias = L.b1; f1.out = e->h_t; f1.ldo = F;
        smallm_gemm<SM_PRO_LN, SM_EPI_GELU_BF16>(e, "sm_fc1", f1);
        SmallMParams f2{};
        f2.rows = n; f2.K = F; f2.a_bf16 = W(e->h_t); f2.w = W(L.w2); f2.N = D; f2.bias = L.b2; f2.out = e->x_f32; f2.ldo = D;
        f2.resid = e->c_f32; f2.resid_stats = st2; f2.resid_g = L.ln2g; f2.resid_b = L.ln2b;
        smallm_gemm<SM_PRO_PLAIN, SM_EPI_SUM>(e, "sm_fc2", f2);                       // s3 = h W2^T + b2 + LN2(s2)
    }
    const DecLayerW& Z = w.dec[e->cfg.dec_layers - 1];
    SmallMParams tr{};
    tr.rows = n; tr.K = D; tr.a_f32 = e->x_f32; tr.ln_g = Z.ln3g; tr.ln_b = Z.ln3b;
    tr.w = W(w.wt); tr.N = D; tr.bias = w.bt; tr.out = e->a_f32; tr.ldo = D;
    smallm_gemm<SM_PRO_LN, SM_EPI_GELU_F32>(e, "sm_transform", tr);                   // gelu(LN3(s3) Wt^T + bt), pre-LayerNorm
    SmallMParams v{};
    v.rows = n; v.K = D; v.a_f32 = e->a_f32; v.ln_g = w.lntg; v.ln_b = w.lntb;
    v.w = W(w.wv); v.N = e->V; v.out = e->slabs; v.ldo = e->V;
    smallm_gemm<SM_PRO_LN, SM_EPI_RAW>(e, "sm_vocab", v);
    dec_token<T, false>(e, st, 1, n);
}

// One greedy step for n rows; `t` is only used for the profiler's byte estimate.
template <typename T>
void decode_step(mocr_engine* e, const DecState& st, int n, int t) {
    if constexpr (sizeof(T) == 2) {
        if (e->use_smallm(e->rrows(n))) { decode_step_smallm(e, st, n, t); return; }
    }
    const int D = e->D, F = e->F;
    auto& w = e->w;
    const void* xin = e->x_t;
    const float* xres = e->x_f32;
    const int rn = e->rrows(n);          // the row count the kernel choices are made by (the batch's regime)
    for (int l = 0; l < e->cfg.dec_layers; ++l) {
        const DecLayerW& L = w.dec[l];
        int ns;
        const size_t esz = sizeof(T);
        if (e->use_latent(rn)) {
            latent_block(e, true, l, n, t, xin, L.wqkv, L.bqkv, L.wkT_s, reinterpret_cast<const char*>(L.wqkv) + (size_t)2 * D * D * esz,
                         L.bqkv + 2 * D);
        } else {
            ns = dec_gemm<T>(e, "gemm_dec_qkv", xin, D, L.wqkv, 3 * D, D, n);
            dec_attn<T, true>(e, l, ns, n, L.bqkv, t + 1);   // t = step index = keys already cached
        }
        ns = dec_gemm<T>(e, "gemm_dec_proj", e->ctx_t, D, L.wo, D, D, n);
        dec_add_ln<T>(e, ns, D, L.bo, xres, L.ln1g, L.ln1b, e->a_f32, e->a_t, n, false);
        if (e->use_latent(rn)) {
            latent_block(e, false, l, n, t, e->a_t, L.wqc, L.bqc, L.wkT_c,
                         reinterpret_cast<const char*>(w.wckv) + (size_t)(2 * l + 1) * D * D * esz, w.bckv + (2 * l + 1) * D);
        } else {
            ns = dec_gemm<T>(e, "gemm_dec_proj", e->a_t, D, L.wqc, D, D, n);
            dec_attn<T, false>(e, l, ns, n, L.bqc, e->S);
        }
        ns = dec_gemm<T>(e, "gemm_dec_proj", e->ctx_t, D, L.woc, D, D, n);
        dec_add_ln<T>(e, ns, D, L.boc, e->a_f32, L.ln2g, L.ln2b, e->c_f32, e->c_t, n, false);
        if (pick_split(F, D, 128 / (int)sizeof(T), rn, e->slab_cap / e->Bp) == 1) {
            gemm<T>(e, "gemm_dec_fc1", e->c_t, D, L.w1, L.b1, e->h_t, F, nullptr, n, F, D, EPI_BIAS_GELU, dec_launch_tile(e, n), 1);
        } else {
            ns = dec_gemm<T>(e, "gemm_dec_fc1", e->c_t, D, L.w1, F, D, n);
            ProfScope ps(e, "dec_bias_gelu", 0, (double)n * F * (4.0 * ns + sizeof(T)));
            hipLaunchKernelGGL((dec_bias_gelu_kernel<T>), dim3((n * F / 4 + 255) / 256), dim3(256), 0, e->stream, e->slabs, ns,
                               (long long)e->Bp * F, L.b1, reinterpret_cast<T*>(e->h_t), n, F);
            HIPCHECK(hipGetLastError());
        }
        ns = dec_gemm<T>(e, "gemm_dec_fc2", e->h_t, F, L.w2, D, F, n);
        dec_add_ln<T>(e, ns, D, L.b2, e->c_f32, L.ln3g, L.ln3b, e->x_f32, e->x_t, n, false,
                      (e->use_latent(rn) && l + 1 < e->cfg.dec_layers) ? l + 1 : -1);
        xin = e->x_t; xres = e->x_f32;
    }
    int ns = dec_gemm<T>(e, "gemm_dec_proj", e->x_t, D, w.wt, D, D, n);
    dec_add_ln<T>(e, ns, D, w.bt, nullptr, w.lntg, w.lntb, nullptr, e->z_t, n, true);
    // LM head.  When the GEMM is not split over K and nobody asked for the logits, its epilogue reduces every N-tile
    // to (max, column) and the token kernel picks among V/tile candidates: the [n, V] fp32 logits (100 MB at 4096
    // rows) are neither written nor read.  acc + bias is the same fp32 value either way, so the argmax is identical.
    const int vt = dec_launch_tile(e, n);
    if (!st.logits_out && !(e->cfg.flags & MOCR_FLAG_NO_FUSED_ARGMAX) && (vt == 64 || vt == 128) &&
        pick_split(e->V, D, 128 / (int)sizeof(T), rn, e->slab_cap / e->Bp) == 1) {
        gemm<T>(e, "gemm_dec_vocab", e->z_t, D, w.wv, w.bv, e->cand_val, e->V, nullptr, n, e->V, D, EPI_ARGMAX, vt, 1, 0, nullptr, 0,
                nullptr, 0, e->cand_idx);
        dec_token<T, false>(e, st, 1, n, e->V / vt);
    } else {
        ns = dec_gemm<T>(e, "gemm_dec_vocab", e->z_t, D, w.wv, e->V, D, n);
        dec_token<T, false>(e, st, ns, n);
    }
}

template <typename T>
void run_cross_kv(mocr_engine* e, int n) {
    const int M = n * e->S;
    static const int enc_tile_env = env_int("MOCR_ENC_TILE", 0);
    const bool few_blocks = (long long)((M + 127) / 128) * (e->NCKV / 128) * 5 <= 4LL * e->num_cus;      // see run_encoder
    const bool big = sizeof(T) == 2 && (long long)((M + 255) / 256) * (e->NCKV / 256) >= e->num_cus && e->NCKV % 256 == 0;
    const int ET = enc_tile_env ? enc_tile_env : big ? 4096 : few_blocks ? 64 : 128;
    gemm<T>(e, "gemm_cross_kv", e->ENC, e->D, e->w.wckv, e->w.bckv, e->CKV, e->NCKV, nullptr, M, e->NCKV, e->D,
            EPI_BIAS, ET, 1);
}

// fp8 attention: the batch's encoder output as e4m3 rows (static scale), once per batch
void quantize_enc(mocr_engine* e, int n) {
    const long long n16 = (long long)n * e->S * e->D / 16;
    ProfScope ps(e, "quant_enc_fp8", 0, (double)n * e->S * e->D * 3);
    hipLaunchKernelGGL(quant_rows_fp8_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, e->stream,
                       reinterpret_cast<const bf16_t*>(e->ENC), e->enc8, n16, 1.0f / e->w.sx_enc);
    HIPCHECK(hipGetLastError());
}

// Raise the dynamic-LDS limit of every kernel that needs it (done once, outside any capture).
template <typename T> void init_kernel_attrs() {
    constexpr int l128 = 2 * (128 + 128) * 128, l64 = 2 * (64 + 64) * 128;
    set_max_lds(gemm_kernel<T, 128, 128, EPI_SLAB>, l128);
    set_max_lds(gemm_kernel<T, 128, 128, EPI_BIAS>, l128);
    set_max_lds(gemm_kernel<T, 128, 128, EPI_BIAS_GELU>, l128);
    set_max_lds(gemm_kernel<T, 128, 128, EPI_BIAS_RESID>, l128);
    set_max_lds(gemm_kernel<T, 128, 128, EPI_PATCH>, l128);
    set_max_lds(gemm_kernel<T, 128, 128, EPI_BIAS_F32>, l128);
    set_max_lds(gemm_kernel<T, 128, 128, EPI_ARGMAX>, l128);
    set_max_lds(gemm_kernel<T, 64, 64, EPI_SLAB, 2>, l64);
    set_max_lds(gemm_kernel<T, 64, 64, EPI_BIAS, 2>, l64);
    set_max_lds(gemm_kernel<T, 64, 64, EPI_BIAS_GELU, 2>, l64);
    set_max_lds(gemm_kernel<T, 64, 64, EPI_BIAS_RESID, 2>, l64);
    set_max_lds(gemm_kernel<T, 64, 64, EPI_PATCH, 2>, l64);
    set_max_lds(gemm_kernel<T, 64, 64, EPI_BIAS_F32, 2>, l64);
    set_max_lds(gemm_kernel<T, 64, 64, EPI_ARGMAX, 2>, l64);
    set_max_lds(gemm_kernel<T, 128, 128, EPI_SLAB, 4>, 2 * l128);
    set_max_lds(gemm_kernel<T, 128, 128, EPI_BIAS, 4>, 2 * l128);
    set_max_lds(gemm_kernel<T, 128, 128, EPI_BIAS_GELU, 4>, 2 * l128);
    set_max_lds(gemm_kernel<T, 128, 128, EPI_BIAS_RESID, 4>, 2 * l128);
    set_max_lds(gemm_kernel<T, 128, 128, EPI_PATCH, 4>, 2 * l128);
    set_max_lds(gemm_kernel<T, 128, 128, EPI_BIAS_F32, 4>, 2 * l128);
    set_max_lds(gemm_kernel<T, 128, 128, EPI_ARGMAX, 4>, 2 * l128);
    set_max_lds(gemm_kernel<T, 64, 64, EPI_SLAB, 4>, 2 * l64);
    set_max_lds(gemm_kernel<T, 64, 64, EPI_BIAS, 4>, 2 * l64);
    set_max_lds(gemm_kernel<T, 64, 64, EPI_BIAS_GELU, 4>, 2 * l64);
    set_max_lds(gemm_kernel<T, 64, 64, EPI_BIAS_RESID, 4>, 2 * l64);
    set_max_lds(gemm_kernel<T, 64, 64, EPI_PATCH, 4>, 2 * l64);
    set_max_lds(gemm_kernel<T, 64, 64, EPI_BIAS_F32, 4>, 2 * l64);
    set_max_lds(gemm_kernel<T, 64, 64, EPI_ARGMAX, 4>, 2 * l64);
    set_max_lds(enc_attn_simple_kernel<T>, (200 * 65 + 200 * 64 + 4 * 64 + 4 * 256) * 4);
    set_max_lds(enc_attn2_kernel, EA2_LDS);
    set_max_lds(enc_attn_f32_kernel, EAF_LDS);
#ifdef MOCR_EXPERIMENTS
    set_max_lds(enc_attn_mfma_kernel, ENC_SP * 128 + 64 * ENC_VT_LD * 2);
#endif
    set_max_lds(dec_qqt_kernel<128>, 160 * 1024);
    set_max_lds(dec_qqt_kernel<64>, 160 * 1024);
#ifdef MOCR_EXPERIMENTS
    constexpr int l256 = 3 * (256 + 128) * 128;
    set_max_lds(gemm_wide_kernel<EPI_BIAS, 2>, 3 * (256 + 128) * 64);
    set_max_lds(gemm_wide_kernel<EPI_BIAS_GELU, 2>, 3 * (256 + 128) * 64);
    set_max_lds(gemm_wide_kernel<EPI_BIAS_RESID, 2>, 3 * (256 + 128) * 64);
    set_max_lds(gemm_wide_kernel<EPI_BIAS, 4>, 3 * (256 + 256) * 64);
    set_max_lds(gemm_wide_kernel<EPI_BIAS_GELU, 4>, 3 * (256 + 256) * 64);
    set_max_lds(gemm_wide_kernel<EPI_BIAS_RESID, 4>, 3 * (256 + 256) * 64);
    set_max_lds(gemm_wide2_kernel<EPI_BIAS>, 4 * (256 + 256) * 64);
    set_max_lds(gemm_wide2_kernel<EPI_BIAS_GELU>, 4 * (256 + 256) * 64);
    set_max_lds(gemm_wide2_kernel<EPI_BIAS_RESID>, 4 * (256 + 256) * 64);
#endif
    set_max_lds(gemm_pers_kernel<EPI_BIAS_RESID, true, true>, PERS_LDS);
    set_max_lds(gemm_pers_kernel<EPI_BIAS_RESID, true, true, true>, PERS_LDS);
    set_max_lds(gemm_pers_kernel<EPI_BIAS_RESID, true, true, true, true>, PERS_LDS);
    set_max_lds(gemm_pers_kernel<EPI_BIAS_RESID, true, true, false, true>, PERS_LDS);
    set_max_lds(gemm_pers_kernel<EPI_BIAS, true, true>, PERS_LDS);
    set_max_lds(gemm_pers_kernel<EPI_BIAS_GELU, true, true>, PERS_LDS);
    set_max_lds(gemm_pers_kernel<EPI_BIAS, true, true, false, true>, PERS_LDS);
    set_max_lds(gemm_pers_kernel<EPI_BIAS_GELU, true, true, false, true>, PERS_LDS);
    set_max_lds(gemm_pers_kernel<EPI_BIAS, true, true, true, true>, PERS_LDS);
    set_max_lds(gemm_pers_kernel<EPI_BIAS_GELU, true, true, true, true>, PERS_LDS);
    set_max_lds(gemm_pers_kernel<EPI_BIAS, true, true, true>, PERS_LDS);
    set_max_lds(gemm_pers_kernel<EPI_BIAS_GELU, true, true, true>, PERS_LDS);
#ifdef MOCR_EXPERIMENTS
    // (the one-barrier-per-K-tile loop on the 32-deep image: A/B partner of the pair loop)
    set_max_lds(gemm_pers_kernel<EPI_BIAS, true>, PERS_LDS);
    set_max_lds(gemm_pers_kernel<EPI_BIAS_GELU, true>, PERS_LDS);
    set_max_lds(gemm_pers_kernel<EPI_BIAS_RESID, true>, PERS_LDS);
    set_max_lds(gemm_pers_kernel<EPI_BIAS, true, false, false, true>, PERS_LDS);
    set_max_lds(gemm_pers_kernel<EPI_BIAS_GELU, true, false, false, true>, PERS_LDS);
    set_max_lds(gemm_pers_kernel<EPI_BIAS, true, false, true, true>, PERS_LDS);
    set_max_lds(gemm_pers_kernel<EPI_BIAS_GELU, true, false, true, true>, PERS_LDS);
    set_max_lds(gemm_pers_kernel<EPI_BIAS, true, false, true>, PERS_LDS);
    set_max_lds(gemm_pers_kernel<EPI_BIAS_GELU, true, false, true>, PERS_LDS);
    set_max_lds(gemm_pers_kernel<EPI_BIAS_RESID, true, false, true>, PERS_LDS);
    set_max_lds(gemm_pers_kernel<EPI_BIAS, false, true>, PERS_LDS);
    set_max_lds(gemm_pers_kernel<EPI_BIAS_GELU, false, true>, PERS_LDS);
    set_max_lds(gemm_pers_kernel<EPI_BIAS_RESID, false, true>, PERS_LDS);
    set_max_lds(gemm_pers_kernel<EPI_BIAS, false>, PERS_LDS);
    set_max_lds(gemm_pers_kernel<EPI_BIAS_GELU, false>, PERS_LDS);
    set_max_lds(gemm_pers_kernel<EPI_BIAS_RESID, false>, PERS_LDS);
    set_max_lds(gemm256_kernel<EPI_BIAS>, l256);
    set_max_lds(gemm256_kernel<EPI_BIAS_GELU>, l256);
    set_max_lds(gemm256_kernel<EPI_BIAS_RESID>, l256);
    set_max_lds(gemm256_kernel<EPI_PATCH>, l256);
    set_max_lds(gemm256_kernel<EPI_BIAS_F32>, l256);
#endif
    set_max_lds(smallm_gemm_kernel<SM_PRO_LN, SM_EPI_RAW, 2, 3>, SM_LDS(SM_PRO_LN, 2));
    set_max_lds(smallm_gemm_kernel<SM_PRO_LN, SM_EPI_GELU_BF16, 2, 3>, SM_LDS(SM_PRO_LN, 2));
    set_max_lds(smallm_gemm_kernel<SM_PRO_LN, SM_EPI_GELU_F32, 2, 3>, SM_LDS(SM_PRO_LN, 2));
    set_max_lds(latent_attn_kernel<true, 32>, LatCfg<32>::LDS);
    set_max_lds(latent_attn_kernel<false, 32>, LatCfg<32>::LDS);
#ifdef MOCR_EXPERIMENTS
    set_max_lds(latent_attn_kernel<true, 16>, LatCfg<16>::LDS);
    set_max_lds(latent_attn_kernel<false, 16>, LatCfg<16>::LDS);
    set_max_lds(latent_attnT_kernel<true>, LATT_LDS);
    set_max_lds(latent_attnT_kernel<false>, LATT_LDS);
#endif
    set_max_lds((latent_attnT_kernel<true, 2>), LATT_LDS_OF(2));
    set_max_lds((latent_attnT_kernel<false, 2>), LATT_LDS_OF(2));
    set_max_lds(latent_attn_fp8_kernel<true>, LAT8_LDS);
    set_max_lds(latent_attn_fp8_kernel<false>, LAT8_LDS);
    set_max_lds(latent_attnT8_kernel<true>, LATT8_LDS);
    set_max_lds(latent_attnT8_kernel<false>, LATT8_LDS);
}

// `steps` consecutive greedy steps captured once and replayed: every per-step value (position,
// token, finished flags) lives in device memory, so the launch sequence is identical each step.
template <typename T>
hipGraphExec_t decode_graph(mocr_engine* e, const DecState& st, int n, int steps, int t0) {
    // the self-attention variant depends on the context length, so graphs are bucketed by it
    const int need = ((t0 + steps + 3) / 4 + 7) / 8;
    const int bucket = need <= 3 ? 3 : need <= 5 ? 5 : need <= 8 ? 8 : 10;
    const int t_hi = std::min(bucket * 32, st.max_len) - 1;      // largest context this bucket covers
    const auto key = std::make_tuple(e->lane_id, n, st.max_len * 16 + bucket, steps, e->rrows(n));
    auto it = e->graphs.find(key);
    if (it != e->graphs.end()) return it->second;
    hipGraph_t g = nullptr;
    HIPCHECK(hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal));
    try {
        for (int i = 0; i < steps; ++i) decode_step<T>(e, st, n, t_hi - 1);
    } catch (...) {
        (void)hipStreamEndCapture(e->stream, &g);
        if (g) (void)hipGraphDestroy(g);
        throw;
    }
    HIPCHECK(hipStreamEndCapture(e->stream, &g));
    hipGraphExec_t ge = nullptr;
    HIPCHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    HIPCHECK(hipGraphDestroy(g));
    e->graphs[key] = ge;
    return ge;
}

// Teacher-forced decode (test hook): eager launches, logits of every step kept.
template <typename T>
void run_decode_forced(mocr_engine* e, int n, const int* forced, int forced_T, float* logits_out) {
    DecState st = make_state(e, e->cfg.max_len, forced, forced_T, logits_out, n);
    dec_token<T, true>(e, st, 0, n);
    for (int t = 0; t < forced_T; ++t) decode_step<T>(e, st, n, t);
}

// ---------------------------------------------------------------------------------------- scheduler
// A job moves through: start (input staging, encoder, cross-K/V, start token) -> chunks of CHUNK
// greedy steps (one HIP-graph replay each) -> finish (copy ids/lengths out).  After each chunk the
// lane's unfinished-row counter is copied to pinned memory; the flag of chunk c-2 is examined
// before chunk c is enqueued, so a lane always has work queued while the host looks at a flag, and
// a batch whose rows have all emitted EOS stops at most one chunk late.
constexpr int CHUNK = 8;
// Small batches use shorter chunks: their steps are launch-bound (105-165 us whatever the rows), a batch whose rows have
// all finished is noticed (c_f + 2) chunks in, and ordinary speech-bubble texts are 10-30 tokens.  Measured (r02,
// tools/short_text_probe.py, <= 16 rows, every row ending at once): 8-step chunks 3.03 ms, 4-step 2.19 ms, 2-step
// 1.71 ms, i.e. a 20-token batch 4.9 -> 3.9 ms; long rows are unaffected up to 16 rows (106 tokens alone: 14.4 -> 13.7 ms),
// while at 64 rows 2-step chunks cost +3 % on 300-token rows and 4-step chunks nothing.  A replay costs the host
// 10-16 us against >= 210 us of GPU work per chunk.  MOCR_SMALL_CHUNK (with MOCR_SMALL_CHUNK_ROWS): experiments.
static int chunk_steps(int rows) {
    static const int forced = env_int("MOCR_SMALL_CHUNK", 0), forced_rows = env_int("MOCR_SMALL_CHUNK_ROWS", 128);
    if (forced > 0) return rows <= forced_rows ? std::min(forced, CHUNK) : CHUNK;
    return rows <= 16 ? 2 : rows <= 128 ? 4 : CHUNK;
}

// Decode graphs are keyed by row count.  Callers submit any n in 1..max_batch (the batcher of MangaOcr, the crop-job
// queue), so n is rounded up to a coarse grid before it becomes a key: at most ~50 distinct row counts per engine
// instead of max_batch, i.e. a bounded number of captures / instantiated graphs, and a batch of 37 crops replays the
// graph a batch of 40 captured.  The padding costs <= 12.5 % more rows in the (latency-bound) decode steps.
static int graph_rows(int n, int max_batch) {
    int q;
    if (n <= 8) q = 1;
    else if (n <= 64) q = 8;
    else if (n <= 256) q = 32;
    else if (n <= 1024) {
        // (r04: 64 instead of 128 - a 288-row batch decoded on 384 slots.  tools/r04_graphq_ab.sh, isolated batches, 128 / 64:
        // 272-320 rows 100-102 / 95-97 ms, 416-448 rows 113-116 / 109-111, 544 rows 128.6 / 126.6, 700 rows 148 / 145)
        static const int q_mid = env_int("MOCR_GRAPH_Q_MID", 64);
        q = q_mid;
    }
    else q = 256;       // (r04: 512 until rows were compacted - a batch now passes through these row counts on its way down)
    return std::min(round_up(n, q), max_batch);
}

template <typename T>
void start_batch(mocr_engine* e, Lane& L) {
    const int IMG = e->cfg.image_size;
    const size_t plane = (size_t)IMG * IMG;
    int row0 = 0;
    for (const Job& j : L.jobs) {
        uint8_t* gdst = e->d_in + (size_t)row0 * plane;
        if (j.wait_ev) HIPCHECK(hipStreamWaitEvent(e->stream, j.wait_ev, 0));
        if (!j.src_host) {
            HIPCHECK(hipMemcpyAsync(gdst, j.src, plane * j.n, hipMemcpyDeviceToDevice, e->stream));
        } else {
            const size_t rowb = (size_t)IMG * j.channels;
            uint8_t* dst = j.channels == 1 ? gdst : e->d_rgb + (size_t)row0 * plane * 3;
            if (j.row_stride == (int64_t)rowb && j.image_stride == (int64_t)(rowb * IMG)) {
                HIPCHECK(hipMemcpyAsync(dst, j.src, rowb * IMG * j.n, hipMemcpyHostToDevice, e->stream));
            } else {
                for (int i = 0; i < j.n; ++i)
                    HIPCHECK(hipMemcpy2DAsync(dst + (size_t)i * IMG * rowb, rowb, j.src + (size_t)i * j.image_stride, j.row_stride,
                                              rowb, IMG, hipMemcpyHostToDevice, e->stream));
            }
            if (j.channels == 3) {
                const long long npix = (long long)j.n * IMG * IMG;
                hipLaunchKernelGGL(rgb_to_l_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, e->stream, dst, gdst, npix);
                HIPCHECK(hipGetLastError());
            }
        }
        row0 += j.n;
    }
    run_encoder<T>(e, e->d_in, L.n);
    if (!e->use_latent(L.np0)) run_cross_kv<T>(e, L.n);
    else if (e->fp8attn) quantize_enc(e, L.n);
    // rows read pad_id (= 0) beyond what the loop writes
    HIPCHECK(hipMemsetAsync(e->ids, 0, (size_t)L.np * e->cfg.max_len * sizeof(int), e->stream));
    // The decode steps run on np >= n rows (graph_rows): the padding rows are born finished, emit pad_id and read
    // whatever the workspace holds for them (finite values; no kernel mixes rows).
    DecState st = make_state(e, L.max_len, nullptr, 0, nullptr, L.n);
    dec_token<T, true>(e, st, 0, L.np);
    L.t = 0; L.steps = L.max_len - 1; L.chunk = 0;
    L.finishing = false;
    L.flag_pending[0] = L.flag_pending[1] = false;
}

void finish_batch(mocr_engine* e, Lane& L) {
    int row0 = 0;
    for (const Job& j : L.jobs) {
        const hipMemcpyKind kind = j.out_host ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
        HIPCHECK(hipMemcpyAsync(j.out_ids, e->ids + (size_t)row0 * e->cfg.max_len, (size_t)j.n * e->cfg.max_len * sizeof(int), kind, e->stream));
        HIPCHECK(hipMemcpyAsync(j.out_len, e->len + row0, (size_t)j.n * sizeof(int), kind, e->stream));
        row0 += j.n;
    }
    L.jobs.clear();
    L.active = false;
}

// Finished rows stop costing (r04).  `unfinished` is the lane's unfinished-row count as of two chunks ago (the early-exit
// flag): rows only ever finish, so at most that many are unfinished now.  When the decode steps could run on a smaller
// graph-friendly row count, the unfinished rows are moved to the first slots (kernels_decode.h: compact_plan_kernel /
// compact_move_kernel, three small launches between two graph replays) and the next chunks run on that many slots.  The
// batch keeps its kernel regime (mocr_engine::regime = the row count it started with): the attention paths keep different
// caches, and with the GEMM tiles and split-K slabs unchanged a row's ids are bit-identical to an uncompacted run's.
// Worth it from a 1/8 cut: every distinct row count is a decode graph of its own.
template <typename T>
void compact_rows(mocr_engine* e, Lane& L, int unfinished) {
    if (e->cfg.flags & MOCR_FLAG_NO_COMPACTION) return;
    // (batches of the one-launch-per-projection path - <= 32 rows - stay as they are: their steps are launch-bound)
    if (e->use_smallm(L.np0)) return;
    const int want = graph_rows(std::max(unfinished, 1), e->cfg.max_batch);
    if (want >= L.np || (long long)want * 8 > (long long)L.np * 7) return;
    ProfScope ps(e, "compact_rows", 0, (double)want * e->D * (4 + sizeof(T)) * 4);
    int* const new_map = e->rowmap_tmp;
    int* const src_slot = e->rowmap_tmp + e->Bp;
    hipLaunchKernelGGL(compact_plan_kernel, dim3(1), dim3(1024), 0, e->stream, (const int*)e->rowmap, (const int*)e->finished, L.np,
                       new_map, src_slot);
    for (int phase = 0; phase < 2; ++phase)
        hipLaunchKernelGGL((compact_move_kernel<T, 768>), dim3(want), dim3(192), 0, e->stream, phase, (const int*)new_map,
                           (const int*)src_slot, e->x_f32, reinterpret_cast<T*>(e->x_t), e->a_f32, reinterpret_cast<T*>(e->a_t), e->rowmap,
                           want, L.np);
    HIPCHECK(hipGetLastError());
    L.np = want;
    e->n_compactions += 1;
}

// Enqueue the next chunk of lane L (or finish it).  Blocks only on a flag two chunks old.
template <typename T>
void advance(mocr_engine* e, Lane& L) {
    const bool early = !(e->cfg.flags & MOCR_FLAG_NO_EARLY_EXIT);
    const int slot = L.chunk & 1;
    if (early) {
        // The unfinished-row count this chunk is planned with: the flag of the chunk just before it when that has already
        // landed (it has whenever another lane's work ran in between - no wait), otherwise the flag two chunks back, waited
        // for as always (the lane then still has a chunk queued while the host looks).  A fresher count compacts a batch -
        // and ends it - up to a chunk earlier; when the rows move does not change any id (compact_rows).
        int unf = -1;
        const int newer = slot ^ 1;
        if (L.flag_pending[newer]) {
            const hipError_t q = hipEventQuery(L.flag_ev[newer]);
            if (q == hipSuccess) {
                unf = e->h_pinned[newer];
                L.flag_pending[newer] = L.flag_pending[slot] = false;      // (the older flag's event precedes it in the stream)
            } else {
                (void)hipGetLastError();                                   // hipErrorNotReady is not a failure
                if (q != hipErrorNotReady) HIPCHECK(q);
            }
        }
        if (unf < 0 && L.flag_pending[slot]) {
            HIPCHECK(hipEventSynchronize(L.flag_ev[slot]));
            L.flag_pending[slot] = false;
            unf = e->h_pinned[slot];
        }
        if (unf >= 0) {
            if (unf <= 0) { finish_batch(e, L); return; }
            if (unf < L.n) L.finishing = true;
            if (e->D == 768) compact_rows<T>(e, L, unf);
        }
    }
    if (L.t >= L.steps) { finish_batch(e, L); return; }
    DecState st = make_state(e, L.max_len, nullptr, 0, nullptr, L.n);
    // (while rows are leaving, half-length chunks: the count a compaction acts on is at most 4 + 4 steps old instead of 8 + 8;
    // a batch none of whose rows has finished - the synthetic-weights headline - keeps the long chunks)
    const int chunk = L.finishing ? std::min(chunk_steps(L.np), CHUNK / 2) : chunk_steps(L.np);
    const int k = std::min(chunk, L.steps - L.t);
    const bool use_graph = !e->prof_on && !(e->cfg.flags & MOCR_FLAG_NO_GRAPH);
    if (use_graph && k == chunk) {
        HIPCHECK(hipGraphLaunch(decode_graph<T>(e, st, L.np, chunk, L.t), e->stream));
    } else {
        for (int i = 0; i < k; ++i) {
            if (use_graph) HIPCHECK(hipGraphLaunch(decode_graph<T>(e, st, L.np, 1, L.t + i), e->stream));
            else decode_step<T>(e, st, L.np, L.t + i);
        }
    }
    L.t += k;
    e->n_slot_steps += (long long)L.np * k;
    if (early) {
        HIPCHECK(hipMemcpyAsync(e->h_pinned + slot, e->n_unf, sizeof(int), hipMemcpyDeviceToHost, e->stream));
        HIPCHECK(hipEventRecord(L.flag_ev[slot], e->stream));
        L.flag_pending[slot] = true;
    }
    L.chunk += 1;
}

// One scheduling pass over the lanes: an idle lane takes as many pending requests as fit in
// max_batch rows (FIFO, same max_len) and runs them as ONE batch; busy lanes get one chunk.
template <typename T>
bool pump_once(mocr_engine* e) {
    bool any = false;
    for (size_t i = 0; i < e->lanes.size(); ++i) {
        Lane& L = e->lanes[i];
        if (!L.active && !e->pending.empty()) {
            L.jobs.clear();
            L.n = 0;
            L.max_len = e->pending.front().max_len;
            // An idle lane takes as many queued requests as fit in max_batch rows.  When SEVERAL lanes are idle and the
            // queue would fit into fewer of them, it is split evenly over the idle lanes as long as every part keeps
            // >= SPLIT_MIN rows: two fat batches in flight overlap each other's latency-bound phases (+6 % at 2 x 2560
            // against 1 x 5120 rows, r02), while below that merging beats overlapping.
            // (r03: 600 instead of 1024 - the 1250 rows a rank of the 8-GPU queue run gets decoded as 2 x 625 in 258 ms
            // instead of 270 ms as one batch.  r04, with three attention blocks per CU and the fused query kernel from 512
            // rows, tools/r04_probe_lanes.sh: 1250 rows as ONE batch 216-221 ms against 229-231 ms as 2 x 625; 2500 rows 370-378
            // as one against 374-381 as 2 x 1250 (3 x 833: 390); 2 x 2560 against 1 x 5120: +5.5 %.  Parts below ~1280 rows lose.)
            static const long long SPLIT_MIN = env_int("MOCR_SPLIT_MIN", 1280);
            long long rows_pending = 0;
            for (const Job& p : e->pending) rows_pending += p.n;
            int idle = 0;
            for (size_t k = i; k < e->lanes.size(); ++k) idle += e->lanes[k].active ? 0 : 1;
            long long cap = e->cfg.max_batch;
            if (idle > 1 && !e->prof_on && rows_pending < (long long)idle * cap) {     // (instrumented passes: one batch, no overlap)
                const int parts = (int)std::max<long long>(1, std::min<long long>(idle, rows_pending / SPLIT_MIN));
                cap = std::min<long long>(cap, (rows_pending + parts - 1) / parts);
            }
            size_t take = 0;
            while (take < e->pending.size() && e->pending[take].max_len == L.max_len &&
                   (take == 0 || L.n + e->pending[take].n <= cap) && L.n + e->pending[take].n <= e->cfg.max_batch) {
                L.n += e->pending[take].n;
                L.jobs.push_back(e->pending[take]);
                ++take;
            }
            e->pending.erase(e->pending.begin(), e->pending.begin() + take);
            L.np = L.np0 = graph_rows(L.n, e->cfg.max_batch);
            L.active = true;
            e->bind((int)i);
            e->regime = L.np0;
            start_batch<T>(e, L);
            e->regime = 0;
            e->unbind((int)i);
        }
        if (L.active) {
            e->bind((int)i);
            e->regime = L.np0;
            advance<T>(e, L);
            e->regime = 0;
            e->unbind((int)i);
            any = true;
        }
    }
    return any || !e->pending.empty();
}

// Run every submitted job to completion and wait for the GPU.
void drive(mocr_engine* e) {
    try {
        if (e->cfg.dtype == MOCR_BF16) { while (pump_once<bf16_t>(e)) {} }
        else { while (pump_once<float>(e)) {} }
    } catch (...) {
        e->regime = 0;
        e->pending.clear();
        for (auto& L : e->lanes) { L.active = false; L.jobs.clear(); (void)hipStreamSynchronize(L.ctx.stream); }
        throw;
    }
    for (auto& L : e->lanes) HIPCHECK(hipStreamSynchronize(L.ctx.stream));
}

void submit(mocr_engine* e, const Job& j) {
    e->pending.push_back(j);
    long long rows = 0;
    for (const Job& p : e->pending) rows += p.n;
    if (rows >= e->cfg.max_batch) {   // a full batch is waiting: get the GPU going; else wait for more to merge
        if (e->cfg.dtype == MOCR_BF16) pump_once<bf16_t>(e); else pump_once<float>(e);
    }
}

// ---------------------------------------------------------------------------------------- weights
struct Uploader {
    mocr_engine* e;
    const std::vector<float>& get(const std::string& name, std::initializer_list<int64_t> shape) {
        auto it = e->host_w.find(name);
        if (it == e->host_w.end()) throw ArgError{"missing tensor " + name, MOCR_ERR_STATE};
        const auto& shp = e->host_shape[name];
        std::vector<int64_t> want(shape);
        if (shp != want) throw ArgError{"bad shape for tensor " + name, MOCR_ERR_ARG};
        return it->second;
    }
    float* f32(const std::vector<float>& v) {
        float* d = e->dalloc<float>(v.size());
        HIPCHECK(hipMemcpy(d, v.data(), v.size() * 4, hipMemcpyHostToDevice));
        return d;
    }
    void* mat(const std::vector<float>& v) {  // GEMM operand in the engine dtype
        if (e->cfg.dtype == MOCR_F32) return f32(v);
        std::vector<uint16_t> h(v.size());
        for (size_t i = 0; i < v.size(); ++i) h[i] = host_f2bf(v[i]);
        uint16_t* d = e->dalloc<uint16_t>(h.size());
        HIPCHECK(hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        return d;
    }
};

static std::vector<float> concat(std::initializer_list<const std::vector<float>*> parts) {
    std::vector<float> out;
    for (auto* p : parts) out.insert(out.end(), p->begin(), p->end());
    return out;
}

// Is this checkpoint's residual stream one the LayerNorm fold may round to bf16?  Eight probe crops (six of seeded noise, one
// white, one black) go through the encoder with the LayerNorms as launches; in front of each of the 24 foldable LayerNorms
// the stream is copied out and the rounding-noise ratio of calib_observe is taken.  Synthetic N(0, 0.02^2) weights: ~1.0-1.1.
// A stream with a DC offset of several sigma (trained ViTs): several - the fold would cost that factor in GEMM input noise,
// and stays off.  Run once per engine, at commit (~30 ms); mocr_ln_fold_state reports the outcome.
constexpr double FOLD_RATIO_MAX = 1.5;
void calibrate_ln_fold(mocr_engine* e, mocr_engine::FoldCalib& cal) {
    const int n = 8;
    if (e->cfg.max_batch < 56 || e->D != 768 || e->lanes.empty()) return;      // batches this small never take the persistent GEMMs
    const size_t plane = (size_t)e->cfg.image_size * e->cfg.image_size;
    std::vector<uint8_t> probe(n * plane);
    uint32_t lcg = 12345u;
    for (size_t i = 0; i < 6 * plane; ++i) { lcg = lcg * 1664525u + 1013904223u; probe[i] = (uint8_t)(lcg >> 24); }
    std::fill(probe.begin() + 6 * plane, probe.begin() + 7 * plane, (uint8_t)255);
    std::fill(probe.begin() + 7 * plane, probe.end(), (uint8_t)0);
    e->bind(0);
    HIPCHECK(hipMemcpyAsync(e->d_in, probe.data(), probe.size(), hipMemcpyHostToDevice, e->stream));
    cal.idx = 0; cal.worst = 0.0;
    e->calib = &cal;
    try {
        run_encoder<bf16_t>(e, e->d_in, n);
        HIPCHECK(hipStreamSynchronize(e->stream));
    } catch (...) {
        e->calib = nullptr;
        e->unbind(0);
        throw;
    }
    e->calib = nullptr;
    e->unbind(0);
    e->fold_ratio = (float)cal.worst;
    e->fold_ok = cal.worst <= FOLD_RATIO_MAX;
}

void commit_weights(mocr_engine* e) {
    const auto& c = e->cfg;
    const int64_t D = c.hidden, F = c.ffn, V = c.vocab, P = c.patch_size, S = e->S;
    Uploader up{e};
    mocr_engine::FoldCalib ln_gb;
    auto& w = e->w;
    // pixel LUT, exactly the HF image processor's arithmetic (float64 rescale, float32 normalise)
    {
        std::vector<float> lut(256);
        for (int u = 0; u < 256; ++u) {
            const float x = (float)((double)u * (1.0 / 255.0));
            lut[u] = (x - 0.5f) / 0.5f;
        }
        w.lut = up.f32(lut);
    }
    // patch embedding: the three input channels are identical, so sum the kernel over channels
    {
        const auto& pw = up.get("encoder.embeddings.patch_embeddings.projection.weight", {D, 3, P, P});
        std::vector<float> ws((size_t)D * P * P);
        for (int64_t o = 0; o < D; ++o)
            for (int64_t k = 0; k < P * P; ++k) {
                double s = 0;
                for (int ch = 0; ch < 3; ++ch) s += pw[(o * 3 + ch) * P * P + k];
                ws[o * P * P + k] = (float)s;
            }
        w.wpe = up.mat(ws);
        w.bpe = up.f32(up.get("encoder.embeddings.patch_embeddings.projection.bias", {D}));
        w.cls = up.f32(up.get("encoder.embeddings.cls_token", {1, 1, D}));
        w.pos_enc = up.f32(up.get("encoder.embeddings.position_embeddings", {1, S, D}));
    }
    w.enc.resize(c.enc_layers);
    for (int l = 0; l < c.enc_layers; ++l) {
        const std::string p = "encoder.layers." + std::to_string(l) + ".";
        EncLayerW& L = w.enc[l];
        L.wqkv = up.mat(concat({&up.get(p + "attention.q_proj.weight", {D, D}), &up.get(p + "attention.k_proj.weight", {D, D}),
                                &up.get(p + "attention.v_proj.weight", {D, D})}));
        L.bqkv = up.f32(concat({&up.get(p + "attention.q_proj.bias", {D}), &up.get(p + "attention.k_proj.bias", {D}),
                                &up.get(p + "attention.v_proj.bias", {D})}));
        L.wo = up.mat(up.get(p + "attention.o_proj.weight", {D, D}));
        L.bo = up.f32(up.get(p + "attention.o_proj.bias", {D}));
        L.ln1g = up.f32(up.get(p + "layernorm_before.weight", {D}));
        L.ln1b = up.f32(up.get(p + "layernorm_before.bias", {D}));
        L.ln2g = up.f32(up.get(p + "layernorm_after.weight", {D}));
        L.ln2b = up.f32(up.get(p + "layernorm_after.bias", {D}));
        L.w1 = up.mat(up.get(p + "mlp.fc1.weight", {F, D}));
        L.b1 = up.f32(up.get(p + "mlp.fc1.bias", {F}));
        L.w2 = up.mat(up.get(p + "mlp.fc2.weight", {D, F}));
        L.b2 = up.f32(up.get(p + "mlp.fc2.bias", {D}));
        if (c.dtype == MOCR_BF16) {
            // LN(x) W^T + b = rstd (x (W o gamma)^T - mean colsum) + (b + W beta): the folded operands (colsum of the ROUNDED
            // folded weight - what the MFMAs multiply -, in double)
            auto fold = [&](const std::vector<float>& W, const std::vector<float>& b, const std::vector<float>& g, const std::vector<float>& be,
                            int64_t N, void*& wf, float*& cs, float*& bf) {
                std::vector<float> Wf((size_t)N * D), csum(N), bias(N);
                for (int64_t n = 0; n < N; ++n) {
                    double sc = 0.0, sb = 0.0;
                    for (int64_t k = 0; k < D; ++k) {
                        const float v = W[n * D + k] * g[k];
                        Wf[n * D + k] = v;
                        sc += (double)host_bf2f(host_f2bf(v));
                        sb += (double)W[n * D + k] * (double)be[k];
                    }
                    csum[n] = (float)sc;
                    bias[n] = (float)((double)b[n] + sb);
                }
                wf = up.mat(Wf); cs = up.f32(csum); bf = up.f32(bias);
            };
            const auto wqkv_h = concat({&up.get(p + "attention.q_proj.weight", {D, D}), &up.get(p + "attention.k_proj.weight", {D, D}),
                                        &up.get(p + "attention.v_proj.weight", {D, D})});
            const auto bqkv_h = concat({&up.get(p + "attention.q_proj.bias", {D}), &up.get(p + "attention.k_proj.bias", {D}),
                                        &up.get(p + "attention.v_proj.bias", {D})});
            fold(wqkv_h, bqkv_h, up.get(p + "layernorm_before.weight", {D}), up.get(p + "layernorm_before.bias", {D}), 3 * D,
                 L.wqkv_f, L.sqkv, L.bqkv_f);
            ln_gb.g.push_back(up.get(p + "layernorm_before.weight", {D})); ln_gb.b.push_back(up.get(p + "layernorm_before.bias", {D}));
            ln_gb.g.push_back(up.get(p + "layernorm_after.weight", {D})); ln_gb.b.push_back(up.get(p + "layernorm_after.bias", {D}));
            fold(up.get(p + "mlp.fc1.weight", {F, D}), up.get(p + "mlp.fc1.bias", {F}), up.get(p + "layernorm_after.weight", {D}),
                 up.get(p + "layernorm_after.bias", {D}), F, L.w1_f, L.s1, L.b1_f);
        }
    }
    w.lnfg = up.f32(up.get("encoder.layernorm.weight", {D}));
    w.lnfb = up.f32(up.get("encoder.layernorm.bias", {D}));
    // Static e4m3 scale of a LayerNorm output: |gamma_k z_k + beta_k| <= max|gamma| sqrt(D - 1) + max|beta| for ANY input
    // (a normalised vector's element is at most sqrt(D - 1)), mapped onto e4m3's largest finite value 448.
    auto ln_scale = [&](const std::string& gname, const std::string& bname) {
        float gm = 0.f, bm = 0.f;
        for (float v : up.get(gname, {D})) gm = std::max(gm, std::fabs(v));
        for (float v : up.get(bname, {D})) bm = std::max(bm, std::fabs(v));
        return (gm * std::sqrt((float)(D - 1)) + bm) / 448.0f;
    };
    w.sx_enc = ln_scale("encoder.layernorm.weight", "encoder.layernorm.bias");
    w.sx_self.assign(c.dec_layers, 1.f);
    w.sx_self[0] = ln_scale("decoder.bert.embeddings.LayerNorm.weight", "decoder.bert.embeddings.LayerNorm.bias");
    for (int l = 1; l < c.dec_layers; ++l) {
        const std::string pp = "decoder.bert.encoder.layer." + std::to_string(l - 1) + ".output.LayerNorm.";
        w.sx_self[l] = ln_scale(pp + "weight", pp + "bias");
    }
    const std::string d = "decoder.bert.";
    w.word = up.f32(up.get(d + "embeddings.word_embeddings.weight", {V, D}));
    w.posd = up.f32(up.get(d + "embeddings.position_embeddings.weight", {(int64_t)c.max_pos, D}));
    {
        const auto& tt = e->host_w.at(d + "embeddings.token_type_embeddings.weight");
        std::vector<float> t0(tt.begin(), tt.begin() + D);
        w.type0 = up.f32(t0);
    }
    w.embg = up.f32(up.get(d + "embeddings.LayerNorm.weight", {D}));
    w.embb = up.f32(up.get(d + "embeddings.LayerNorm.bias", {D}));
    w.dec.resize(c.dec_layers);
    std::vector<float> ckv_w, ckv_b;
    for (int l = 0; l < c.dec_layers; ++l) {
        const std::string p = d + "encoder.layer." + std::to_string(l) + ".";
        DecLayerW& L = w.dec[l];
        const std::string a = p + "attention.", x = p + "crossattention.";
        L.wqkv = up.mat(concat({&up.get(a + "self.query.weight", {D, D}), &up.get(a + "self.key.weight", {D, D}),
                                &up.get(a + "self.value.weight", {D, D})}));
        L.bqkv = up.f32(concat({&up.get(a + "self.query.bias", {D}), &up.get(a + "self.key.bias", {D}),
                                &up.get(a + "self.value.bias", {D})}));
        if (e->latent) {
            auto transposed_scaled = [&](const std::vector<float>& wk) {
                std::vector<float> t((size_t)D * D);
                for (int64_t o = 0; o < D; ++o)
                    for (int64_t i = 0; i < D; ++i) t[i * D + o] = wk[o * D + i] * 0.125f;   // [in][out], 1/sqrt(64) folded in
                return t;
            };
            L.wkT_s = up.mat(transposed_scaled(up.get(a + "self.key.weight", {D, D})));
            L.wkT_c = up.mat(transposed_scaled(up.get(x + "self.key.weight", {D, D})));
        }
        L.wo = up.mat(up.get(a + "output.dense.weight", {D, D}));
        L.bo = up.f32(up.get(a + "output.dense.bias", {D}));
        L.ln1g = up.f32(up.get(a + "output.LayerNorm.weight", {D}));
        L.ln1b = up.f32(up.get(a + "output.LayerNorm.bias", {D}));
        L.wqc = up.mat(up.get(x + "self.query.weight", {D, D}));
        L.bqc = up.f32(up.get(x + "self.query.bias", {D}));
        for (const char* kv : {"self.key.", "self.value."}) {
            const auto& ww = up.get(x + kv + "weight", {D, D});
            const auto& bb = up.get(x + kv + "bias", {D});
            ckv_w.insert(ckv_w.end(), ww.begin(), ww.end());
            ckv_b.insert(ckv_b.end(), bb.begin(), bb.end());
        }
        L.woc = up.mat(up.get(x + "output.dense.weight", {D, D}));
        L.boc = up.f32(up.get(x + "output.dense.bias", {D}));
        L.ln2g = up.f32(up.get(x + "output.LayerNorm.weight", {D}));
        L.ln2b = up.f32(up.get(x + "output.LayerNorm.bias", {D}));
        L.w1 = up.mat(up.get(p + "intermediate.dense.weight", {F, D}));
        L.b1 = up.f32(up.get(p + "intermediate.dense.bias", {F}));
        L.w2 = up.mat(up.get(p + "output.dense.weight", {D, F}));
        L.b2 = up.f32(up.get(p + "output.dense.bias", {D}));
        L.ln3g = up.f32(up.get(p + "output.LayerNorm.weight", {D}));
        L.ln3b = up.f32(up.get(p + "output.LayerNorm.bias", {D}));
    }
    w.wckv = up.mat(ckv_w);
    w.bckv = up.f32(ckv_b);
    w.zero_bias = up.f32(std::vector<float>((size_t)D, 0.f));
    const std::string cl = "decoder.cls.predictions.";
    w.wt = up.mat(up.get(cl + "transform.dense.weight", {D, D}));
    w.bt = up.f32(up.get(cl + "transform.dense.bias", {D}));
    w.lntg = up.f32(up.get(cl + "transform.LayerNorm.weight", {D}));
    w.lntb = up.f32(up.get(cl + "transform.LayerNorm.bias", {D}));
    w.wv = up.mat(up.get(cl + "decoder.weight", {V, D}));
    w.bv = up.f32(up.get(cl + "decoder.bias", {V}));
    if (c.dtype == MOCR_BF16) init_kernel_attrs<bf16_t>(); else init_kernel_attrs<float>();
    e->host_w.clear();
    e->host_shape.clear();
    e->committed = true;
    if (c.dtype == MOCR_BF16 && !ln_gb.g.empty()) calibrate_ln_fold(e, ln_gb);
}

void compute_geometry(mocr_engine* e) {
    const auto& c = e->cfg;
    e->S = (c.image_size / c.patch_size) * (c.image_size / c.patch_size) + 1;
    e->G = c.image_size / c.patch_size;
    e->D = c.hidden; e->H = c.heads; e->F = c.ffn; e->V = c.vocab;
    e->esz = c.dtype == MOCR_BF16 ? 2 : 4;
    e->Bp = round_up(c.max_batch, 128);
    e->Mp = round_up(c.max_batch * e->S, 256) + 256;
    e->NCKV = c.dec_layers * 2 * c.hidden;
}

// Workspace of ONE lane, allocated into the engine's bound LaneCtx (then saved with unbind()).
void allocate_lane(mocr_engine* e, int lane_id) {
    const auto& c = e->cfg;
    static_cast<LaneCtx&>(*e) = LaneCtx{};
    e->lane_id = lane_id;
    HIPCHECK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    const size_t Mp = e->Mp, D = e->D, Bp = e->Bp, esz = e->esz;
    e->d_in = e->dalloc<uint8_t>((size_t)c.max_batch * c.image_size * c.image_size);
    e->d_rgb = e->dalloc<uint8_t>((size_t)c.max_batch * c.image_size * c.image_size * 3);
    e->X = e->dalloc<float>(Mp * D);
    e->Xn = e->dalloc<char>(Mp * D * esz);
    if (c.dtype == MOCR_BF16) {
        e->ln_part = e->dalloc<float>(Mp * 8);
        HIPCHECK(hipMemset(e->ln_part, 0, Mp * 8 * sizeof(float)));      // (partial 3 stays zero: N = 768 has three column slices)
    }
    e->QKV = e->dalloc<char>(Mp * 3 * D * esz);
    e->CTX = e->dalloc<char>(Mp * D * esz);
    e->Hb = e->dalloc<char>(Mp * (size_t)e->F * esz);
    e->ENC = e->dalloc<char>(Mp * D * esz);
    if (e->latent && e->classic_rows < c.max_batch) {      // else every batch takes the classic kernels
        e->q_t = e->dalloc<char>(Bp * D * esz);
        e->qt = e->dalloc<char>(Bp * 16 * D * esz);
        e->et = e->dalloc<char>(Bp * 16 * D * esz);
        if (!e->fp8attn) e->xcache = e->dalloc<char>(((size_t)c.dec_layers * Bp * c.max_len + 64) * D * esz);
        else {
            e->x8cache = e->dalloc<uint8_t>(((size_t)c.dec_layers * Bp * c.max_len + 64) * D);
            e->enc8 = e->dalloc<uint8_t>((Mp + 64) * D);
        }
    }
    if (e->Bc > 0) {      // classic K/V: the whole engine (fp32 / MOCR_FLAG_CLASSIC_ATTENTION) or its small batches
        const size_t Mc = (size_t)round_up(e->Bc * e->S, 256) + 256;
        e->CKV = e->dalloc<char>(Mc * (size_t)e->NCKV * esz);
        const size_t cache = (size_t)c.dec_layers * e->Bc * e->H * c.max_len * 64 * esz;
        e->kcache = e->dalloc<char>(cache);
        e->vcache = e->dalloc<char>(cache);
    }
    e->slab_cap = (long long)Bp * 12288;
    e->slabs = e->dalloc<float>((size_t)e->slab_cap);
    e->cand_val = e->dalloc<float>((size_t)Bp * (e->V / 64)); e->cand_idx = e->dalloc<int>((size_t)Bp * (e->V / 64));
    e->x_f32 = e->dalloc<float>(Bp * D); e->a_f32 = e->dalloc<float>(Bp * D); e->c_f32 = e->dalloc<float>(Bp * D);
    e->ln_stats = e->dalloc<float>(3 * Bp * 2);
    e->x_t = e->dalloc<char>(Bp * D * esz); e->a_t = e->dalloc<char>(Bp * D * esz); e->c_t = e->dalloc<char>(Bp * D * esz);
    e->ctx_t = e->dalloc<char>(Bp * D * esz); e->z_t = e->dalloc<char>(Bp * D * esz);
    e->h_t = e->dalloc<char>(Bp * (size_t)e->F * esz);
    e->ids = e->dalloc<int>(Bp * (size_t)c.max_len);
    e->step = e->dalloc<int>(Bp); e->finished = e->dalloc<int>(Bp); e->len = e->dalloc<int>(Bp); e->n_unf = e->dalloc<int>(4);
    e->rowmap = e->dalloc<int>(Bp); e->rowmap_tmp = e->dalloc<int>(2 * Bp);
    HIPCHECK(hipHostMalloc(reinterpret_cast<void**>(&e->h_pinned), 64, hipHostMallocDefault));
}

void allocate_lanes(mocr_engine* e) {
    compute_geometry(e);
    e->latent = e->cfg.dtype == MOCR_BF16 && !(e->cfg.flags & MOCR_FLAG_CLASSIC_ATTENTION);
    e->fp8attn = e->latent && (e->cfg.flags & MOCR_FLAG_FP8_ATTENTION);
    e->lat_tk = (e->cfg.flags & MOCR_FLAG_LATENT_TILE32) ? 32 : env_int("MOCR_LAT_TK", 18);
    // Small batches of a latent engine take the classic kernels: the persistent latent kernel walks a sequence's key
    // tiles serially on ONE CU (~20 us per call whatever the batch), the classic one spreads a row over 12 blocks.
    // Measured (r01, 300 tokens): 8 rows 36 vs 73 ms, 64 rows 50 vs 80 ms, 256 rows 111 vs 116 ms; r02, both paths with
    // non-temporal key loads: 320 rows 116 vs 126 ms, 384 rows 119 vs 129 ms, 448 / 512 rows (one graph shape) 159 vs 148 ms;
    // r04, latent on three blocks per CU (tools/r04_classic_rows_ab.sh, isolated batch, classic / latent, ms): 192 rows 76.8 /
    // 90.2, 256 rows 90.3 / 96.1, 320 rows 113.3 / 107.7, 384 rows 116.1 / 110.5 - the switch moved from 384 to 256 rows.
    e->classic_rows = !e->latent ? 0 : (e->cfg.flags & MOCR_FLAG_LATENT_ALWAYS) ? 0 : std::min(env_int("MOCR_CLASSIC_ROWS", 256), e->cfg.max_batch);
    e->Bc = e->latent ? e->classic_rows : e->Bp;
    e->smallm_rows = (e->cfg.dtype == MOCR_BF16 && e->D == 768 && e->F == 3072 && e->V % SM_NT == 0 && !(e->cfg.flags & MOCR_FLAG_NO_SMALL_BATCH_PATH))
                         ? std::min(SM_MAX_ROWS, env_int("MOCR_SMALLM_ROWS", SM_MAX_ROWS)) : 0;
    const int nl = std::max(1, std::min(16, (int)e->cfg.lanes));
    e->lanes.resize(nl);
    for (int i = 0; i < nl; ++i) {
        allocate_lane(e, i);
        e->unbind(i);
        HIPCHECK(hipEventCreateWithFlags(&e->lanes[i].flag_ev[0], hipEventDisableTiming));
        HIPCHECK(hipEventCreateWithFlags(&e->lanes[i].flag_ev[1], hipEventDisableTiming));
    }
    e->bind(0);
    HIPCHECK(hipStreamCreateWithFlags(&e->prep_stream, hipStreamNonBlocking));
}

static void set_error(mocr_engine* e, const std::string& msg) {
    std::lock_guard<std::mutex> lk(e->err_mu);
    e->err = msg;
}

template <typename F> int guarded(mocr_engine* e, F&& f) {
    if (!e) return MOCR_ERR_ARG;
    if (e->poisoned.load(std::memory_order_acquire)) {
        // after a failed HIP call (a fault, a lost device) the context is not trustworthy and HIP keeps
        // returning the same error: refuse instead of serving from a half-dead engine
        return MOCR_ERR_STATE;
    }
    try {
        f();
        return MOCR_OK;
    } catch (const HipError& h) {
        char buf[512];
        snprintf(buf, sizeof(buf), "HIP error %d (%s) at engine.hip:%d: %s - the engine refuses further calls (MOCR_ERR_STATE) until it is destroyed",
                 (int)h.code, hipGetErrorString(h.code), h.line, h.what);
        set_error(e, buf);
        e->poisoned.store(true, std::memory_order_release);
        return MOCR_ERR_HIP;
    } catch (const ArgError& a) {
        set_error(e, a.msg);
        return a.code;
    } catch (const std::bad_alloc&) {
        set_error(e, "host allocation failed");
        return MOCR_ERR_NOMEM;
    } catch (const std::exception& x) {
        set_error(e, x.what());
        return MOCR_ERR_STATE;
    }
}

void require_ready(mocr_engine* e, int n, bool bounded = true) {
    if (!e->committed) throw ArgError{"weights not committed (mocr_commit_weights)", MOCR_ERR_STATE};
    if (n <= 0) throw ArgError{"n must be positive", MOCR_ERR_ARG};
    if (bounded && n > e->cfg.max_batch) throw ArgError{"n exceeds max_batch", MOCR_ERR_ARG};
}

// run fn with T = storage type of the engine: fn(bf16_t{}) or fn(float{})
template <typename Fn> void dispatch(mocr_engine* e, Fn&& fn) {
    if (e->cfg.dtype == MOCR_BF16) fn(bf16_t{}); else fn(float{});
}

}  // namespace

// ============================================================================================ C ABI
extern "C" {

int mocr_abi_version(void) { return MOCR_ABI_VERSION; }

int mocr_create(const mocr_config* cfg, mocr_engine** out) {
    if (!cfg || !out || cfg->struct_size != (int32_t)sizeof(mocr_config)) return MOCR_ERR_ARG;
    if (cfg->hidden != 768 || cfg->heads != 12 || cfg->image_size != 224 || cfg->patch_size != 16 || cfg->ffn % 128 ||
        cfg->vocab != 6144 || cfg->max_len < 2 || cfg->max_len > 320 || cfg->max_len > cfg->max_pos || cfg->max_batch < 1 ||
        (cfg->dtype != MOCR_F32 && cfg->dtype != MOCR_BF16) || cfg->enc_layers < 1 || cfg->dec_layers < 1 ||
        cfg->pad_id != 0 || cfg->lanes < 0)
        return MOCR_ERR_UNSUPPORTED;
    mocr_engine* e = new (std::nothrow) mocr_engine();
    if (!e) return MOCR_ERR_NOMEM;
    e->cfg = *cfg;
    e->gen_max_len = cfg->max_len;
    if (e->cfg.lanes == 0) e->cfg.lanes = 1;
    int rc = guarded(e, [&] {
        int ndev = 0;
        HIPCHECK(hipGetDeviceCount(&ndev));
        if (ndev <= 0) throw ArgError{"no HIP device visible: the Manga-OCR engine needs a GPU", MOCR_ERR_HIP};
        HIPCHECK(hipSetDevice(cfg->device));
        {
            int cus = 0;
            HIPCHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, cfg->device));
            e->num_cus = cus >= 8 ? cus / 8 * 8 : 8;
        }
        allocate_lanes(e);
    });
    if (rc != MOCR_OK) {
        fprintf(stderr, "mocr_create failed: %s\n", mocr_last_error(e));
        mocr_destroy(e);
        return rc;
    }
    *out = e;
    return MOCR_OK;
}

void mocr_destroy(mocr_engine* e) {
    if (!e) return;
    (void)hipSetDevice(e->cfg.device);
    for (auto& L : e->lanes)
        if (L.ctx.stream) (void)hipStreamSynchronize(L.ctx.stream);
    for (auto& r : e->recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    for (auto ev : e->ev_pool) (void)hipEventDestroy(ev);
    for (auto* sc : {&e->rs_src, &e->rs_tmp, &e->rs_desc, &e->rs_coef, &e->rs_bounds, &e->rs_gray})
        if (sc->p) (void)hipFree(sc->p);
    for (auto& sp : e->rs_pin)
        if (sp.p) (void)hipHostFree(sp.p);
    if (e->prep_stream) (void)hipStreamDestroy(e->prep_stream);
    for (auto& kv : e->graphs) (void)hipGraphExecDestroy(kv.second);
    for (void* p : e->allocs) (void)hipFree(p);
    for (auto& L : e->lanes) {
        if (L.ctx.h_pinned) (void)hipHostFree(L.ctx.h_pinned);
        for (auto ev : L.flag_ev) if (ev) (void)hipEventDestroy(ev);
        if (L.ctx.stream) (void)hipStreamDestroy(L.ctx.stream);
    }
    delete e;
}

const char* mocr_last_error(const mocr_engine* e) {
    if (!e) return "null engine";
    // a copy taken under the error mutex, owned by the calling thread: valid until that thread's next call of this
    // function, whatever other threads do to the engine meanwhile
    static thread_local std::string copy;
    mocr_engine* m = const_cast<mocr_engine*>(e);
    std::lock_guard<std::mutex> lk(m->err_mu);
    copy = m->err;
    return copy.c_str();
}

int mocr_set_tensor(mocr_engine* e, const char* name, const float* data, const int64_t* shape, int32_t ndim) {
    return guarded(e, [&] {
        if (!name || !data || !shape || ndim < 1 || ndim > 4) throw ArgError{"mocr_set_tensor: bad argument", MOCR_ERR_ARG};
        if (e->committed) throw ArgError{"weights already committed", MOCR_ERR_STATE};
        std::lock_guard<std::mutex> lk(e->mu);
        int64_t count = 1;
        for (int i = 0; i < ndim; ++i) count *= shape[i];
        e->host_w[name].assign(data, data + count);
        e->host_shape[name].assign(shape, shape + ndim);
    });
}

int mocr_commit_weights(mocr_engine* e) {
    return guarded(e, [&] {
        std::lock_guard<std::mutex> lk(e->mu);
        if (e->committed) throw ArgError{"weights already committed", MOCR_ERR_STATE};
        HIPCHECK(hipSetDevice(e->cfg.device));
        commit_weights(e);
    });
}

void* mocr_stream(mocr_engine* e) { return (e && !e->lanes.empty()) ? (void*)e->lanes[0].ctx.stream : nullptr; }

int mocr_synchronize(mocr_engine* e) {
    return guarded(e, [&] {
        std::lock_guard<std::mutex> lk(e->mu);
        HIPCHECK(hipSetDevice(e->cfg.device));
        drive(e);
    });
}

int mocr_recognize_device(mocr_engine* e, const void* d_gray, int32_t n, void* d_out_ids, void* d_out_len) {
    return guarded(e, [&] {
        std::lock_guard<std::mutex> lk(e->mu);
        require_ready(e, n);
        if (!d_gray || !d_out_ids || !d_out_len) throw ArgError{"null device pointer", MOCR_ERR_ARG};
        HIPCHECK(hipSetDevice(e->cfg.device));
        Job j;
        j.src = reinterpret_cast<const uint8_t*>(d_gray); j.src_host = false;
        j.n = n; j.max_len = e->gen_max_len;
        j.out_ids = reinterpret_cast<int32_t*>(d_out_ids); j.out_len = reinterpret_cast<int32_t*>(d_out_len); j.out_host = false;
        submit(e, j);
    });
}

static void recognize_host_chunks(mocr_engine* e, const uint8_t* images, int n, int h, int w, int64_t row_stride,
                                  int64_t image_stride, int channels, int max_len, int32_t* out_ids, int32_t* out_len) {
    const int IMG = e->cfg.image_size;
    if (h != IMG || w != IMG)
        throw ArgError{"crops must be image_size x image_size (resize with PIL BILINEAR on the caller side)", MOCR_ERR_UNSUPPORTED};
    if (channels != 1 && channels != 3) throw ArgError{"channels must be 1 (L) or 3 (RGB)", MOCR_ERR_ARG};
    if (!images || !out_ids || !out_len) throw ArgError{"null pointer", MOCR_ERR_ARG};
    if (max_len < 2 || max_len > e->cfg.max_len) throw ArgError{"bad max_len override", MOCR_ERR_ARG};
    for (int base = 0; base < n; base += e->cfg.max_batch) {
        Job j;
        j.src = images + (size_t)base * image_stride; j.src_host = true; j.channels = channels;
        j.row_stride = row_stride; j.image_stride = image_stride;
        j.n = std::min(e->cfg.max_batch, n - base); j.max_len = max_len;
        j.out_ids = out_ids + (size_t)base * e->cfg.max_len; j.out_len = out_len + base; j.out_host = true;
        e->pending.push_back(j);
    }
    drive(e);
}

int mocr_recognize(mocr_engine* e, const uint8_t* images, int32_t n, int32_t h, int32_t w, int64_t row_stride,
                   int64_t image_stride, int32_t channels, int32_t* out_ids, int32_t* out_len) {
    return guarded(e, [&] {
        std::lock_guard<std::mutex> lk(e->mu);
        require_ready(e, n, false);
        HIPCHECK(hipSetDevice(e->cfg.device));
        recognize_host_chunks(e, images, n, h, w, row_stride, image_stride, channels, e->gen_max_len, out_ids, out_len);
    });
}

int mocr_recognize_gray_host(mocr_engine* e, const uint8_t* gray, int32_t n, int32_t max_len_override, int32_t* out_ids,
                             int32_t* out_len) {
    return guarded(e, [&] {
        std::lock_guard<std::mutex> lk(e->mu);
        require_ready(e, n, false);
        HIPCHECK(hipSetDevice(e->cfg.device));
        const int IMG = e->cfg.image_size;
        recognize_host_chunks(e, gray, n, IMG, IMG, IMG, (int64_t)IMG * IMG, 1, max_len_override, out_ids, out_len);
    });
}

// Host pixels uploaded once (an image, or a whole page several crops are cut from) ...
struct PrepSource { const uint8_t* data; int h, w; int64_t row_stride; int ch; int bgr; int rot; };
// ... and what one crop reads of its source: the rectangle [x, x+w) x [y, y+h), then rotated (MOCR_ROTATE_*)
struct PrepView { int source, x, y, w, h, rot; };

static PrepSource source_of(const mocr_image& im) {
    const int ch = im.channels == MOCR_CHANNELS_BGR ? 3 : im.channels;
    if (!im.data || im.height < 1 || im.width < 1 || im.height > 16384 || im.width > 16384 || (ch != 1 && ch != 3) ||
        im.row_stride < (int64_t)im.width * ch)
        throw ArgError{"bad image descriptor (channels must be 1 = L, 3 = RGB or MOCR_CHANNELS_BGR)", MOCR_ERR_ARG};
    if (im.rotate != MOCR_ROTATE_NONE && im.rotate != MOCR_ROTATE_90_CW && im.rotate != MOCR_ROTATE_90_CCW)
        throw ArgError{"bad image descriptor (rotate must be MOCR_ROTATE_NONE, _90_CW or _90_CCW)", MOCR_ERR_ARG};
    return PrepSource{im.data, im.height, im.width, im.row_stride, ch, im.channels == MOCR_CHANNELS_BGR ? 1 : 0, im.rotate};
}

// L conversion + Pillow-exact BILINEAR resize of the views (any sizes) into d_out [views][IMG][IMG] u8 on the
// device; synchronous (lane 0's stream).  Every source is uploaded once, however many views read it: the crops of
// a page's detected regions are cut ON THE DEVICE (descriptor = offset + page stride), not copied out on the host.
// prep_enqueue packs the sources into pinned buffer `slot`, and enqueues H2D + the two resize launches on the engine's
// preparation stream; it does NOT wait for them.  Device scratch (packed sources, horizontal-pass planes, descriptors,
// coefficient tables) is reused from call to call: everything runs on the ONE preparation stream, in order.  The pinned
// buffer of a slot may be repacked only once the copy that reads it has finished (the caller's business).
struct PrepHold { std::vector<ResizeDesc> descs; std::vector<int> coef, bounds; };     // host sources of a pass's small uploads: alive until it has run
static void prep_enqueue(mocr_engine* e, const std::vector<PrepSource>& srcs, const PrepView* views, int n, uint8_t* d_out, int slot, bool prof,
                         PrepHold& hold) {
    const int IMG = e->cfg.image_size;
    if (n > 4096) throw ArgError{"prep_enqueue: at most 4096 crops per pass", MOCR_ERR_ARG};   // bounded scratch and grid.y
    if (IMG != 224) throw ArgError{"device preprocessing is instantiated for image_size 224", MOCR_ERR_UNSUPPORTED};
    std::vector<ResizeDesc>& descs = hold.descs;
    std::vector<int>&coef = hold.coef, &bounds = hold.bounds;
    descs.assign(n, ResizeDesc{});
    coef.clear(); bounds.clear();
    std::map<int, std::pair<int, int>> placed;        // input size -> (coef offset, bounds offset) in this call's buffers
    std::map<int, long long> src_off;                 // sources this pass reads -> byte offset in the packed upload
    size_t src_bytes = 0, tmp_bytes = 0;
    int max_h = 0;
    auto place = [&](int in_size, int& k_off, int& b_off, int& ks) {
        if (in_size == IMG) { k_off = b_off = ks = 0; return; }           // Pillow skips a pass whose size is unchanged
        auto it = e->rs_tables.find(in_size);
        if (it == e->rs_tables.end()) it = e->rs_tables.emplace(in_size, make_resample_table(in_size, IMG)).first;
        const ResampleTable& t = it->second;
        auto pl = placed.find(in_size);
        if (pl == placed.end()) {
            pl = placed.emplace(in_size, std::make_pair((int)coef.size(), (int)bounds.size())).first;
            coef.insert(coef.end(), t.kk.begin(), t.kk.end());
            bounds.insert(bounds.end(), t.bounds.begin(), t.bounds.end());
        }
        k_off = pl->second.first; b_off = pl->second.second; ks = t.ksize;
    };
    for (int i = 0; i < n; ++i) {
        const PrepView& v = views[i];
        if (v.source < 0 || v.source >= (int)srcs.size()) throw ArgError{"view of an unknown source", MOCR_ERR_ARG};
        const PrepSource& sc = srcs[v.source];
        if (v.w < 1 || v.h < 1 || v.x < 0 || v.y < 0 || v.x + v.w > sc.w || v.y + v.h > sc.h)
            throw ArgError{"view outside its source image", MOCR_ERR_ARG};
        auto so = src_off.find(v.source);
        if (so == src_off.end()) {
            so = src_off.emplace(v.source, (long long)src_bytes).first;
            src_bytes += (size_t)sc.h * sc.w * sc.ch;
        }
        ResizeDesc& d = descs[i];
        d.channels = sc.ch; d.bgr = sc.bgr;
        const int stride = sc.w * sc.ch;                        // the packed copy has no row padding
        const long long origin = so->second + ((long long)v.y * sc.w + v.x) * sc.ch;
        if (v.rot == MOCR_ROTATE_90_CW) {                       // R[y'][x'] = S[h - 1 - x'][y']   (np.rot90(k = -1))
            d.h = v.w; d.w = v.h; d.row_step = sc.ch; d.pix_step = -stride;
            d.src_off = origin + (long long)(v.h - 1) * stride;
        } else if (v.rot == MOCR_ROTATE_90_CCW) {               // R[y'][x'] = S[x'][w - 1 - y']   (np.rot90(k = 1))
            d.h = v.w; d.w = v.h; d.row_step = -sc.ch; d.pix_step = stride;
            d.src_off = origin + (long long)(v.w - 1) * sc.ch;
        } else {
            d.h = v.h; d.w = v.w; d.row_step = stride; d.pix_step = sc.ch;
            d.src_off = origin;
        }
        d.tmp_off = (long long)tmp_bytes;
        tmp_bytes += (size_t)d.h * IMG;
        place(d.w, d.kx_off, d.bx_off, d.ksx);
        place(d.h, d.ky_off, d.by_off, d.ksy);
        max_h = std::max(max_h, d.h);
    }
    // pack the pixel rows of every source (drops the callers' row padding) into the pinned staging buffer, the
    // sources dealt to a few host threads by bytes (2048 crops of 224 x 224 x 3 are 300 MB: ~50 ms on one core)
    uint8_t* const packed = (uint8_t*)e->grow_pinned(slot, src_bytes);
    {
        std::vector<std::pair<int, long long>> items(src_off.begin(), src_off.end());
        auto pack_range = [&](size_t i0, size_t i1) {
            for (size_t i = i0; i < i1; ++i) {
                const PrepSource& sc = srcs[items[i].first];
                uint8_t* dst = packed + items[i].second;
                const size_t rowb = (size_t)sc.w * sc.ch;
                if ((int64_t)rowb == sc.row_stride) memcpy(dst, sc.data, rowb * sc.h);
                else
                    for (int y = 0; y < sc.h; ++y) memcpy(dst + (size_t)y * rowb, sc.data + (size_t)y * sc.row_stride, rowb);
            }
        };
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        const size_t nthr = std::min<size_t>({(size_t)8, (size_t)hw, items.size(), src_bytes / (4u << 20) + 1});
        if (nthr <= 1) pack_range(0, items.size());
        else {
            std::vector<std::thread> pool;
            const size_t per = (items.size() + nthr - 1) / nthr;
            for (size_t k = 0; k < nthr; ++k) {
                const size_t i0 = k * per, i1 = std::min(items.size(), i0 + per);
                if (i0 < i1) pool.emplace_back(pack_range, i0, i1);
            }
            for (auto& th : pool) th.join();
        }
    }
    uint8_t* d_src = (uint8_t*)e->grow(e->rs_src, src_bytes);
    uint8_t* d_tmp = (uint8_t*)e->grow(e->rs_tmp, tmp_bytes);
    ResizeDesc* d_desc = (ResizeDesc*)e->grow(e->rs_desc, descs.size() * sizeof(ResizeDesc));
    int* d_coef = (int*)e->grow(e->rs_coef, std::max<size_t>(coef.size(), 1) * sizeof(int));
    int* d_bounds = (int*)e->grow(e->rs_bounds, std::max<size_t>(bounds.size(), 1) * sizeof(int));
    hipStream_t st = e->prep_stream;
    HIPCHECK(hipMemcpyAsync(d_src, packed, src_bytes, hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(d_desc, descs.data(), descs.size() * sizeof(ResizeDesc), hipMemcpyHostToDevice, st));
    if (!coef.empty()) HIPCHECK(hipMemcpyAsync(d_coef, coef.data(), coef.size() * sizeof(int), hipMemcpyHostToDevice, st));
    if (!bounds.empty()) HIPCHECK(hipMemcpyAsync(d_bounds, bounds.data(), bounds.size() * sizeof(int), hipMemcpyHostToDevice, st));
    constexpr int ROWS = 8;
    hipEvent_t t0 = nullptr, t1 = nullptr, t2 = nullptr;
    if (prof) { t0 = e->get_event(); t1 = e->get_event(); t2 = e->get_event(); HIPCHECK(hipEventRecord(t0, st)); }
    hipLaunchKernelGGL((resize_h_kernel<224, ROWS>), dim3((max_h + ROWS - 1) / ROWS, n), dim3(256), 0, st, d_src, d_desc, d_coef, d_bounds, d_tmp);
    HIPCHECK(hipGetLastError());
    if (prof) HIPCHECK(hipEventRecord(t1, st));
    hipLaunchKernelGGL((resize_v_kernel<224, ROWS>), dim3(224 / ROWS, n), dim3(256), 0, st, d_tmp, d_desc, d_coef, d_bounds, d_out);
    HIPCHECK(hipGetLastError());
    if (prof) {
        HIPCHECK(hipEventRecord(t2, st));
        e->recs.push_back(ProfRec{e->kid("resize_h"), t0, t1, 0, (double)src_bytes + (double)tmp_bytes});
        hipEvent_t t1b = e->get_event();         // (a record owns both of its events)
        HIPCHECK(hipEventRecord(t1b, st));
        e->recs.push_back(ProfRec{e->kid("resize_v"), t1b, t2, 0, (double)tmp_bytes + (double)n * IMG * IMG});
    }
}

// Synchronous form (test hook mocr_preprocess, small calls): L conversion + Pillow-exact BILINEAR resize of the views
// into d_out [views][IMG][IMG] u8 on the device.
static void preprocess_views(mocr_engine* e, const std::vector<PrepSource>& srcs, const PrepView* views, int n, uint8_t* d_out) {
    const size_t plane = (size_t)e->cfg.image_size * e->cfg.image_size;
    PrepHold hold;
    for (int b = 0; b < n; b += 4096) {
        prep_enqueue(e, srcs, views + b, std::min(4096, n - b), d_out + (size_t)b * plane, 0, e->prof_on, hold);
        HIPCHECK(hipStreamSynchronize(e->prep_stream));     // the staging buffer and the descriptors are reused by the next pass
    }
}

static void preprocess_images(mocr_engine* e, const mocr_image* imgs, int n, uint8_t* d_out) {
    std::vector<PrepSource> srcs(n);
    std::vector<PrepView> views(n);
    for (int i = 0; i < n; ++i) {
        srcs[i] = source_of(imgs[i]);
        views[i] = PrepView{i, 0, 0, srcs[i].w, srcs[i].h, srcs[i].rot};
    }
    preprocess_views(e, srcs, views.data(), n, d_out);
}

// THE host entry points' engine: crops (views of host sources) -> ids.  The views are cut into chunks of max_batch
// rows = one decode job each.  A producer thread prepares chunk k + 1 (pack into the other pinned buffer, H2D, resize, on
// the preparation stream) while the lanes decode chunk k; a job's lane stream waits ON THE DEVICE for its chunk's event,
// so the host never blocks on a preparation.  r02 prepared ALL crops, synchronised, and only then started to decode.
static void prepare_and_decode(mocr_engine* e, const std::vector<PrepSource>& srcs, const PrepView* views, int n, int32_t* out_ids,
                               int32_t* out_len) {
    const size_t plane = (size_t)e->cfg.image_size * e->cfg.image_size;
    const int C = std::min(e->cfg.max_batch, 4096), nchunks = (n + C - 1) / C;
    uint8_t* const d_gray = (uint8_t*)e->grow(e->rs_gray, (size_t)n * plane);
    auto push_job = [&](int k, hipEvent_t ev) {
        Job j;
        j.wait_ev = ev;
        j.src = d_gray + (size_t)k * C * plane; j.src_host = false; j.channels = 1;
        j.row_stride = e->cfg.image_size; j.image_stride = (int64_t)plane;
        j.n = std::min(C, n - k * C); j.max_len = e->gen_max_len;
        j.out_ids = out_ids + (size_t)k * C * e->cfg.max_len; j.out_len = out_len + (size_t)k * C; j.out_host = true;
        e->pending.push_back(j);
    };
    std::vector<PrepHold> holds(nchunks);
    if (nchunks == 1 || e->prof_on) {              // nothing to overlap (or an instrumented pass: one thread records the events)
        for (int k = 0; k < nchunks; ++k) {
            prep_enqueue(e, srcs, views + (size_t)k * C, std::min(C, n - k * C), d_gray + (size_t)k * C * plane, 0, e->prof_on, holds[k]);
            HIPCHECK(hipStreamSynchronize(e->prep_stream));
            push_job(k, nullptr);
        }
        drive(e);
        return;
    }
    std::vector<hipEvent_t> ev(nchunks, nullptr);
    for (auto& x : ev) HIPCHECK(hipEventCreateWithFlags(&x, hipEventDisableTiming));
    auto cleanup = [&] {
        (void)hipStreamSynchronize(e->prep_stream);
        for (auto x : ev) if (x) (void)hipEventDestroy(x);
    };
    run_prep_pipeline(
        nchunks,
        [&](int k) {                                   // producer thread: pinned buffer k & 1 is free once chunk k - 2's copy has run
            if (k == 0) HIPCHECK(hipSetDevice(e->cfg.device));
            if (k >= 2) HIPCHECK(hipEventSynchronize(ev[k - 2]));
        },
        [&](int k) {                                   // producer thread
            prep_enqueue(e, srcs, views + (size_t)k * C, std::min(C, n - k * C), d_gray + (size_t)k * C * plane, k & 1, false, holds[k]);
            HIPCHECK(hipEventRecord(ev[k], e->prep_stream));
        },
        [&](int k) { push_job(k, ev[k]); },            // calling thread
        [&] { return (e->cfg.dtype == MOCR_BF16) ? pump_once<bf16_t>(e) : pump_once<float>(e); },
        [&] {                                          // either side failed: nothing stays queued, nothing stays in flight
            e->pending.clear();
            for (auto& L : e->lanes) { L.active = false; L.jobs.clear(); (void)hipStreamSynchronize(L.ctx.stream); }
            cleanup();
        });
    for (auto& L : e->lanes) HIPCHECK(hipStreamSynchronize(L.ctx.stream));
    cleanup();
}

int mocr_preprocess(mocr_engine* e, const mocr_image* images, int32_t n, uint8_t* out_gray) {
    return guarded(e, [&] {
        std::lock_guard<std::mutex> lk(e->mu);
        if (!images || !out_gray || n < 1) throw ArgError{"bad argument", MOCR_ERR_ARG};
        HIPCHECK(hipSetDevice(e->cfg.device));
        drive(e);
        e->bind(0);
        const size_t plane = (size_t)e->cfg.image_size * e->cfg.image_size;
        uint8_t* d_gray = (uint8_t*)e->grow(e->rs_gray, (size_t)n * plane);
        preprocess_images(e, images, n, d_gray);
        HIPCHECK(hipMemcpy(out_gray, d_gray, (size_t)n * plane, hipMemcpyDeviceToHost));
        e->unbind(0);
    });
}

int mocr_recognize_images(mocr_engine* e, const mocr_image* images, int32_t n, int32_t* out_ids, int32_t* out_len) {
    return guarded(e, [&] {
        std::lock_guard<std::mutex> lk(e->mu);
        require_ready(e, n, false);
        if (!images || !out_ids || !out_len) throw ArgError{"null pointer", MOCR_ERR_ARG};
        HIPCHECK(hipSetDevice(e->cfg.device));
        drive(e);
        std::vector<PrepSource> srcs(n);
        std::vector<PrepView> views(n);
        for (int i = 0; i < n; ++i) {
            srcs[i] = source_of(images[i]);
            views[i] = PrepView{i, 0, 0, srcs[i].w, srcs[i].h, srcs[i].rot};
        }
        prepare_and_decode(e, srcs, views.data(), n, out_ids, out_len);
    });
}

// The crop a detected text region gets (src/ui/main_window.py:9530-9540): its bounding box grown by
// int(max(w, h) * 0.08) on every side, clipped to the page; false when no more than a 1-pixel sliver is left
// (the reference then returns '' without calling the recogniser).
static bool padded_region(const mocr_region& r, int page_h, int page_w, PrepView& v) {
    if (r.width < 0 || r.height < 0) throw ArgError{"region with a negative size", MOCR_ERR_ARG};
    const int pad = (int)((double)std::max(r.width, r.height) * 0.08);
    const long long x1 = std::max<long long>((long long)r.x - pad, 0), y1 = std::max<long long>((long long)r.y - pad, 0);
    const long long x2 = std::min<long long>((long long)r.x + r.width + pad, page_w), y2 = std::min<long long>((long long)r.y + r.height + pad, page_h);
    if (x2 - x1 <= 1 || y2 - y1 <= 1) return false;
    v.x = (int)x1; v.y = (int)y1; v.w = (int)(x2 - x1); v.h = (int)(y2 - y1);
    return true;
}

int mocr_recognize_regions(mocr_engine* e, const mocr_image* pages, int32_t n_pages, const mocr_region* regions, int32_t n_regions,
                           int32_t* out_ids, int32_t* out_len) {
    return guarded(e, [&] {
        std::lock_guard<std::mutex> lk(e->mu);
        if (!e->committed) throw ArgError{"weights not committed (mocr_commit_weights)", MOCR_ERR_STATE};
        if (!pages || n_pages < 1 || n_regions < 0 || (n_regions > 0 && (!regions || !out_ids || !out_len)))
            throw ArgError{"bad argument", MOCR_ERR_ARG};
        if (n_regions == 0) return;
        HIPCHECK(hipSetDevice(e->cfg.device));
        drive(e);
        std::vector<PrepSource> srcs(n_pages);
        for (int i = 0; i < n_pages; ++i) srcs[i] = source_of(pages[i]);
        std::vector<PrepView> views;
        std::vector<int> where(n_regions, -1);          // region -> its row among the recognised crops (-1: sliver)
        for (int i = 0; i < n_regions; ++i) {
            const mocr_region& r = regions[i];
            if (r.page < 0 || r.page >= n_pages) throw ArgError{"region of an unknown page", MOCR_ERR_ARG};
            PrepView v{r.page, 0, 0, 0, 0, MOCR_ROTATE_NONE};
            if (!padded_region(r, srcs[r.page].h, srcs[r.page].w, v)) continue;
            where[i] = (int)views.size();
            views.push_back(v);
        }
        const int L = e->cfg.max_len, nv = (int)views.size();
        std::vector<int32_t> ids((size_t)nv * L), lens(nv);
        if (nv > 0) prepare_and_decode(e, srcs, views.data(), nv, ids.data(), lens.data());
        for (int i = 0; i < n_regions; ++i) {
            int32_t* row = out_ids + (size_t)i * L;
            if (where[i] < 0) {
                for (int t = 0; t < L; ++t) row[t] = e->cfg.pad_id;
                out_len[i] = 0;
            } else {
                memcpy(row, ids.data() + (size_t)where[i] * L, (size_t)L * sizeof(int32_t));
                out_len[i] = lens[where[i]];
            }
        }
    });
}

int mocr_set_generate_max_length(mocr_engine* e, int32_t max_len) {
    return guarded(e, [&] {
        std::lock_guard<std::mutex> lk(e->mu);
        if (max_len < 2 || max_len > e->cfg.max_len) throw ArgError{"generate max_length must be in [2, max_len]", MOCR_ERR_ARG};
        e->gen_max_len = max_len;
    });
}

int mocr_ln_fold_state(mocr_engine* e, float* noise_ratio) {
    if (!e) return 0;
    std::lock_guard<std::mutex> lk(e->mu);
    if (noise_ratio) *noise_ratio = e->fold_ratio;
    if (e->cfg.dtype != MOCR_BF16 || (e->cfg.flags & MOCR_FLAG_NO_LN_FOLD)) return 0;
    return (e->fold_ok || (e->cfg.flags & MOCR_FLAG_FORCE_LN_FOLD)) ? 1 : 0;
}

int64_t mocr_decode_slot_steps(mocr_engine* e) {
    if (!e) return 0;
    std::lock_guard<std::mutex> lk(e->mu);
    return e->n_slot_steps;
}

int64_t mocr_compaction_count(mocr_engine* e) {
    if (!e) return 0;
    std::lock_guard<std::mutex> lk(e->mu);
    return e->n_compactions;
}

int mocr_graph_count(mocr_engine* e) {
    if (!e) return MOCR_ERR_ARG;
    std::lock_guard<std::mutex> lk(e->mu);
    return (int)e->graphs.size();
}

int mocr_device_memory(int32_t device, int64_t* free_bytes, int64_t* total_bytes) {
    if (!free_bytes || !total_bytes) return MOCR_ERR_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return MOCR_ERR_HIP;
    size_t f = 0, t = 0;
    if (hipSetDevice(device) != hipSuccess || hipMemGetInfo(&f, &t) != hipSuccess) return MOCR_ERR_HIP;
    *free_bytes = (int64_t)f; *total_bytes = (int64_t)t;
    return MOCR_OK;
}

int mocr_encode(mocr_engine* e, const void* d_gray, int32_t n, float* h_out) {
    return guarded(e, [&] {
        std::lock_guard<std::mutex> lk(e->mu);
        require_ready(e, n);
        if (!d_gray || !h_out) throw ArgError{"null pointer", MOCR_ERR_ARG};
        HIPCHECK(hipSetDevice(e->cfg.device));
        drive(e);
        e->bind(0);
        const uint8_t* g = reinterpret_cast<const uint8_t*>(d_gray);
        dispatch(e, [&](auto tag) { run_encoder<decltype(tag)>(e, g, n); });
        const size_t count = (size_t)n * e->S * e->D;
        if (e->cfg.dtype == MOCR_F32) {
            HIPCHECK(hipMemcpyAsync(h_out, e->ENC, count * 4, hipMemcpyDeviceToHost, e->stream));
            HIPCHECK(hipStreamSynchronize(e->stream));
        } else {
            std::vector<uint16_t> tmp(count);
            HIPCHECK(hipMemcpyAsync(tmp.data(), e->ENC, count * 2, hipMemcpyDeviceToHost, e->stream));
            HIPCHECK(hipStreamSynchronize(e->stream));
            for (size_t i = 0; i < count; ++i) {
                const uint32_t u = (uint32_t)tmp[i] << 16;
                memcpy(&h_out[i], &u, 4);
            }
        }
    });
}

int mocr_decode_logits(mocr_engine* e, const void* d_gray, int32_t n, const int32_t* forced_ids, int32_t T, float* h_logits) {
    return guarded(e, [&] {
        std::lock_guard<std::mutex> lk(e->mu);
        require_ready(e, n);
        if (!d_gray || !forced_ids || !h_logits || T < 1 || T >= e->cfg.max_len) throw ArgError{"bad argument", MOCR_ERR_ARG};
        HIPCHECK(hipSetDevice(e->cfg.device));
        drive(e);
        e->bind(0);
        const size_t fcount = (size_t)n * T, lcount = (size_t)n * T * e->V;
        if (fcount > e->forced_cap) { e->forced = e->dalloc<int>(fcount); e->forced_cap = fcount; }
        if (lcount > e->logits_cap) { e->logits_dbg = e->dalloc<float>(lcount); e->logits_cap = lcount; }
        e->unbind(0);
        HIPCHECK(hipMemcpyAsync(e->forced, forced_ids, fcount * sizeof(int), hipMemcpyHostToDevice, e->stream));
        const uint8_t* g = reinterpret_cast<const uint8_t*>(d_gray);
        dispatch(e, [&](auto tag) {
            using T_ = decltype(tag);
            run_encoder<T_>(e, g, n);
            if (!e->use_latent(n)) run_cross_kv<T_>(e, n);
            else if (e->fp8attn) quantize_enc(e, n);
            run_decode_forced<T_>(e, n, e->forced, T, e->logits_dbg);
        });
        HIPCHECK(hipMemcpyAsync(h_logits, e->logits_dbg, lcount * 4, hipMemcpyDeviceToHost, e->stream));
        HIPCHECK(hipStreamSynchronize(e->stream));
    });
}

int mocr_op_gemm(mocr_engine* e, const void* dA, const void* dW, const float* d_bias, void* d_out, const float* d_resid,
                 int32_t M, int32_t N, int32_t K, int32_t epilogue, int32_t tile, int32_t split_k) {
    return guarded(e, [&] {
        std::lock_guard<std::mutex> lk(e->mu);
        HIPCHECK(hipSetDevice(e->cfg.device));
        drive(e);
        e->bind(0);
        if (epilogue == EPI_PATCH) throw ArgError{"EPI_PATCH is not exposed through mocr_op_gemm", MOCR_ERR_ARG};
        const long long slab = (long long)M * N;
        if (e->cfg.dtype == MOCR_BF16)
            gemm<bf16_t>(e, "op_gemm", dA, K, dW, d_bias, d_out, N, d_resid, M, N, K, epilogue, tile, split_k, slab);
        else
            gemm<float>(e, "op_gemm", dA, K, dW, d_bias, d_out, N, d_resid, M, N, K, epilogue, tile, split_k, slab);
        HIPCHECK(hipStreamSynchronize(e->stream));
    });
}

int mocr_op_gemm_ln(mocr_engine* e, const void* dA, const void* dW, const float* d_bias, void* d_out, const float* d_resid,
                    int32_t M, int32_t N, int32_t K, int32_t epilogue, int32_t tile, float* d_part, const float* d_csum, void* d_xb) {
    return guarded(e, [&] {
        std::lock_guard<std::mutex> lk(e->mu);
        HIPCHECK(hipSetDevice(e->cfg.device));
        drive(e);
        e->bind(0);
        if (e->cfg.dtype != MOCR_BF16) throw ArgError{"mocr_op_gemm_ln: bf16 engines only", MOCR_ERR_UNSUPPORTED};
        LnFold f{d_part, d_csum, d_xb};
        gemm<bf16_t>(e, "op_gemm_ln", dA, K, dW, d_bias, d_out, N, d_resid, M, N, K, epilogue, tile, 1, 0, nullptr, 0, nullptr, 0, nullptr, &f);
        HIPCHECK(hipStreamSynchronize(e->stream));
    });
}

int mocr_op_ln_prep(mocr_engine* e, const float* d_x, void* d_xb, float* d_part, int32_t M) {
    return guarded(e, [&] {
        std::lock_guard<std::mutex> lk(e->mu);
        HIPCHECK(hipSetDevice(e->cfg.device));
        drive(e);
        e->bind(0);
        if (e->D != 768) throw ArgError{"mocr_op_ln_prep: hidden size 768 only", MOCR_ERR_UNSUPPORTED};
        hipLaunchKernelGGL((ln_prep_kernel<768>), dim3((M + 3) / 4), dim3(256), 0, e->stream, d_x, reinterpret_cast<bf16_t*>(d_xb), d_part, M);
        HIPCHECK(hipGetLastError());
        HIPCHECK(hipStreamSynchronize(e->stream));
    });
}

int mocr_op_layernorm(mocr_engine* e, const float* d_x, const float* d_gamma, const float* d_beta, void* d_out, int32_t M) {
    return guarded(e, [&] {
        std::lock_guard<std::mutex> lk(e->mu);
        HIPCHECK(hipSetDevice(e->cfg.device));
        drive(e);
        e->bind(0);
        dispatch(e, [&](auto tag) { layernorm<decltype(tag)>(e, d_x, d_gamma, d_beta, d_out, M); });
        HIPCHECK(hipStreamSynchronize(e->stream));
    });
}

int mocr_op_enc_attention(mocr_engine* e, const void* d_qkv, void* d_ctx, int32_t n, int32_t impl) {
    return guarded(e, [&] {
        std::lock_guard<std::mutex> lk(e->mu);
        HIPCHECK(hipSetDevice(e->cfg.device));
        drive(e);
        e->bind(0);
        dispatch(e, [&](auto tag) { enc_attention<decltype(tag)>(e, d_qkv, d_ctx, n, impl); });
        HIPCHECK(hipStreamSynchronize(e->stream));
    });
}

int mocr_op_latent_attention(mocr_engine* e, const void* d_qt, const void* d_x, void* d_out, int32_t n, int32_t len,
                             int64_t x_batch_stride) {
    return guarded(e, [&] {
        std::lock_guard<std::mutex> lk(e->mu);
        HIPCHECK(hipSetDevice(e->cfg.device));
        drive(e);
        e->bind(0);
        if (e->cfg.dtype != MOCR_BF16 || !d_qt || !d_x || !d_out || n < 1 || len < 1) throw ArgError{"bad argument", MOCR_ERR_ARG};
        LatentParams p{};
        p.qt = reinterpret_cast<const bf16_t*>(d_qt); p.x = reinterpret_cast<const bf16_t*>(d_x);
        p.out = reinterpret_cast<bf16_t*>(d_out); p.x_batch_stride = x_batch_stride; p.fixed_len = len; p.heads = e->H;
        p.ablate = env_int("MOCR_LAT_ABLATE", 0); p.rows = n;
        static unsigned long long* dbg = nullptr;
        if (env_int("MOCR_LAT_STAMP", 0)) { if (!dbg) dbg = e->dalloc<unsigned long long>(8 + 512); p.dbg = dbg; }
        ProfScope ps(e, "op_latent", 0, (double)n * len * 1536);
        launch_latent(e, false, p);
        HIPCHECK(hipStreamSynchronize(e->stream));
        if (p.dbg) {
            unsigned long long h[40];
            HIPCHECK(hipMemcpy(h, p.dbg, sizeof(h), hipMemcpyDeviceToHost));
            if (env_int("MOCR_LAT_DUMP", 0)) {
                std::vector<float> f(1024);
                HIPCHECK(hipMemcpy(f.data(), p.dbg + 8, 1024 * 4, hipMemcpyDeviceToHost));
                FILE* fp = fopen("gpurun_out/lat_dump.bin", "wb");
                if (fp) { fwrite(f.data(), 4, 1024, fp); fclose(fp); }
            }
            for (int w = 0; w < 4; ++w)
                fprintf(stderr, "[lat stamps, cycles, block 0 wave %d] wait+issue %llu  S %llu  exchange %llu  softmax %llu  PX %llu  loop/Qt %llu  rowend-b1 %llu  stage-b2 %llu  store %llu\n",
                        w, h[10 * w], h[10 * w + 1], h[10 * w + 2], h[10 * w + 3], h[10 * w + 4], h[10 * w + 5], h[10 * w + 6], h[10 * w + 7], h[10 * w + 8]);
        }
    });
}

int mocr_op_quant_fp8(mocr_engine* e, const void* d_x, void* d_x8, int64_t n_elems, float inv_sx) {
    return guarded(e, [&] {
        std::lock_guard<std::mutex> lk(e->mu);
        HIPCHECK(hipSetDevice(e->cfg.device));
        drive(e);
        e->bind(0);
        if (!d_x || !d_x8 || n_elems < 16 || n_elems % 16) throw ArgError{"bad argument (n_elems must be a positive multiple of 16)", MOCR_ERR_ARG};
        const long long n16 = n_elems / 16;
        hipLaunchKernelGGL(quant_rows_fp8_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, e->stream,
                           reinterpret_cast<const bf16_t*>(d_x), reinterpret_cast<uint8_t*>(d_x8), n16, inv_sx);
        HIPCHECK(hipGetLastError());
        HIPCHECK(hipStreamSynchronize(e->stream));
    });
}

int mocr_op_latent_attention_fp8(mocr_engine* e, const void* d_qt, const void* d_x8, void* d_out, int32_t n, int32_t len,
                                 int64_t x_batch_stride_bytes, float sx) {
    return guarded(e, [&] {
        std::lock_guard<std::mutex> lk(e->mu);
        HIPCHECK(hipSetDevice(e->cfg.device));
        drive(e);
        e->bind(0);
        if (e->cfg.dtype != MOCR_BF16 || !d_qt || !d_x8 || !d_out || n < 1 || len < 1 || !(sx > 0.f)) throw ArgError{"bad argument", MOCR_ERR_ARG};
        Latent8Params p{};
        p.qt = reinterpret_cast<const bf16_t*>(d_qt); p.x8 = reinterpret_cast<const uint8_t*>(d_x8);
        p.out = reinterpret_cast<bf16_t*>(d_out); p.x_batch_stride = x_batch_stride_bytes; p.fixed_len = len; p.heads = e->H;
        p.rows = n; p.sx = sx;
        ProfScope ps(e, "op_latent8", 0, (double)n * len * 768);
        launch_latent8(e, false, p);
        HIPCHECK(hipStreamSynchronize(e->stream));
    });
}

int mocr_op_qqt(mocr_engine* e, const void* d_x, const void* d_wq, const float* d_bq, const void* d_wkT, void* d_qt, int32_t n) {
    return guarded(e, [&] {
        std::lock_guard<std::mutex> lk(e->mu);
        HIPCHECK(hipSetDevice(e->cfg.device));
        drive(e);
        e->bind(0);
        if (e->cfg.dtype != MOCR_BF16 || !d_x || !d_wq || !d_bq || !d_wkT || !d_qt || n < 1) throw ArgError{"bad argument", MOCR_ERR_ARG};
        QqtParams q{};
        q.x = reinterpret_cast<const bf16_t*>(d_x); q.wq = reinterpret_cast<const bf16_t*>(d_wq); q.bq = d_bq;
        q.wkT = reinterpret_cast<const bf16_t*>(d_wkT); q.qt = reinterpret_cast<bf16_t*>(d_qt);
        ProfScope ps(e, "op_qqt", 4.0 * n * 768 * 768, 0);
        launch_qqt(e, q, n, n);
        HIPCHECK(hipStreamSynchronize(e->stream));
    });
}

int mocr_profile_enable(mocr_engine* e, int32_t on) {
    return guarded(e, [&] {
        std::lock_guard<std::mutex> lk(e->mu);
        HIPCHECK(hipSetDevice(e->cfg.device));
        drive(e);
        e->prof_collect();
        e->prof_on = on != 0;
    });
}

int mocr_profile_reset(mocr_engine* e) {
    return guarded(e, [&] {
        std::lock_guard<std::mutex> lk(e->mu);
        HIPCHECK(hipSetDevice(e->cfg.device));
        drive(e);
        e->prof_collect();
        for (auto& s : e->stats) { s.launches = 0; s.total_ms = 0; s.flops = 0; s.bytes = 0; }
    });
}

int mocr_profile_get(mocr_engine* e, mocr_kernel_stat* out, int32_t cap, int32_t* n_out) {
    return guarded(e, [&] {
        std::lock_guard<std::mutex> lk(e->mu);
        if (!out || !n_out || cap < 0) throw ArgError{"bad argument", MOCR_ERR_ARG};
        HIPCHECK(hipSetDevice(e->cfg.device));
        drive(e);
        e->prof_collect();
        int k = 0;
        for (auto& s : e->stats)
            if (s.launches > 0 && k < cap) out[k++] = s;
        *n_out = k;
    });
}

}  // extern "C"
