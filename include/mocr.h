/*
 * mocr.h - C ABI of the MI355X-native Manga-OCR recogniser engine (libmocr_hip.so).
 *
 * This is the drop-in boundary for ONE call of the reference application:
 *
 *     raw_text = self.manga_ocr_reader(pil_img)          reference src/ui/main_window.py:9801
 *
 * i.e. `manga_ocr.MangaOcr.__call__(PIL.Image) -> str`, constructed once at
 * src/ui/main_window.py:3394 (`MangaOcr()`), imported at src/core/config.py:431-436 and
 * called concurrently, without a lock, by up to MAX_WORKERS QueueProcessorWorker threads
 * (src/core/workers.py:209-247, 318-327, 383-402) and by the text-detect path
 * (src/ui/main_window.py:9462-9476, 9530-9549).  The reference has no native interface of
 * its own (it is pure Python); the entry points below are what a binding for that call
 * needs, and the Python class manga_ocr.MangaOcr in this repo is that binding (ctypes).
 * INTEGRATION.md shows the stub.
 *
 * Conventions: plain C, no exceptions, no Python or torch types.  Every function returns
 * MOCR_OK (0) or a negative error code; mocr_last_error() returns a message for the last
 * failure on that engine.  The caller owns every buffer it passes.  Token-id blocks are
 * int32, row-major [n, max_len]; a row holds start_id, the generated ids, eos_id, then
 * pad_id up to max_len - exactly the rows `generate(max_length=max_len)` returns
 * (TF/generation/utils.py:2929-2937), padded to the fixed width.
 */
#ifndef MOCR_H
#define MOCR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MOCR_ABI_VERSION 2      /* 2: mocr_image.rotate (occupies what was padding: the struct keeps its size) */

enum {
    MOCR_OK = 0,
    MOCR_ERR_ARG = -1,          /* bad argument (null, size, unsupported shape) */
    MOCR_ERR_STATE = -2,        /* call order (weights not committed, ...) */
    MOCR_ERR_HIP = -3,          /* a HIP runtime call failed */
    MOCR_ERR_UNSUPPORTED = -4,  /* valid request this build does not implement */
    MOCR_ERR_NOMEM = -5
};

enum { MOCR_F32 = 0, MOCR_BF16 = 1 };

/* mocr_config.flags */
enum {
    MOCR_FLAG_SIMPLE_ATTENTION = 1 << 0, /* encoder attention on the VALU kernel even in bf16 mode */
    MOCR_FLAG_NO_GRAPH = 1 << 1,         /* launch decode steps eagerly instead of replaying a HIP graph */
    MOCR_FLAG_NO_EARLY_EXIT = 1 << 2,    /* always run max_len-1 decode steps */
    MOCR_FLAG_CLASSIC_ATTENTION = 1 << 3, /* bf16: projected K/V caches instead of the latent (absorbed) decode attention */
    MOCR_FLAG_NO_FUSED_ARGMAX = 1 << 4,   /* always write the logits and take the argmax in the token kernel */
    MOCR_FLAG_NO_FUSED_QQT = 1 << 5,      /* latent attention: query and absorbed query as two GEMM launches even for fat batches */
    MOCR_FLAG_FP8_ATTENTION = 1 << 7,     /* bf16 engines, opt-in (BASELINE configs[4]): the latent decode attention reads its
                                           * key/value rows as OCP e4m3 (768 B per key instead of 1,536 B, static per-source
                                           * scales) and runs both products on fp8 MFMA, softmax in fp32.  NOT the parity
                                           * configuration: its accuracy is reported by tests/test_gpu_fp8_attention.py */
    MOCR_FLAG_LATENT_ALWAYS = 1 << 6,     /* bf16: latent attention for every batch size (default: batches of <= 256 rows take the
                                           * classic projected-K/V kernels, whose grid - one block per (row, head) - has half the
                                           * step latency there: 50 instead of 80 ms for 64 crops) */
    MOCR_FLAG_NO_SMALL_BATCH_PATH = 1 << 8, /* bf16: batches of <= 32 rows through the generic split-K projections + add/LayerNorm
                                           * launches (28 per decode step) instead of the one-launch-per-projection path (19) */
    MOCR_FLAG_LATENT_TILE32 = 1 << 10,    /* bf16 latent attention on r03's kernel shape (32-key tiles, one persistent block per CU, three
                                           * barriers per tile) instead of 16-key tiles on two blocks per CU with the score tile
                                           * transposed (two barriers per tile): the A/B partner */
    MOCR_FLAG_NO_COMPACTION = 1 << 11,    /* keep every row of a batch in the decode steps until the whole batch has finished (r01-r03
                                           * behaviour) instead of compacting the unfinished rows between chunks of steps */
    MOCR_FLAG_FORCE_LN_FOLD = 1 << 13,    /* bf16: fold the encoder's LayerNorms even where this checkpoint's residual stream failed the
                                           * commit-time check (mocr_ln_fold_state): A/B and tests only */
    MOCR_FLAG_NO_LN_FOLD = 1 << 9         /* bf16: the encoder's LayerNorms as launches of their own even where the layer GEMMs run on
                                           * the persistent kernel (default there: folded into the GEMMs on both sides of them) */
};

typedef struct mocr_engine mocr_engine;

/* Replaces the (argument-less) constructor call at src/ui/main_window.py:3394.  The model
 * hyper-parameters are those BASELINE.json fixes (ViT-B/16-224 + 2-layer BERT decoder). */
typedef struct mocr_config {
    int32_t struct_size; /* sizeof(mocr_config), for forward compatibility */
    int32_t device;      /* HIP device ordinal of this process's GPU */
    int32_t dtype;       /* MOCR_F32 (parity mode) or MOCR_BF16 (bf16 storage, fp32 accumulate) */
    int32_t max_batch;   /* crops per internal batch (rows of one decode step) */
    int32_t max_len;     /* generate(max_length): 300 */
    int32_t image_size;  /* 224 */
    int32_t patch_size;  /* 16 */
    int32_t hidden;      /* 768 */
    int32_t enc_layers;  /* 12 */
    int32_t dec_layers;  /* 2 */
    int32_t heads;       /* 12 */
    int32_t ffn;         /* 3072 */
    int32_t vocab;       /* 6144 */
    int32_t max_pos;     /* 512 */
    int32_t start_id;    /* 2 */
    int32_t eos_id;      /* 3 */
    int32_t pad_id;      /* 0 */
    float ln_eps;        /* 1e-12 */
    int32_t flags;
    int32_t lanes;       /* batches kept in flight on separate HIP streams (0 = 1); each has its own workspace */
} mocr_config;

int mocr_abi_version(void);

/* Construct an engine on cfg->device.  Allocates all device memory for max_batch. */
int mocr_create(const mocr_config* cfg, mocr_engine** out);
void mocr_destroy(mocr_engine* e);
const char* mocr_last_error(const mocr_engine* e);

/* Weights: one call per tensor of the model (canonical transformers-5.x state_dict names,
 * e.g. "encoder.layers.0.attention.q_proj.weight"), host float32, row-major; then commit.
 * Replaces the `from_pretrained(...)` inside the reference's recogniser constructor. */
int mocr_set_tensor(mocr_engine* e, const char* name, const float* data, const int64_t* shape, int32_t ndim);
int mocr_commit_weights(mocr_engine* e);

/* THE HOT PATH (host buffers).  images: n crops, uint8, each h x w with `channels` (1 = the
 * luminance plane the recogniser's convert('L') would produce, 3 = RGB as handed over at
 * src/ui/main_window.py:9800; converted on the device with Pillow's fixed-point formula),
 * row_stride bytes between rows, image_stride bytes between crops.  h and w must equal
 * image_size (crops of other sizes: mocr_recognize_images below).
 * out_ids [n, max_len] int32, out_len [n] int32.  Blocking; thread-safe (calls from
 * several threads are serialised per engine).  n may exceed max_batch. */
int mocr_recognize(mocr_engine* e, const uint8_t* images, int32_t n, int32_t h, int32_t w,
                   int64_t row_stride, int64_t image_stride, int32_t channels,
                   int32_t* out_ids, int32_t* out_len);

/* THE HOT PATH from crops of ANY size (BASELINE configs[4]'s variable-resolution crops).
 * The device does what the reference's recogniser and image processor do in front of the encoder:
 * img.convert('L') [.convert('RGB')] and resize((224,224), BILINEAR) (MangaOcr.__call__, SURVEY row a10;
 * TF/models/vit/image_processing_pil_vit.py:20-27, TF/image_processing_backends.py:521-570 -> Pillow's
 * ImagingResample) - in Pillow's own fixed-point arithmetic, i.e. bit-exact with the CPU path.
 * One descriptor per crop: host pointer, size, bytes between rows, channels (1 = L, 3 = RGB as handed over at
 * src/ui/main_window.py:9800).  The callee neither keeps nor modifies the pixels. */
typedef struct mocr_image {
    const uint8_t* data;
    int32_t height, width;
    int64_t row_stride;
    int32_t channels;   /* 1 = L, 3 = RGB, MOCR_CHANNELS_BGR = 3 bytes per pixel in OpenCV's B,G,R order */
    int32_t rotate;     /* MOCR_ROTATE_NONE, or the reference's orientation-only rotation applied ON THE DEVICE before the
                         * resize: MOCR_ROTATE_90_CW = cv2.ROTATE_90_CLOCKWISE ("Vertical" setting, landscape crop),
                         * MOCR_ROTATE_90_CCW = cv2.ROTATE_90_COUNTERCLOCKWISE ("Horizontal" setting, portrait crop);
                         * src/core/workers.py:320-326, src/ui/main_window.py:9787-9795.  height / width / row_stride
                         * describe the crop as it lies in memory (before the rotation).  Ignored for pages of
                         * mocr_recognize_regions. */
} mocr_image;
enum { MOCR_ROTATE_NONE = 0, MOCR_ROTATE_90_CW = 1, MOCR_ROTATE_90_CCW = 2 };
/* BGR pixels as the reference's crop tools and pages hold them (`cropped_cv_img`, `cv_image`: src/ui/main_window.py:6431,
 * src/core/workers.py:461): the BGR -> RGB swap of src/ui/main_window.py:9800 is folded into the luminance conversion. */
#define MOCR_CHANNELS_BGR (-3)
/* out_ids [n, max_len] int32, out_len [n] int32 (host).  Blocking, thread-safe; n may exceed max_batch. */
int mocr_recognize_images(mocr_engine* e, const mocr_image* images, int32_t n, int32_t* out_ids, int32_t* out_len);
/* THE HOT PATH for the Text-detect callers: whole pages plus the bounding
 * rectangles of the regions a detector found on them, replacing the serial loop of `_collect_manga_detections` ->
 * `_recognize_polygon` -> `perform_ocr` (src/ui/main_window.py:9462-9476, 9530-9549, 9774-9803) and, over several
 * pages, of `AutoDetectorWorker.run` (src/core/workers.py:448-482).  Every page is uploaded ONCE; each region's crop -
 * its rectangle grown by int(max(w, h) * 0.08) on every side and clipped to the page, the rule of
 * src/ui/main_window.py:9533-9537 - is cut on the device by the resize kernel's descriptor (offset + page stride), and
 * all regions of all pages decode as one job queue.  A region that leaves no more than a 1-pixel sliver gets
 * out_len = 0 and a row of pad_id (the reference returns '' for it without calling the recogniser, :9538-9539).
 * region.{x, y, width, height} = QPolygon.boundingRect() of the detected polygon, in page pixels. */
typedef struct mocr_region {
    int32_t page;                /* index into pages[] */
    int32_t x, y, width, height;
} mocr_region;
int mocr_recognize_regions(mocr_engine* e, const mocr_image* pages, int32_t n_pages, const mocr_region* regions,
                           int32_t n_regions, int32_t* out_ids, int32_t* out_len);

/* Preprocessing only (test hook): out_gray [n, image_size, image_size] uint8 (host) = the plane the encoder sees
 * in each of its three equal input channels before the 1/255 and (x - 0.5)/0.5 scaling. */
int mocr_preprocess(mocr_engine* e, const mocr_image* images, int32_t n, uint8_t* out_gray);

/* THE HOT PATH (device buffers, asynchronous): submits one batch.  d_gray is a device pointer to
 * n contiguous image_size x image_size uint8 luminance planes, d_out_ids / d_out_len device
 * pointers ([n,max_len] / [n] int32), n <= max_batch; all three must stay valid until
 * mocr_synchronize() returns.  Batches submitted back to back run concurrently on the engine's
 * lanes; mocr_synchronize() schedules every submitted batch to completion (greedy steps in
 * chunks, stopping a batch once all of its rows have emitted EOS) and waits for the GPU. */
int mocr_recognize_device(mocr_engine* e, const void* d_gray, int32_t n, void* d_out_ids, void* d_out_len);
/* generate(max_length=...) of every batch submitted from now on, whatever the entry point (2 <= max_len <= the
 * engine's max_len; rows are still max_len wide; mocr_recognize_gray_host's own argument overrides it).  The reference always calls generate with 300; a speech bubble is
 * typically ~32 tokens (SURVEY.md §8d reports both regimes). */
int mocr_set_generate_max_length(mocr_engine* e, int32_t max_len);
int mocr_synchronize(mocr_engine* e);
/* The hipStream_t of lane 0 (the stream the test hooks and single-lane engines launch on). */
void* mocr_stream(mocr_engine* e);

/* ---- test hooks (fp32 out; used by tests/ and __graft_entry__.smoke()) -------------------- */
/* Encoder only: h_out [n, 197, hidden] float32 (final LayerNorm output). */
int mocr_encode(mocr_engine* e, const void* d_gray, int32_t n, float* h_out);
/* Teacher-forced decode: inputs forced_ids [n, T] (forced_ids[:,0] must be start_id);
 * h_logits [n, T, vocab] float32 = logits after consuming forced_ids[:, :t+1]. */
int mocr_decode_logits(mocr_engine* e, const void* d_gray, int32_t n, const int32_t* forced_ids,
                       int32_t T, float* h_logits);
/* Greedy decode limited to max_len_override tokens (<= max_len); host outputs. */
int mocr_recognize_gray_host(mocr_engine* e, const uint8_t* gray, int32_t n, int32_t max_len_override,
                             int32_t* out_ids, int32_t* out_len);

/* Single operators on device buffers of the engine's dtype (kernel unit tests). */
int mocr_op_gemm(mocr_engine* e, const void* dA, const void* dW, const float* d_bias, void* d_out,
                 const float* d_resid, int32_t M, int32_t N, int32_t K, int32_t epilogue,
                 int32_t tile, int32_t split_k);
int mocr_op_layernorm(mocr_engine* e, const float* d_x, const float* d_gamma, const float* d_beta,
                      void* d_out, int32_t M);
/* bf16 engines: the persistent encoder GEMM (tile 4096 / 4097 / 4099 / 4100) with the LayerNorm folded in.
 * epilogue 3 (bias + residual, fp32 out): also writes d_xb [M,N] bf16 = the output rows and d_part [M,4,2] float32 = each
 * row's (sum, sum of squares) per 256-column slice.  epilogue 1 / 2 (bias / bias + GELU, bf16 out): dA = the rows x as bf16,
 * dW = bf16(W o gamma), d_bias = b + W beta, d_csum [N] = column sums of dW, d_part = the statistics of the fp32 rows:
 * out = LN(x) W^T + b.  mocr_op_ln_prep: d_x [M,768] float32 -> d_xb bf16 + d_part (partial 0 = the row's sums). */
int mocr_op_gemm_ln(mocr_engine* e, const void* dA, const void* dW, const float* d_bias, void* d_out,
                    const float* d_resid, int32_t M, int32_t N, int32_t K, int32_t epilogue, int32_t tile,
                    float* d_part, const float* d_csum, void* d_xb);
int mocr_op_ln_prep(mocr_engine* e, const float* d_x, void* d_xb, float* d_part, int32_t M);
int mocr_op_enc_attention(mocr_engine* e, const void* d_qkv, void* d_ctx, int32_t n, int32_t impl);
/* Latent decode attention (bf16 engines): d_qt [n,16,768], keys d_x with x_batch_stride elements between
 * sequences, context length len for every row; d_out [n,16,768] = softmax(qt . x^T) x per head. */
int mocr_op_latent_attention(mocr_engine* e, const void* d_qt, const void* d_x, void* d_out, int32_t n, int32_t len,
                             int64_t x_batch_stride);

/* fp8 attention (MOCR_FLAG_FP8_ATTENTION) operators: d_x bf16 [n_elems] -> d_x8 e4m3 [n_elems] = e4m3(x * inv_sx);
 * and the latent attention on e4m3 key rows (768 B per key, x = x8 * sx): d_qt [n,16,768] bf16, d_out [n,16,768] bf16. */
int mocr_op_quant_fp8(mocr_engine* e, const void* d_x, void* d_x8, int64_t n_elems, float inv_sx);
int mocr_op_latent_attention_fp8(mocr_engine* e, const void* d_qt, const void* d_x8, void* d_out, int32_t n, int32_t len,
                                 int64_t x_batch_stride_bytes, float sx);

/* Fused query path of the latent attention (bf16 engines): d_x [rows_pad,768] bf16 (rows_pad = n rounded up to 128),
 * d_wq [768,768] bf16, d_bq [768] f32, d_wkT [768,768] bf16 (row n, column 64h+k = Wk_h[k][n]/8), d_qt [rows_pad,16,768] bf16:
 * d_qt[m][h] = bf16(x[m] . Wq_h^T + bq_h) . wkT_h   for the 12 heads. */
int mocr_op_qqt(mocr_engine* e, const void* d_x, const void* d_wq, const float* d_bq, const void* d_wkT, void* d_qt, int32_t n);

/* Decode-step HIP graphs this engine holds (test hook: the count must stay bounded whatever row counts callers submit). */
int mocr_graph_count(mocr_engine* e);
/* Row compactions this engine has performed (r04): between two chunks of decode steps the unfinished rows of a batch are
 * moved to the first decode slots and the following steps run on fewer slots - the counterpart of the reference's
 * one-generate()-per-crop loop, where a short text stops at its own EOS (TF/generation/utils.py:2929-2937 via
 * src/ui/main_window.py:9801).  Test hook / statistic; MOCR_FLAG_NO_COMPACTION keeps it at 0. */
int64_t mocr_compaction_count(mocr_engine* e);
/* Decode slots x greedy steps enqueued since mocr_create (statistic): the row-steps the decode launches were sized for.  The
 * tokens a caller got (sum of out_len - 1) over this number is the useful fraction of the decode work - 1.0 for the
 * reference's one-generate()-per-crop loop; mean / max length for a lock-step batch without compaction. */
int64_t mocr_decode_slot_steps(mocr_engine* e);
/* LayerNorm folding of the bf16 encoder (r03: the 24 LayerNorms between the persistent layer GEMMs applied in those GEMMs'
 * epilogues, the GEMMs reading bf16(x) instead of bf16(LN(x)); TF/models/vit/modeling_vit.py:266-286 is the unfolded form).
 * mocr_commit_weights measures, on eight probe crops, how much more input-rounding noise that costs on THIS checkpoint's
 * residual stream: *noise_ratio = sqrt(sum (x g rstd)^2 / sum LN(x)^2), worst LayerNorm (~1 on a near-normalised stream,
 * ~|mean| / spread on one with a DC offset).  Returns 1 when fat batches fold (ratio <= 1.5, or MOCR_FLAG_FORCE_LN_FOLD), 0
 * when the LayerNorms stay launches (ratio above, fp32 engine, MOCR_FLAG_NO_LN_FOLD). */
int mocr_ln_fold_state(mocr_engine* e, float* noise_ratio);
/* Free / total bytes of HBM on a device (the Python constructor sizes its default max_batch from it). */
int mocr_device_memory(int32_t device, int64_t* free_bytes, int64_t* total_bytes);

/* ---- per-kernel timing (HIP events on the engine's stream) -------------------------------- */
typedef struct mocr_kernel_stat {
    char name[48];
    int64_t launches;
    double total_ms;     /* sum of event-measured durations */
    double flops;        /* algorithmic FLOPs over those launches (2 per MAC) */
    double bytes;        /* algorithmic HBM bytes over those launches */
} mocr_kernel_stat;

int mocr_profile_enable(mocr_engine* e, int32_t on); /* on: record an event pair around every launch */
int mocr_profile_reset(mocr_engine* e);
int mocr_profile_get(mocr_engine* e, mocr_kernel_stat* out, int32_t cap, int32_t* n_out);

#ifdef __cplusplus
}
#endif
#endif /* MOCR_H */
