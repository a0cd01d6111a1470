// Device-side preprocessing in front of the recogniser: what MangaOcr.__call__ and the HF image
// processor do to a crop before the encoder sees it:
//
//   img.convert('L')                    Pillow libImaging/Convert.c:  L = (19595 R + 38470 G + 7471 B + 0x8000) >> 16
//   .convert('RGB') -> resize((224,224), BILINEAR)
//                                       TF/models/vit/image_processing_pil_vit.py:20-27, TF/image_processing_backends.py:521-570
//                                       -> Pillow libImaging/Resample.c (ImagingResample, 8 bits per channel):
//                                       separable triangle filter whose support grows with the down-scale
//                                       factor, coefficients rounded to 22 fractional bits, horizontal pass
//                                       (rounded, clamped to uint8) and then vertical pass.
//
// The three channels are equal after convert('L').convert('RGB'), so one luminance plane is resized.
// Everything is integer arithmetic with Pillow's rounding, so the result is BIT-EXACT with Pillow
// (tests/test_gpu_preprocess.py against oracle/pil_ops.py and tests/golden/preprocess.npz, which was
// written by Pillow itself).  The coefficient tables are computed on the host in double precision in
// exactly Resample.c's operation order (precompute_coeffs + normalize_coeffs_8bpc); they depend only
// on (input size, 224) and are cached per input size.
#pragma once
#include <cmath>
#include <cstdint>
#include <map>
#include <vector>

#include "common.h"

#define MOCR_RS_PRECISION_BITS (32 - 8 - 2)

struct ResampleTable {
    int ksize = 0;
    std::vector<int> bounds;   // [out][2] = (first input index, tap count)
    std::vector<int> kk;       // [out][ksize] fixed-point coefficients
};

// precompute_coeffs(inSize, in0 = 0, in1 = inSize, outSize, BILINEAR) + normalize_coeffs_8bpc
static ResampleTable make_resample_table(int in_size, int out_size) {
#pragma clang fp contract(off)      // Resample.c's doubles are not fused: keep a*b+c as two roundings
    ResampleTable t;
    const double scale = (double)in_size / out_size;
    double filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 1.0 * filterscale;            // triangle filter: support 1.0
    t.ksize = (int)std::ceil(support) * 2 + 1;
    t.bounds.assign((size_t)out_size * 2, 0);
    t.kk.assign((size_t)out_size * t.ksize, 0);
    const double ss = 1.0 / filterscale;
    std::vector<double> w(t.ksize);
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale;
        double ww = 0.0;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        for (int x = 0; x < xmax; ++x) {
            double a = (x + xmin - center + 0.5) * ss;
            if (a < 0.0) a = -a;
            const double v = a < 1.0 ? 1.0 - a : 0.0;
            w[x] = v;
            ww += v;
        }
        for (int x = 0; x < xmax; ++x) {
            if (ww != 0.0) w[x] /= ww;
            const double v = w[x] * (double)(1 << MOCR_RS_PRECISION_BITS);
            t.kk[(size_t)xx * t.ksize + x] = w[x] < 0 ? (int)(-0.5 + v) : (int)(0.5 + v);
        }
        t.bounds[2 * xx] = xmin;
        t.bounds[2 * xx + 1] = xmax;
    }
    return t;
}

// One image of a preprocessing batch (device-side descriptor).  The image the resize SEES has h rows of w pixels; pixel
// (y, x) of it sits at src_off + y * row_step + x * pix_step in the packed source buffer.  An unrotated crop has
// (row_step, pix_step) = (bytes per source row, bytes per pixel); the reference's orientation-only rotation
// (src/core/workers.py:320-326, src/ui/main_window.py:9787-9795: cv2.ROTATE_90_CLOCKWISE / _COUNTERCLOCKWISE before the
// recogniser) is the same crop read with swapped, signed steps - no copy on the host, no extra pass on the device.
struct ResizeDesc {
    long long src_off;     // byte offset of pixel (0, 0) of the (rotated) image in the packed source buffer
    long long tmp_off;     // byte offset of its [h][OUT] horizontal-pass plane
    int h, w, row_step, pix_step, channels;
    int bgr;               // 3-channel pixels are stored B,G,R (OpenCV order, what the reference's crop tools hold)
    int kx_off, bx_off, ksx;   // int offsets into the coefficient / bounds buffers, taps per output (0: pass skipped)
    int ky_off, by_off, ksy;
};

__device__ __forceinline__ int rs_luma(const uint8_t* p, int channels, int bgr) {
    if (channels == 1) return p[0];
    // BGR input: the reference converts BGR -> RGB on the host first (src/ui/main_window.py:9800, cv2.COLOR_BGR2RGB),
    // a pure channel swap, so reading the channels in the other order is the same arithmetic
    const unsigned r = bgr ? p[2] : p[0], b = bgr ? p[0] : p[2];
    return (int)((19595u * r + 38470u * p[1] + 7471u * b + 0x8000u) >> 16);
}
__device__ __forceinline__ uint8_t rs_clip8(int v) {
    v >>= MOCR_RS_PRECISION_BITS;              // arithmetic shift, like clip8_lookups[in >> PRECISION_BITS]
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// Horizontal pass (with the luminance conversion fused): tmp[img][y][xx], xx < OUT.  grid (row blocks, images),
// 256 threads = one row of OUT (<= 256) outputs at a time, ROWS rows per block.
template <int OUT, int ROWS>
__global__ __launch_bounds__(256) void resize_h_kernel(const uint8_t* __restrict__ src, const ResizeDesc* __restrict__ descs,
                                                       const int* __restrict__ coef, const int* __restrict__ bounds,
                                                       uint8_t* __restrict__ tmp) {
    const ResizeDesc d = descs[blockIdx.y];
    const int xx = threadIdx.x;
    if (xx >= OUT) return;
    const int y0 = blockIdx.x * ROWS;
    if (y0 >= d.h) return;
    int xmin = xx, cnt = 1;
    const int* k = nullptr;
    if (d.ksx) {
        xmin = bounds[d.bx_off + 2 * xx];
        cnt = bounds[d.bx_off + 2 * xx + 1];
        k = coef + d.kx_off + xx * d.ksx;
    }
    for (int y = y0; y < min(y0 + ROWS, d.h); ++y) {
        const uint8_t* row = src + d.src_off + (long long)y * d.row_step;
        uint8_t o;
        if (!d.ksx) {                        // width already OUT: Pillow skips the pass
            o = (uint8_t)rs_luma(row + (long long)xx * d.pix_step, d.channels, d.bgr);
        } else {
            int acc = 1 << (MOCR_RS_PRECISION_BITS - 1);
            for (int x = 0; x < cnt; ++x) acc += rs_luma(row + (long long)(xmin + x) * d.pix_step, d.channels, d.bgr) * k[x];
            o = rs_clip8(acc);
        }
        tmp[d.tmp_off + (size_t)y * OUT + xx] = o;
    }
}

// Vertical pass: out[img][yy][xx].  grid (OUT / ROWS, images).
template <int OUT, int ROWS>
__global__ __launch_bounds__(256) void resize_v_kernel(const uint8_t* __restrict__ tmp, const ResizeDesc* __restrict__ descs,
                                                       const int* __restrict__ coef, const int* __restrict__ bounds,
                                                       uint8_t* __restrict__ out) {
    const ResizeDesc d = descs[blockIdx.y];
    const int xx = threadIdx.x;
    if (xx >= OUT) return;
    const uint8_t* plane = tmp + d.tmp_off;
    uint8_t* o = out + (size_t)blockIdx.y * OUT * OUT;
    for (int yy = blockIdx.x * ROWS; yy < min((int)(blockIdx.x + 1) * ROWS, OUT); ++yy) {
        if (!d.ksy) {                        // height already OUT
            o[(size_t)yy * OUT + xx] = plane[(size_t)yy * OUT + xx];
            continue;
        }
        const int ymin = bounds[d.by_off + 2 * yy], cnt = bounds[d.by_off + 2 * yy + 1];
        const int* k = coef + d.ky_off + yy * d.ksy;
        int acc = 1 << (MOCR_RS_PRECISION_BITS - 1);
        for (int y = 0; y < cnt; ++y) acc += (int)plane[(size_t)(ymin + y) * OUT + xx] * k[y];
        o[(size_t)yy * OUT + xx] = rs_clip8(acc);
    }
}
