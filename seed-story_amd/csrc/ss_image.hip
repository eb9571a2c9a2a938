// Image pre-processing on the device: PIL-exact separable resample of a uint8 HWC image +
// ToTensor + Normalize + cast, i.e. the reference's `image_transform(image)` (src/processer/transforms.py:4-19 —
// torchvision Resize on a PIL image IS Pillow's ImagingResample — then ToTensor /255, Normalize(mean, std)) followed by
// `.to(device, dtype)` (gen_george.py:166), producing the ViT's input [3, S, S] directly in HBM.
//
// Pillow's 8-bit resampler is integer arithmetic, so the bar is bit-exact on the uint8 stage:
//   * per output coordinate a window [xmin, xmin+n) of taps; tap weights = filter((x - center + 0.5) / filterscale)
//     in double, normalised to sum 1, then quantised to 22-bit fixed point with round-half-away (precompute + normalize
//     coefficient steps of Pillow's Resample.c — computed on the HOST here by ss_resample_coeffs, exactly that way);
//   * each pass accumulates  (1 << 21) + sum(pixel * coeff)  in int32 and stores clip8(acc >> 22);
//   * horizontal pass first into a uint8 intermediate, then the vertical pass.
// The float tail is fp32 in torch's op order: v = u8 / 255.0f; v = (v - mean[c]) / std[c]; cast RNE to the model dtype.
//
// HBM-trivial (a 1024x1024 RGB source is 3 MB): two gather kernels, reads coalesced along x, no LDS needed.
#include <math.h>

#include "ss_common.h"

namespace ss {

constexpr int kPrecisionBits = 32 - 8 - 2;

__device__ __forceinline__ uint8_t clip8(int v) {
    v >>= kPrecisionBits;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// src [H, W, 3] u8 -> tmp [H, OW, 3] u8 (rows y0 .. y0+rows of the source only)
__global__ void resample_h_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ tmp, int W, int OW, int rows, int y0,
                                  const int* __restrict__ coef, const int* __restrict__ bounds, int ksize) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     // over rows * OW * 3
    if (i >= (int64_t)rows * OW * 3) return;
    const int c = (int)(i % 3), ox = (int)((i / 3) % OW), r = (int)(i / (3 * (int64_t)OW));
    const int xmin = bounds[2 * ox], n = bounds[2 * ox + 1];
    const int* k = coef + (int64_t)ox * ksize;
    const uint8_t* row = src + ((int64_t)(y0 + r) * W + xmin) * 3 + c;
    int acc = 1 << (kPrecisionBits - 1);
    for (int x = 0; x < n; ++x) acc += (int)row[3 * x] * k[x];
    tmp[i] = clip8(acc);
}

// tmp [rows, OW, 3] u8 (row 0 = source row y0) -> dst [3, OH', OW'] T over the crop window, optionally also u8 HWC
template <typename T>
__global__ void resample_v_norm_kernel(const uint8_t* __restrict__ tmp, T* __restrict__ dst, uint8_t* __restrict__ dst_u8,
                                       int OW, int y0, int crop_top, int crop_left, int CH, int CW,
                                       const int* __restrict__ coef, const int* __restrict__ bounds, int ksize,
                                       float m0, float m1, float m2, float s0, float s1, float s2) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     // over CH * CW * 3
    if (i >= (int64_t)CH * CW * 3) return;
    const int c = (int)(i % 3), cx = (int)((i / 3) % CW), cy = (int)(i / (3 * (int64_t)CW));
    const int oy = cy + crop_top, ox = cx + crop_left;
    const int ymin = bounds[2 * oy], n = bounds[2 * oy + 1];
    const int* k = coef + (int64_t)oy * ksize;
    const uint8_t* col = tmp + ((int64_t)(ymin - y0) * OW + ox) * 3 + c;
    int acc = 1 << (kPrecisionBits - 1);
    for (int y = 0; y < n; ++y) acc += (int)col[(int64_t)y * OW * 3] * k[y];
    const uint8_t u = clip8(acc);
    if (dst_u8) dst_u8[i] = u;
    const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
    float v = (float)u / 255.0f;            // ToTensor
    v = (v - mean) / sd;                    // Normalize: sub_ then div_
    Tr<T>::st(dst + ((int64_t)c * CH + cy) * CW + cx, v);
}

template <typename T>
int preprocess_launch(const void* src, int64_t /*H*/, int64_t W, void* dst, void* dst_u8, int64_t /*OH*/, int64_t OW, int64_t crop_top,
                      int64_t crop_left, int64_t CH, int64_t CW, const int* coef_h, const int* bounds_h, int ksize_h,
                      const int* coef_v, const int* bounds_v, int ksize_v, int y0, int rows, void* tmp, const float* mean,
                      const float* std, hipStream_t stream) {
    const int64_t n1 = (int64_t)rows * OW * 3, n2 = CH * CW * 3;
    hipLaunchKernelGGL(resample_h_kernel, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, stream, (const uint8_t*)src,
                       (uint8_t*)tmp, (int)W, (int)OW, rows, y0, coef_h, bounds_h, ksize_h);
    SS_LAUNCH_CHECK("resample_h");
    hipLaunchKernelGGL(resample_v_norm_kernel<T>, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, stream, (const uint8_t*)tmp,
                       (T*)dst, (uint8_t*)dst_u8, (int)OW, y0, (int)crop_top, (int)crop_left, (int)CH, (int)CW, coef_v,
                       bounds_v, ksize_v, mean[0], mean[1], mean[2], std[0], std[1], std[2]);
    SS_LAUNCH_CHECK("resample_v_norm");
    return SS_OK;
}

static double filt_bilinear(double x) {
    if (x < 0.0) x = -x;
    return x < 1.0 ? 1.0 - x : 0.0;
}
static double filt_bicubic(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

}  // namespace ss

extern "C" {

int ss_resample_ksize(int64_t in_size, int64_t out_size, int filter) {
    if (in_size <= 0 || out_size <= 0 || (filter != SS_FILTER_BILINEAR && filter != SS_FILTER_BICUBIC)) return -1;
    double filterscale = (double)in_size / (double)out_size;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = (filter == SS_FILTER_BICUBIC ? 2.0 : 1.0) * filterscale;
    return (int)ceil(support) * 2 + 1;
}

int ss_resample_coeffs(int64_t in_size, int64_t out_size, int filter, int32_t* coef, int32_t* bounds) {
    const int ksize = ss_resample_ksize(in_size, out_size, filter);
    SS_REQUIRE(ksize > 0 && coef && bounds, "ss_resample_coeffs: bad arguments");
    double (*f)(double) = filter == SS_FILTER_BICUBIC ? ss::filt_bicubic : ss::filt_bilinear;
    const double scale = (double)in_size / (double)out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = (filter == SS_FILTER_BICUBIC ? 2.0 : 1.0) * filterscale;
    const double ss = 1.0 / filterscale;
    double* k = new double[ksize];
    for (int64_t xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale;
        double ww = 0.0;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = (int)in_size;
        xmax -= xmin;
        int x = 0;
        for (; x < xmax; ++x) {
            const double w = f((x + xmin - center + 0.5) * ss);
            k[x] = w;
            ww += w;
        }
        for (x = 0; x < xmax; ++x)
            if (ww != 0.0) k[x] /= ww;
        for (; x < ksize; ++x) k[x] = 0.0;
        for (x = 0; x < ksize; ++x) {
            const double v = k[x] * (double)(1 << ss::kPrecisionBits);
            coef[xx * ksize + x] = k[x] < 0 ? (int32_t)(-0.5 + v) : (int32_t)(0.5 + v);
        }
        bounds[2 * xx] = xmin;
        bounds[2 * xx + 1] = xmax;
    }
    delete[] k;
    return SS_OK;
}

int ss_image_preprocess(const void* src_u8_hwc, int64_t H, int64_t W, void* dst_chw, void* dst_u8_hwc, int64_t OH, int64_t OW,
                        int64_t crop_top, int64_t crop_left, int64_t CH, int64_t CW, const int32_t* coef_h,
                        const int32_t* bounds_h, int ksize_h, const int32_t* coef_v, const int32_t* bounds_v, int ksize_v,
                        int64_t first_row, int64_t n_rows, void* tmp_u8, const float mean[3], const float std[3], int dtype,
                        void* stream) {
    SS_REQUIRE(src_u8_hwc && dst_chw && tmp_u8 && coef_h && bounds_h && coef_v && bounds_v && mean && std,
               "ss_image_preprocess: NULL argument");
    SS_REQUIRE(H > 0 && W > 0 && OH > 0 && OW > 0 && CH > 0 && CW > 0 && crop_top >= 0 && crop_left >= 0 &&
                   crop_top + CH <= OH && crop_left + CW <= OW && first_row >= 0 && n_rows > 0 && first_row + n_rows <= H,
               "ss_image_preprocess: bad geometry");
    return SS_DISPATCH(dtype, ss::preprocess_launch, src_u8_hwc, H, W, dst_chw, dst_u8_hwc, OH, OW, crop_top, crop_left, CH, CW,
                       coef_h, bounds_h, ksize_h, coef_v, bounds_v, ksize_v, (int)first_row, (int)n_rows, tmp_u8, mean, std,
                       (hipStream_t)stream);
}

}  // extern "C"
