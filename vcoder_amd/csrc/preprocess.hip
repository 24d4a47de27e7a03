// preprocess.hip — device-side image preprocessing: the step right before the encoders.
//
// Reproduces, bit for bit on the uint8 stages, what the reference does on the host with PIL + HF CLIPImageProcessor:
//   expand2square(image, mean colour)                          vcoder_llava/mm_utils.py:14-25,31-35
//   resize (bicubic, shortest edge -> S) + center crop S x S   CLIPImageProcessor.preprocess -> PIL Image.resize
//   rescale 1/255, normalise (x - mean) / std, HWC -> CHW      [HF] image_transforms.rescale / normalize
// PIL's 8-bit resampler is separable with per-output-pixel windows and 22-bit fixed-point coefficients that are
// normalised in double, rounded half away from zero, accumulated from 1<<21 and shifted down — horizontal pass first,
// with the intermediate image rounded to uint8.  The coefficient tables are built on the host (engine.hip,
// `resample_coeffs`); these kernels apply them.  HBM-bound and tiny next to one ViT forward; the point is that 8 GPUs
// at > 100 images/s are not fed by one Python thread running PIL.
#include "vc_device.h"
#include "kernels.h"

namespace vc {

// square canvas filled with `fill`, source pasted at (ox, oy)     (Image.new + paste of expand2square)
__global__ __launch_bounds__(256) void pad_square_kernel(const uint8_t* src, int h, int w, uint8_t* dst, int side, int ox,
                                                         int oy, int f0, int f1, int f2) {
    const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= (size_t)side * side) return;
    const int y = (int)(id / side), x = (int)(id % side);
    const int sy = y - oy, sx = x - ox;
    uint8_t* d = dst + id * 3;
    if (sy >= 0 && sy < h && sx >= 0 && sx < w) {
        const uint8_t* s = src + ((size_t)sy * w + sx) * 3;
        d[0] = s[0]; d[1] = s[1]; d[2] = s[2];
    } else {
        d[0] = (uint8_t)f0; d[1] = (uint8_t)f1; d[2] = (uint8_t)f2;
    }
}
void launch_pad_square(const uint8_t* src, int h, int w, uint8_t* dst, int side, int ox, int oy, const int fill[3],
                       hipStream_t s) {
    const size_t n = (size_t)side * side;
    VC_LAUNCH(pad_square_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, h, w, dst, side, ox, oy, fill[0],
              fill[1], fill[2]);
}

VC_DEV uint8_t clip8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

// one pass of the separable resampler along x (HORIZ) or y: out[i][o][c] = clip8((2^21 + sum_t in[.. x0+t ..]*kk[o][t]) >> 22)
template <bool HORIZ>
__global__ __launch_bounds__(256) void resample_kernel(const uint8_t* in, int in_h, int in_w, uint8_t* out, int out_h,
                                                       int out_w, const int* bounds, const int* kk, int ksize) {
    const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= (size_t)out_h * out_w) return;
    const int y = (int)(id / out_w), x = (int)(id % out_w);
    const int o = HORIZ ? x : y;
    const int x0 = bounds[2 * o], n = bounds[2 * o + 1];
    const int* k = kk + (size_t)o * ksize;
    int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
    for (int t = 0; t < n; ++t) {
        const uint8_t* p = HORIZ ? in + ((size_t)y * in_w + x0 + t) * 3 : in + ((size_t)(x0 + t) * in_w + x) * 3;
        const int c = k[t];
        s0 += p[0] * c; s1 += p[1] * c; s2 += p[2] * c;
    }
    uint8_t* d = out + id * 3;
    d[0] = clip8(s0 >> 22); d[1] = clip8(s1 >> 22); d[2] = clip8(s2 >> 22);
}
void launch_resample(const uint8_t* in, int in_h, int in_w, uint8_t* out, int out_h, int out_w, const int* bounds,
                     const int* kk, int ksize, int horizontal, hipStream_t s) {
    const size_t n = (size_t)out_h * out_w;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (horizontal) VC_LAUNCH((resample_kernel<true>), grid, block, 0, s, in, in_h, in_w, out, out_h, out_w, bounds, kk, ksize);
    else VC_LAUNCH((resample_kernel<false>), grid, block, 0, s, in, in_h, in_w, out, out_h, out_w, bounds, kk, ksize);
}

// center crop + rescale + normalise + HWC->CHW:  float32( double(u8) * (1/255) ), then (x - mean) / std in fp32
__global__ __launch_bounds__(256) void crop_normalize_kernel(const uint8_t* in, int in_h, int in_w, int top, int left,
                                                             float* out, int S, float m0, float m1, float m2, float d0,
                                                             float d1, float d2) {
    const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= (size_t)S * S) return;
    const int y = (int)(id / S), x = (int)(id % S);
    const uint8_t* p = in + ((size_t)(y + top) * in_w + (x + left)) * 3;
    const float mean[3] = {m0, m1, m2}, sd[3] = {d0, d1, d2};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = (float)((double)p[c] * 0.00392156862745098);
        out[(size_t)c * S * S + id] = (v - mean[c]) / sd[c];
    }
}
void launch_crop_normalize(const uint8_t* in, int in_h, int in_w, int top, int left, float* out, int S, const float mean[3],
                           const float stdv[3], hipStream_t s) {
    const size_t n = (size_t)S * S;
    VC_LAUNCH(crop_normalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, in_h, in_w, top, left, out, S,
              mean[0], mean[1], mean[2], stdv[0], stdv[1], stdv[2]);
}

}  // namespace vc
