// Image half of the input pipeline: decoded RGB uint8 images of arbitrary size -> bicubic 224 x 224
// (bit-exact with Pillow's Image.resize, which torchvision's F.resize calls for PIL inputs: UniMP transforms.py:102-136)
// -> ToTensor (/255) -> Normalize(mean, std) (rec_dataset.py:30-31, 91-107) -> [n][3][224][224] bf16 / f32.
//
// Pillow's 8-bit resampler is two separable integer passes (horizontal first, uint8 in between) with 22-bit fixed-point
// taps; the host builds the tap tables exactly as Pillow does (double arithmetic, unimp_amd/data.py) and the kernels do
// the byte work: HBM-bound, one thread per output pixel (3 channels), taps and source bytes served by L1/L2.
#include "common.h"
#include "unimp_hip.h"

#define PBITS 22

__device__ __forceinline__ uint8_t clip8(int v) { v >>= PBITS; return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// horizontal pass: src [H][W][3] -> tmp [H][out_w][3]
__global__ __launch_bounds__(256) void resize_h_kernel(const uint8_t* __restrict__ src, const unimp_image_desc* __restrict__ descs,
                                                       const int32_t* __restrict__ tables, uint8_t* __restrict__ tmp, int out_w) {
  const unimp_image_desc d = descs[blockIdx.y];
  if (d.ksx == 0) return;
  const uint8_t* im = src + d.src_off;
  uint8_t* to = tmp + d.tmp_off;
  const int rowlen = (int)d.ksx + 2;
  const long total = d.H * out_w;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int y = (int)(i / out_w), xo = (int)(i - (long)y * out_w);
    const int32_t* k = tables + d.kx_off + (long)xo * rowlen;
    int first = k[0], n = k[1];
    const uint8_t* p = im + ((long)y * d.W + first) * 3;
    int a0 = 1 << (PBITS - 1), a1 = a0, a2 = a0;
    for (int t = 0; t < n; ++t) { int w = k[2 + t]; a0 += p[3 * t] * w; a1 += p[3 * t + 1] * w; a2 += p[3 * t + 2] * w; }
    uint8_t* q = to + i * 3;
    q[0] = clip8(a0); q[1] = clip8(a1); q[2] = clip8(a2);
  }
}

// vertical pass + ToTensor + Normalize: in [H][out_w][3] (tmp, or src when the width was not resampled) -> out [3][out_h][out_w]
template <bool F32>
__global__ __launch_bounds__(256) void resize_v_norm_kernel(const uint8_t* __restrict__ src, const uint8_t* __restrict__ tmp,
                                                            const unimp_image_desc* __restrict__ descs, const int32_t* __restrict__ tables,
                                                            int out_h, int out_w, float m0, float m1, float m2, float s0, float s1, float s2,
                                                            void* __restrict__ out, uint8_t* __restrict__ out_u8) {
  const int img = blockIdx.y;
  const unimp_image_desc d = descs[img];
  const uint8_t* in = d.ksx == 0 ? src + d.src_off : tmp + d.tmp_off;
  const int rowlen = (int)d.ksy + 2;
  const int total = out_h * out_w;
  const long plane = (long)total;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    int yo = i / out_w, xo = i - yo * out_w;
    uint8_t c0, c1, c2;
    if (d.ksy == 0) {
      const uint8_t* p = in + ((long)yo * out_w + xo) * 3;
      c0 = p[0]; c1 = p[1]; c2 = p[2];
    } else {
      const int32_t* k = tables + d.ky_off + (long)yo * rowlen;
      int first = k[0], n = k[1];
      const uint8_t* p = in + ((long)first * out_w + xo) * 3;
      int a0 = 1 << (PBITS - 1), a1 = a0, a2 = a0;
      for (int t = 0; t < n; ++t) { int w = k[2 + t]; const uint8_t* r = p + (long)t * out_w * 3; a0 += r[0] * w; a1 += r[1] * w; a2 += r[2] * w; }
      c0 = clip8(a0); c1 = clip8(a1); c2 = clip8(a2);
    }
    if (out_u8) { uint8_t* q = out_u8 + ((long)img * total + i) * 3; q[0] = c0; q[1] = c1; q[2] = c2; }
    // ToTensor: uint8 -> float32 / 255;  Normalize: (x - mean) / std, float32 (IEEE division, no fast-math)
    float v0 = ((float)c0 / 255.0f - m0) / s0, v1 = ((float)c1 / 255.0f - m1) / s1, v2 = ((float)c2 / 255.0f - m2) / s2;
    long base = (long)img * 3 * plane + i;
    if (F32) { float* o = (float*)out; o[base] = v0; o[base + plane] = v1; o[base + 2 * plane] = v2; }
    else { bf16* o = (bf16*)out; o[base] = f2bf(v0); o[base + plane] = f2bf(v1); o[base + 2 * plane] = f2bf(v2); }
  }
}

extern "C" int unimp_image_resize_normalize(const uint8_t* src, const unimp_image_desc* descs, int n_images, int max_h,
                                            const int32_t* tables, uint8_t* tmp, int out_h, int out_w, const float* mean,
                                            const float* std, void* out, int out_f32, uint8_t* out_u8, void* stream) {
  if (n_images <= 0) return UNIMP_OK;
  if (!src || !descs || !tables || !tmp || !out || !mean || !std) return unimp_set_error(UNIMP_ERR_ARG, "image_resize_normalize: null pointer");
  if (out_h <= 0 || out_w <= 0 || max_h <= 0) return unimp_set_error(UNIMP_ERR_SHAPE, "image_resize_normalize: empty output / max_h");
  for (int c = 0; c < 3; ++c) if (!(std[c] > 0.f)) return unimp_set_error(UNIMP_ERR_ARG, "image_resize_normalize: std must be positive");
  hipStream_t s = (hipStream_t)stream;
  long hpix = (long)max_h * out_w;
  dim3 gh((unsigned)((hpix + 255) / 256 > 4096 ? 4096 : (hpix + 255) / 256), n_images), gv((out_h * out_w + 255) / 256, n_images);
  hipLaunchKernelGGL(resize_h_kernel, gh, dim3(256), 0, s, src, descs, tables, tmp, out_w);
  if (out_f32) hipLaunchKernelGGL((resize_v_norm_kernel<true>), gv, dim3(256), 0, s, src, tmp, descs, tables, out_h, out_w,
                                  mean[0], mean[1], mean[2], std[0], std[1], std[2], out, out_u8);
  else hipLaunchKernelGGL((resize_v_norm_kernel<false>), gv, dim3(256), 0, s, src, tmp, descs, tables, out_h, out_w,
                          mean[0], mean[1], mean[2], std[0], std[1], std[2], out, out_u8);
  return unimp_check_launch("image_resize_normalize");
}
