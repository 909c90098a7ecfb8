// LayerNorm / RMSNorm forward + backward for gfx950.  HBM-bound: one wave64 per row, the row lives in
// registers (16-byte bf16x8 loads, D/8 chunks striped over the 64 lanes), statistics in fp32 through
// wave shuffles, no LDS in the forward.  Backward computes dx per row and keeps per-wave dgamma/dbeta
// partial sums in registers over a grid-stride loop, combines the block's 4 waves through LDS and writes
// one fp32 partial row per block; a second kernel column-sums the partials (deterministic, no atomics).
#include "common.h"
#include "unimp_hip.h"

template <int MAXC>
struct RowRegs { float v[MAXC][8]; };

__device__ __forceinline__ long map_row(int r, int grp, int grp_stride, int grp_off) {
  return grp ? (long)(r / grp) * grp_stride + (r % grp) + grp_off : (long)r;
}

template <int MAXC>
__device__ __forceinline__ void load_row(const bf16* __restrict__ p, int nch, float (&v)[MAXC][8]) {
  int lane = lane_id();
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    int ch = c * 64 + lane;
    if (ch < nch) {
      bf16x8 t = *(const bf16x8*)(p + ch * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[c][j] = bf2f(t[j]);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[c][j] = 0.f;
    }
  }
}

template <int MAXC>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const bf16* __restrict__ x, long ldx, const bf16* __restrict__ gamma,
                                                     const bf16* __restrict__ beta, bf16* __restrict__ y, long ldy,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int rows, int D,
                                                     float eps, int rms, int grp, int grp_stride, int grp_off) {
  int lane = lane_id();
  int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int nwaves = (gridDim.x * blockDim.x) >> 6;
  int nch = D >> 3;
  float inv_d = 1.f / (float)D;
  for (int r = wave; r < rows; r += nwaves) {
    float v[MAXC][8];
    load_row<MAXC>(x + (long)r * ldx, nch, v);
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[c][j];
    float mu = rms ? 0.f : wave_sum(s) * inv_d;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      bool ok = c * 64 + lane < nch;
#pragma unroll
      for (int j = 0; j < 8; ++j) { float d = ok ? v[c][j] - mu : 0.f; q += d * d; }
    }
    float var = wave_sum(q) * inv_d;
    float rs = rsqrtf(var + eps);
    if (lane == 0) { if (mean) mean[r] = mu; rstd[r] = rs; }
    bf16* yo = y + map_row(r, grp, grp_stride, grp_off) * ldy;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      int ch = c * 64 + lane;
      if (ch < nch) {
        bf16x8 g = *(const bf16x8*)(gamma + ch * 8);
        bf16x8 o;
        if (beta) {
          bf16x8 b = *(const bf16x8*)(beta + ch * 8);
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = f2bf((v[c][j] - mu) * rs * bf2f(g[j]) + bf2f(b[j]));
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = f2bf((v[c][j] - mu) * rs * bf2f(g[j]));
        }
        *(bf16x8*)(yo + ch * 8) = o;
      }
    }
  }
}

// dx = rstd * (g*dy - mean(g*dy) - xhat * mean(g*dy*xhat))      [LayerNorm]
// dx = rstd * (g*dy - xhat * mean(g*dy*xhat))                   [RMSNorm, xhat = x*rstd]
template <int MAXC, bool WGRAD>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16* __restrict__ dy, long lddy, const bf16* __restrict__ dy2, long lddy2,
                                                     const bf16* __restrict__ x, long ldx,
                                                     const bf16* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const bf16* __restrict__ dres, long lddres,
                                                     bf16* __restrict__ dx, long lddx, float* __restrict__ partial,
                                                     int rows, int D, int rms, int grp, int grp_stride, int grp_off) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int lane = lane_id();
  int wib = threadIdx.x >> 6;
  int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int nwaves = (gridDim.x * blockDim.x) >> 6;
  int nch = D >> 3;
  float inv_d = 1.f / (float)D;
  float ag[MAXC][8], ab[MAXC][8];
  if (WGRAD) {
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j) { ag[c][j] = 0.f; ab[c][j] = 0.f; }
  }
  for (int r = wave; r < rows; r += nwaves) {
    float xv[MAXC][8], gy[MAXC][8];
    load_row<MAXC>(x + (long)r * ldx, nch, xv);
    load_row<MAXC>(dy + map_row(r, grp, grp_stride, grp_off) * lddy, nch, gy);
    if (dy2) {
      float g2[MAXC][8];
      load_row<MAXC>(dy2 + (long)r * lddy2, nch, g2);
#pragma unroll
      for (int c = 0; c < MAXC; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) gy[c][j] += g2[c][j];
    }
    float mu = rms ? 0.f : mean[r];
    float rs = rstd[r];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      int ch = c * 64 + lane;
      if (ch < nch) {
        bf16x8 g = *(const bf16x8*)(gamma + ch * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float xh = (xv[c][j] - mu) * rs;
          float d = gy[c][j];
          if (WGRAD) { ag[c][j] += d * xh; ab[c][j] += d; }
          float gd = d * bf2f(g[j]);
          xv[c][j] = xh; gy[c][j] = gd;
          s1 += gd; s2 += gd * xh;
        }
      }
    }
    s1 = rms ? 0.f : wave_sum(s1) * inv_d;
    s2 = wave_sum(s2) * inv_d;
    bf16* o = dx + (long)r * lddx;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      int ch = c * 64 + lane;
      if (ch < nch) {
        bf16x8 ov;
        if (dres) {
          bf16x8 rr = *(const bf16x8*)(dres + (long)r * lddres + ch * 8);
#pragma unroll
          for (int j = 0; j < 8; ++j) ov[j] = f2bf(rs * (gy[c][j] - s1 - xv[c][j] * s2) + bf2f(rr[j]));
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) ov[j] = f2bf(rs * (gy[c][j] - s1 - xv[c][j] * s2));
        }
        *(bf16x8*)(o + ch * 8) = ov;
      }
    }
  }
  if (WGRAD) {
    // combine the 4 waves of the block through LDS [4][D] fp32, dgamma then dbeta
    float* sh = (float*)smem;
#pragma unroll
    for (int which = 0; which < 2; ++which) {
#pragma unroll
      for (int c = 0; c < MAXC; ++c) {
        int ch = c * 64 + lane;
        if (ch < nch) {
#pragma unroll
          for (int j = 0; j < 8; ++j) sh[wib * D + ch * 8 + j] = which ? ab[c][j] : ag[c][j];
        }
      }
      __syncthreads();
      for (int col = threadIdx.x; col < D; col += blockDim.x)
        partial[((long)blockIdx.x * 2 + which) * D + col] = sh[col] + sh[D + col] + sh[2 * D + col] + sh[3 * D + col];
      __syncthreads();
    }
  }
}

// column sums of the per-block partials [nblk][2][D]: a block owns 32 columns of dgamma or dbeta, 8 row groups x 32 lanes
// read 128 contiguous bytes per partial row, LDS combines the 8 groups.
__global__ __launch_bounds__(256) void ln_wgrad_reduce_kernel(const float* __restrict__ partial, int nblk, int D, bf16* __restrict__ dgamma,
                                                              bf16* __restrict__ dbeta) {
  __shared__ float sh[8][32];
  int cb = blockIdx.x * 32, which = blockIdx.y;
  int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
  int col = cb + cl;
  float t = 0.f;
  if (col < D)
    for (int b = rg; b < nblk; b += 8) t += partial[((long)b * 2 + which) * D + col];
  sh[rg][cl] = t;
  __syncthreads();
  if (rg == 0 && col < D) {
    float a = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) a += sh[g][cl];
    if (which == 0) dgamma[col] = f2bf(a);
    else if (dbeta) dbeta[col] = f2bf(a);
  }
}

#include <stdlib.h>
static inline int ln_grid(int rows) {
  static const int cap = [] { const char* e = getenv("UNIMP_LN_GRID"); return e ? atoi(e) : 8192; }();
  int blocks = (rows + 3) / 4;
  return blocks < 1 ? 1 : (blocks > cap ? cap : blocks);
}

extern "C" int unimp_layernorm_fwd(const void* x, int64_t ldx, const void* gamma, const void* beta, void* y, int64_t ldy,
                                   float* mean, float* rstd, int rows, int D, float eps, int rms, int grp, int grp_stride,
                                   int grp_off, void* stream) {
  if (!x || !gamma || !y || !rstd) return unimp_set_error(UNIMP_ERR_ARG, "layernorm_fwd: null pointer");
  if (rows <= 0) return UNIMP_OK;
  if ((D & 7) || D > 4096 || (ldx & 7) || (ldy & 7)) return unimp_set_error(UNIMP_ERR_SHAPE, "layernorm_fwd: need D%8==0, D<=4096, ld%8==0");
  hipStream_t s = (hipStream_t)stream;
  dim3 g(ln_grid(rows)), b(256);
#define LN_FWD(MC) hipLaunchKernelGGL((ln_fwd_kernel<MC>), g, b, 0, s, (const bf16*)x, (long)ldx, (const bf16*)gamma, (const bf16*)beta, \
                                      (bf16*)y, (long)ldy, mean, rstd, rows, D, eps, rms, grp, grp_stride, grp_off)
  if (D <= 1024) LN_FWD(2); else if (D <= 2560) LN_FWD(5); else LN_FWD(8);
#undef LN_FWD
  return unimp_check_launch("layernorm_fwd");
}

extern "C" int unimp_layernorm_bwd(const void* dy, int64_t lddy, const void* dy2, int64_t lddy2, const void* x, int64_t ldx, const void* gamma,
                                   const float* mean, const float* rstd, const void* dres, int64_t lddres, void* dx,
                                   int64_t lddx, void* dgamma, void* dbeta, float* partial, int partial_blocks, int rows,
                                   int D, int rms, int grp, int grp_stride, int grp_off, void* stream) {
  if (!dy || !x || !gamma || !rstd || !dx) return unimp_set_error(UNIMP_ERR_ARG, "layernorm_bwd: null pointer");
  if (rows <= 0) return UNIMP_OK;
  if ((D & 7) || D > 4096 || (ldx & 7) || (lddy & 7) || (lddx & 7)) return unimp_set_error(UNIMP_ERR_SHAPE, "layernorm_bwd: need D%8==0, D<=4096, ld%8==0");
  hipStream_t s = (hipStream_t)stream;
  bool wg = dgamma != nullptr;
  int nb = ln_grid(rows);
  if (wg) {
    if (!partial || partial_blocks < 1) return unimp_set_error(UNIMP_ERR_ARG, "layernorm_bwd: dgamma needs a partial workspace");
    if (nb > partial_blocks) nb = partial_blocks;
  }
  dim3 g(nb), b(256);
  size_t lds = wg ? (size_t)4 * D * sizeof(float) : 0;
#define LN_BWD(MC, WG) hipLaunchKernelGGL((ln_bwd_kernel<MC, WG>), g, b, lds, s, (const bf16*)dy, (long)lddy, (const bf16*)dy2, (long)lddy2, (const bf16*)x, (long)ldx, \
      (const bf16*)gamma, mean, rstd, (const bf16*)dres, (long)lddres, (bf16*)dx, (long)lddx, partial, rows, D, rms, grp, grp_stride, grp_off)
  if (wg) { if (D <= 1024) LN_BWD(2, true); else if (D <= 2560) LN_BWD(5, true); else LN_BWD(8, true); }
  else    { if (D <= 1024) LN_BWD(2, false); else if (D <= 2560) LN_BWD(5, false); else LN_BWD(8, false); }
#undef LN_BWD
  int e = unimp_check_launch("layernorm_bwd");
  if (e) return e;
  if (wg) {
    hipLaunchKernelGGL(ln_wgrad_reduce_kernel, dim3((D + 31) / 32, 2), dim3(256), 0, s, partial, nb, D, (bf16*)dgamma, (bf16*)dbeta);
    return unimp_check_launch("layernorm_wgrad_reduce");
  }
  return UNIMP_OK;
}
