// LayerNorm / RMSNorm forward + backward for gfx950.  HBM-bound: one wave64 per row, the row lives in
// registers (16-byte bf16x8 loads, D/8 chunks striped over the 64 lanes, all of a row's loads issued before the first
// wait), statistics in fp32 through wave shuffles, no LDS in the forward.  Backward computes dx per row and keeps per-wave dgamma/dbeta
// partial sums in registers over a grid-stride loop, combines the block's 4 waves through LDS and writes
// one fp32 partial row per block; a second kernel column-sums the partials (deterministic, no atomics).
#include "common.h"
#include "unimp_hip.h"

__device__ __forceinline__ long map_row(int r, int grp, int grp_stride, int grp_off) {
  return grp ? (long)(r / grp) * grp_stride + (r % grp) + grp_off : (long)r;
}

// Chunk c of a lane is 8 consecutive elements at (c * 64 + lane) * 8.  Loads are branch-free: a lane whose chunk lies beyond
// the row reads chunk 0 instead (a valid address) and its values are masked afterwards, so the MAXC loads of a row - and the
// rows of every operand - go out back to back and are waited for once.  (With a branch per chunk the compiler put a
// vmcnt(0) behind every single load: five dependent HBM round trips per row, 4.0 TB/s at best.)  FULL = every lane owns
// all MAXC chunks (D = MAXC * 512: 1024, 2560, 4096): the masks vanish at compile time.
template <int MAXC, bool FULL>
struct RowMap {
  int off[MAXC];
  bool ok[MAXC];
  __device__ __forceinline__ RowMap(int nch) {
    int lane = lane_id();
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      int ch = c * 64 + lane;
      ok[c] = FULL || ch < nch;
      off[c] = (ok[c] ? ch : 0) * 8;
    }
  }
};

#define LN_KEEP_RAW(R_) asm volatile("" : "+v"(R_))

template <int MAXC>
__device__ __forceinline__ void load_raw(const bf16* __restrict__ p, const int (&off)[MAXC], bf16x8 (&t)[MAXC]) {
#pragma unroll
  for (int c = 0; c < MAXC; ++c) t[c] = *(const bf16x8*)(p + off[c]);
}

// MXOUT: the normalised row leaves as an MX-fp8 operand (e4m3 bytes at y, ldy in bytes; one E8M0 scale per 32 consecutive elements at
// `sc`) -- exactly the bytes mx_quantize_kernel (mx.hip) makes of the bf16 row this replaces: the 32-element block is the 8-element
// chunks of 4 ADJACENT lanes, its amax one quad exchange.  D % 32 == 0.
template <int CTRL> __device__ __forceinline__ float ln_quad(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
template <int MAXC, bool FULL, bool MXOUT = false>
__device__ __forceinline__ void ln_fwd_rows(const bf16* __restrict__ x, long ldx, const bf16* __restrict__ gamma,
                                            const bf16* __restrict__ beta, bf16* __restrict__ y, long ldy,
                                            float* __restrict__ mean, float* __restrict__ rstd, int rows, int D,
                                            float eps, int rms, int grp, int grp_stride, int grp_off,
                                            uint8_t* __restrict__ sc = nullptr, long ldsc = 0) {
  int lane = lane_id();
  int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int nwaves = (gridDim.x * blockDim.x) >> 6;
  if (wave >= rows) return;
  RowMap<MAXC, FULL> m(D >> 3);
  float inv_d = 1.f / (float)D;
  bf16x8 raw[MAXC];
  load_raw<MAXC>(x + (long)wave * ldx, m.off, raw);
  for (int r = wave; r < rows; r += nwaves) {
    // gamma / beta per row (L2 hits, in flight with the row): held across rows the compiler keeps them widened to fp32,
    // 80 registers that halve the occupancy
    bf16x8 g[MAXC], b[MAXC];
    load_raw<MAXC>(gamma, m.off, g);
    if (beta) load_raw<MAXC>(beta, m.off, b);
    else {
#pragma unroll
      for (int c = 0; c < MAXC; ++c) b[c] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
    float v[MAXC][8];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { v[c][j] = m.ok[c] ? bf2f(raw[c][j]) : 0.f; s += v[c][j]; }
    }
    int rn = r + nwaves;
    if (rn < rows) load_raw<MAXC>(x + (long)rn * ldx, m.off, raw);      // the next row flies under this row's arithmetic
    float mu = rms ? 0.f : wave_sum(s) * inv_d;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float d = m.ok[c] ? v[c][j] - mu : 0.f;
        // the FULL bf16 instantiations compile to packed d * d products followed by sequential adds (no fma); this one would contract to
        // scalar fmas and differ in the last bit of rstd (the ragged forms contract in both): spelled out so that both outputs of a shape
        // come from the same statistics (tests/test_kernels_gpu.py compares them bitwise for every (MAXC, FULL) pair)
        if (MXOUT && FULL) q = add_rn(q, mul_rn(d, d)); else q += d * d;
      }
    float var = wave_sum(q) * inv_d;
    float rs = rsqrtf(var + eps);
    if (lane == 0) { if (mean) mean[r] = mu; rstd[r] = rs; }
    bf16* yo = MXOUT ? (bf16*)((uint8_t*)y + (long)r * ldy) : y + map_row(r, grp, grp_stride, grp_off) * ldy;
    // gamma / beta stay in their bf16 load registers until the chunk that uses them (LN_KEEP_RAW: left to itself the
    // scheduler widens all of them to fp32 as soon as they land - 80 more live registers, half the waves per SIMD)
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      LN_KEEP_RAW(g[c]); LN_KEEP_RAW(b[c]);
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = f2bf((v[c][j] - mu) * rs * bf2f(g[c][j]) + bf2f(b[c][j]));
      if (MXOUT) {
        float w[8], amax = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { w[j] = bf2f(o[j]); amax = fmaxf(amax, fabsf(w[j])); }
        amax = fmaxf(amax, ln_quad<0xB1>(amax));                 // lane ^ 1
        amax = fmaxf(amax, ln_quad<0x4E>(amax));                 // lane ^ 2
        int eb = (int)((__float_as_uint(amax) >> 23) & 0xff);
        int sbyte = max(eb - 8, 0);
        if (!(amax == amax) || eb == 255) sbyte = 254;
        float inv = __uint_as_float((uint32_t)(254 - sbyte) << 23);
        if (sbyte == 254) inv = 1.17549435e-38f;
        uint32_t pk[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          float a0 = fminf(fmaxf(w[4 * h] * inv, -448.f), 448.f), a1 = fminf(fmaxf(w[4 * h + 1] * inv, -448.f), 448.f);
          float a2 = fminf(fmaxf(w[4 * h + 2] * inv, -448.f), 448.f), a3 = fminf(fmaxf(w[4 * h + 3] * inv, -448.f), 448.f);
          int t = __builtin_amdgcn_cvt_pk_fp8_f32(a0, a1, 0, false);
          t = __builtin_amdgcn_cvt_pk_fp8_f32(a2, a3, t, true);
          pk[h] = (uint32_t)t;
        }
        if (m.ok[c]) {
          *(uint2*)((uint8_t*)yo + m.off[c]) = uint2{pk[0], pk[1]};          // yo advanced in BYTES: ldy is the byte pitch (see the kernel)
          if (!(lane & 3)) sc[(long)r * ldsc + (m.off[c] >> 5)] = (uint8_t)sbyte;
        }
      } else if (m.ok[c]) *(bf16x8*)(yo + m.off[c]) = o;
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

template <int MAXC, bool FULL>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const bf16* __restrict__ x, long ldx, const bf16* __restrict__ gamma,
                                                     const bf16* __restrict__ beta, bf16* __restrict__ y, long ldy,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int rows, int D,
                                                     float eps, int rms, int grp, int grp_stride, int grp_off) {
  ln_fwd_rows<MAXC, FULL>(x, ldx, gamma, beta, y, ldy, mean, rstd, rows, D, eps, rms, grp, grp_stride, grp_off);
}

template <int MAXC, bool FULL>
__global__ __launch_bounds__(256) void ln_fwd_mx_kernel(const bf16* __restrict__ x, long ldx, const bf16* __restrict__ gamma,
                                                        const bf16* __restrict__ beta, uint8_t* __restrict__ yq, long ldyq,
                                                        uint8_t* __restrict__ sc, long ldsc, float* __restrict__ mean,
                                                        float* __restrict__ rstd, int rows, int D, float eps, int rms) {
  ln_fwd_rows<MAXC, FULL, true>(x, ldx, gamma, beta, (bf16*)yq, ldyq, mean, rstd, rows, D, eps, rms, 0, 0, 0, sc, ldsc);
}

// dx = rstd * (g*dy - mean(g*dy) - xhat * mean(g*dy*xhat))      [LayerNorm]
// dx = rstd * (g*dy - xhat * mean(g*dy*xhat))                   [RMSNorm, xhat = x*rstd]
// The row stays in its bf16 load registers; xhat and g*dy are recomputed in the output pass (three operations per element,
// the arithmetic units are a quarter busy) instead of being held as two fp32 copies: 80 registers fewer, so the
// weight-gradient form - 80 more for its dgamma / dbeta partial sums - keeps more than two waves per SIMD.
template <int MAXC, bool WGRAD, bool FULL>
__device__ __forceinline__ void ln_bwd_rows(const bf16* __restrict__ dy, long lddy, const bf16* __restrict__ dy2, long lddy2,
                                            const bf16* __restrict__ x, long ldx,
                                            const bf16* __restrict__ gamma, const float* __restrict__ mean,
                                            const float* __restrict__ rstd, const bf16* __restrict__ dres, long lddres,
                                            bf16* __restrict__ dx, long lddx, float* __restrict__ partial,
                                            int rows, int D, int rms, int grp, int grp_stride, int grp_off, char* smem) {
  int wib = threadIdx.x >> 6;
  int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int nwaves = (gridDim.x * blockDim.x) >> 6;
  RowMap<MAXC, FULL> m(D >> 3);
  float inv_d = 1.f / (float)D;
  // weight gradient: the wave's dgamma / dbeta partial sums live in its own LDS region [2][MAXC][2][64 lanes][4] fp32
  // (16 bytes per lane, lane-contiguous: conflict-free b128), read-add-written once per row and chunk - 80 KiB per block at
  // D = 2560, two blocks per CU - instead of 80 more registers, which left one wave per SIMD.  (ds_add_f32 is no
  // alternative: 160 LDS atomics per row ran 5x slower than the whole kernel.)  A wave's LDS operations execute in order and
  // nobody else touches its region, so the sums are reproducible.
  constexpr int ACC = MAXC * 512;                       // floats per sum per wave
  float* acc = (float*)smem + wib * (2 * ACC) + lane_id() * 4;
  if (WGRAD) {
#pragma unroll
    for (int k = 0; k < 2 * MAXC * 2; ++k) *(f32x4*)(acc + k * 256) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (int r = wave; r < rows; r += nwaves) {
    bf16x8 xr[MAXC], dr[MAXC], d2[MAXC], rr[MAXC], g[MAXC];
    load_raw<MAXC>(x + (long)r * ldx, m.off, xr);
    load_raw<MAXC>(gamma, m.off, g);
    load_raw<MAXC>(dy + map_row(r, grp, grp_stride, grp_off) * lddy, m.off, dr);
    if (dy2) load_raw<MAXC>(dy2 + (long)r * lddy2, m.off, d2);
    if (dres) load_raw<MAXC>(dres + (long)r * lddres, m.off, rr);
    float mu = rms ? 0.f : mean[r];
    float rs = rstd[r];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      f32x4 wg[2][2];                           // [dgamma, dbeta][half of the chunk]
      if (WGRAD) {
#pragma unroll
        for (int h = 0; h < 2; ++h) { wg[0][h] = *(const f32x4*)(acc + (c * 2 + h) * 256); wg[1][h] = *(const f32x4*)(acc + ACC + (c * 2 + h) * 256); }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float xh = (bf2f(xr[c][j]) - mu) * rs;
        float d = bf2f(dr[c][j]);
        if (dy2) d += bf2f(d2[c][j]);
        if (!m.ok[c]) { xh = 0.f; d = 0.f; }
        if (WGRAD) { wg[0][j >> 2][j & 3] += d * xh; wg[1][j >> 2][j & 3] += d; }
        float gd = d * bf2f(g[c][j]);
        s1 += gd; s2 += gd * xh;
      }
      if (WGRAD) {
#pragma unroll
        for (int h = 0; h < 2; ++h) { *(f32x4*)(acc + (c * 2 + h) * 256) = wg[0][h]; *(f32x4*)(acc + ACC + (c * 2 + h) * 256) = wg[1][h]; }
      }
      __builtin_amdgcn_sched_barrier(0);       // one chunk's fp32 temporaries at a time
    }
    s1 = rms ? 0.f : wave_sum(s1) * inv_d;
    s2 = wave_sum(s2) * inv_d;
    // make the row opaque so that the output pass really recomputes from the bf16 registers (otherwise the common
    // subexpressions of the two passes are kept alive as fp32 and the registers are back)
    bf16* o = dx + (long)r * lddx;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      LN_KEEP_RAW(xr[c]); LN_KEEP_RAW(dr[c]); LN_KEEP_RAW(g[c]);
      if (dy2) LN_KEEP_RAW(d2[c]);
      if (dres) LN_KEEP_RAW(rr[c]);
      bf16x8 ov;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float xh = (bf2f(xr[c][j]) - mu) * rs;
        float d = bf2f(dr[c][j]);
        if (dy2) d += bf2f(d2[c][j]);
        float gd = d * bf2f(g[c][j]);
        float t = rs * (gd - s1 - xh * s2);
        ov[j] = f2bf(dres ? t + bf2f(rr[c][j]) : t);
      }
      if (m.ok[c]) *(bf16x8*)(o + m.off[c]) = ov;
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (WGRAD) {
    // combine the block's 4 wave regions; LDS index k = ((c * 2 + h) * 64 + lane) * 4 + jj  <->  column (c * 64 + lane) * 8 + h * 4 + jj
    __syncthreads();
    const float* sh = (const float*)smem;
    for (int i = threadIdx.x; i < 2 * ACC; i += blockDim.x) {
      int which = i / ACC, k = i - which * ACC;
      int jj = k & 3, ln = (k >> 2) & 63, ch = k >> 8;
      int col = ((ch >> 1) * 64 + ln) * 8 + (ch & 1) * 4 + jj;
      if (col < D) partial[((long)blockIdx.x * 2 + which) * D + col] = sh[i] + sh[2 * ACC + i] + sh[4 * ACC + i] + sh[6 * ACC + i];
    }
  }
}

template <int MAXC, bool WGRAD, bool FULL>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16* __restrict__ dy, long lddy, const bf16* __restrict__ dy2, long lddy2,
                                                     const bf16* __restrict__ x, long ldx,
                                                     const bf16* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const bf16* __restrict__ dres, long lddres,
                                                     bf16* __restrict__ dx, long lddx, float* __restrict__ partial,
                                                     int rows, int D, int rms, int grp, int grp_stride, int grp_off) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  ln_bwd_rows<MAXC, WGRAD, FULL>(dy, lddy, dy2, lddy2, x, ldx, gamma, mean, rstd, dres, lddres, dx, lddx, partial, rows, D, rms, grp, grp_stride, grp_off, smem);
}

// column sums of the per-block partials [nblk][2][D]: a block owns 64 columns of dgamma or dbeta (a wave reads 256 contiguous
// bytes of a partial row), its 8 waves take every 8th row with 8 independent loads in flight, LDS combines the 8 waves in a
// fixed order.
__global__ __launch_bounds__(512) void ln_wgrad_reduce_kernel(const float* __restrict__ partial, int nblk, int D, bf16* __restrict__ dgamma,
                                                              bf16* __restrict__ dbeta, int accumulate) {
  __shared__ float sh[8][64];
  int which = blockIdx.y;
  int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  int col = blockIdx.x * 64 + cl;
  float t = 0.f;
  if (col < D) {
    const float* p = partial + (long)which * D + col;
    int b = rg;
    for (; b + 56 < nblk; b += 64) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[(long)(b + 8 * u) * 2 * D];
#pragma unroll
      for (int u = 0; u < 8; ++u) t += v[u];
    }
    for (; b < nblk; b += 8) t += p[(long)b * 2 * D];
  }
  sh[rg][cl] = t;
  __syncthreads();
  if (rg == 0 && col < D) {
    float a = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) a += sh[g][cl];
    bf16* d = which == 0 ? dgamma : dbeta;           // accumulate: the destination is the parameter's slot of the gradient buffer
    if (d) d[col] = f2bf(accumulate ? a + bf2f(d[col]) : a);
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
#define LN_FWD_(MC, FU) hipLaunchKernelGGL((ln_fwd_kernel<MC, FU>), g, b, 0, s, (const bf16*)x, (long)ldx, (const bf16*)gamma, (const bf16*)beta, \
                                      (bf16*)y, (long)ldy, mean, rstd, rows, D, eps, rms, grp, grp_stride, grp_off)
#define LN_FWD(MC) do { if (D == MC * 512) LN_FWD_(MC, true); else LN_FWD_(MC, false); } while (0)
  if (D <= 1024) LN_FWD(2); else if (D <= 2560) LN_FWD(5); else LN_FWD(8);
#undef LN_FWD_
#undef LN_FWD
  return unimp_check_launch("layernorm_fwd");
}

extern "C" int unimp_layernorm_fwd_mx(const void* x, int64_t ldx, const void* gamma, const void* beta, void* yq, int64_t ldyq, void* scales,
                                      int64_t ldsc, float* mean, float* rstd, int rows, int D, float eps, int rms, void* stream) {
  if (!x || !gamma || !yq || !scales || !rstd) return unimp_set_error(UNIMP_ERR_ARG, "layernorm_fwd_mx: null pointer");
  if (rows <= 0) return UNIMP_OK;
  if ((D & 31) || D > 4096 || (ldx & 7) || (ldyq & 7)) return unimp_set_error(UNIMP_ERR_SHAPE, "layernorm_fwd_mx: need D%32==0, D<=4096, ldx%8==0, ldyq%8==0");
  hipStream_t s = (hipStream_t)stream;
  dim3 g(ln_grid(rows)), b(256);
#define LN_MX_(MC, FU) hipLaunchKernelGGL((ln_fwd_mx_kernel<MC, FU>), g, b, 0, s, (const bf16*)x, (long)ldx, (const bf16*)gamma, (const bf16*)beta, \
                                      (uint8_t*)yq, (long)ldyq, (uint8_t*)scales, (long)ldsc, mean, rstd, rows, D, eps, rms)
#define LN_MX(MC) do { if (D == MC * 512) LN_MX_(MC, true); else LN_MX_(MC, false); } while (0)
  if (D <= 1024) LN_MX(2); else if (D <= 2560) LN_MX(5); else LN_MX(8);
#undef LN_MX_
#undef LN_MX
  return unimp_check_launch("layernorm_fwd_mx");
}

extern "C" int unimp_layernorm_bwd(const void* dy, int64_t lddy, const void* dy2, int64_t lddy2, const void* x, int64_t ldx, const void* gamma,
                                   const float* mean, const float* rstd, const void* dres, int64_t lddres, void* dx,
                                   int64_t lddx, void* dgamma, void* dbeta, float* partial, int partial_blocks, int rows,
                                   int D, int rms, int grp, int grp_stride, int grp_off, int wgrad_accumulate, void* stream) {
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
  const int maxc = D <= 1024 ? 2 : D <= 2560 ? 5 : 8;
  size_t lds = wg ? (size_t)4 * 2 * maxc * 512 * sizeof(float) : 0;          // 4 waves x [2][MAXC * 512] partial sums
#define LN_BWD(MC, WG) do { if (D == MC * 512) LN_BWD_(MC, WG, true); else LN_BWD_(MC, WG, false); } while (0)
#define LN_BWD_(MC, WG, FU) do { if (WG) { static bool once = false; if (!once) { once = true;                                            \
      (void)hipFuncSetAttribute((const void*)ln_bwd_kernel<MC, WG, FU>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 2 * MC * 512 * 4); } } \
    hipLaunchKernelGGL((ln_bwd_kernel<MC, WG, FU>), g, b, lds, s, (const bf16*)dy, (long)lddy, (const bf16*)dy2, (long)lddy2, (const bf16*)x, (long)ldx, \
      (const bf16*)gamma, mean, rstd, (const bf16*)dres, (long)lddres, (bf16*)dx, (long)lddx, partial, rows, D, rms, grp, grp_stride, grp_off); } while (0)
  if (wg) { if (D <= 1024) LN_BWD(2, true); else if (D <= 2560) LN_BWD(5, true); else LN_BWD(8, true); }
  else    { if (D <= 1024) LN_BWD(2, false); else if (D <= 2560) LN_BWD(5, false); else LN_BWD(8, false); }
#undef LN_BWD
#undef LN_BWD_
  int e = unimp_check_launch("layernorm_bwd");
  if (e) return e;
  if (wg) {
    hipLaunchKernelGGL(ln_wgrad_reduce_kernel, dim3((D + 63) / 64, 2), dim3(512), 0, s, partial, nb, D, (bf16*)dgamma, (bf16*)dbeta, wgrad_accumulate);
    return unimp_check_launch("layernorm_wgrad_reduce");
  }
  return UNIMP_OK;
}
