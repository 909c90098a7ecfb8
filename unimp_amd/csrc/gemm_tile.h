// Shared by the large-tile GEMM kernels (gemm2.hip, gemm3.hip): parameters, LDS-DMA helper, row-major epilogue.
#pragma once
#include "common.h"
#include "unimp_hip.h"

struct Gemm2Params {
  const bf16* A; const bf16* B; void* C;
  int M, N, K;
  long lda, ldb, ldc;
  const bf16* bias;
  const bf16* res;  long ldres;
  const bf16* aux;  long ldaux;
  bf16* pre;        long ldpre;
  const bf16* gate;
  float alpha;
  int act, dact, out_f32, accumulate, pre_deriv;
  int nbm, nbn;
};

static __device__ uint4 g_zero16[4];      // zero-initialised: source of every out-of-range LDS-DMA chunk

__device__ __forceinline__ void glds16(const void* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// 8 consecutive columns of one output row
template <bool FAST>
__device__ __forceinline__ void epi8(const Gemm2Params& p, float (&v)[8], int m, int n, float gate) {
  int nv = FAST ? 8 : min(8, p.N - n);
#pragma unroll
  for (int r = 0; r < 8; ++r) v[r] *= p.alpha;
  if (p.bias) {
    if (FAST) { bf16x8 b = *(const bf16x8*)(p.bias + n);
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] += bf2f(b[r]); }
    else { for (int r = 0; r < nv; ++r) v[r] += bf2f(p.bias[n + r]); }
  }
  if (p.pre && p.pre_deriv) {                 // y = act(v) and act'(v) from one exponential; store the derivative
    float dv[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) act_fwd_deriv(p.act, v[r], v[r], dv[r]);
    bf16* d = p.pre + (long)m * p.ldpre + n;
    if (FAST) { bf16x8 o;
#pragma unroll
      for (int r = 0; r < 8; ++r) o[r] = f2bf(dv[r]);
      *(bf16x8*)d = o; }
    else { for (int r = 0; r < nv; ++r) d[r] = f2bf(dv[r]); }
  } else {
    if (p.pre) {
      bf16* d = p.pre + (long)m * p.ldpre + n;
      if (FAST) { bf16x8 o;
#pragma unroll
        for (int r = 0; r < 8; ++r) o[r] = f2bf(v[r]);
        *(bf16x8*)d = o; }
      else { for (int r = 0; r < nv; ++r) d[r] = f2bf(v[r]); }
    }
    if (p.act) {
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] = act_fwd(p.act, v[r]);
    }
  }
  if (p.aux) {
    const bf16* s = p.aux + (long)m * p.ldaux + n;
    if (FAST) { bf16x8 x = *(const bf16x8*)s;
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] *= act_bwd(p.dact, bf2f(x[r])); }
    else { for (int r = 0; r < nv; ++r) v[r] *= act_bwd(p.dact, bf2f(s[r])); }
  }
#pragma unroll
  for (int r = 0; r < 8; ++r) v[r] *= gate;
  if (p.res) {
    const bf16* s = p.res + (long)m * p.ldres + n;
    if (FAST) { bf16x8 x = *(const bf16x8*)s;
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] += bf2f(x[r]); }
    else { for (int r = 0; r < nv; ++r) v[r] += bf2f(s[r]); }
  }
  if (p.out_f32) {
    float* d = (float*)p.C + (long)m * p.ldc + n;
    if (FAST) {
      f32x4 o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
      if (p.accumulate) { o0 += *(const f32x4*)d; o1 += *(const f32x4*)(d + 4); }
      *(f32x4*)d = o0; *(f32x4*)(d + 4) = o1;
    } else { for (int r = 0; r < nv; ++r) d[r] = p.accumulate ? d[r] + v[r] : v[r]; }
  } else {
    bf16* d = (bf16*)p.C + (long)m * p.ldc + n;
    if (FAST) {
      if (p.accumulate) { bf16x8 c = *(const bf16x8*)d;
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] += bf2f(c[r]); }
      bf16x8 o;
#pragma unroll
      for (int r = 0; r < 8; ++r) o[r] = f2bf(v[r]);
      *(bf16x8*)d = o;
    } else { for (int r = 0; r < nv; ++r) d[r] = f2bf(p.accumulate ? bf2f(d[r]) + v[r] : v[r]); }
  }
}

