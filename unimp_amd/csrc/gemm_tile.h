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

// 8 consecutive columns of one output row.  PRE: the caller already fetched this row group's aux / res chunks (FAST only).
template <bool FAST, bool PRE = false>
__device__ __forceinline__ void epi8(const Gemm2Params& p, float (&v)[8], int m, int n, float gate,
                                     bf16x8 auxv = bf16x8{}, bf16x8 resv = bf16x8{}) {
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
    act_fwd_deriv_n<8>(p.act, v, dv);
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
    if (p.act) act_fwd_n<8>(p.act, v);
  }
  if (p.aux) {
    const bf16* s = p.aux + (long)m * p.ldaux + n;
    if (FAST) { bf16x8 x = PRE ? auxv : *(const bf16x8*)s;
      float xf[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) xf[r] = bf2f(x[r]);
      act_bwd_mul_n<8>(p.dact, v, xf); }
    else { for (int r = 0; r < nv; ++r) v[r] *= act_bwd(p.dact, bf2f(s[r])); }
  }
#pragma unroll
  for (int r = 0; r < 8; ++r) v[r] *= gate;
  if (p.res) {
    const bf16* s = p.res + (long)m * p.ldres + n;
    if (FAST) { bf16x8 x = PRE ? resv : *(const bf16x8*)s;
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


// One pass of the LDS-staged epilogue: the wave's [64][WN] f32 region (16-B units XOR-swizzled by row) -> global memory.
// Row groups are processed U at a time with their aux / res chunks fetched up front: one global-load latency per U groups
// instead of one per group (the dX GEMM that multiplies by the stored act'(z), and every GEMM with a residual, spent
// 10-25 % of their time there).  U = 8 (a whole pass of a 64-column wave tile in one round trip; the fragment registers
// are dead by now).
template <int WN, int U = (64 / (64 / (WN / 8)) < 8 ? 64 / (64 / (WN / 8)) : 8)>
__device__ __forceinline__ void epi_pass(const Gemm2Params& p, const char* er, int lane, int mbase, int nbase, float gate, bool fast) {
  constexpr int ESTR = WN * 4, UNITS = WN / 4, LPR = WN / 8, RPI = 64 / LPR, NIT = 64 / RPI;
  static_assert(NIT % U == 0, "row groups per pass must be a multiple of U");
  const int cg = lane % LPR, n = nbase + cg * 8;
  if (!fast || n >= p.N) {                       // ragged N / odd leading dimensions: element-wise path, one group at a time
    for (int it = 0; it < NIT; ++it) {
      int row = it * RPI + lane / LPR, m = mbase + row, sw = row & (UNITS - 1);
      f32x4 x0 = *(const f32x4*)(er + row * ESTR + (((2 * cg) ^ sw) << 4));
      f32x4 x1 = *(const f32x4*)(er + row * ESTR + (((2 * cg + 1) ^ sw) << 4));
      if (m < p.M && n < p.N) {
        float v[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
        if (fast) epi8<true>(p, v, m, n, gate); else epi8<false>(p, v, m, n, gate);
      }
    }
    return;
  }
  for (int it0 = 0; it0 < NIT; it0 += U) {
    bf16x8 av[U], rv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int m = min(mbase + (it0 + u) * RPI + lane / LPR, p.M - 1);          // clamped: rows beyond M are loaded, never stored
      av[u] = bf16x8{}; rv[u] = bf16x8{};
      if (p.aux) av[u] = *(const bf16x8*)(p.aux + (long)m * p.ldaux + n);
      if (p.res) rv[u] = *(const bf16x8*)(p.res + (long)m * p.ldres + n);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int row = (it0 + u) * RPI + lane / LPR, m = mbase + row, sw = row & (UNITS - 1);
      f32x4 x0 = *(const f32x4*)(er + row * ESTR + (((2 * cg) ^ sw) << 4));
      f32x4 x1 = *(const f32x4*)(er + row * ESTR + (((2 * cg + 1) ^ sw) << 4));
      if (m < p.M) {
        float v[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
        epi8<true, true>(p, v, m, n, gate, av[u], rv[u]);
      }
    }
  }
}


// Whole-pass variant for wave tiles of <= 64 columns (<= 8 row groups per pass): the aux / residual chunks of a pass are
// fetched by epi_fetch() BEFORE the accumulators of that pass are staged through LDS -- and, for the second pass, before the
// first pass is computed and stored -- so their global-load latency is not on the epilogue's critical path.
template <int WN>
struct EpiPre {
  static constexpr int LPR = WN / 8, RPI = 64 / LPR, NIT = 64 / RPI;
  bf16x8 av[NIT], rv[NIT];
};
template <int WN>
__device__ __forceinline__ void epi_fetch(const Gemm2Params& p, int lane, int mbase, int nbase, bool fast, EpiPre<WN>& e) {
  constexpr int LPR = EpiPre<WN>::LPR, RPI = EpiPre<WN>::RPI, NIT = EpiPre<WN>::NIT;
  const int n = nbase + (lane % LPR) * 8;
  if (!fast || n >= p.N) return;
#pragma unroll
  for (int u = 0; u < NIT; ++u) {
    int m = min(mbase + u * RPI + lane / LPR, p.M - 1);                    // clamped: rows beyond M are loaded, never stored
    e.av[u] = bf16x8{}; e.rv[u] = bf16x8{};
    if (p.aux) e.av[u] = *(const bf16x8*)(p.aux + (long)m * p.ldaux + n);
    if (p.res) e.rv[u] = *(const bf16x8*)(p.res + (long)m * p.ldres + n);
  }
}
template <int WN>
__device__ __forceinline__ void epi_pass_pre(const Gemm2Params& p, const char* er, int lane, int mbase, int nbase, float gate, bool fast,
                                             const EpiPre<WN>& e) {
  constexpr int ESTR = WN * 4, UNITS = WN / 4, LPR = EpiPre<WN>::LPR, RPI = EpiPre<WN>::RPI, NIT = EpiPre<WN>::NIT;
  const int cg = lane % LPR, n = nbase + cg * 8;
  if (!fast || n >= p.N) { epi_pass<WN>(p, er, lane, mbase, nbase, gate, fast); return; }
#pragma unroll
  for (int u = 0; u < NIT; ++u) {
    int row = u * RPI + lane / LPR, m = mbase + row, sw = row & (UNITS - 1);
    f32x4 x0 = *(const f32x4*)(er + row * ESTR + (((2 * cg) ^ sw) << 4));
    f32x4 x1 = *(const f32x4*)(er + row * ESTR + (((2 * cg + 1) ^ sw) << 4));
    if (m < p.M) {
      float v[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
      epi8<true, true>(p, v, m, n, gate, e.av[u], e.rv[u]);
    }
  }
}
