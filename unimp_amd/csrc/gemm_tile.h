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
  int ksplit;                       // > 0 (gemm3 only): blockIdx.y reduces k in [y*ksplit, (y+1)*ksplit) into f32 slab y of C
  int gm;                           // gemm3 only: tile rows per raster group (0 = the default 4); UNIMP_GEMM_GM, measurement knob
  int rope_rot, rope_hd, rope_period, rope_span, rope_L;      // rotary epilogue (EK_PLAIN only; include/unimp_hip.h), rope_rot == 0: none
  float rope_step, rope_invL;       // 2 log2(base) / rope_rot;  1 / rope_L
  const int* rope_tab;              // null: position = m % rope_L; else position = rope_tab[m] (packed rows)
};

#define GEMM2_FILL_ROPE(P_, D_) do { (P_).rope_rot = (D_)->rope_rot; (P_).rope_hd = (D_)->rope_hd; (P_).rope_period = (D_)->rope_period;   \
    (P_).rope_span = (D_)->rope_span; (P_).rope_L = (D_)->rope_L;                                                                      \
    (P_).rope_step = (D_)->rope_rot > 0 ? 2.f * (D_)->rope_log2_base / (float)(D_)->rope_rot : 0.f;                                     \
    (P_).rope_invL = (D_)->rope_L > 0 ? 1.f / (float)(D_)->rope_L : 0.f; (P_).rope_tab = (D_)->rope_pos; } while (0)

// cos / sin of position * base^(-2 i / rot) for the 4 frequencies i0 .. i0 + 3, from the fractional number of turns
__device__ __forceinline__ void rope_cs4(float pos, int i0, float step, float (&co)[4], float (&si)[4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float turns = pos * (__builtin_amdgcn_exp2f(-(float)(i0 + j) * step) * 0.15915494309189535f);
    turns = __builtin_amdgcn_fractf(turns);
    co[j] = __builtin_amdgcn_cosf(turns); si[j] = __builtin_amdgcn_sinf(turns);
  }
}
// m % L for 0 <= m < 2^24 without an integer division
__device__ __forceinline__ int rope_pos(int m, int L, float invL) {
  int q = (int)((float)m * invL);
  int r = m - q * L;
  r += r < 0 ? L : 0;
  r -= r >= L ? L : 0;
  return r;
}

static __device__ uint4 g_zero16[4];      // zero-initialised: source of every out-of-range LDS-DMA chunk

__device__ __forceinline__ void glds16(const void* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// 8 consecutive columns of one output row.  PRE: the caller already fetched this row group's aux / res chunks (FAST only).
// alpha / bias / gate / residual use individually rounded multiplies and adds (common.h mul_rn / add_rn: no fma contraction): every kernel variant and
// every epilogue form must produce the same bits, or the result of a sample would depend on which variant its batch size tuned to.
template <bool FAST, bool PRE = false>
__device__ __forceinline__ void epi8(const Gemm2Params& p, float (&v)[8], int m, int n, float gate,
                                     bf16x8 auxv = bf16x8{}, bf16x8 resv = bf16x8{}, bf16x8 biasv = bf16x8{}) {
  int nv = FAST ? 8 : min(8, p.N - n);
#pragma unroll
  for (int r = 0; r < 8; ++r) v[r] = mul_rn(v[r], p.alpha);
  if (p.bias) {
    if (FAST) { bf16x8 b = PRE ? biasv : *(const bf16x8*)(p.bias + n);
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] = add_rn(v[r], bf2f(b[r])); }
    else { for (int r = 0; r < nv; ++r) v[r] = add_rn(v[r], bf2f(p.bias[n + r])); }
  }
  if (p.pre && p.pre_deriv) {                 // y = act(v) and act'(v) from one exponential; store the derivative
    float dv[8];
    act_fwd_deriv_n<8>(p.act, v, dv);
    bf16* d = p.pre + (long)m * p.ldpre + n;    // (the uint8 derivative never reaches this form: gemm.hip validate(), epi_kind)
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
  for (int r = 0; r < 8; ++r) v[r] = mul_rn(v[r], gate);
  if (p.res) {
    const bf16* s = p.res + (long)m * p.ldres + n;
    if (FAST) { bf16x8 x = PRE ? resv : *(const bf16x8*)s;
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] = add_rn(v[r], bf2f(x[r])); }
    else { for (int r = 0; r < nv; ++r) v[r] = add_rn(v[r], bf2f(s[r])); }
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
// GENERIC form: every fused option behind run-time branches, one row group at a time in a ROLLED loop.  It is the fallback
// (accumulate into C, aux AND residual, ragged N / odd leading dimensions); the hot combinations use epi_groups<KIND> below.
// Code size matters here: with the row-group loop unrolled around the fully general epi8 the epilogue was 185 KiB of a
// 191 KiB kernel -- three times the instruction cache two CUs share -- and a 256 x 256 tile spent 9.5 us (idle chip) to
// 15 us (loaded) in it against 18-26 us for its whole K = 1024 main loop (s_memrealtime stamps, -DG3_STAMP).
template <int WN, int ROWS = 64>
__device__ __forceinline__ void epi_pass(const Gemm2Params& p, const char* er, int lane, int mbase, int nbase, float gate, bool fast) {
  constexpr int ESTR = WN * 4, UNITS = WN / 4, LPR = WN / 8, RPI = 64 / LPR, NIT = ROWS / RPI;
  const int cg = lane % LPR, n = nbase + cg * 8;
#pragma unroll 1
  for (int it = 0; it < NIT; ++it) {
    int row = it * RPI + lane / LPR, m = mbase + row, sw = row & (UNITS - 1);
    f32x4 x0 = *(const f32x4*)(er + row * ESTR + (((2 * cg) ^ sw) << 4));
    f32x4 x1 = *(const f32x4*)(er + row * ESTR + (((2 * cg + 1) ^ sw) << 4));
    if (m < p.M && n < p.N) {
      float v[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
      if (fast && n + 8 <= p.N) epi8<true>(p, v, m, n, gate); else epi8<false>(p, v, m, n, gate);
    }
  }
}

// ---- specialised epilogues ---------------------------------------------------------------------------------------------
// The combinations the step actually uses, chosen once per tile (epi_kind) and compiled without inner option branches:
//   EK_PLAIN  alpha (+bias)                                   -> C                 (qkv, dX of plain linears, weight gradients)
//   EK_ACT    alpha (+bias), act, optional second output      -> C, pre            (MLP up-projection: act(z) and act'(z) as uint8)
//   EK_AUX    alpha (+bias), x stored act'(z) (uint8)         -> C                 (dX through the activation)
//   EK_RES    alpha (+bias), x tanh(gate), + residual         -> C (, raw pre)     (attention-out / MLP down-projection, gated xattn)
// Every global load (bias, aux / residual chunks) is issued by epi_fetch() BEFORE the accumulators are staged and waited for
// once (epi_inputs_ready) before the first store: a load consumed inside the store loop makes hipcc emit `s_waitcnt vmcnt(0)`
// there (it cannot count in-flight stores across branches), which also waits for the previous row group's store to be
// acknowledged.  EK_PLAIN / EK_ACT have no per-row-group input and run as a rolled loop; EK_AUX / EK_RES unroll their <= 8
// row groups around the prefetched registers.
enum { EK_PLAIN = 0, EK_ACT = 1, EK_AUX = 2, EK_RES = 3, EK_GENERIC = 4, EK_ROPE = 5 /* EK_PLAIN + rotary pairs (QKV projection) */,
       EK_GELU2 = 6 /* EK_ACT with act == GELU and the uint8 derivative as second output, both fixed at compile time (the LM's up-projection) */ };
__device__ __forceinline__ int epi_kind(const Gemm2Params& p, bool fast) {
  if (!fast || p.accumulate || (p.aux && p.res)) return EK_GENERIC;
  // the specialised kinds serve the stored derivative in its 8-bit form only (what the MLP blocks use); a bf16 derivative or a raw
  // pre-activation output takes the generic form -- one copy of each in the instruction cache instead of two
  if (p.aux) return (p.dact == ACT_DERIV_U8 && !p.act && !p.pre) ? EK_AUX : EK_GENERIC;
  if (p.res) return (!p.act && !(p.pre && p.pre_deriv)) ? EK_RES : EK_GENERIC;     // a raw (pre-gate) second output is part of EK_RES
  if (p.pre) return p.pre_deriv == 2 ? EK_ACT : EK_GENERIC;
  return p.act ? EK_ACT : EK_PLAIN;                                                   // every kind applies tanh(gate) (1 when absent)
}
template <int WN, int ROWS = 64>
struct EpiPre {
  static constexpr int LPR = WN / 8, RPI = 64 / LPR, NIT = ROWS / RPI;
  bf16x8 xv[NIT];                   // the aux OR the residual chunk of each row group
};
template <int WN>
__device__ __forceinline__ bf16x8 epi_bias(const Gemm2Params& p, int lane, int nbase, int kind) {
  const int n = nbase + (lane % (WN / 8)) * 8;
  bf16x8 b = bf16x8{};
  if (kind != EK_GENERIC && n + 8 <= p.N && p.bias) b = *(const bf16x8*)(p.bias + n);      // a partial last group goes through epi8<false>
  return b;
}
template <int WN, int ROWS = 64>
__device__ __forceinline__ void epi_fetch(const Gemm2Params& p, int lane, int mbase, int nbase, int kind, EpiPre<WN, ROWS>& e) {
  constexpr int LPR = EpiPre<WN, ROWS>::LPR, RPI = EpiPre<WN, ROWS>::RPI, NIT = EpiPre<WN, ROWS>::NIT;
  const int n = nbase + (lane % LPR) * 8;
  if ((kind != EK_AUX && kind != EK_RES) || n >= p.N) return;
  const bf16* src = kind == EK_AUX ? p.aux : p.res;
  const long ld = kind == EK_AUX ? p.ldaux : p.ldres;
  if (kind == EK_AUX) {                                                    // uint8 derivative: 8 bytes per chunk, kept in the low half of the register
#pragma unroll
    for (int u = 0; u < NIT; ++u) {
      int m = min(mbase + u * RPI + lane / LPR, p.M - 1);
      uint2 w = *(const uint2*)((const uint8_t*)src + (long)m * ld + n);
      union { uint2 q[2]; bf16x8 b; } cv; cv.q[0] = w; cv.q[1] = uint2{0, 0};
      e.xv[u] = cv.b;
    }
    return;
  }
#pragma unroll
  for (int u = 0; u < NIT; ++u) {
    int m = min(mbase + u * RPI + lane / LPR, p.M - 1);                    // clamped: rows beyond M are loaded, never stored
    e.xv[u] = *(const bf16x8*)(src + (long)m * ld + n);
  }
}
// all epilogue inputs have landed; from here on only stores are in flight (vmcnt(0); expcnt / lgkmcnt untouched)
__device__ __forceinline__ void epi_inputs_ready() { __builtin_amdgcn_s_waitcnt(0x0f70); }

// alpha and tanh(gate) are exactly 1.0 in most launches of the step (every frozen-tower GEMM): x * 1.0f is x bit for bit, so the two
// multiplies are skipped behind wave-uniform branches -- 8 of the ~16 packed instructions a PLAIN row group costs, 10 % of a GELU one
template <int KIND>
__device__ __forceinline__ void epi8k(const Gemm2Params& p, float (&v)[8], int m, int n, float gate, bf16x8 x, bf16x8 biasv) {
  if (p.alpha != 1.f) {
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = mul_rn(v[r], p.alpha);
  }
  if (p.bias) {
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = add_rn(v[r], bf2f(biasv[r]));
  }
  if (KIND == EK_ROPE) {                      // rotary epilogue: the 8 columns are 4 adjacent pairs (j, j + 4)
    int pp = n % p.rope_hd;
    if (n % p.rope_period < p.rope_span && pp < p.rope_rot) {
      float co[4], si[4];
      rope_cs4((float)(p.rope_tab ? p.rope_tab[m] : rope_pos(m, p.rope_L, p.rope_invL)), (pp >> 3) * 4, p.rope_step, co, si);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float x1 = v[j], x2 = v[j + 4];
        v[j] = add_rn(mul_rn(x1, co[j]), -mul_rn(x2, si[j]));
        v[j + 4] = add_rn(mul_rn(x2, co[j]), mul_rn(x1, si[j]));
      }
    }
  }
  if (KIND == EK_GELU2) {                       // one activation, no switch: a quarter of EK_ACT's code per row group
    float dv[8];
    act_fwd_deriv_n<8>(ACT_GELU, v, dv);
    *(uint2*)((uint8_t*)p.pre + (long)m * p.ldpre + n) = uint2{deriv_u8_pack4(dv[0], dv[1], dv[2], dv[3]), deriv_u8_pack4(dv[4], dv[5], dv[6], dv[7])};
  }
  if (KIND == EK_ACT) {
    if (p.pre) {                                // epi_kind: the second output of this kind is the uint8 derivative
      float dv[8];
      act_fwd_deriv_n<8>(p.act, v, dv);
      *(uint2*)((uint8_t*)p.pre + (long)m * p.ldpre + n) = uint2{deriv_u8_pack4(dv[0], dv[1], dv[2], dv[3]), deriv_u8_pack4(dv[4], dv[5], dv[6], dv[7])};
    } else act_fwd_n<8>(p.act, v);
  }
  if (KIND == EK_AUX) {                         // epi_kind: the uint8 derivative
    union { bf16x8 b; uint2 q[2]; } cv; cv.b = x;
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] *= deriv_u8_get(r < 4 ? cv.q[0].x : cv.q[0].y, r & 3);
  }
  if (KIND != EK_RES && gate != 1.f) {        // the gated cross-attention's dX / dW GEMMs: x tanh(gate), after act / aux like the general form
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = mul_rn(v[r], gate);
  }
  if (KIND == EK_RES) {
    if (p.pre) {                            // gated blocks keep the un-gated value for the gate's gradient
      bf16x8 o;
#pragma unroll
      for (int r = 0; r < 8; ++r) o[r] = f2bf(v[r]);
      *(bf16x8*)(p.pre + (long)m * p.ldpre + n) = o;
    }
    if (gate != 1.f) {
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] = mul_rn(v[r], gate);
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = add_rn(v[r], bf2f(x[r]));
  }
  if (p.out_f32) {
    float* d = (float*)p.C + (long)m * p.ldc + n;
    *(f32x4*)d = f32x4{v[0], v[1], v[2], v[3]}; *(f32x4*)(d + 4) = f32x4{v[4], v[5], v[6], v[7]};
  } else {
    bf16x8 o;
#pragma unroll
    for (int r = 0; r < 8; ++r) o[r] = f2bf(v[r]);
    *(bf16x8*)((bf16*)p.C + (long)m * p.ldc + n) = o;
  }
}
// FIXED: called from a fixed-kind kernel (room to unroll); false = the run-time-dispatch kernels, which hold every kind and must stay
// inside the instruction cache (tests/test_cabi_cpu.py): their activation kind runs as a rolled loop
// UNR > 0 (gemm7.hip: one wave per SIMD, nobody to overlap a wave's LDS / store latency but its own other row groups): row groups in flight
// for the kinds without a per-row-group input; 0 = the depths chosen for the 8-wave kernels
template <int WN, int KIND, int ROWS = 64, bool FIXED = false, int UNR = 0>
__device__ __forceinline__ void epi_groups(const Gemm2Params& p, const char* er, int lane, int mbase, int nbase, float gate,
                                           const EpiPre<WN, ROWS>& e, bf16x8 biasv) {
  constexpr int ESTR = WN * 4, UNITS = WN / 4, LPR = EpiPre<WN, ROWS>::LPR, RPI = EpiPre<WN, ROWS>::RPI, NIT = EpiPre<WN, ROWS>::NIT;
  const int cg = lane % LPR, n = nbase + cg * 8;
  if (n >= p.N) return;
  if (n + 8 > p.N) return;                // the partial last group of a ragged N is written by the generic path (epi_pass_kind)
#define EPI_GROUP(U_, X_) do {                                                                                     \
    int row = (U_) * RPI + lane / LPR, m = mbase + row, sw = row & (UNITS - 1);                                    \
    f32x4 x0 = *(const f32x4*)(er + row * ESTR + (((2 * cg) ^ sw) << 4));                                          \
    f32x4 x1 = *(const f32x4*)(er + row * ESTR + (((2 * cg + 1) ^ sw) << 4));                                      \
    if (m < p.M) {                                                                                                 \
      float v[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};                                       \
      epi8k<KIND>(p, v, m, n, gate, (X_), biasv);                                                                  \
    } } while (0)
  if (KIND == EK_AUX || KIND == EK_RES) {
#pragma unroll
    for (int u = 0; u < NIT; ++u) EPI_GROUP(u, e.xv[u]);
  } else if (UNR > 0) {
#pragma unroll (UNR > 0 ? UNR : 1)
    for (int u = 0; u < NIT; ++u) EPI_GROUP(u, bf16x8{});
  } else if (KIND == EK_PLAIN) {          // small body: 4 row groups per trip so their LDS reads overlap
#pragma unroll 4
    for (int u = 0; u < NIT; ++u) EPI_GROUP(u, bf16x8{});
  } else if (KIND == EK_ROPE) {           // one rolled copy: the kernels must stay inside the instruction cache
#pragma unroll 1
    for (int u = 0; u < NIT; ++u) EPI_GROUP(u, bf16x8{});
  } else if (KIND == EK_GELU2) {          // fixed-kind kernels only: four row groups in flight (the erf-GELU pair is a long dependent chain)
#pragma unroll 4
    for (int u = 0; u < NIT; ++u) EPI_GROUP(u, bf16x8{});
  } else if (FIXED) {
#pragma unroll 2
    for (int u = 0; u < NIT; ++u) EPI_GROUP(u, bf16x8{});
  } else {
#pragma unroll 1
    for (int u = 0; u < NIT; ++u) EPI_GROUP(u, bf16x8{});
  }
#undef EPI_GROUP
}
// Kernel instantiations with the epilogue kind FIXED at compile time (gemm3.hip, template parameter EPI >= 0): the host has checked the
// kind's conditions (epi_kind_host) plus N % 8 == 0, so neither the run-time kind dispatch nor the generic element-wise form is compiled
// in -- 8-15 KiB of code instead of 62, which leaves room to unroll the activation kinds for instruction-level parallelism.
template <int WN, int KIND, int ROWS = 64, int UNR = 0>
__device__ __forceinline__ void epi_pass_fixed(const Gemm2Params& p, const char* er, int lane, int mbase, int nbase, float gate,
                                               const EpiPre<WN, ROWS>& e, bf16x8 biasv) {
  epi_groups<WN, KIND, ROWS, true, UNR>(p, er, lane, mbase, nbase, gate, e, biasv);
}
// host-side twin of epi_kind() + the fixed kinds' extra conditions; returns the EPI template value to launch, -1 = the generic kernel
static inline int epi_kind_host(const Gemm2Params& p) {
  bool fast = ((p.ldc | p.ldres | p.ldaux | p.ldpre) & 7) == 0;
  if (!fast || (p.N & 7) || p.accumulate || (p.aux && p.res) || p.rope_rot) return -1;
  if (p.aux) return (p.dact == ACT_DERIV_U8 && !p.act && !p.pre) ? EK_AUX : -1;
  if (p.res) return (!p.act && !(p.pre && p.pre_deriv)) ? EK_RES : -1;
  if (p.pre) return (p.pre_deriv == 2 && p.act == ACT_GELU) ? EK_GELU2 : -1;
  return p.act ? EK_ACT : EK_PLAIN;
}

// one pass of the chosen kind
// ROPE: the kernel instantiation that serves the rotary epilogue (and nothing else: the host validated a plain alpha / bias
// epilogue, N % 8 == 0) -- a separate instantiation because the ordinary kernels sit within 1 KiB of the instruction cache.
template <int WN, int ROWS = 64, bool ROPE = false, int ROPE_UNR = 0>
__device__ __forceinline__ void epi_pass_kind(const Gemm2Params& p, const char* er, int lane, int mbase, int nbase, float gate, bool fast,
                                              int kind, const EpiPre<WN, ROWS>& e, bf16x8 biasv) {
  if (ROPE) { epi_groups<WN, EK_ROPE, ROWS, false, ROPE_UNR>(p, er, lane, mbase, nbase, gate, e, biasv); return; }
  // generic form: the whole wave for EK_GENERIC; otherwise only the lanes that own the partial last 8-column group of a ragged N
  // (the 74 053-column LM head) -- one copy of the general code serves both
  const int n_ = nbase + (lane % (WN / 8)) * 8;
  if (kind == EK_GENERIC || (n_ < p.N && n_ + 8 > p.N)) epi_pass<WN, ROWS>(p, er, lane, mbase, nbase, gate, fast);
  switch (kind) {
    case EK_PLAIN: epi_groups<WN, EK_PLAIN, ROWS>(p, er, lane, mbase, nbase, gate, e, biasv); break;
    case EK_ACT:   epi_groups<WN, EK_ACT, ROWS>(p, er, lane, mbase, nbase, gate, e, biasv); break;
    case EK_AUX:   epi_groups<WN, EK_AUX, ROWS>(p, er, lane, mbase, nbase, gate, e, biasv); break;
    case EK_RES:   epi_groups<WN, EK_RES, ROWS>(p, er, lane, mbase, nbase, gate, e, biasv); break;
    default: break;
  }
}
