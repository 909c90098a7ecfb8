// MX-fp8 path for the FROZEN towers (BASELINE config 5 "9B ... fp8 MFMA weights"): OCP e4m3 elements with one
// E8M0 scale per 32 consecutive elements of the contraction dimension, multiplied on the block-scaled matrix instruction
// v_mfma_scale_f32_16x16x128_f8f6f4 (fp32 accumulate; twice the bf16 MFMA rate per clock).
//
// Operand layout of that instruction, measured with exact integer data (tools/micro/mx_probe.hip; the guides give none):
//   lane l (r = l & 15, g = l >> 4) supplies 32 bytes of row r:  bytes 0..15 = k 16 g .. 16 g + 15,  bytes 16..31 = k 64 + 16 g ..;
//   the scale operand's lane (r, t = l >> 4) carries, in the byte chosen by op_sel, the E8M0 scale of row r's block k 32 t .. 32 t + 31;
//   B likewise with r = column; C / D as every 16x16 MFMA (col = l & 15, row = 4 (l >> 4) + reg).
// So a lane reads two 16-byte chunks (g and 4 + g) of its row's 128-byte K-step, and shifts its row's scale word by 8 t.
//
// unimp_mx_quantize : bf16 [R, K] -> e4m3 [R, K] + E8M0 [R, K / 32] (MX rule: shared exponent = floor(log2 amax) - 8, saturating cast)
// unimp_gemm_mxfp8  : C[M, N] = epi(A[M, K] B[N, K]^T), both operands quantised along K and k-contiguous.  The frozen weights are
//                     quantised ONCE in both orientations (W along K for y = x W^T; W^T along N for dx = dy W), so this one form
//                     serves forward and dX.  128 x 128 tile, 4 waves (64 x 64 each), 128-byte K-steps, two LDS stages filled by
//                     LDS-DMA (data + scale words; the XOR swizzle of the 128-byte rows is applied to the DMA's source address),
//                     epilogue through LDS in 16-byte row chunks: + bias, GELU / ReLU with the derivative as a second output,
//                     x aux (activation backward), + residual, bf16 out.
// Round 4: the 256 x 256 tile runs as a ping-pong kernel (gemm_mx_pp_kernel) whose epilogue is the bf16 family's fixed-kind code (gemm_tile.h;
// uint8 act'(z)); the tower's launches run at 0.45 of the 5 PF peak inside the train step (profiles/r04_mx_step_shapes.txt).
#include <stdlib.h>
#include "common.h"
#include "unimp_hip.h"
#include "gemm_tile.h"

typedef __attribute__((ext_vector_type(8))) int i32x8;

// ------------------------------------------------------------------------------------------- quantiser
// one thread per 32-element block: 64 bytes in, 32 + 1 bytes out
__global__ __launch_bounds__(256) void mx_quantize_kernel(const bf16* __restrict__ x, long ldx, uint8_t* __restrict__ q, long ldq,
                                                          uint8_t* __restrict__ sc, long lds_, int rows, int kblocks) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)rows * kblocks;
  if (i >= total) return;
  int kb = i % kblocks; long r = i / kblocks;
  const bf16* p = x + r * ldx + kb * 32;
  float v[32];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    bf16x8 t = *(const bf16x8*)(p + c * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[c * 8 + j] = bf2f(t[j]);
  }
  float amax = 0.f;
#pragma unroll
  for (int j = 0; j < 32; ++j) amax = fmaxf(amax, fabsf(v[j]));
  // shared exponent e = floor(log2 amax) - emax(e4m3 = 8), E8M0 byte = e + 127 in [0, 254]
  int eb = (int)((__float_as_uint(amax) >> 23) & 0xff);          // biased exponent of amax (0 for zero / subnormal bf16 magnitudes)
  int sbyte = max(eb - 8, 0);
  if (!(amax == amax) || eb == 255) sbyte = 254;                // NaN / inf in the block: largest finite scale; elements saturate / NaN
  float inv = __uint_as_float((uint32_t)(254 - sbyte) << 23);   // 2^-(sbyte - 127); sbyte = 0 -> 2^127
  if (sbyte == 254) inv = 1.17549435e-38f;
  uint32_t out[8];
#pragma unroll
  for (int w = 0; w < 8; ++w) {
    float a = fminf(fmaxf(v[4 * w] * inv, -448.f), 448.f), b = fminf(fmaxf(v[4 * w + 1] * inv, -448.f), 448.f);
    float c = fminf(fmaxf(v[4 * w + 2] * inv, -448.f), 448.f), d = fminf(fmaxf(v[4 * w + 3] * inv, -448.f), 448.f);
    int pk = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    pk = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, pk, true);
    out[w] = (uint32_t)pk;
  }
  uint8_t* qo = q + r * ldq + kb * 32;
  *(u32x4*)qo = u32x4{out[0], out[1], out[2], out[3]};
  *(u32x4*)(qo + 16) = u32x4{out[4], out[5], out[6], out[7]};
  sc[r * lds_ + kb] = (uint8_t)sbyte;
}

extern "C" int unimp_mx_quantize(const void* x, int64_t ldx, void* q, int64_t ldq, void* scales, int64_t lds_, int rows, int K,
                                 void* stream) {
  if (!x || !q || !scales) return unimp_set_error(UNIMP_ERR_ARG, "mx_quantize: null pointer");
  if (rows <= 0 || K <= 0) return UNIMP_OK;
  if ((K & 31) || (ldx & 7) || (ldq & 15) || ((uintptr_t)x & 15) || ((uintptr_t)q & 15))
    return unimp_set_error(UNIMP_ERR_SHAPE, "mx_quantize: K %% 32 == 0, ldx %% 8 == 0, ldq %% 16 == 0, 16-byte aligned pointers");
  long total = (long)rows * (K / 32);
  hipLaunchKernelGGL(mx_quantize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (long)ldx,
                     (uint8_t*)q, (long)ldq, (uint8_t*)scales, (long)lds_, rows, K / 32);
  return unimp_check_launch("mx_quantize");
}

// ------------------------------------------------------------------------------------------- GEMM
struct MxP {
  const uint8_t* A; const uint8_t* B; const uint8_t* sA; const uint8_t* sB;
  long lda, ldb, ldsa, ldsb;
  bf16* C; long ldc;
  const bf16* bias; const bf16* res; long ldres; const bf16* aux; long ldaux; bf16* pre; long ldpre;
  int M, N, K, act, nbm, nbn;
  int deriv_u8;              // the stored derivative act'(z) (`pre` written / `aux` read) is the 8-bit form of the bf16 GEMMs (common.h)
  uint8_t* sC; long ldsc;    // != null: C is written as an MX operand (e4m3 bytes at C, ldc in bytes, + E8M0 per 32 columns here)
};

// ---- fused MX output: 8 consecutive columns of one row, held by one lane; the 32-column scale block spans 4 ADJACENT lanes of the same row (both
// epilogue forms below hand consecutive 8-column chunks of a row to consecutive lanes, 8 chunks per 64-column wave tile).  The value is rounded
// to bf16 first, so the bytes are exactly what mx_quantize_kernel makes of the bf16 tensor this replaces.
template <int CTRL> __device__ __forceinline__ float mx_quad(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ void mx_store8(const MxP& p, const float (&v)[8], int m, int n) {
  float w[8], amax = 0.f;
#pragma unroll
  for (int r = 0; r < 8; ++r) { w[r] = bf2f(f2bf(v[r])); amax = fmaxf(amax, fabsf(w[r])); }
  amax = fmaxf(amax, mx_quad<0xB1>(amax));                       // quad_perm [1, 0, 3, 2]: lane ^ 1
  amax = fmaxf(amax, mx_quad<0x4E>(amax));                       // quad_perm [2, 3, 0, 1]: lane ^ 2
  int eb = (int)((__float_as_uint(amax) >> 23) & 0xff);
  int sbyte = max(eb - 8, 0);
  if (!(amax == amax) || eb == 255) sbyte = 254;
  float inv = __uint_as_float((uint32_t)(254 - sbyte) << 23);
  if (sbyte == 254) inv = 1.17549435e-38f;
  uint32_t out[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    float a = fminf(fmaxf(w[4 * h] * inv, -448.f), 448.f), b = fminf(fmaxf(w[4 * h + 1] * inv, -448.f), 448.f);
    float c = fminf(fmaxf(w[4 * h + 2] * inv, -448.f), 448.f), d = fminf(fmaxf(w[4 * h + 3] * inv, -448.f), 448.f);
    int pk = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    pk = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, pk, true);
    out[h] = (uint32_t)pk;
  }
  *(uint2*)((uint8_t*)p.C + (long)m * p.ldc + n) = uint2{out[0], out[1]};
  if (!(n & 31)) p.sC[(long)m * p.ldsc + (n >> 5)] = (uint8_t)sbyte;
}

// ---- general epilogue (every option behind a run-time branch) through a wave-private f32 staging area, 32 rows per pass, then 16-byte
// chunks of 8 consecutive columns.  acc[i][j]: row m = 16 i + r16 (the swapped product puts A's row on the lane), columns n = 16 j + 4 g + e
template <int MI, int NJ, int NWAVE, int STAGE>
__device__ __forceinline__ void mx_epilogue_general(const MxP& p, const f32x4 (&acc)[MI][NJ], char* smem, int wave, int em, int en) {
  constexpr int WNC = 16 * NJ, FP = WNC + 4, CPRW = WNC / 8;          // wave tile width, row pitch in floats, 8-column chunks per row
  static_assert(NWAVE * 32 * FP * 4 <= 2 * STAGE, "epilogue staging fits");
  const int l = lane_id(), r16 = l & 15, g = l >> 4;
  float* ewf = (float*)(smem + wave * (32 * FP * 4));
  const bool vec = !((p.ldc | p.ldres | p.ldaux | p.ldpre) & 7);
#pragma unroll
  for (int pass = 0; pass < MI / 2; ++pass) {
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
      for (int j = 0; j < NJ; ++j) *(f32x4*)(ewf + (16 * i2 + r16) * FP + 16 * j + 4 * g) = acc[2 * pass + i2][j];
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < 32 * CPRW / 64; ++it) {
      int idx = l + 64 * it;                    // 32 rows x CPRW chunks of 8 columns
      int r = idx / CPRW, c = idx - r * CPRW;
      int gm = em + 32 * pass + r, gn = en + c * 8;
      if (gm >= p.M || gn >= p.N) continue;
      float v[8], d[8];
      f32x4 x0 = *(const f32x4*)(ewf + r * FP + c * 8), x1 = *(const f32x4*)(ewf + r * FP + c * 8 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = x0[e]; v[4 + e] = x1[e]; }
      bool full = vec && gn + 8 <= p.N;
      if (p.bias) {
#pragma unroll
        for (int e = 0; e < 8; ++e) if (gn + e < p.N) v[e] += bf2f(p.bias[gn + e]);
      }
      if (p.act) {
        if (p.pre) {
          act_fwd_deriv_n<8>(p.act, v, d);
          if (p.deriv_u8) {
            uint32_t w0 = deriv_u8_pack4(d[0], d[1], d[2], d[3]), w1 = deriv_u8_pack4(d[4], d[5], d[6], d[7]);
            uint8_t* dst = (uint8_t*)p.pre + (long)gm * p.ldpre + gn;
            if (full) *(uint2*)dst = uint2{w0, w1};
            else for (int e = 0; e < 8; ++e) if (gn + e < p.N) dst[e] = (uint8_t)(((e < 4 ? w0 : w1) >> (8 * (e & 3))) & 0xffu);
          } else if (full) { bf16x8 o; for (int e = 0; e < 8; ++e) o[e] = f2bf(d[e]); *(bf16x8*)(p.pre + (long)gm * p.ldpre + gn) = o; }
          else for (int e = 0; e < 8; ++e) if (gn + e < p.N) p.pre[(long)gm * p.ldpre + gn + e] = f2bf(d[e]);
        } else act_fwd_n<8>(p.act, v);
      }
      if (p.aux) {
        if (p.deriv_u8) {
          const uint8_t* src = (const uint8_t*)p.aux + (long)gm * p.ldaux + gn;
          if (full) { uint2 w = *(const uint2*)src; for (int e = 0; e < 8; ++e) v[e] *= deriv_u8_get(e < 4 ? w.x : w.y, e & 3); }
          else for (int e = 0; e < 8; ++e) if (gn + e < p.N) v[e] *= deriv_u8_get(src[e], 0);
        } else if (full) { bf16x8 a = *(const bf16x8*)(p.aux + (long)gm * p.ldaux + gn); for (int e = 0; e < 8; ++e) v[e] *= bf2f(a[e]); }
        else for (int e = 0; e < 8; ++e) if (gn + e < p.N) v[e] *= bf2f(p.aux[(long)gm * p.ldaux + gn + e]);
      }
      if (p.res) {
        if (full) { bf16x8 a = *(const bf16x8*)(p.res + (long)gm * p.ldres + gn); for (int e = 0; e < 8; ++e) v[e] += bf2f(a[e]); }
        else for (int e = 0; e < 8; ++e) if (gn + e < p.N) v[e] += bf2f(p.res[(long)gm * p.ldres + gn + e]);
      }
      if (p.sC) mx_store8(p, v, gm, gn);                   // host: N % 32 == 0, so the 4 lanes of a scale block are in range together
      else if (full) { bf16x8 o; for (int e = 0; e < 8; ++e) o[e] = f2bf(v[e]); *(bf16x8*)(p.C + (long)gm * p.ldc + gn) = o; }
      else for (int e = 0; e < 8; ++e) if (gn + e < p.N) p.C[(long)gm * p.ldc + gn + e] = f2bf(v[e]);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
  }
}

// fixed kinds with the fused MX output (the frozen MLP: up-projection + GELU + uint8 GELU' -> MX, and dX x GELU' -> MX): the arithmetic of
// gemm_tile.h's epi8k<EK_GELU2> / <EK_AUX> (bias by add_rn, unit alpha / gate skipped), then mx_store8 instead of the bf16 store
enum { MXK_GELU2_MX = 100, MXK_AUX_MX = 101 };
template <int WN, int KIND>
__device__ __forceinline__ void mx_out_groups(const MxP& p, const char* er, int lane, int mbase, int nbase, const EpiPre<WN>& e, bf16x8 biasv) {
  constexpr int ESTR = WN * 4, UNITS = WN / 4, LPR = EpiPre<WN>::LPR, RPI = EpiPre<WN>::RPI, NIT = EpiPre<WN>::NIT;
  const int cg = lane % LPR, n = nbase + cg * 8;
  if (n >= p.N) return;
#pragma unroll
  for (int u = 0; u < NIT; ++u) {                  // fully unrolled: e.xv[u] must stay in registers
    int row = u * RPI + lane / LPR, m = mbase + row, sw = row & (UNITS - 1);
    f32x4 x0 = *(const f32x4*)(er + row * ESTR + (((2 * cg) ^ sw) << 4));
    f32x4 x1 = *(const f32x4*)(er + row * ESTR + (((2 * cg + 1) ^ sw) << 4));
    if (m < p.M) {
      float v[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
      if (p.bias) {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = add_rn(v[r], bf2f(biasv[r]));
      }
      if (KIND == EK_GELU2) {
        float dv[8];
        act_fwd_deriv_n<8>(ACT_GELU, v, dv);
        *(uint2*)((uint8_t*)p.pre + (long)m * p.ldpre + n) = uint2{deriv_u8_pack4(dv[0], dv[1], dv[2], dv[3]), deriv_u8_pack4(dv[4], dv[5], dv[6], dv[7])};
      } else {
        union { bf16x8 b; uint2 q[2]; } cv; cv.b = e.xv[u];
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] *= deriv_u8_get(r < 4 ? cv.q[0].x : cv.q[0].y, r & 3);
      }
      mx_store8(p, v, m, n);
    }
  }
}

// the bf16 GEMM family's parameter block for the shared fixed-kind epilogues (gemm_tile.h): alpha = 1, no gate, bf16 output
__device__ __forceinline__ Gemm2Params mx_as_gemm2(const MxP& p) {
  Gemm2Params q = {};
  q.C = p.C; q.M = p.M; q.N = p.N; q.K = p.K; q.ldc = p.ldc;
  q.bias = p.bias; q.res = p.res; q.ldres = p.ldres; q.aux = p.aux; q.ldaux = p.ldaux; q.pre = p.pre; q.ldpre = p.ldpre;
  q.alpha = 1.f; q.act = p.act; q.dact = p.deriv_u8 ? ACT_DERIV_U8 : ACT_DERIV; q.pre_deriv = p.pre ? (p.deriv_u8 ? 2 : 1) : 0;
  return q;
}

__device__ __forceinline__ void mx_glds16(const void* sbase, uint32_t voff, uint32_t lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
}
__device__ __forceinline__ void mx_glds4(const void* sbase, uint32_t voff, uint32_t lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
}

// Tile configurations: WMW x WNW waves, each owning (16 MI) x (16 NJ) outputs.
//   <2, 2, 4, 4>: 128 x 128 tile, 4 waves, 66 KiB of LDS (2 workgroups per CU) -- few-tile problems
//   <2, 4, 8, 4>: 256 x 256 tile, 8 waves (128 x 64 each, 2 per SIMD), 132 KiB of LDS -- the tower shapes: per K-step a wave reads
//                 12 fragments for 32 MFMAs instead of 8 for 16 (LDS bytes per MFMA: 0.75 vs 1.0)
template <int WMW, int WNW, int MI, int NJ>
__global__ __launch_bounds__(64 * WMW * WNW) void gemm_mx_kernel(MxP p) {
  constexpr int BM = WMW * 16 * MI, BN = WNW * 16 * NJ, NWAVE = WMW * WNW;
  constexpr int STAGE = (BM + BN) * 128 + (BM + BN) * 4;                 // data rows of 128 B, then one scale word per row
  constexpr int ND = (BM + BN) * 8 / 64 / NWAVE;                        // data DMA instructions per wave and K-step
  constexpr int NS = (BM + BN + 64 * NWAVE - 1) / (64 * NWAVE);         // scale DMA instructions per wave and K-step
  static_assert((BM + BN) * 8 % (64 * NWAVE) == 0, "data slots divide evenly over the waves");
  __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = lane_id(), r16 = l & 15, g = l >> 4;
  const int wm = wave / WNW, wn = wave % WNW;
  // tile mapping: XCD-aware, groups of 8 row tiles share their column sweep
  int nwg = p.nbm * p.nbn;
  int id = xcd_remap(blockIdx.x, nwg);
  constexpr int GM = 8;
  int per_group = GM * p.nbn, grp = id / per_group, first_m = grp * GM, gsz = min(p.nbm - first_m, GM);
  int in_g = id - grp * per_group;
  const int tm = first_m + in_g % gsz, tn = in_g / gsz;
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- DMA plan.  The stage holds the rows [A rows 0..BM) ; B rows 0..BN)], 8 slots of 16 B each; slot s holds
  // (row s >> 3, chunk (s & 7) ^ ((row >> 1) & 7)).  Scales: one dword per row and K-step.
  uint32_t d_off[ND];
#pragma unroll
  for (int t = 0; t < ND; ++t) {
    int s = 64 * (wave * ND + t) + l;
    int row = s >> 3, c = (s & 7) ^ ((row >> 1) & 7);
    bool isb = row >= BM;
    int gr = isb ? min(n0 + row - BM, p.N - 1) : min(m0 + row, p.M - 1);
    d_off[t] = (uint32_t)((long)gr * (isb ? p.ldb : p.lda) + c * 16);
  }
  uint32_t s_off[NS];
#pragma unroll
  for (int t = 0; t < NS; ++t) {
    int row = min(64 * (wave * NS + t) + l, BM + BN - 1);
    bool isb = row >= BM;
    int gr = isb ? min(n0 + row - BM, p.N - 1) : min(m0 + row, p.M - 1);
    s_off[t] = (uint32_t)((long)gr * (isb ? p.ldsb : p.ldsa));
  }
  const uint32_t smem_lds = (uint32_t)(uintptr_t)LDS_PTR(char, smem);
  auto dma_step = [&](int ks, int stage) {
    const uint8_t* ab = p.A + (long)ks * 128;
    const uint8_t* bb = p.B + (long)ks * 128;
#pragma unroll
    for (int t = 0; t < ND; ++t) {
      int i = wave * ND + t;                                  // wave-uniform; 64 slots = 8 rows per instruction
      bool isb = i * 8 >= BM;
      mx_glds16(isb ? bb : ab, d_off[t], __builtin_amdgcn_readfirstlane(smem_lds + stage * STAGE + i * 1024));
    }
#pragma unroll
    for (int t = 0; t < NS; ++t) {
      int i = wave * NS + t;                                  // 64 rows per instruction; BM is a multiple of 64
      if (i * 64 < BM + BN) {
        bool isb = i * 64 >= BM;
        mx_glds4((isb ? p.sB : p.sA) + (long)ks * 4, s_off[t], __builtin_amdgcn_readfirstlane(smem_lds + stage * STAGE + (BM + BN) * 128 + i * 256));
      }
    }
  };

  f32x4 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = p.K >> 7;
  dma_step(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int ks = 0; ks < nk; ++ks) {
    const int st = ks & 1;
    const char* sb = smem + st * STAGE;
    const char* ssc = sb + (BM + BN) * 128;
    if (ks + 1 < nk) dma_step(ks + 1, st ^ 1);
    i32x8 bfr[NJ];
    int sbv[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      int rb = BM + wn * 16 * NJ + j * 16 + r16;
      u32x4 b0 = *(const u32x4*)(sb + kc_off(rb, g)), b1 = *(const u32x4*)(sb + kc_off(rb, 4 + g));
      bfr[j] = i32x8{(int)b0[0], (int)b0[1], (int)b0[2], (int)b0[3], (int)b1[0], (int)b1[1], (int)b1[2], (int)b1[3]};
      sbv[j] = (int)(*(const uint32_t*)(ssc + rb * 4) >> (8 * g));
    }
    // D = B_frag x A_frag (operands swapped, as in the bf16 kernels): a lane then owns 4 consecutive n of one m
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      int ra = wm * 16 * MI + i * 16 + r16;
      u32x4 a0 = *(const u32x4*)(sb + kc_off(ra, g)), a1 = *(const u32x4*)(sb + kc_off(ra, 4 + g));
      i32x8 af = i32x8{(int)a0[0], (int)a0[1], (int)a0[2], (int)a0[3], (int)a1[0], (int)a1[1], (int)a1[2], (int)a1[3]};
      int sa = (int)(*(const uint32_t*)(ssc + ra * 4) >> (8 * g));
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bfr[j], af, acc[i][j], 0, 0, 0, sbv[j], 0, sa);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  mx_epilogue_general<MI, NJ, NWAVE, STAGE>(p, acc, smem, wave, m0 + wm * 16 * MI, n0 + wn * 16 * NJ);
}

// ---- ping-pong form of the 256 x 256 tile (round 4).  The kernel above runs its 8 waves in lockstep: everybody issues the LDS-DMA, everybody
// reads 24 KiB of fragments, everybody issues its 32 MFMAs, vmcnt(0), __syncthreads() -- the matrix pipes idle through the first two.
// Here the two waves of a SIMD (w and w + 4: group A = waves 0-3, group B = waves 4-7) run ONE PHASE APART like the bf16 ping-pong GEMM
// (gemm3.hip): per 128-byte K-step an L phase (fragment + scale reads of the step) and a C phase (32 block-scaled MFMAs = 1024 cycles),
// raw s_barrier between phases, B one interval behind A.  Two LDS stages of a whole K-step each (2 x 66 KiB; the instruction eats K = 128
// at once, so a stage cannot be cut in half): step s + 1 can only be fetched once BOTH groups have read step s - 1 out of its stage,
// i.e. from interval 2 s on -- group A issues its share of the DMA in L(s) after its own reads and waits for it behind its C(s) MFMAs,
// group B issues its share at the head of C(s - 1) (the same interval) and waits at the end of its L(s): every wait sits one phase of
// >= 1000 cycles behind its issue, nothing reads a stage in the phase that retires it.  Same MFMA order per output: same bits as above.
#define MXP_FENCE() do { __builtin_amdgcn_sched_barrier(0); asm volatile("" ::: "memory"); } while (0)
#define MXP_BARRIER() do { MXP_FENCE(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); MXP_FENCE(); } while (0)
// EPI: -1 = the general epilogue above; EK_PLAIN / EK_GELU2 / EK_AUX / EK_RES = that fixed kind of the bf16 family (gemm_tile.h epi_groups: inputs
// prefetched before the first store, uint8 act'(z), no option branches) -- the accumulator layout is the bf16 kernels', so the staging and
// the store loops are shared code and an epilogue costs here what it costs there.
template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_mx_pp_kernel(MxP p) {
  constexpr int BM = 256, BN = 256, NWAVE = 8, MI = 8, NJ = 4;
  constexpr int STAGE = (BM + BN) * 128 + (BM + BN) * 4;
  constexpr int ND = (BM + BN) * 8 / 64 / NWAVE;                        // 8 data DMA instructions per wave and K-step
  constexpr int NS = 1;                                                   // 512 rows of scale words = 8 instructions of 64 rows: one per wave
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = lane_id(), r16 = l & 15, g = l >> 4;
  const int wm = wave >> 2, wn = wave & 3;                                // wm doubles as the ping-pong group
  int nwg = p.nbm * p.nbn;
  int id = xcd_remap(blockIdx.x, nwg);
  constexpr int GM = 4;
  int per_group = GM * p.nbn, grp = id / per_group, first_m = grp * GM, gsz = min(p.nbm - first_m, GM);
  int in_g = id - grp * per_group;
  const int tm = first_m + in_g % gsz, tn = in_g / gsz;
  const int m0 = tm * BM, n0 = tn * BN;

  uint32_t d_off[ND];
#pragma unroll
  for (int t = 0; t < ND; ++t) {
    int s = 64 * (wave * ND + t) + l;
    int row = s >> 3, c = (s & 7) ^ ((row >> 1) & 7);
    bool isb = row >= BM;
    int gr = isb ? min(n0 + row - BM, p.N - 1) : min(m0 + row, p.M - 1);
    d_off[t] = (uint32_t)((long)gr * (isb ? p.ldb : p.lda) + c * 16);
  }
  uint32_t s_off;
  {
    int row = 64 * wave + l;
    bool isb = row >= BM;
    int gr = isb ? min(n0 + row - BM, p.N - 1) : min(m0 + row, p.M - 1);
    s_off = (uint32_t)((long)gr * (isb ? p.ldsb : p.ldsa));
  }
  const uint32_t smem_lds = (uint32_t)(uintptr_t)LDS_PTR(char, smem);
  const bool wave_b = wave * ND * 8 >= BM;                                // this wave's data rows belong to B (waves 4-7), its scale rows too
  auto dma_step = [&](int ks, int stage) {
    const uint8_t* base = (wave_b ? p.B : p.A) + (long)ks * 128;
#pragma unroll
    for (int t = 0; t < ND; ++t)
      mx_glds16(base, d_off[t], __builtin_amdgcn_readfirstlane(smem_lds + stage * STAGE + (wave * ND + t) * 1024));
    mx_glds4((wave_b ? p.sB : p.sA) + (long)ks * 4, s_off, __builtin_amdgcn_readfirstlane(smem_lds + stage * STAGE + (BM + BN) * 128 + wave * 256));
  };

  f32x4 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  i32x8 afr[MI], bfr[NJ];
  int sav[MI], sbv[NJ];
  // per-lane LDS offsets of fragment 0 (chunk g of the row; chunk 4 + g is the same address with bit 6 flipped; row + 16 i = + 2048 i)
  const int a_lane = kc_off(wm * 128 + r16, g), b_lane = kc_off(BM + wn * 64 + r16, g);
  const int sa_lane = (BM + BN) * 128 + (wm * 128 + r16) * 4, sb_lane = (BM + BN) * 128 + (BM + wn * 64 + r16) * 4;
#define MXP_LOADF(ST) do { const char* sb_ = smem + (ST) * STAGE;                                                     \
    _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                                               \
      u32x4 b0 = *(const u32x4*)(sb_ + b_lane + j * 2048), b1 = *(const u32x4*)(sb_ + (b_lane ^ 64) + j * 2048);   \
      bfr[j] = i32x8{(int)b0[0], (int)b0[1], (int)b0[2], (int)b0[3], (int)b1[0], (int)b1[1], (int)b1[2], (int)b1[3]};  \
      sbv[j] = (int)(*(const uint32_t*)(sb_ + sb_lane + j * 64) >> (8 * g)); }                                     \
    _Pragma("unroll") for (int i = 0; i < MI; ++i) {                                                               \
      u32x4 a0 = *(const u32x4*)(sb_ + a_lane + i * 2048), a1 = *(const u32x4*)(sb_ + (a_lane ^ 64) + i * 2048);   \
      afr[i] = i32x8{(int)a0[0], (int)a0[1], (int)a0[2], (int)a0[3], (int)a1[0], (int)a1[1], (int)a1[2], (int)a1[3]};  \
      sav[i] = (int)(*(const uint32_t*)(sb_ + sa_lane + i * 64) >> (8 * g)); } } while (0)
#define MXP_MFMAS() do { __builtin_amdgcn_s_setprio(1);                                                            \
    _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                                 \
      _Pragma("unroll") for (int j = 0; j < NJ; ++j)                                                               \
        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bfr[j], afr[i], acc[i][j], 0, 0, 0, sbv[j], 0, sav[i]);  \
    __builtin_amdgcn_s_setprio(0); } while (0)

  const int nk = p.K >> 7;
  dma_step(0, 0);
  if (nk > 1) { dma_step(1, 1); asm volatile("s_waitcnt vmcnt(%0)" :: "n"(ND + NS) : "memory"); }      // step 0 has landed, step 1 stays in flight
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  MXP_BARRIER();
  if (wm == 0) {
    for (int s = 0; s < nk; ++s) {                         // group A: L(s) at interval 2 s, C(s) at 2 s + 1
      MXP_LOADF(s & 1);
      if (s >= 1 && s + 1 < nk) dma_step(s + 1, (s + 1) & 1);       // the other stage: both groups read step s - 1 out of it two and one intervals ago
      MXP_BARRIER();
      MXP_MFMAS();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's share of step s + 1 (issued one phase ago) has landed
      MXP_BARRIER();
    }
    MXP_BARRIER();
  } else {
    MXP_BARRIER();                                         // group B runs one interval behind
    for (int s = 0; s < nk; ++s) {                         // L(s) at interval 2 s + 1, C(s) at 2 s + 2
      MXP_LOADF(s & 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the share of step s + 1 issued at the head of C(s - 1) (the prologue for s = 0)
      MXP_BARRIER();
      if (s + 2 < nk) dma_step(s + 2, s & 1);              // both groups have read step s out of this stage (intervals 2 s, 2 s + 1)
      MXP_MFMAS();
      MXP_BARRIER();
    }
  }
#undef MXP_LOADF
#undef MXP_MFMAS

  const int em = m0 + wm * 16 * MI, en = n0 + wn * 16 * NJ;
  if constexpr (EPI < 0) {
    mx_epilogue_general<MI, NJ, NWAVE, STAGE>(p, acc, smem, wave, em, en);
  } else {
    // the loop ended with a barrier behind every group's last fragment read: the stages are free.  Wave-private [64][64] f32 region,
    // 16-byte units XOR-swizzled by row, two 64-row passes (gemm3.hip's epilogue, same helpers)
    constexpr int WN = 16 * NJ, ESTR = WN * 4, UNITS = WN / 4;
    static_assert(NWAVE * 64 * ESTR <= 2 * STAGE, "epilogue staging fits");
    const Gemm2Params q = mx_as_gemm2(p);
    char* er = smem + wave * (64 * ESTR);
#define MXP_STAGE(PASS) do {                                                                                      \
    _Pragma("unroll") for (int i2 = 0; i2 < 4; ++i2)                                                               \
      _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                                           \
        int row = i2 * 16 + r16, u = j * 4 + g;                                                                    \
        *(f32x4*)(er + row * ESTR + ((u ^ (row & (UNITS - 1))) << 4)) = acc[(PASS) * 4 + i2][j];                   \
      }                                                                                                            \
    __builtin_amdgcn_s_waitcnt(0xc07f); } while (0)
    constexpr int KIND = EPI == MXK_GELU2_MX ? EK_GELU2 : (EPI == MXK_AUX_MX ? EK_AUX : EPI);
    EpiPre<WN> pre0, pre1;
    bf16x8 biasv = epi_bias<WN>(q, l, en, KIND);
    epi_fetch<WN>(q, l, em, en, KIND, pre0);
    MXP_STAGE(0);
    epi_fetch<WN>(q, l, em + 64, en, KIND, pre1);
    epi_inputs_ready();
    if constexpr (EPI >= 100) mx_out_groups<WN, KIND>(p, er, l, em, en, pre0, biasv);
    else epi_pass_fixed<WN, KIND < 0 ? 0 : KIND>(q, er, l, em, en, 1.f, pre0, biasv);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    MXP_STAGE(1);
    if constexpr (EPI >= 100) mx_out_groups<WN, KIND>(p, er, l, em + 64, en, pre1, biasv);
    else epi_pass_fixed<WN, KIND < 0 ? 0 : KIND>(q, er, l, em + 64, en, 1.f, pre1, biasv);
#undef MXP_STAGE
  }
}

extern "C" int unimp_gemm_mxfp8(const unimp_mx_gemm_desc* d, void* stream) {
  if (!d || !d->A || !d->B || !d->scale_a || !d->scale_b || !d->C) return unimp_set_error(UNIMP_ERR_ARG, "gemm_mxfp8: null pointer");
  if (d->M <= 0 || d->N <= 0) return UNIMP_OK;
  if (d->K <= 0 || (d->K & 127)) return unimp_set_error(UNIMP_ERR_SHAPE, "gemm_mxfp8: K must be a positive multiple of 128");
  if ((d->lda & 15) || (d->ldb & 15) || (d->ldsa & 3) || (d->ldsb & 3) || ((uintptr_t)d->A & 15) || ((uintptr_t)d->B & 15) ||
      ((uintptr_t)d->scale_a & 3) || ((uintptr_t)d->scale_b & 3) || ((uintptr_t)d->C & 15))
    return unimp_set_error(UNIMP_ERR_ALIGN, "gemm_mxfp8: lda, ldb %% 16, scale strides %% 4, 16-byte aligned data / output, 4-byte aligned scales");
  if ((long)d->M * d->lda >= (1L << 32) || (long)d->N * d->ldb >= (1L << 32))
    return unimp_set_error(UNIMP_ERR_SHAPE, "gemm_mxfp8: operand larger than 4 GiB");
  MxP p;
  p.A = (const uint8_t*)d->A; p.B = (const uint8_t*)d->B; p.sA = (const uint8_t*)d->scale_a; p.sB = (const uint8_t*)d->scale_b;
  p.lda = d->lda; p.ldb = d->ldb; p.ldsa = d->ldsa; p.ldsb = d->ldsb;
  p.C = (bf16*)d->C; p.ldc = d->ldc;
  p.bias = (const bf16*)d->bias; p.res = (const bf16*)d->res; p.ldres = d->ldres; p.aux = (const bf16*)d->aux; p.ldaux = d->ldaux;
  p.pre = (bf16*)d->pre; p.ldpre = d->ldpre;
  p.M = d->M; p.N = d->N; p.K = d->K; p.act = d->act; p.deriv_u8 = d->deriv_u8 != 0;
  p.sC = (uint8_t*)d->scale_c; p.ldsc = d->ldsc;
  if (p.sC && ((d->N & 31) || (d->ldc & 7))) return unimp_set_error(UNIMP_ERR_SHAPE, "gemm_mxfp8: MX output needs N %% 32 == 0 and ldc %% 8 == 0 (bytes)");
  if (p.deriv_u8 && p.pre && !p.act) return unimp_set_error(UNIMP_ERR_ARG, "gemm_mxfp8: deriv_u8 with `pre` needs an activation (pre = act'(z))");
  // 256 x 256 tiles once they fill the chip (>= 2 rounds of 256 CUs), 128 x 128 otherwise; env UNIMP_MX_TILE=128|256 forces one (A/B)
  static const int force = [] { const char* e = getenv("UNIMP_MX_TILE"); return e ? atoi(e) : 0; }();
  long t256 = (long)((d->M + 255) / 256) * ((d->N + 255) / 256);
  bool big = force ? force == 256 : t256 >= 512;
  static const int no_pp = [] { const char* e = getenv("UNIMP_MX_PP"); return e && e[0] == '0'; }();       // UNIMP_MX_PP=0: the lockstep kernel (A/B)
  if (big && !no_pp) {
    p.nbm = (d->M + 255) / 256; p.nbn = (d->N + 255) / 256;
    // fixed epilogue kinds (UNIMP_MX_FIXED_EPI=0: the general epilogue everywhere, A/B): the conditions of the bf16 family's epi_kind_host
    static const int no_fixed = [] { const char* e = getenv("UNIMP_MX_FIXED_EPI"); return e && e[0] == '0'; }();
    int kind = -1;
    if (!no_fixed && !((p.ldc | p.ldres | p.ldaux | p.ldpre) & 7) && !(p.N & 7) && !(p.aux && p.res)) {
      if (p.aux) kind = (p.deriv_u8 && !p.act && !p.pre) ? EK_AUX : -1;
      else if (p.res) kind = (!p.act && !p.pre) ? EK_RES : -1;
      else if (p.pre) kind = (p.deriv_u8 && p.act == ACT_GELU) ? EK_GELU2 : -1;
      else kind = p.act ? -1 : EK_PLAIN;
      if (p.sC) kind = kind == EK_GELU2 ? MXK_GELU2_MX : (kind == EK_AUX ? MXK_AUX_MX : -1);     // other MX-output forms: the general epilogue
    }
#define MXPP(E_) do { auto kern = gemm_mx_pp_kernel<E_>; static bool attr_set = false;                                                    \
      if (!attr_set) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr_set = true; }   \
      hipLaunchKernelGGL(kern, dim3(p.nbm * p.nbn), dim3(512), lds, (hipStream_t)stream, p); } while (0)
    constexpr int lds = 2 * ((256 + 256) * 128 + (256 + 256) * 4);
    switch (kind) {
      case EK_PLAIN: MXPP(EK_PLAIN); break;
      case EK_GELU2: MXPP(EK_GELU2); break;
      case EK_AUX:   MXPP(EK_AUX); break;
      case EK_RES:   MXPP(EK_RES); break;
      case MXK_GELU2_MX: MXPP(MXK_GELU2_MX); break;
      case MXK_AUX_MX:   MXPP(MXK_AUX_MX); break;
      default:       MXPP(-1); break;
    }
#undef MXPP
  } else if (big) {
    p.nbm = (d->M + 255) / 256; p.nbn = (d->N + 255) / 256;
    hipLaunchKernelGGL((gemm_mx_kernel<2, 4, 8, 4>), dim3(p.nbm * p.nbn), dim3(512), 0, (hipStream_t)stream, p);
  } else {
    p.nbm = (d->M + 127) / 128; p.nbn = (d->N + 127) / 128;
    hipLaunchKernelGGL((gemm_mx_kernel<2, 2, 4, 4>), dim3(p.nbm * p.nbn), dim3(256), 0, (hipStream_t)stream, p);
  }
  return unimp_check_launch("gemm_mxfp8");
}
