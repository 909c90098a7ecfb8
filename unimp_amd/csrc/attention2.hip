// Flash attention forward, second generation (gfx950): v_mfma_f32_32x32x16_bf16, K / V tiles by LDS-DMA.
//
// Why a second kernel (profiles/r02_pmc_attention.csv, first-generation kernel on the LM shape: 48 x 32 heads x 512 x 80):
// MFMA pipe busy 13 % of SIMD cycles, VALU 35 %, 42 % of wave time parked in waits at 2 waves per SIMD (212 registers), 37 %
// of the LDS cycles bank conflicts, 417 VALU instructions per 44 MFMAs.  Neither pipe was the bound: occupancy and
// instruction overhead were.  This kernel attacks those:
//   * 32x32x16 MFMA: a wave owns 32 query rows; S^T = K Q^T leaves a lane with 16 of the 32 keys of ONE query row per
//     32-key block (its partner lane l ^ 32 holds the other 16), so the row max / sum are in-lane chains plus ONE
//     cross-half exchange, and the exponentiated P registers ARE the next MFMA's B operand (contraction over keys in the
//     permuted order  slot (hi, j) <-> key 8 (j >> 2) + 4 hi + (j & 3)  of each 16-key group; V^T fragments are fetched in
//     that order by two ds_read_b64_tr_b16).  A 32x32x16 MFMA holds the vector issue port for 8 of its 32 cycles (the
//     16x16x32 form: 8 of 16), which leaves the softmax VALU work twice the issue room per flop.  Head dim 80 = 5 k-steps
//     of 16: no padding to 96 on the Q K^T contraction.
//   * K / V tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4): no staging registers, no ds_write pass.  The DMA
//     destination is lane-linear, so the conflict-free images are built on the SOURCE side: 16-byte slot s of the K image
//     holds (key s / PK, chunk s % PK) with an ODD pitch PK (ds_read_b128 of 16 keys at one chunk hits 16 distinct bank
//     groups); the V image uses a pitch with (4 PV) % 64 in {16, 48} (the 4 key rows x 64 B a half-wave reads with
//     ds_read_b64_tr_b16 fall on disjoint banks).  Pad slots are filled from a clamped valid address and never matter.
//   * 2 LDS stages, one barrier per 64-key tile; the softmax row sums stay per-lane partial sums until the epilogue; the
//     O rescale is skipped (wave-uniform) whenever no row's running max moved.
// Registers: Q 4 KS + S 32 + O 16 ND32 + ... -> 3 waves per SIMD at head dim 80 (first generation: 2).
// Same arithmetic contract as the first generation (attention.hip): masks none / causal + kv_len / segment, ALiBi, strided
// [B, S, H, D] views, lse output; tests/test_kernels_gpu.py runs both.
#include "common.h"
#include "unimp_hip.h"
#include "attention_params.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#define EXP2(x) __builtin_amdgcn_exp2f(x)
#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f

// one LDS-DMA wave-instruction: 64 lanes x 16 B, lane i lands at lds_dst + 16 i; saddr form (scalar 64-bit base + per-lane
// 32-bit byte offset).  M0 carries the LDS destination.
__device__ __forceinline__ void a2_glds(const void* sbase, uint32_t voff, uint32_t lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
               :: "v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
}

template <int D> struct A2Cfg {
  static constexpr int CPR = D / 8;                          // 16-byte chunks per row
  static constexpr int PK = CPR + 1 + (CPR & 1);             // odd pitch (slots) of the K image: 9 / 11 / 17
  static constexpr int PV = D == 128 ? 20 : 12;              // V image pitch: (4 PV) % 64 in {16, 48}
  static constexpr int KS = D / 16;                          // k-steps of the Q K^T contraction
  static constexpr int ND = (D + 31) / 32;                   // 32-row blocks of O^T
  static constexpr int NI = PK + PV;                         // DMA wave-instructions per 64-key tile
  static constexpr int STAGE = 64 * (PK + PV) * 16;
};

// key range [lo, hi) attended by query row `qr` of batch b
__device__ __forceinline__ void a2_key_range(const AttnP& p, int b, int qr, int& lo, int& hi) {
  lo = 0; hi = 0;
  if (qr >= p.Sq) return;
  int kvl = p.kv_len ? p.kv_len[b] : p.Sk;
  if (p.mask_mode == UNIMP_MASK_NONE) { hi = kvl; }
  else if (p.mask_mode == UNIMP_MASK_CAUSAL) { hi = min(qr + 1, kvl); }
  else { int t = p.seg[(long)b * p.Sq + qr]; if (t > 0) { lo = (t - 1) * p.seg_len; hi = min(t * p.seg_len, p.Sk); } }
}

template <int D, int NW, bool ALIBI>
__global__ __launch_bounds__(64 * NW) void attn_fwd2_kernel(AttnP p, int row_base) {
  using C = A2Cfg<D>;
  constexpr int CPR = C::CPR, PK = C::PK, PV = C::PV, KS = C::KS, ND = C::ND, NI = C::NI, STAGE = C::STAGE;
  constexpr int NT = (NI + NW - 1) / NW;                     // DMA instructions per wave and tile
  __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE];
  const int b = blockIdx.z, h = blockIdx.y;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = lane_id(), hi5 = l >> 5, ql = l & 31;
  const int qblk0 = row_base + blockIdx.x * (32 * NW);       // first query row of the block
  const int q0 = qblk0 + wave * 32;
  const bf16* qb = p.q + b * p.q_bs + h * p.q_hs;
  const char* kb = (const char*)(p.k + b * p.k_bs + h * p.k_hs);
  const char* vb = (const char*)(p.v + b * p.v_bs + h * p.v_hs);

  // ---- block-uniform key tile range
  int kt_lo = 0, kt_hi = 0;
  {
    int q_last = min(qblk0 + 32 * NW - 1, p.Sq - 1);
    int kvl = p.kv_len ? p.kv_len[b] : p.Sk;
    if (p.mask_mode == UNIMP_MASK_NONE) kt_hi = (kvl + 63) >> 6;
    else if (p.mask_mode == UNIMP_MASK_CAUSAL) kt_hi = (min(q_last + 1, kvl) + 63) >> 6;
    else {
      int t0 = p.seg[(long)b * p.Sq + qblk0], t1 = p.seg[(long)b * p.Sq + q_last];
      if (t1 > 0) { kt_lo = (max(t0 - 1, 0) * p.seg_len) >> 6; kt_hi = (min(t1 * p.seg_len, p.Sk) + 63) >> 6; }
    }
  }

  // ---- DMA plan of this wave: instruction i = wave + NW t; slot s = 64 i + lane -> (row, chunk) of the K or V image
  int d_row[NT]; uint32_t d_col[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    int i = wave + NW * t;
    bool isv = i >= PK;
    int s = 64 * (isv ? i - PK : i) + l;
    int pitch = isv ? PV : PK;
    int row = s / pitch, cs = s - row * pitch;
    d_row[t] = row;
    d_col[t] = (uint32_t)(min(cs, CPR - 1) * 16);
  }
  const uint32_t smem_lds = (uint32_t)(uintptr_t)LDS_PTR(char, smem);
  const long k_sb = p.k_ss * 2, v_sb = p.v_ss * 2;
  auto dma_tile = [&](int kt, int stage) {
    int rows_left = p.Sk - kt * 64;                          // >= 1
    const char* kt_base = kb + (long)kt * 64 * k_sb;
    const char* vt_base = vb + (long)kt * 64 * v_sb;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      int i = wave + NW * t;                                 // wave-uniform
      if (i < NI) {
        bool isv = i >= PK;
        int rc = min(d_row[t], rows_left - 1);               // rows beyond the tensor: clamped (they only meet masked scores)
        uint32_t off = (uint32_t)(rc * (isv ? v_sb : k_sb)) + d_col[t];
        uint32_t dst = smem_lds + stage * STAGE + (isv ? 64 * PK * 16 + (i - PK) * 1024 : i * 1024);
        a2_glds(isv ? vt_base : kt_base, off, __builtin_amdgcn_readfirstlane(dst));
      }
    }
  };

  // ---- Q fragments (B operand: n = query row, k = 16 ks + 8 hi + j) and the per-row key range
  bf16x8 qf[KS];
  int lo, hi;
  {
    int qr = q0 + ql;
    const bf16* qrow = qb + (long)min(qr, p.Sq - 1) * p.q_ss;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const bf16x8*)(qrow + ks * 16 + hi5 * 8);
    a2_key_range(p, b, qr, lo, hi);
  }
  int wlo = lo, whi = hi;                                     // key range seen by ANY row of the wave
#pragma unroll
  for (int o_ = 32; o_ > 0; o_ >>= 1) { wlo = min(wlo, __shfl_xor(wlo, o_, 64)); whi = max(whi, __shfl_xor(whi, o_, 64)); }
  wlo = __builtin_amdgcn_readfirstlane(wlo); whi = __builtin_amdgcn_readfirstlane(whi);
  const float sc2 = p.scale * LOG2E;
  const float ab = ALIBI ? p.alibi[h] / p.scale : 0.f;
  float m = -INFINITY, lsum = 0.f;                           // lsum: this lane's partial row sum (its 32 keys of every tile)
  f32x16 o[ND];
#pragma unroll
  for (int nd = 0; nd < ND; ++nd)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[nd][r] = 0.f;

  if (kt_lo < kt_hi) dma_tile(kt_lo, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // per-lane LDS read offsets (bytes) inside a stage
  const int k_rd = (ql * PK + hi5) * 16;                                      // + kb * 32 * PK * 16 + ks * 32
  const int g16 = (l >> 4) & 1, i16 = l & 15;
  const int v_rd = 64 * PK * 16 + ((4 * hi5 + (i16 >> 2)) * PV) * 16 + (16 * g16 + 4 * (i16 & 3)) * 2;   // + key group * PV * 16 + db * 64

  for (int kt = kt_lo; kt < kt_hi; ++kt) {
    const int st = (kt - kt_lo) & 1;
    const char* sb = smem + st * STAGE;
    if (kt + 1 < kt_hi) dma_tile(kt + 1, st ^ 1);
    if (q0 < p.Sq && kt * 64 < whi && kt * 64 + 64 > wlo) {          // wave-uniform: some row of this wave sees some key of this tile
      f32x16 s0, s1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        bf16x8 k0 = *(const bf16x8*)(sb + k_rd + ks * 32);
        bf16x8 k1 = *(const bf16x8*)(sb + k_rd + 32 * PK * 16 + ks * 32);
        s0 = MFMA32(k0, qf[ks], s0);
        s1 = MFMA32(k1, qf[ks], s1);
      }
      // lane holds, for query row q0 + ql, keys kt*64 + 32 kb + 8 (r >> 2) + 4 hi5 + (r & 3)
      const int key0 = kt * 64 + 4 * hi5;
      if (ALIBI) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float kk = ab * (float)(key0 + 8 * (r >> 2) + (r & 3));
          s0[r] += kk; s1[r] += kk + ab * 32.f;
        }
      }
      bool full = __all(kt * 64 >= lo && kt * 64 + 64 <= hi);
      if (!full) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int key = key0 + 8 * (r >> 2) + (r & 3);
          s0[r] = (key >= lo && key < hi) ? s0[r] : -INFINITY;
          s1[r] = (key + 32 >= lo && key + 32 < hi) ? s1[r] : -INFINITY;
        }
      }
      float mloc = fmaxf(s0[0], s1[0]);
#pragma unroll
      for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, fmaxf(s0[r], s1[r]));
      mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
      float mnew = fmaxf(m, mloc * sc2);                     // running max in the scaled (log2) domain; sc2 > 0
      float muse = (mnew == -INFINITY) ? 0.f : mnew;
      if (__any(mnew != m)) {                                // some row's max moved: rescale (rare after the first tiles)
        float alpha = EXP2(m - muse);
        lsum *= alpha;
#pragma unroll
        for (int nd = 0; nd < ND; ++nd)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[nd][r] *= alpha;
        m = mnew;
      }
      float rs = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s0[r] = EXP2(fmaf(s0[r], sc2, -muse)); s1[r] = EXP2(fmaf(s1[r], sc2, -muse));
        rs += s0[r] + s1[r];
      }
      lsum += rs;
      // P^T as the B operand of the P V products: k-step (kb, half) takes registers 8 half .. 8 half + 7 of block kb
      bf16x8 pf[4];
#pragma unroll
      for (int j = 0; j < 8; ++j) { pf[0][j] = f2bf(s0[j]); pf[1][j] = f2bf(s0[8 + j]); pf[2][j] = f2bf(s1[j]); pf[3][j] = f2bf(s1[8 + j]); }
#pragma unroll
      for (int nd = 0; nd < ND; ++nd)
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) {                     // 16-key group kg of the tile
          const char* a = sb + v_rd + kg * 16 * PV * 16 + nd * 64;
          s16x4 vlo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, a));
          s16x4 vhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, a + 8 * PV * 16));
          union { struct { s16x4 a, b; } s; bf16x8 v; } u;
          u.s.a = vlo; u.s.b = vhi;
          o[nd] = MFMA32(u.v, pf[kg], o[nd]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- epilogue: O^T[d][q] -> O[q][d]; lane holds d = 32 nd + 8 (r >> 2) + 4 hi5 + (r & 3) of its query row
  {
    float ltot = lsum + __shfl_xor(lsum, 32, 64);
    int qr = q0 + ql;
    if (qr < p.Sq) {
      float inv = ltot > 0.f ? 1.f / ltot : 0.f;
      bf16* ob = p.o + b * p.o_bs + (long)qr * p.o_ss + h * p.o_hs;
#pragma unroll
      for (int nd = 0; nd < ND; ++nd)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          int d0 = 32 * nd + 8 * g + 4 * hi5;
          if (d0 < D) {
            bf16x4 w = {f2bf(o[nd][4 * g] * inv), f2bf(o[nd][4 * g + 1] * inv), f2bf(o[nd][4 * g + 2] * inv), f2bf(o[nd][4 * g + 3] * inv)};
            *(bf16x4*)(ob + d0) = w;
          }
        }
      if (hi5 == 0 && p.lse) p.lse[((long)b * p.H + h) * p.Sq + qr] = ltot > 0.f ? (m + log2f(ltot)) * LN2 : -INFINITY;
    }
  }
}

template <int D, bool ALIBI>
static void launch_fwd2(const AttnP& p, hipStream_t s) {
  int full = p.Sq / 128, rem = p.Sq - full * 128;
  // whole 128-row blocks (plus a ragged one when more than one wave's worth of rows is left) by the 4-wave kernel; a tail of
  // at most 32 rows (the ViT's 257th token) by one-wave blocks instead of a fourth-empty 128-row block
  int big = full + (rem > 32 ? 1 : 0);
  if (big > 0) hipLaunchKernelGGL((attn_fwd2_kernel<D, 4, ALIBI>), dim3(big, p.H, p.B), dim3(256), 0, s, p, 0);
  if (rem > 0 && rem <= 32) hipLaunchKernelGGL((attn_fwd2_kernel<D, 1, ALIBI>), dim3(1, p.H, p.B), dim3(64), 0, s, p, full * 128);
}

int unimp_attn_fwd2_dispatch(const AttnP& p, void* stream) {
  hipStream_t s = (hipStream_t)stream;
#define FWD2(A_) do { if (p.D == 64) launch_fwd2<64, A_>(p, s); else if (p.D == 80) launch_fwd2<80, A_>(p, s); else launch_fwd2<128, A_>(p, s); } while (0)
  if (p.alibi) FWD2(true); else FWD2(false);
#undef FWD2
  return unimp_check_launch("attn_fwd2");
}
