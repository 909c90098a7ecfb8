// dK / dV of flash attention, third generation (gfx950): 64 keys per wave, ONE wave per SIMD, accumulators in the AGPR half of
// the register file.  Serves head dim 80 under the causal / no mask (+ kv_len) with Sq, Sk multiples of 32 -- the language
// model's self-attention, 32 launches per step; every other form stays on the earlier kernels (attention.hip dispatch).
//
// Why a third kernel (profiles/r04_attn_dkv_stamps.txt, r04_pmc_attention.csv): the 16-keys-per-wave kernel re-reads every
// Q / dO fragment once per 16 keys -- 22.5 KB of LDS reads per 22 MFMAs of 16 cycles, twelve waves per CU: the LDS pipe alone
// needs 2 100 of the 3 500 cycles a query tile takes, the matrix pipe is busy 15 %.  Here a wave owns TWO 32-key blocks
// (K / V fragments of both in registers, dK^T / dV^T [96][64] fp32 = 192 accumulator registers) and every fragment read feeds
// two v_mfma_f32_32x32x16_bf16: the same 22.5 KB serve 44 MFMAs of 32 cycles (LDS pipe 50 % at the matrix pipe's pace).
//   * workgroup = 4 waves = 256 keys; wave w owns the 32-key blocks w and w + 4 of them (interleaved: under the causal mask the
//     four waves see nearly the same number of (tile, block) pairs; contiguous halves would leave wave 3 idle for 6 of 16 tiles)
//   * query tiles of 32 rows; per tile ONE image of Q and ONE of dO (row pitch 12 x 16 B, the 16-byte chunks of a row XORed with
//     (row >> 2) & 3 inside their aligned group of four) serves both the row-fragment ds_read_b128 of S = Q K^T / dP = dO V^T and
//     the ds_read_b64_tr_b16 of dV^T += dO^T P / dK^T += Q^T dS: 13 LDS-DMA instructions per tile instead of 25 (the second
//     generation staged a row image and a transposed-read image of each).  The swizzle is applied on the SOURCE side of the
//     lane-linear DMA.  Four stages, tiles fetched three ahead, one barrier per tile.
//   * P and dS never leave registers: the 32x32 accumulator layout of S (lane: key = l & 31, 16 query rows) IS the B operand
//     of the dV / dK products, contraction over the query rows in the order the transposed reads deliver them.
//   * the softmax scale is applied once to dK in the epilogue (dS carries p (dP - delta) unscaled): one multiply per element less
//     in the loop whose vector work is as long as its matrix work.
// Same arithmetic contract as the earlier generations otherwise (tests/test_kernels_gpu.py runs all of them against the fp32
// reference): P and dS rounded to bf16 before their products, fp32 accumulation, dk optionally rotated back in the epilogue.
#include "attention2_common.h"

extern "C" int unimp_attn_dkv3_eligible(const AttnP* p);

template <int V> struct A3V { static constexpr int value = V; };
typedef __attribute__((ext_vector_type(8))) short s16x8;
__device__ __forceinline__ s16x8 a3_join(s16x4 lo, s16x4 hi) { return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7); }
// the value lives in the accumulator half of the register file from here on (its only readers are MFMA operands)
template <typename T> __device__ __forceinline__ void a3_pin_acc(T& v) { asm volatile("" : "+a"(v)); }

template <int D>
__global__ __launch_bounds__(256, 1) void attn_dkv3_kernel(AttnP p, int nx) {
  static_assert(D == 80, "image layout below: 10 chunks per row in a pitch of 12");
  constexpr int CPR = D / 8, PT = 12, KS = D / 16, ND = (D + 31) / 32;
  constexpr int IMG = 32 * PT * 16, NJ = IMG / 1024, OFF_DO = IMG, STAGE = 2 * IMG, NST = 4;
  constexpr int NT = (2 * NJ) / 4;                            // DMA instructions per wave and tile
  static_assert((2 * NJ) % 4 == 0, "image pieces divide over the four waves");
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  int kblk, h, b;
  a2_decode(nx, p.H, p.B, kblk, h, b);
  kblk = nx - 1 - kblk;                                      // causal: the FIRST key block sees the most query tiles
  const int kbase = kblk * 256;
  if (kbase >= p.Sk) return;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = lane_id(), hi5 = l >> 5, kl = l & 31;
  const bool causal = p.mask_mode == UNIMP_MASK_CAUSAL;
  const int kvl = p.kv_len ? min(p.kv_len[b], p.Sk) : p.Sk;
  const int key0a = kbase + 32 * wave, key0b = key0a + 128;  // the wave's two 32-key blocks
  const char* qb = (const char*)(p.q + b * p.q_bs + h * p.q_hs);
  const char* dob = (const char*)(p.d_o + b * p.do_bs + h * p.do_hs);
  const bf16* kbp = p.k + b * p.k_bs + h * p.k_hs;
  const bf16* vbp = p.v + b * p.v_bs + h * p.v_hs;
  const char* lse_b = (const char*)(p.lse + ((long)b * p.H + h) * p.SqS);
  const char* dl_b = (const char*)(p.delta + ((long)b * p.H + h) * p.SqS);
  const uint32_t q_sb = (uint32_t)(p.q_ss * 2), do_sb = (uint32_t)(p.do_ss * 2);

  const int nqt = p.Sq >> 5;
  int qt_a = causal ? kbase >> 5 : 0;
  const int qt_b = nqt;
  if (kbase >= kvl) qt_a = qt_b;                              // every key of the workgroup is padding: dk = dv = 0

  // DMA plan: instruction i = wave + 4 t (t < NT) moves piece i % NJ of image i / NJ (0: Q, 1: dO).  Slot s = 64 piece + lane of
  // an image holds row s / PT, chunk (s % PT) ^ ((row >> 2) & 3).  Chunks 10, 11 of a row are padding -- and chunk 10 of the Q
  // image's rows 0-7 / 8-15 carries the tile's lse / delta (four rows of 4 bytes per chunk): no separate row-block instruction,
  // every wave issues exactly NT per tile.  Per lane and t: a 64-bit source pointer for tile qt_a and its step per tile.
  const char* src0[NT];
  uint32_t step[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    int i = wave + 4 * t;
    int img = i >= NJ, j = i - img * NJ;
    int s_ = 64 * j + l;
    int r = s_ / PT, cp = s_ - r * PT;
    int c = cp ^ ((r >> 2) & 3);
    const uint32_t sb_ = img ? do_sb : q_sb;
    const char* base = (img ? dob : qb) + (long)qt_a * 32 * sb_ + (long)r * sb_ + min(c, CPR - 1) * 16;
    uint32_t st_ = 32 * sb_;
    if (!img && c == CPR && r < 16) { base = (r < 8 ? lse_b : dl_b) + ((long)qt_a * 32 + 4 * (r & 7)) * 4; st_ = 128; }
    src0[t] = base; step[t] = st_;
  }
  const uint32_t smem_lds = (uint32_t)(uintptr_t)LDS_PTR(char, smem);
  auto dma_tile = [&](int qt, int stage) {
    const uint32_t n = (uint32_t)(min(qt, qt_b - 1) - qt_a);   // past the last tile: fetch it again (uniform counts; the stage is never read)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      int i = wave + 4 * t;                                   // wave-uniform
      int img = i >= NJ, j = i - img * NJ;
      uint32_t dst = smem_lds + stage * STAGE + img * IMG + j * 1024;
      a2_glds_v(src0[t] + (unsigned long)n * step[t], __builtin_amdgcn_readfirstlane(dst));
    }
  };

  bf16x8 kf[2][KS], vf[2][KS];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    int key = min((kb ? key0b : key0a) + kl, p.Sk - 1);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      kf[kb][ks] = *(const bf16x8*)(kbp + (long)key * p.k_ss + ks * 16 + hi5 * 8);
      vf[kb][ks] = *(const bf16x8*)(vbp + (long)key * p.v_ss + ks * 16 + hi5 * 8);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { a2_pin(kf[kb][ks]); a3_pin_acc(vf[kb][ks]); }
  f32x16 dk[2][ND], dv[2][ND];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int nd = 0; nd < ND; ++nd)
#pragma unroll
      for (int r = 0; r < 16; ++r) { dk[kb][nd][r] = 0.f; dv[kb][nd][r] = 0.f; }
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int nd = 0; nd < ND; ++nd) { a3_pin_acc(dk[kb][nd]); a3_pin_acc(dv[kb][nd]); }
  const float sc2 = p.scale * LOG2E;

  // fragment addresses inside a stage.  Row fragments (ds_read_b128, lane: query row kl, chunk 2 ks + hi5): chunks hi5, 4 + hi5,
  // 8 + hi5 (ks 0, 2, 4) share the swizzled low bits, chunks 2 + hi5, 6 + hi5 (ks 1, 3) the other pair.
  const int kq = (kl >> 2) & 3;
  const int r_a = (kl * PT + (hi5 ^ kq)) * 16, r_b = (kl * PT + ((hi5 ^ kq) ^ 2)) * 16;
  // transposed fragments (two ds_read_b64_tr_b16: rows 4 hi5 + (i16 >> 2) and + 8 of a 16-row group; 32 g16 + 8 (i16 & 3) bytes
  // into a 64-byte group of columns): a row's swizzle key is 2 half + hi5
  const int g16 = (l >> 4) & 1, i16 = l & 15;
  const int trow = 4 * hi5 + (i16 >> 2), tc = 2 * g16 + ((i16 & 3) >> 1);
  const int t_0 = (trow * PT + (tc ^ hi5)) * 16 + 8 * (i16 & 1);
  const int t_1 = ((trow + 8) * PT + (tc ^ (2 + hi5))) * 16 + 8 * (i16 & 1);
  const int mykey_a = key0a + kl, mykey_b = key0b + kl;

  // one query tile for the wave's two key blocks; V = 0: the block sees no row of the tile, 1: every element is visible,
  // 2: per-element mask (the diagonal, or keys beyond kv_len).  The instruction stream is attention3_sched.inc (generated and
  // checked by tools/gen_attn3.py); this side prepares the per-row constants: nl = -lse log2(e) (the exponent offset; -1e30 for a
  // masked element of this lane's key: p = 0, dS = 0) and ndl = -delta, the value the dP accumulators START from.
  auto tile = [&](auto v0c, auto v1c, int stage, int q0) {
    constexpr int V0 = decltype(v0c)::value, V1 = decltype(v1c)::value;
    const uint32_t sbo = smem_lds + stage * STAGE;
    const uint32_t a_ra = sbo + r_a, a_rb = sbo + r_b, a_t0 = sbo + t_0, a_t1 = sbo + t_1;
    // lse / delta of rows 8 g + 4 hi5 .. + 3: chunk 10 of the Q image's row 2 g + hi5 (lse) and row 8 + 2 g + hi5 (delta)
    const char* ax = smem + stage * STAGE + hi5 * (PT * 16);
    f32x16 nl, ndl;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 a = *(const f32x4*)(ax + (2 * g * PT + (CPR ^ (g >> 1))) * 16);
      f32x4 d = *(const f32x4*)(ax + ((8 + 2 * g) * PT + (CPR ^ ((2 + (g >> 1)) & 3))) * 16);
#pragma unroll
      for (int e = 0; e < 4; ++e) { nl[4 * g + e] = a[e] * -LOG2E; ndl[4 * g + e] = -d[e]; }
    }
    // lane holds key = mykey and query rows q0 + 8 g + 4 hi5 + e (r = 4 g + e); element visible <=> qmin <= row (and the key is real)
    auto masked = [&](int mykey) {
      const int qmin = causal ? mykey : 0;
      const int a0 = q0 + 4 * hi5 - qmin;
      const unsigned rng = mykey < kvl ? (unsigned)(p.Sq - qmin) : 0u;
      f32x16 m;
#pragma unroll
      for (int r = 0; r < 16; ++r) m[r] = (unsigned)(a0 + 8 * (r >> 2) + (r & 3)) < rng ? nl[r] : -1e30f;
      return m;
    };
    f32x16 nl0 = nl, nl1 = nl;
    if (V0 == 2) nl0 = masked(mykey_a);
    if (V1 == 2) nl1 = masked(mykey_b);
    f32x16 s0, s1, dp0, dp1;
    bf16x8 qf[KS], dof[KS];
    s16x4 udl[2 * ND], udh[2 * ND], uql[2 * ND], uqh[2 * ND];
    u32x4 pf[2][2], dsf[2][2];
#include "attention3_sched.inc"
  };
  // visibility of a 32-key block for the 32 rows from q0: 0 none, 1 all, 2 mixed
  auto vis = [&](int k0, int q0) {
    if (k0 >= kvl || (causal && k0 > q0 + 31)) return 0;
    return (k0 + 32 <= kvl && (!causal || k0 + 31 <= q0)) ? 1 : 2;
  };

  // prologue: tiles qt_a, +1, +2 in flight, the first one waited for
  if (qt_a < qt_b) {
    dma_tile(qt_a, 0); dma_tile(qt_a + 1, 1); dma_tile(qt_a + 2, 2);
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NT) : "memory");
  }
  __syncthreads();

  // The tile loop, one straight-line loop per visibility state of the wave's two blocks.  A block's state over the tiles of a
  // sequence only moves none -> mixed -> all (causal: the diagonal passes once; keys beyond kv_len stay invisible), so the pairs
  // are visited in an order that is a chain of the product order and every loop below runs at most once.  (One loop with a
  // switch over the states made the structurizer merge the 192 accumulator registers through copies after every tile.)
  int stage = 0, qt = qt_a;
  auto run = [&](auto v0c, auto v1c) {
    constexpr int V0 = decltype(v0c)::value, V1 = decltype(v1c)::value;
    while (qt < qt_b && vis(key0a, qt * 32) == V0 && vis(key0b, qt * 32) == V1) {
      dma_tile(qt + 3, (stage + 3) & 3);
      if (V0 || V1) tile(v0c, v1c, stage, qt * 32);
      // tile qt + 1 must have landed before the barrier: the two newest (qt + 2, qt + 3) may stay in flight
      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NT) : "memory");
      __syncthreads();
      stage = (stage + 1) & 3;
      ++qt;
    }
  };
  run(A3V<0>{}, A3V<0>{});
  run(A3V<2>{}, A3V<0>{}); run(A3V<0>{}, A3V<2>{});
  run(A3V<1>{}, A3V<0>{}); run(A3V<2>{}, A3V<2>{}); run(A3V<0>{}, A3V<1>{});
  run(A3V<1>{}, A3V<2>{}); run(A3V<2>{}, A3V<1>{});
  run(A3V<1>{}, A3V<1>{});

  // epilogue through a wave-private LDS region, once the tiles fetched past the end have landed
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  char* ew = smem + wave * (32 * (D * 2 + 16));
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    const int key0 = kb ? key0b : key0a;
    if (key0 < p.Sk) {
      a2_store_rows<D, ND>(ew, dk[kb], p.scale, p.dk + b * p.dk_bs + h * p.dk_hs, p.dk_ss, key0, p.Sk, p.rope_cos, p.rope_sin, p.rope_half, p.rope_step);
      __builtin_amdgcn_wave_barrier();
      a2_store_rows<D, ND>(ew, dv[kb], 1.f, p.dv + b * p.dv_bs + h * p.dv_hs, p.dv_ss, key0, p.Sk);
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// forms this kernel serves (everything else: the earlier generations)
extern "C" int unimp_attn_dkv3_eligible(const AttnP* p) {
  if (p->D != 80 || p->alibi || p->q_off || p->k_off || p->seg) return 0;
  if (p->mask_mode != UNIMP_MASK_NONE && p->mask_mode != UNIMP_MASK_CAUSAL) return 0;
  if ((p->Sq & 31) || (p->Sk & 31) || p->Sq < 32) return 0;
  if ((p->q_ss | p->do_ss) & 7) return 0;
  if ((p->q_ss * 2 * 32) >> 31 || (p->do_ss * 2 * 32) >> 31) return 0;      // 32-bit per-lane source offsets inside a tile
  if (((uintptr_t)p->lse | (uintptr_t)p->delta) & 15) return 0;
  return 1;
}

int unimp_attn_dkv3_launch(const AttnP& p, void* stream) {
  constexpr int D = 80;
  constexpr int STAGE = 2 * (32 * 12 * 16);
  constexpr size_t lds = 4 * STAGE;
  static_assert(4 * 32 * (D * 2 + 16) <= lds, "epilogue staging fits");
  static bool attr_set = false;
  auto kern = attn_dkv3_kernel<D>;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; }
  const int nk = (p.Sk + 255) / 256;
  hipLaunchKernelGGL(kern, dim3(nk * p.H * p.B), dim3(256), lds, (hipStream_t)stream, p, nk);
  return unimp_check_launch("attn_dkv3");
}
