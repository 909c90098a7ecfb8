// gemm7.hip's hand-ordered two-register-set loop at TWO waves per SIMD (round 5, experiment): 256 x 256 tiles, 512 threads = 8 waves, each wave
// owns 128 x 64 (the ping-pong kernels' blocking: 32 accumulator tiles = 128 AGPRs), K in 64-k stages of whole rows, two stages in LDS.
//
// Why: gemm7's K loop runs at 81 % of the matrix pipe with ONE wave per SIMD and pays for it in the epilogue (a lone wave issues vector
// instructions at half rate); the ping-pong kernels have the fast 8-wave epilogue and a phase-separated loop (65 - 70 %).  Here every wave
// runs gemm7's self-interleaved stream -- fragment reads of the other k-half, fused MFMA + LDS-DMA statements, counted waits, all as riders in
// its own MFMA gaps -- and the two waves of a SIMD simply compete for the matrix pipe: what one loses at a wait or a barrier the other takes.
// No phases, three workgroup barriers per 64 MFMAs per wave.  Register budget per wave at two waves per SIMD: 256 in all = 128 AGPRs
// (accumulators) + 128 VGPRs (two fragment sets 96, DMA offsets 6 - 8, read addresses 6, the rest scalar).
//
// Schedule of one iteration (tile t in stage s; F0 = its k-half 0, read during the previous iteration; a block = 32 MFMAs, gap g follows MFMA g):
//   block 0 (F0): A reads (k-half 1 -> F1) gaps 0..7 | gap 9 lgkmcnt(0) + barrier (A region of stage s free) | B reads gaps 10..17 |
//                 A DMA of tile t+2 fused at MFMA 18..21 | gap 23 lgkmcnt(0) + barrier (B region free) | B DMA fused at MFMA 24, 26, 28, 30 |
//                 stage toggles of the read addresses in the odd gaps 25..31
//   block 1 (F1): gap 4 vmcnt(8) + barrier (tile t+1 has landed everywhere) | reads of its k-half 0 -> F0 (B then A) gaps 5.. | lgkmcnt(0)
// Same k grouping inside every MFMA and the same k order per accumulator as every other variant: bit-identical results.
// Serves: k-contiguous A, B k-contiguous or k-strided (not packed), K % 64 == 0, K >= 128.
#include <stdlib.h>
#include <type_traits>
#include "gemm_half.h"

#define G8_STG 65536
#define G8_ASUB 32768

__device__ __forceinline__ void g8_mfma(f32x4& c, bf16x8 b, bf16x8 a) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(b), "v"(a));
}
template <int IMM>
__device__ __forceinline__ void g8_mfma_dma(f32x4& c, bf16x8 b, bf16x8 a, uint32_t m0base, const void* sbase, uint32_t voff) {
  asm volatile("s_add_u32 m0, %3, %4\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\tglobal_load_lds_dwordx4 %5, %6"
               : "+a"(c) : "v"(b), "v"(a), "s"(m0base), "n"(IMM), "v"(voff), "s"(sbase) : "memory", "m0", "scc");
}
template <int OFF>
__device__ __forceinline__ void g8_read128(bf16x8& out, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(out) : "v"(addr), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void g8_read_tr(s16x4& out, uint32_t addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(out) : "v"(addr), "n"(OFF) : "memory");
}
__device__ __forceinline__ void g8_dma(const void* sbase, uint32_t voff, uint32_t lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
}
__device__ __forceinline__ void g8_xor_stage(uint32_t& x) { asm volatile("v_xor_b32 %0, 0x10000, %0" : "+v"(x)); }
#define G8_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define G8_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory")
#define G8_BAR() asm volatile("s_barrier" ::: "memory")

template <bool BKS, int EPI>
__global__ __launch_bounds__(512, 2) void gemm8_bf16_kernel(Gemm2Params p) {
  constexpr bool ROPE = EPI == EK_ROPE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nwg = p.nbm * p.nbn;
  const int id = xcd_remap(blockIdx.x, nwg);
  const int GM = p.gm > 0 ? p.gm : 4;
  const int per_group = GM * p.nbn;
  const int grp_ = id / per_group;
  const int first_m = grp_ * GM;
  const int gsz = min(p.nbm - first_m, GM);
  const int in_g = id - grp_ * per_group;
  const int tm = first_m + in_g % gsz, tn = in_g / gsz;
  const int m0 = tm * 256, n0 = tn * 256;

  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = lane_id();
  const int wm = wave >> 2, wn = wave & 3;

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 fa0[8], fa1[8], fb0[4], fb1[4];
  s16x4 bl0[4], bh0[4], bl1[4], bh1[4];

  uint32_t aoff[4], boff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int P = (wave * 4 + i) * 64 + lane, row = P >> 3, c = (P & 7) ^ ((row >> 1) & 7);
    aoff[i] = (uint32_t)(((long)min(m0 + row, p.M - 1) * p.lda + c * 8) * 2);
    if (!BKS) boff[i] = (uint32_t)(((long)min(n0 + row, p.N - 1) * p.ldb + c * 8) * 2);
  }
  uint32_t boffs[2];
  if (BKS) dma_setup<true, 256, 8>(p.ldb, n0, p.N, wave, boffs);
  const uint32_t smem_lds = lds_addr(smem);
  const uint32_t dA = __builtin_amdgcn_readfirstlane(smem_lds + wave * 4096);
  const uint32_t dB = __builtin_amdgcn_readfirstlane(smem_lds + G8_ASUB + (BKS ? wave * 2048 : wave * 4096));
  // DMA number I of a stage and wave: 0..3 A, 4..7 B (k-strided: 4, 5 = k-half 0, 6, 7 = k-half 1)
#define G8_DMA(T, S_, I) do { constexpr int I_ = (I) & 7, J_ = I_ & 3;                                                             \
    if (I_ < 4) g8_dma((const char*)p.A + (long)(T) * 128, aoff[J_], dA + (S_) * G8_STG + J_ * 1024);                              \
    else if (!BKS) g8_dma((const char*)p.B + (long)(T) * 128, boff[J_], dB + (S_) * G8_STG + J_ * 1024);                           \
    else g8_dma((const char*)p.B + ((long)(2 * (T) + (J_ >> 1)) * 32 * p.ldb) * 2, boffs[J_ & 1],                                  \
                dB + (S_) * G8_STG + (J_ >> 1) * 16384 + (J_ & 1) * 1024); } while (0)

  uint32_t va0 = smem_lds + kc_off(wm * 128 + (lane & 15), lane >> 4), va1 = va0 ^ 64;
  uint32_t vb0 = 0, vb1 = 0, vbs[4];
  if (!BKS) { vb0 = smem_lds + G8_ASUB + kc_off(wn * 64 + (lane & 15), lane >> 4); vb1 = vb0 ^ 64; }
  else {
    const uint32_t lb = ks32_lane_base<256>(wn * 64);
#pragma unroll
    for (int j = 0; j < 4; ++j) vbs[j] = smem_lds + G8_ASUB + (lb ^ ((uint32_t)j << 5));
  }
#define G8_RD_A(F, KH, I) g8_read128<(I) * 2048>(fa##F[I], (KH) ? va1 : va0)
#define G8_RD_BC(F, KH, J) g8_read128<(J) * 2048>(fb##F[J], (KH) ? vb1 : vb0)
#define G8_RD_BSL(F, KH, J) g8_read_tr<(KH) * 16384>(bl##F[J], vbs[J])
#define G8_RD_BSH(F, KH, J) g8_read_tr<(KH) * 16384 + 2048>(bh##F[J], vbs[J])
#define G8_FB(F, J) (BKS ? join_halves(bl##F[J], bh##F[J]) : fb##F[J])
#define G8_MF(F, M) g8_mfma(acc[(M) / 4][(M) % 4], G8_FB(F, (M) % 4), fa##F[(M) / 4])
  // read number R of a k-half's B fragments (k-strided: 8 transposed halves; k-contiguous: 4) and A fragments (8)
#define G8_RD_B(F, KH, R) do { if (BKS) { if ((R) & 1) G8_RD_BSH(F, KH, ((R) >> 1) & 3); else G8_RD_BSL(F, KH, ((R) >> 1) & 3); }            \
                               else G8_RD_BC(F, KH, (R) & 3); } while (0)
  constexpr int NRB = BKS ? 8 : 4;

  const int nt = p.K >> 6;
  // ---- prologue: tiles 0 and 1 on their way, k-half 0 of tile 0 in F0
  G8_DMA(0, 0, 0); G8_DMA(0, 0, 1); G8_DMA(0, 0, 2); G8_DMA(0, 0, 3); G8_DMA(0, 0, 4); G8_DMA(0, 0, 5); G8_DMA(0, 0, 6); G8_DMA(0, 0, 7);
  G8_DMA(1, 1, 0); G8_DMA(1, 1, 1); G8_DMA(1, 1, 2); G8_DMA(1, 1, 3); G8_DMA(1, 1, 4); G8_DMA(1, 1, 5); G8_DMA(1, 1, 6); G8_DMA(1, 1, 7);
  G8_WAIT_VM(8);
  G8_BAR();
  G8_RD_B(0, 0, 0); G8_RD_B(0, 0, 1); G8_RD_B(0, 0, 2); G8_RD_B(0, 0, 3);
  if (BKS) { G8_RD_B(0, 0, 4); G8_RD_B(0, 0, 5); G8_RD_B(0, 0, 6); G8_RD_B(0, 0, 7); }
  G8_RD_A(0, 0, 0); G8_RD_A(0, 0, 1); G8_RD_A(0, 0, 2); G8_RD_A(0, 0, 3); G8_RD_A(0, 0, 4); G8_RD_A(0, 0, 5); G8_RD_A(0, 0, 6); G8_RD_A(0, 0, 7);
  G8_WAIT_LGKM0();

  uint32_t dAs = dA, dBs = dB;
  auto iter = [&](auto mode_c, int t) __attribute__((always_inline)) {
    constexpr int MODE = decltype(mode_c)::value;         // 0 steady state, 1 second to last (no DMA), 2 last (no DMA, no next reads)
    const char* srcA = (const char*)p.A + (long)(t + 2) * 128;
    const char* srcB0 = BKS ? (const char*)p.B + ((long)(2 * (t + 2)) * 32 * p.ldb) * 2 : (const char*)p.B + (long)(t + 2) * 128;
    const char* srcB1 = BKS ? srcB0 + (long)32 * p.ldb * 2 : srcB0;
#define G8_MFD(F, M, D) do { constexpr int D_ = (D) & 7, J_ = D_ & 3;                                                                \
      if (D_ < 4) g8_mfma_dma<J_ * 1024>(acc[(M) / 4][(M) % 4], G8_FB(F, (M) % 4), fa##F[(M) / 4], dAs, srcA, aoff[J_]);               \
      else if (!BKS) g8_mfma_dma<J_ * 1024>(acc[(M) / 4][(M) % 4], G8_FB(F, (M) % 4), fa##F[(M) / 4], dBs, srcB0, boff[J_]);           \
      else g8_mfma_dma<(J_ >> 1) * 16384 + (J_ & 1) * 1024>(acc[(M) / 4][(M) % 4], G8_FB(F, (M) % 4), fa##F[(M) / 4], dBs,             \
                                                            (J_ >> 1) ? srcB1 : srcB0, boffs[J_ & 1]); } while (0)
#define G8_XOR(X) do { constexpr int X_ = (X);                                                                                       \
      if (X_ == 0) g8_xor_stage(va0); else if (X_ == 1) g8_xor_stage(va1);                                                           \
      else if (!BKS) { if (X_ == 2) g8_xor_stage(vb0); else if (X_ == 3) g8_xor_stage(vb1); }                                        \
      else if (X_ < 6) g8_xor_stage(vbs[(X_ - 2) & 3]); } while (0)
    // ---- block 0
#define G8_B0(M) do {                                                                                                                \
      if (MODE == 0 && (M) >= 18 && (M) <= 21) G8_MFD(0, M, (M) - 18);                                                               \
      else if (MODE == 0 && (M) >= 24 && ((M) & 1) == 0) G8_MFD(0, M, 4 + (((M) - 24) >> 1));                                        \
      else G8_MF(0, M);                                                                                                              \
      if ((M) <= 7) G8_RD_A(1, 1, (M) & 7);                                                                                          \
      if ((M) == 9) { G8_WAIT_LGKM0(); if (MODE == 0) G8_BAR(); }                                                                     \
      if (BKS && (M) >= 10 && (M) <= 17) G8_RD_B(1, 1, (M) - 10);                                                                    \
      if (!BKS && (M) >= 10 && (M) <= 16 && ((M) & 1) == 0) G8_RD_B(1, 1, ((M) - 10) >> 1);                                          \
      if ((M) == 23) { G8_WAIT_LGKM0(); if (MODE == 0) G8_BAR(); }                                                                    \
      if (MODE != 2 && (M) >= 25 && ((M) & 1)) G8_XOR(((M) - 25) >> 1); } while (0)
#define G8_ROW0(I) do { G8_B0((I) * 4 + 0); G8_B0((I) * 4 + 1); G8_B0((I) * 4 + 2); G8_B0((I) * 4 + 3); } while (0)
    G8_ROW0(0); G8_ROW0(1); G8_ROW0(2); G8_ROW0(3); G8_ROW0(4); G8_ROW0(5); G8_ROW0(6); G8_ROW0(7);
    // ---- block 1
#define G8_B1(M) do {                                                                                                                \
      G8_MF(1, M);                                                                                                                   \
      if (MODE != 2 && (M) <= 1) G8_XOR(4 + (M));                                                                                    \
      if ((M) == 4 && MODE != 2) { if (MODE == 0) G8_WAIT_VM(8); else G8_WAIT_VM(0); G8_BAR(); }                                      \
      if (MODE != 2 && (M) >= 5 && (M) < 5 + NRB) G8_RD_B(0, 0, (M) - 5);                                                             \
      if (MODE != 2 && (M) >= 5 + NRB && (M) < 13 + NRB) G8_RD_A(0, 0, ((M) - 5 - NRB) & 7); } while (0)
#define G8_ROW1(I) do { G8_B1((I) * 4 + 0); G8_B1((I) * 4 + 1); G8_B1((I) * 4 + 2); G8_B1((I) * 4 + 3); } while (0)
    G8_ROW1(0); G8_ROW1(1); G8_ROW1(2); G8_ROW1(3); G8_ROW1(4); G8_ROW1(5); G8_ROW1(6); G8_ROW1(7);
    if (MODE != 2) G8_WAIT_LGKM0();
    dAs ^= G8_STG; dBs ^= G8_STG;
  };
  {
    int t = 0;
#pragma unroll 1
    for (; t + 2 < nt; ++t) iter(std::integral_constant<int, 0>{}, t);
    iter(std::integral_constant<int, 1>{}, t);
    iter(std::integral_constant<int, 2>{}, t + 1);
  }
#define G8_SETTLE(I) asm volatile("s_nop 7" : "+a"(acc[I][0]), "+a"(acc[I][1]), "+a"(acc[I][2]), "+a"(acc[I][3]))
  asm volatile("s_nop 15" ::: "memory");
  G8_SETTLE(0); G8_SETTLE(1); G8_SETTLE(2); G8_SETTLE(3); G8_SETTLE(4); G8_SETTLE(5); G8_SETTLE(6); G8_SETTLE(7);
  G8_BAR();                                          // every wave is done with the ring: the epilogue reuses it

  // ---- epilogue through LDS (gemm3.hip's): wave-private [64][64] f32 region, 16-B units XOR-swizzled by row, two passes of 64 rows
  float gate = 1.f;
  if (p.gate) gate = tanhf(bf2f(*p.gate));
  constexpr int WN = 64, NJ = 4, ESTR = WN * 4, UNITS = WN / 4;
  char* er = smem + wave * (64 * ESTR);
  const bool fast = ((p.ldc | p.ldres | p.ldaux | p.ldpre) & 7) == 0;
#define G8_EPI_STAGE(PASS) do {                                                                                    \
    _Pragma("unroll") for (int i2 = 0; i2 < 4; ++i2)                                                               \
      _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                                             \
        int row = i2 * 16 + (lane & 15), u = j * 4 + (lane >> 4);                                                  \
        *(f32x4*)(er + row * ESTR + ((u ^ (row & (UNITS - 1))) << 4)) = acc[(PASS) * 4 + i2][j];                   \
      }                                                                                                            \
    __builtin_amdgcn_s_waitcnt(0xc07f); } while (0)
  const int em = m0 + wm * 128, en = n0 + wn * WN;
  const int kind = EPI >= 0 ? EPI : epi_kind(p, fast);
  EpiPre<WN> pre0, pre1;
  bf16x8 biasv = epi_bias<WN>(p, lane, en, kind);
  epi_fetch<WN>(p, lane, em, en, kind, pre0);
  G8_EPI_STAGE(0);
  epi_fetch<WN>(p, lane, em + 64, en, kind, pre1);
  if (kind != EK_GENERIC) epi_inputs_ready();
  if (EPI >= 0 && !ROPE) epi_pass_fixed<WN, EPI < 0 ? 0 : EPI>(p, er, lane, em, en, gate, pre0, biasv);
  else epi_pass_kind<WN, 64, ROPE>(p, er, lane, em, en, gate, fast, kind, pre0, biasv);
  __builtin_amdgcn_s_waitcnt(0xc07f);
  G8_EPI_STAGE(1);
  if (EPI >= 0 && !ROPE) epi_pass_fixed<WN, EPI < 0 ? 0 : EPI>(p, er, lane, em + 64, en, gate, pre1, biasv);
  else epi_pass_kind<WN, 64, ROPE>(p, er, lane, em + 64, en, gate, fast, kind, pre1, biasv);
}

template <bool BKS, int EPI>
static void launch8(const Gemm2Params& p, hipStream_t s) {
  static bool attr_set = false;
  constexpr size_t lds = 2 * G8_STG;
  auto kern = gemm8_bf16_kernel<BKS, EPI>;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; }
  hipLaunchKernelGGL(kern, dim3(p.nbm * p.nbn), dim3(512), lds, s, p);
}

// returns 1 if launched, 0 if this form is not served
extern "C" int unimp_gemm8_launch(const unimp_gemm_desc* d, void* stream) {
  if (d->a_kstrided || d->b_kstrided == 2 || (d->K & 63) || d->K < 128) return 0;
  if ((long)d->M * d->lda * 2 >= (1L << 32) || (long)(d->b_kstrided ? 32 : d->N) * d->ldb * 2 >= (1L << 32)) return 0;
  Gemm2Params p;
  p.A = (const bf16*)d->A; p.B = (const bf16*)d->B; p.C = d->C;
  p.M = d->M; p.N = d->N; p.K = d->K; p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc;
  p.bias = (const bf16*)d->bias; p.res = (const bf16*)d->res; p.ldres = d->ldres;
  p.aux = (const bf16*)d->aux; p.ldaux = d->ldaux; p.pre = (bf16*)d->pre; p.ldpre = d->ldpre;
  p.gate = (const bf16*)d->gate; p.alpha = d->alpha; p.act = d->act; p.dact = d->dact;
  p.out_f32 = d->out_f32; p.accumulate = d->accumulate; p.pre_deriv = d->pre_deriv; GEMM2_FILL_ROPE(p, d); p.ksplit = 0;
  { static int gm = -1; if (gm < 0) { const char* e = getenv("UNIMP_GEMM_GM"); gm = e ? atoi(e) : 0; } p.gm = gm; }
  p.nbm = (d->M + 255) / 256;
  p.nbn = (d->N + 255) / 256;
  hipStream_t s = (hipStream_t)stream;
  const int b = d->b_kstrided;
#define L8(K_) do { if (b) launch8<true, K_>(p, s); else launch8<false, K_>(p, s); return 1; } while (0)
  if (p.rope_rot) L8(EK_ROPE);
  switch (epi_kind_host(p)) {
    case EK_PLAIN: L8(EK_PLAIN);
    case EK_ACT:   L8(EK_ACT);
    case EK_GELU2: L8(EK_GELU2);
    case EK_RES:   L8(EK_RES);
    case EK_AUX:   L8(EK_AUX);
    default: break;
  }
  L8(-1);
#undef L8
}
