// Two-waves-per-SIMD bf16 MFMA GEMM for gfx950 WITHOUT phase alternation: 256 x 256 tiles, 512 threads = 8 waves of
// 128 x 64, 32-k half-stages in a 4-slot LDS-DMA ring, ONE workgroup barrier per half-stage.
//
// gemm3.hip separates memory work (L phase) and MFMAs (C phase) per wave and alternates two wave groups; its measured
// half-step is L + C + two barrier turnarounds because a wave's own L and C are serial.  Here every wave
// interleaves the fragment reads of half-stage h+1 and its 4 DMA instructions of half-stage h+3 between the 32 MFMAs of
// half-stage h (as gemm4.hip does with one wave per SIMD); with two waves on a SIMD, whenever one is held by an LDS-DMA
// or ds_read issue the other one's MFMAs go out.
#include "gemm_half.h"

#define G5_BM 256
#define G5_BN 256
#ifndef G5_NST
#define G5_NST 4
#endif

#define G5_FENCE() do { __builtin_amdgcn_sched_barrier(0); asm volatile("" ::: "memory"); } while (0)

template <bool AKS, bool BKS>
__global__ __launch_bounds__(512, 2) void gemm5_bf16_kernel(Gemm2Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int A_SUB = G5_BM * 64, B_SUB = G5_BN * 64, SUB = A_SUB + B_SUB;
  constexpr int NIA = G5_BM / 128, NIB = G5_BN / 128, NEW = NIA + NIB;   // LDS-DMA instructions a wave issues per half-stage

  int nwg = p.nbm * p.nbn;
  int id = xcd_remap(blockIdx.x, nwg);
  constexpr int GM = 4;
  int per_group = GM * p.nbn;
  int grp_ = id / per_group;
  int first_m = grp_ * GM;
  int gsz = min(p.nbm - first_m, GM);
  int in_g = id - grp_ * per_group;
  int tm = first_m + in_g % gsz, tn = in_g / gsz;
  int m0 = tm * G5_BM, n0 = tn * G5_BN;

  int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = lane_id();
  int wm = wave >> 2, wn = wave & 3;

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 ra0[8], rb0[4], ra1[8], rb1[4];
  s16x4 la0[8], ha0[8], lb0[4], hb0[4], la1[8], ha1[8], lb1[4], hb1[4];

  int nh = (p.K + 31) >> 5;
  uint32_t aoff[NIA], boff[NIB];
  dma_setup<AKS, G5_BM, 8>(p.lda, m0, p.M, wave, aoff);
  dma_setup<BKS, G5_BN, 8>(p.ldb, n0, p.N, wave, boff);
#define DMA(H) do { char* b_ = smem + ((H) % G5_NST) * SUB;                                                             \
    dma_issue<AKS, G5_BM, 8>(p.A, p.lda, (H), p.K, b_, wave, aoff);                                                \
    dma_issue<BKS, G5_BN, 8>(p.B, p.ldb, (H), p.K, b_ + A_SUB, wave, boff); } while (0)
  const uint32_t smem_lds = (uint32_t)(uintptr_t)LDS_PTR(char, smem);
  const uint32_t lbA = AKS ? ks32_lane_base<G5_BM>(wm * 128) : 0u, lbB = BKS ? ks32_lane_base<G5_BN>(wn * 64) : 0u;
#define LOADA(S, H, I) do { if (AKS) frag_ks32_asm<G5_BM>(lbA + smem_lds + ((H) % G5_NST) * SUB, (I), la##S[I], ha##S[I]);  \
    else ra##S[I] = frag_kc32(smem + ((H) % G5_NST) * SUB, wm * 128 + (I) * 16); } while (0)
#define LOADB(S, H, J) do { if (BKS) frag_ks32_asm<G5_BN>(lbB + smem_lds + ((H) % G5_NST) * SUB + A_SUB, (J), lb##S[J], hb##S[J]);  \
    else rb##S[J] = frag_kc32(smem + ((H) % G5_NST) * SUB + A_SUB, wn * 64 + (J) * 16); } while (0)
#define FA(S, I) (AKS ? join_halves(la##S[I], ha##S[I]) : ra##S[I])
#define FB(S, J) (BKS ? join_halves(lb##S[J], hb##S[J]) : rb##S[J])
// row I of the wave tile: 4 MFMAs sharing one A fragment.  Riding along: rows 0-3 fetch the next half-stage's fragments
// (2 of A + 1 of B each; unconditionally -- past the end of K they read a ring slot nobody uses into registers nobody
// reads).  ds_read issue between MFMAs is nearly free (tools/micro/mfma_mix.hip: +4 %); an LDS-DMA instruction is not --
// it holds its wave ~75 cycles while the CU's 64 B/clk load path takes the 1 KB -- so the 4 DMA instructions of
// half-stage H+PD are issued as one block OUTSIDE the MFMA stream: by group A (waves 0-3) before its MFMAs, by group B
// after, so that one group's DMA block runs under the other group's MFMAs.
#define ROW(SC, SN, H, I) do {                                                                                      \
    bf16x8 fa_ = FA(SC, I);                                                                                         \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[I][j] = MFMA16(fb_[j], fa_, acc[I][j]);                       \
    if ((I) < 4) { LOADA(SN, (H) + 1, (2 * (I)) & 7); LOADA(SN, (H) + 1, (2 * (I) + 1) & 7); LOADB(SN, (H) + 1, (I) & 3); }  \
  } while (0)
#define STEP(H, SC, SN) do {                                                                                        \
    if ((H) + PD <= nh) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((PD - 2) * NEW) : "memory");                      \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                           \
    G5_FENCE(); __builtin_amdgcn_s_barrier(); G5_FENCE();                                                           \
    if (wm == 0 && (H) + PD < nh) DMA((H) + PD);                                                                    \
    G5_FENCE();                                                                                                     \
    bf16x8 fb_[4];                                                                                                  \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) fb_[j] = FB(SC, j);                                               \
    ROW(SC, SN, H, 0); ROW(SC, SN, H, 1); ROW(SC, SN, H, 2); ROW(SC, SN, H, 3);                                     \
    ROW(SC, SN, H, 4); ROW(SC, SN, H, 5); ROW(SC, SN, H, 6); ROW(SC, SN, H, 7);                                     \
    G5_FENCE();                                                                                                     \
    if (wm == 1 && (H) + PD < nh) DMA((H) + PD);                                                                    \
    G5_FENCE(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); G5_FENCE(); } while (0)

  // K is eaten in PAIRS of half-stages by one straight-line loop body; an odd count is padded with a half-stage of
  // zeros -- the K-tail path of dma_issue sources zeros for k >= K.
  nh = (nh + 1) & ~1;
  constexpr int PD = G5_NST - 1;                        // prefetch distance in half-stages
  for (int h0 = 0; h0 < PD && h0 < nh; ++h0) DMA(h0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  G5_FENCE(); __builtin_amdgcn_s_barrier(); G5_FENCE();
#define LOAD0(I) do { LOADA(0, 0, I); if ((I) < 4) LOADB(0, 0, (I) & 3); } while (0)
  LOAD0(0); LOAD0(1); LOAD0(2); LOAD0(3); LOAD0(4); LOAD0(5); LOAD0(6); LOAD0(7);
  G5_FENCE(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); G5_FENCE();
#pragma unroll 1
  for (int h = 0; h < nh; h += 2) { STEP(h, 0, 1); STEP(h + 1, 1, 0); }
  G5_FENCE(); __builtin_amdgcn_s_barrier(); G5_FENCE();   // every wave is done with the ring: the epilogue reuses it
#undef DMA
#undef LOADA
#undef LOADB
#undef LOAD0
#undef FA
#undef FB
#undef ROW
#undef STEP

  // ---- epilogue through LDS: wave-private [64][64] f32 region (16 KiB), 16-B units XOR-swizzled by row, two passes
  constexpr int WN = 64, ESTR = WN * 4, UNITS = WN / 4;
  char* er = smem + wave * (64 * ESTR);
  float gate = 1.f;
  if (p.gate) gate = tanhf(bf2f(*p.gate));
  bool fast = ((p.ldc | p.ldres | p.ldaux | p.ldpre) & 7) == 0;      // N may be ragged: only its last 8-column group is element-wise
  // the two 64-row passes are written out (a loop the compiler declines to unroll would index acc at run time and
  // demote the whole accumulator array to scratch)
#define EPI_STAGE(PASS) do {                                                                                      \
    _Pragma("unroll") for (int i2 = 0; i2 < 4; ++i2)                                                               \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                            \
        int row = i2 * 16 + (lane & 15), u = j * 4 + (lane >> 4);                                                  \
        *(f32x4*)(er + row * ESTR + ((u ^ (row & (UNITS - 1))) << 4)) = acc[(PASS) * 4 + i2][j];                   \
      }                                                                                                            \
    __builtin_amdgcn_s_waitcnt(0xc07f); } while (0)
  // the epilogue kind is chosen once per tile; every global load goes out before the first store (see gemm_tile.h)
  const int em = m0 + wm * 128, en = n0 + wn * WN;
  const int kind = epi_kind(p, fast);
  EpiPre<WN> pre0, pre1;
  bf16x8 biasv = epi_bias<WN>(p, lane, en, kind);
  epi_fetch<WN>(p, lane, em, en, kind, pre0);
  EPI_STAGE(0);
  epi_fetch<WN>(p, lane, em + 64, en, kind, pre1);
  if (kind != EK_GENERIC) epi_inputs_ready();
  epi_pass_kind<WN>(p, er, lane, em, en, gate, fast, kind, pre0, biasv);
  __builtin_amdgcn_s_waitcnt(0xc07f);
  EPI_STAGE(1);
  epi_pass_kind<WN>(p, er, lane, em + 64, en, gate, fast, kind, pre1, biasv);
#undef EPI_STAGE
}

template <bool AKS, bool BKS>
static void launch5(const Gemm2Params& p, hipStream_t s) {
  static bool attr_set = false;
  constexpr size_t lds = G5_NST * (G5_BM * 64 + G5_BN * 64);
  auto kern = gemm5_bf16_kernel<AKS, BKS>;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; }
  hipLaunchKernelGGL(kern, dim3(p.nbm * p.nbn), dim3(512), lds, s, p);
}

extern "C" int unimp_gemm5_launch(const unimp_gemm_desc* d, void* stream) {
  Gemm2Params p;
  p.A = (const bf16*)d->A; p.B = (const bf16*)d->B; p.C = d->C;
  p.M = d->M; p.N = d->N; p.K = d->K; p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc;
  p.bias = (const bf16*)d->bias; p.res = (const bf16*)d->res; p.ldres = d->ldres;
  p.aux = (const bf16*)d->aux; p.ldaux = d->ldaux; p.pre = (bf16*)d->pre; p.ldpre = d->ldpre;
  p.gate = (const bf16*)d->gate; p.alpha = d->alpha; p.act = d->act; p.dact = d->dact;
  p.out_f32 = d->out_f32; p.accumulate = d->accumulate; p.pre_deriv = d->pre_deriv; GEMM2_FILL_ROPE(p, d); p.ksplit = 0; p.gm = 0;
  p.nbm = (d->M + G5_BM - 1) / G5_BM;
  p.nbn = (d->N + G5_BN - 1) / G5_BN;
  hipStream_t s = (hipStream_t)stream;
  int a = d->a_kstrided, b = d->b_kstrided;
  if (!a && !b) launch5<false, false>(p, s); else if (!a && b) launch5<false, true>(p, s);
  else if (a && b) launch5<true, true>(p, s); else launch5<true, false>(p, s);
  return 1;
}
