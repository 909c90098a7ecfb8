// One-wave-per-SIMD bf16 MFMA GEMM for gfx950: 256 x 256 tiles, 256 threads = 4 waves, each wave owns 128 x 128 of the
// tile (64 accumulator tiles = 256 registers; the unified 512-entry register file of a SIMD belongs to that one wave).
//
// Why: in the 8-wave ping-pong kernel (gemm3.hip) a wave owns 128 x 64, so every 32-k half-stage costs the block
// 8 x 12 = 96 KiB of fragment reads + 32 KiB of DMA writes against 1024 MFMA cycles -- the LDS is busy ~60 % of the time
// and every hiccup stalls the matrix pipe.  With 128 x 128 per wave the same half-stage needs 4 x 16 = 64 KiB of fragment
// reads (-33 %), and a wave hides its own LDS latency: MFMAs only wait for issue, so the ds_reads of half-stage h+1 and
// the DMA of half-stage h+3 are issued between the 64 MFMAs of half-stage h.
//
// Per half-stage h (all four waves run the same schedule):
//     s_waitcnt vmcnt(..)  my part of half-stage h+1 has landed;  s_barrier  -> everybody's has, and everybody has
//                          finished reading half-stage h-1 (its fragments were waited for before the previous barrier)
//     DMA(h+3) into the slot of h-1;  ds_read fragments of h+1 -> R[(h+1)&1], interleaved with the 64 MFMAs on R[h&1]
//     s_waitcnt lgkmcnt(0)
// LDS images, DMA addressing and the epilogue are those of gemm3.hip (gemm_half.h / gemm_tile.h).
#include "gemm_half.h"

#define G4_BM 256
#define G4_BN 256
#ifndef G4_NST
#define G4_NST 4
#endif


// The 64 accumulator tiles must live in AGPRs (256 of them; the VGPR half holds fragments and addresses).  Left to the
// builtin, hipcc keeps ~40 tiles in VGPRs across the loop back-edge and shuttles them with v_accvgpr_write + s_nop before
// every use; an "a"-constrained asm operand pins the class.  No MFMA hazard needs software help inside the loop (a tile
// is touched once per half-stage, operands come from ds_read under s_waitcnt); the epilogue waits out the last MFMAs.
__device__ __forceinline__ void mfma_agpr(f32x4& c, bf16x8 a, bf16x8 b) {
  asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

#ifdef G4_STAMP      // debug build: cycle stamps of one block's waves (round-2 tool, removed; see docs/history); LDS bytes [128 KiB, +8 KiB) hold them
__device__ unsigned long long g4_stamps[4 * 64 * 4];
extern "C" int unimp_debug_g4_stamps(void* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g4_stamps), sizeof(g4_stamps)); }
#define STAMP(H, K_) do { if (blockIdx.x == 300 && (H) < 64) { unsigned long long t_;                               \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");                                     \
    if (lane == 0) ((unsigned long long*)(smem + 131072))[(wave * 64 + (H)) * 4 + (K_)] = t_; } } while (0)
#else
#define STAMP(H, K_) do {} while (0)
#endif
#define G4_FENCE() do { __builtin_amdgcn_sched_barrier(0); asm volatile("" ::: "memory"); } while (0)

template <bool AKS, bool BKS>
__global__ __launch_bounds__(256, 1) void gemm4_bf16_kernel(Gemm2Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int A_SUB = G4_BM * 64, B_SUB = G4_BN * 64, SUB = A_SUB + B_SUB;
  constexpr int NIA = G4_BM / 64, NIB = G4_BN / 64, NEW = NIA + NIB;   // LDS-DMA instructions a wave issues per half-stage

  int nwg = p.nbm * p.nbn;
  int id = xcd_remap(blockIdx.x, nwg);
  constexpr int GM = 4;
  int per_group = GM * p.nbn;
  int grp_ = id / per_group;
  int first_m = grp_ * GM;
  int gsz = min(p.nbm - first_m, GM);
  int in_g = id - grp_ * per_group;
  int tm = first_m + in_g % gsz, tn = in_g / gsz;
  int m0 = tm * G4_BM, n0 = tn * G4_BN;

  int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = lane_id();
  int wm = wave >> 1, wn = wave & 1;

  f32x4 acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 ra0[8], rb0[8], ra1[8], rb1[8];
  s16x4 la0[8], ha0[8], lb0[8], hb0[8], la1[8], ha1[8], lb1[8], hb1[8];

  int nh = (p.K + 31) >> 5;
  uint32_t aoff[NIA], boff[NIB];
  dma_setup<AKS, G4_BM, 4>(p.lda, m0, p.M, wave, aoff);
  dma_setup<BKS, G4_BN, 4>(p.ldb, n0, p.N, wave, boff);
#define DMA(H) do { char* b_ = smem + ((H) % G4_NST) * SUB;                                                             \
    dma_issue<AKS, G4_BM, 4>(p.A, p.lda, (H), p.K, b_, wave, aoff);                                                \
    dma_issue<BKS, G4_BN, 4>(p.B, p.ldb, (H), p.K, b_ + A_SUB, wave, boff); } while (0)
  const uint32_t smem_lds = (uint32_t)(uintptr_t)LDS_PTR(char, smem);
  const uint32_t lbA = AKS ? ks32_lane_base<G4_BM>(wm * 128) : 0u, lbB = BKS ? ks32_lane_base<G4_BN>(wn * 128) : 0u;
#define LOADA(S, H, I) do { if (AKS) frag_ks32_asm<G4_BM>(lbA + smem_lds + ((H) % G4_NST) * SUB, (I), la##S[I], ha##S[I]);  \
    else ra##S[I] = frag_kc32(smem + ((H) % G4_NST) * SUB, wm * 128 + (I) * 16); } while (0)
#define LOADB(S, H, J) do { if (BKS) frag_ks32_asm<G4_BN>(lbB + smem_lds + ((H) % G4_NST) * SUB + A_SUB, (J), lb##S[J], hb##S[J]);  \
    else rb##S[J] = frag_kc32(smem + ((H) % G4_NST) * SUB + A_SUB, wn * 128 + (J) * 16); } while (0)
#define FA(S, I) (AKS ? join_halves(la##S[I], ha##S[I]) : ra##S[I])
#define FB(S, J) (BKS ? join_halves(lb##S[J], hb##S[J]) : rb##S[J])
// row I of the wave tile: 8 MFMAs sharing one A fragment.  An MFMA occupies the matrix pipe for 16 cycles, so ~3 cheap
// instructions issue in its shadow for free while a CLUSTER of them idles the pipe (measured with G4_STAMP: with the 4
// fragment reads + 1 DMA + their address math bunched at the end of a row the 64 MFMAs of a half-stage took 1 610 cycles
// to issue instead of 1 088).  So everything that rides along is spread one piece per MFMA gap: row I fetches fragment I
// of A and of B for the next half-stage (unconditionally -- past the end of K they read a ring slot nobody uses into
// registers nobody reads) and issues one of the 8 DMA instructions of half-stage H+PD.
#define MF(I, J) mfma_agpr(acc[I][J], fb_[J], fa_)
#define ROW(SC, SN, H, I) do {                                                                                      \
    bf16x8 fa_ = FA(SC, I);                                                                                         \
    MF(I, 0); G4_FENCE(); LOADA(SN, (H) + 1, I); G4_FENCE();                                                        \
    MF(I, 1); MF(I, 2); G4_FENCE(); LOADB(SN, (H) + 1, I); G4_FENCE();                                              \
    MF(I, 3); MF(I, 4); G4_FENCE(); if (fast_) DMA_ROW(H, I); G4_FENCE();                                           \
    MF(I, 5); MF(I, 6); MF(I, 7); } while (0)
#define DMA_ROW(H, I) do {                                                                                           \
      if ((I) < 4) dma_one<AKS, G4_BM, 4>(p.A, p.lda, (H) + PD, smem + (((H) + PD) % G4_NST) * SUB, wave, aoff, (I) & 3);          \
      else dma_one<BKS, G4_BN, 4>(p.B, p.ldb, (H) + PD, smem + (((H) + PD) % G4_NST) * SUB + A_SUB, wave, boff, (I) & 3); } while (0)
#define STEP(H, SC, SN) do {                                                                                        \
    STAMP(H, 0);                                                                                                    \
    if ((H) + PD <= nh) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((PD - 2) * NEW) : "memory");                      \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                           \
    STAMP(H, 1);                                                                                                    \
    G4_FENCE(); __builtin_amdgcn_s_barrier(); G4_FENCE();                                                           \
    STAMP(H, 2);                                                                                                    \
    bool fast_ = ((H) + PD) * 32 + 32 <= p.K;                 /* full half-stage: DMA spread over the rows */       \
    if (!fast_ && (H) + PD < nh) DMA((H) + PD);               /* ragged / zero half-stage: predicated path */       \
    bf16x8 fb_[8];                                                                                                  \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) fb_[j] = FB(SC, j);                                               \
    ROW(SC, SN, H, 0); ROW(SC, SN, H, 1); ROW(SC, SN, H, 2); ROW(SC, SN, H, 3);                                     \
    ROW(SC, SN, H, 4); ROW(SC, SN, H, 5); ROW(SC, SN, H, 6); ROW(SC, SN, H, 7);                                     \
    STAMP(H, 3);                                                                                                    \
    G4_FENCE(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); G4_FENCE(); } while (0)

  // K is eaten in PAIRS of half-stages by one straight-line loop body (any control flow that forks the 64 accumulator
  // tiles makes the allocator copy and spill AGPRs); an odd count is padded with a half-stage of zeros -- the K-tail path
  // of dma_issue sources zeros for k >= K.
  nh = (nh + 1) & ~1;
  constexpr int PD = G4_NST - 1;                        // prefetch distance in half-stages
  for (int h0 = 0; h0 < PD && h0 < nh; ++h0) DMA(h0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  G4_FENCE(); __builtin_amdgcn_s_barrier(); G4_FENCE();
#define LOAD0(I) do { LOADA(0, 0, I); LOADB(0, 0, I); } while (0)
  LOAD0(0); LOAD0(1); LOAD0(2); LOAD0(3); LOAD0(4); LOAD0(5); LOAD0(6); LOAD0(7);
  G4_FENCE(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); G4_FENCE();
#pragma unroll 1
  for (int h = 0; h < nh; h += 2) { STEP(h, 0, 1); STEP(h + 1, 1, 0); }
  // the asm MFMAs are opaque to the hazard recogniser: wait out the last results before anything reads an accumulator
  // (each statement is ordered after the previous one and before every read of its row)
#define SETTLE(I) asm volatile("s_nop 7" : "+a"(acc[I][0]), "+a"(acc[I][1]), "+a"(acc[I][2]), "+a"(acc[I][3]),    \
                                             "+a"(acc[I][4]), "+a"(acc[I][5]), "+a"(acc[I][6]), "+a"(acc[I][7]))
  asm volatile("s_nop 15" ::: "memory");
  SETTLE(0); SETTLE(1); SETTLE(2); SETTLE(3); SETTLE(4); SETTLE(5); SETTLE(6); SETTLE(7);
#undef SETTLE
  G4_FENCE(); __builtin_amdgcn_s_barrier(); G4_FENCE();   // every wave is done with the ring: the epilogue reuses it
#ifdef G4_STAMP
  if (blockIdx.x == 300) for (int i = threadIdx.x; i < 1024; i += 256) g4_stamps[i] = ((unsigned long long*)(smem + 131072))[i];
#endif
#undef DMA
#undef LOADA
#undef LOADB
#undef LOAD0
#undef FA
#undef FB
#undef ROW
#undef MF
#undef DMA_ROW
#undef STEP

  // ---- epilogue through LDS: wave-private [64][128] f32 region (32 KiB), 16-B units XOR-swizzled by row, two passes
  constexpr int WN = 128, ESTR = WN * 4, UNITS = WN / 4;
  char* er = smem + wave * (64 * ESTR);
  float gate = 1.f;
  if (p.gate) gate = tanhf(bf2f(*p.gate));
  bool fast = ((p.N & 7) == 0) && (((p.ldc | p.ldres | p.ldaux | p.ldpre) & 7) == 0);
  // the two 64-row passes are written out (a loop the compiler declines to unroll would index acc at run time and
  // demote the whole accumulator array to scratch)
#define EPI_PASS(PASS) do {                                                                                       \
    _Pragma("unroll") for (int i2 = 0; i2 < 4; ++i2)                                                               \
      _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                           \
        int row = i2 * 16 + (lane & 15), u = j * 4 + (lane >> 4);                                                  \
        *(f32x4*)(er + row * ESTR + ((u ^ (row & (UNITS - 1))) << 4)) = acc[(PASS) * 4 + i2][j];                   \
      }                                                                                                            \
    __builtin_amdgcn_s_waitcnt(0xc07f);                                                                            \
    epi_pass<WN>(p, er, lane, m0 + wm * 128 + (PASS) * 64, n0 + wn * WN, gate, fast);                              \
    __builtin_amdgcn_s_waitcnt(0xc07f); } while (0)
  EPI_PASS(0);
  EPI_PASS(1);
#undef EPI_PASS
}

template <bool AKS, bool BKS>
static void launch4(const Gemm2Params& p, hipStream_t s) {
  static bool attr_set = false;
#ifdef G4_STAMP
  constexpr size_t lds = G4_NST * (G4_BM * 64 + G4_BN * 64) + 8192;
#else
  constexpr size_t lds = G4_NST * (G4_BM * 64 + G4_BN * 64);
#endif
  auto kern = gemm4_bf16_kernel<AKS, BKS>;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; }
  hipLaunchKernelGGL(kern, dim3(p.nbm * p.nbn), dim3(256), lds, s, p);
}

extern "C" int unimp_gemm4_launch(const unimp_gemm_desc* d, void* stream) {
  Gemm2Params p;
  p.A = (const bf16*)d->A; p.B = (const bf16*)d->B; p.C = d->C;
  p.M = d->M; p.N = d->N; p.K = d->K; p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc;
  p.bias = (const bf16*)d->bias; p.res = (const bf16*)d->res; p.ldres = d->ldres;
  p.aux = (const bf16*)d->aux; p.ldaux = d->ldaux; p.pre = (bf16*)d->pre; p.ldpre = d->ldpre;
  p.gate = (const bf16*)d->gate; p.alpha = d->alpha; p.act = d->act; p.dact = d->dact;
  p.out_f32 = d->out_f32; p.accumulate = d->accumulate; p.pre_deriv = d->pre_deriv; GEMM2_FILL_ROPE(p, d); p.ksplit = 0; p.gm = 0;
  p.nbm = (d->M + G4_BM - 1) / G4_BM;
  p.nbn = (d->N + G4_BN - 1) / G4_BN;
  hipStream_t s = (hipStream_t)stream;
  int a = d->a_kstrided, b = d->b_kstrided;
  if (!a && !b) launch4<false, false>(p, s); else if (!a && b) launch4<false, true>(p, s);
  else if (a && b) launch4<true, true>(p, s); else launch4<true, false>(p, s);
  return 1;
}
