// Two-workgroups-per-CU bf16 MFMA GEMM for gfx950 (round 6 experiment, variant `dw`): 128 x 256 tiles, 256 threads = 4 waves, each wave a
// 128 x 64 strip of the tile (the ping-pong kernel's wave tile: 32 accumulator tiles = 128 registers), 72 KiB of LDS -- so TWO workgroups
// share a CU, one wave of each per SIMD.
//
// Why: with one 8-wave workgroup per CU nothing runs on a CU while its workgroup is in its prologue (2.5 us),
// its epilogue (4 - 14 us) or waiting for the round's stragglers -- 15 - 35 % of a tile at K = 2560 / 1024 (gemm3 stamps) -- and a problem's
// last, partial round costs a whole tile time.  The ping-pong kernel already is two 4-wave groups one phase apart that happen to share a
// barrier, an A panel and a B panel; here they are two INDEPENDENT workgroups: while one is between tiles the other keeps the matrix pipe, and
// the tile count doubles (the ViT's 2 056 tiles of 256 x 256 become 4 112: the partial round is half as long).  The price is the shared B panel:
// per 32-k half-stage a workgroup stages A 8 KiB + B 16 KiB for 128 MFMAs where the 256 x 256 tile stages 32 KiB for 256 -- 1.5 x the LDS-DMA
// bytes per MFMA -- and a ring of three half-stages instead of four.
//
// Per half-step h (one barrier):   s_waitcnt vmcnt(6)  this wave's part of half-stage h has landed (h + 1 stays in flight)
//                                  s_barrier           everybody's has, and everybody has finished reading half-stage h - 1
//                                  LDS-DMA of half-stage h + 2 into the slot of h - 1;  ds_read the fragments of h (one register set)
//                                  s_waitcnt lgkmcnt(0);  32 MFMAs
// The phases of the two co-resident workgroups are not coupled: whichever has fragments issues MFMAs.
// LDS images, DMA addressing and the epilogue (fixed kinds of gemm_tile.h; the wave tile is gemm3's) are shared code: same k grouping inside every
// MFMA, same bits as the other variants.
#include <stdlib.h>
#include "gemm_half.h"

#define G9_BM 128
#define G9_BN 256
#define G9_NST 3
#define G9_FENCE() do { __builtin_amdgcn_sched_barrier(0); asm volatile("" ::: "memory"); } while (0)
// issue priority between the two co-resident workgroups' waves of a SIMD (measurement switch): 1 = the MFMA block runs at raised priority (the ping-pong
// kernels' setting), 2 = the MEMORY side of a step does (fragment reads, B loads, LDS-DMA go out between the other workgroup's MFMAs), 0 = none.
// Measured (gpurun r06_o, dwpk on the step's shapes): the three settings are within 1 % of each other; 0 is the default
#ifndef G9_PRIO
#define G9_PRIO 0
#endif
#ifndef G9_PIPE
#define G9_PIPE 1          // packed-B form, k-contiguous A: the rolling half-set pipeline (0: the plain one-set steps; A/B switch)
#endif
#define G9_PRIO_MFMA(x) do { if (G9_PRIO == 1) __builtin_amdgcn_s_setprio(x); } while (0)
#define G9_PRIO_MEM(x) do { if (G9_PRIO == 2) __builtin_amdgcn_s_setprio(x); } while (0)

template <bool AKS, bool BKS, int EPI = -1>
__global__ __launch_bounds__(256, 2) void gemm9_bf16_kernel(Gemm2Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NJ = 4, WN = 64;
  constexpr int A_SUB = G9_BM * 64, B_SUB = G9_BN * 64, SUB = A_SUB + B_SUB;
  constexpr int NIA = G9_BM / 64, NIB = G9_BN / 64, NEW = NIA + NIB;       // LDS-DMA instructions a wave issues per half-stage: 2 + 4

  int nwg = p.nbm * p.nbn;
  int id = xcd_remap(blockIdx.x, nwg);
  constexpr int GM = 8;                                  // 8 row tiles of 128 = the 1024-row raster group of the 256-row kernels
  int per_group = GM * p.nbn;
  int grp_ = id / per_group;
  int first_m = grp_ * GM;
  int gsz = min(p.nbm - first_m, GM);
  int in_g = id - grp_ * per_group;
  int tm = first_m + in_g % gsz, tn = in_g / gsz;
  int m0 = tm * G9_BM, n0 = tn * G9_BN;

  int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = lane_id();
  const int wn = wave;

  f32x4 acc[8][NJ];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 ra0[8], rb0[NJ];
  s16x4 la0[8], ha0[8], lb0[NJ], hb0[NJ];

  const int nh = (p.K + 31) >> 5;
  uint32_t aoff[NIA], boff[NIB];
  dma_setup<AKS, G9_BM, 4>(p.lda, m0, p.M, wave, aoff);
  dma_setup<BKS, G9_BN, 4>(p.ldb, n0, p.N, wave, boff);
#define DMA(H) do { char* b_ = smem + ((H) % G9_NST) * SUB;                                                             \
    dma_issue<AKS, G9_BM, 4>(p.A, p.lda, (H), p.K, b_, wave, aoff);                                                \
    dma_issue<BKS, G9_BN, 4>(p.B, p.ldb, (H), p.K, b_ + A_SUB, wave, boff); } while (0)
  const uint32_t smem_lds = (uint32_t)(uintptr_t)LDS_PTR(char, smem);
  const uint32_t lbA = AKS ? ks32_lane_base<G9_BM>(0) : 0u, lbB = BKS ? ks32_lane_base<G9_BN>(wn * WN) : 0u;
#define LOADF(H) do { const char* b_ = smem + ((H) % G9_NST) * SUB;                                                     \
    uint32_t ub_ = smem_lds + ((H) % G9_NST) * SUB;                                                                     \
    _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                                               \
      if (BKS) frag_ks32_asm<G9_BN>(lbB + ub_ + A_SUB, j, lb0[j], hb0[j]);                                         \
      else rb0[j] = frag_kc32(b_ + A_SUB, wn * WN + j * 16); }                                                     \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                                \
      if (AKS) frag_ks32_asm<G9_BM>(lbA + ub_, i, la0[i], ha0[i]);                                                 \
      else ra0[i] = frag_kc32(b_, i * 16); } } while (0)
#define MFMAS() do { G9_PRIO_MFMA(1);                                                                \
    bf16x8 fb_[NJ];                                                                                                \
    _Pragma("unroll") for (int j = 0; j < NJ; ++j) fb_[j] = BKS ? join_halves(lb0[j], hb0[j]) : rb0[j];            \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                                \
      bf16x8 fa_ = AKS ? join_halves(la0[i], ha0[i]) : ra0[i];                                                     \
      _Pragma("unroll") for (int j = 0; j < NJ; ++j) acc[i][j] = MFMA16(fb_[j], fa_, acc[i][j]); }                 \
    G9_PRIO_MFMA(0); } while (0)

#define DMAF(H) do { char* b_ = smem + ((H) % G9_NST) * SUB;                                                            \
    dma_full<AKS, G9_BM, 4>(p.A, p.lda, (H), b_, wave, aoff);                                                      \
    dma_full<BKS, G9_BN, 4>(p.B, p.ldb, (H), b_ + A_SUB, wave, boff); } while (0)
// one half-step: fragment reads FIRST (they land while the wave is held by its six LDS-DMA issues), then the DMA of h + 2
#define STEP(H, ISSUE, VM) do {                                                                                     \
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(VM) : "memory");                                                      \
    G9_FENCE(); __builtin_amdgcn_s_barrier(); G9_FENCE();                                                           \
    LOADF(H);                                                                                                       \
    G9_FENCE(); ISSUE; G9_FENCE();                                                                                  \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); G9_FENCE();                                                  \
    MFMAS(); } while (0)
  if ((p.K & 31) == 0 && nh >= 3) {          // whole half-stages: the steady state has no decision left (gemm3's round-5 lever), the last two steps are written out
    DMAF(0); DMAF(1);
    int h = 0;
#pragma unroll 1
    for (; h + 2 < nh; ++h) STEP(h, DMAF(h + 2), NEW);
    STEP(h, (void)0, NEW); ++h;
    STEP(h, (void)0, 0);
  } else {
    DMA(0);
    if (nh > 1) DMA(1);
#pragma unroll 1
    for (int h = 0; h < nh; ++h) {
      if (h + 1 < nh) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NEW) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      G9_FENCE(); __builtin_amdgcn_s_barrier(); G9_FENCE();
      LOADF(h);
      G9_FENCE();
      if (h + 2 < nh) DMA(h + 2);
      G9_FENCE(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); G9_FENCE();
      MFMAS();
    }
  }
#undef STEP
#undef DMAF
  G9_FENCE(); __builtin_amdgcn_s_barrier(); G9_FENCE();          // every wave is done with the ring: the epilogue stages through it
#undef DMA
#undef LOADF
#undef MFMAS

  float gate = 1.f;
  if (p.gate) gate = tanhf(bf2f(*p.gate));
  // ---- epilogue through LDS: gemm3's (wave tile 128 x 64: wave-private [64][64] f32 region, two passes of 64 rows)
  constexpr int ESTR = WN * 4, UNITS = WN / 4;
  char* er = smem + wave * (64 * ESTR);
  bool fast = ((p.ldc | p.ldres | p.ldaux | p.ldpre) & 7) == 0;
#define EPI_STAGE(PASS) do {                                                                                      \
    _Pragma("unroll") for (int i2 = 0; i2 < 4; ++i2)                                                               \
      _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                                           \
        int row = i2 * 16 + (lane & 15), u = j * 4 + (lane >> 4);                                                  \
        *(f32x4*)(er + row * ESTR + ((u ^ (row & (UNITS - 1))) << 4)) = acc[(PASS) * 4 + i2][j];                   \
      }                                                                                                            \
    __builtin_amdgcn_s_waitcnt(0xc07f); } while (0)
  const int em = m0, en = n0 + wn * WN;
  const int kind = EPI >= 0 ? EPI : epi_kind(p, fast);
  EpiPre<WN> pre0, pre1;
  bf16x8 biasv = epi_bias<WN>(p, lane, en, kind);
  epi_fetch<WN>(p, lane, em, en, kind, pre0);
  EPI_STAGE(0);
  epi_fetch<WN>(p, lane, em + 64, en, kind, pre1);
  if (kind != EK_GENERIC) epi_inputs_ready();
  if (EPI >= 0) epi_pass_fixed<WN, EPI < 0 ? 0 : EPI>(p, er, lane, em, en, gate, pre0, biasv);
  else epi_pass_kind<WN, 64, false>(p, er, lane, em, en, gate, fast, kind, pre0, biasv);
  __builtin_amdgcn_s_waitcnt(0xc07f);
  EPI_STAGE(1);
  if (EPI >= 0) epi_pass_fixed<WN, EPI < 0 ? 0 : EPI>(p, er, lane, em + 64, en, gate, pre1, biasv);
  else epi_pass_kind<WN, 64, false>(p, er, lane, em + 64, en, gate, fast, kind, pre1, biasv);
#undef EPI_STAGE
}

// ---- packed-B form (b_kstrided == 2: the pre-packed image of a frozen weight, unimp_pack_b_bf16): B never touches the LDS.  In the 128 x 256 tile the
// four waves own DISJOINT 64-column strips, so each fragment of the image is fetched by exactly one wave of the workgroup (in the 256 x 256 ping-pong tile
// both wave groups fetched every fragment): per half-step a wave issues 4 coalesced 1-KiB global loads for its B fragments of half-step h + 2 (three
// register sets) and 2 LDS-DMA instructions for the A half-stage h + 3 (ring of four 8-KiB half-stages), reads its 8 A fragments of h, and the LDS carries
// 32 KiB of fragment reads + 8 KiB of DMA writes per 128 MFMAs where the B-through-LDS form carries 48 + 24 (141 B / clk against the pipe's 128).
// vmcnt: a step issues B(h + 2) x 4 then A(h + 3) x 2; at the top of step h everything but A(h + 2), B(h + 1), A(h + 1) -- the 8 youngest -- must have
// landed.  Past the end of K the same instructions are issued at the last half-stage's addresses (into a ring slot nobody reads / registers nobody uses):
// the count stays uniform and the loop has no decision.  K % 32 == 0.
__device__ __forceinline__ void g9_frag_packed(const void* sbase, uint32_t voff, bf16x8& out) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(out) : "v"(voff), "s"(sbase) : "memory");
}

template <bool AKS, int EPI = -1>
__global__ __launch_bounds__(256, 2) void gemm9pk_bf16_kernel(Gemm2Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NJ = 4, WN = 64, NSTA = 4;
  constexpr int A_SUB = G9_BM * 64;
  constexpr int NIA = G9_BM / 64;

  int nwg = p.nbm * p.nbn;
  int id = xcd_remap(blockIdx.x, nwg);
  constexpr int GM = 8;
  int per_group = GM * p.nbn;
  int grp_ = id / per_group;
  int first_m = grp_ * GM;
  int gsz = min(p.nbm - first_m, GM);
  int in_g = id - grp_ * per_group;
  int tm = first_m + in_g % gsz, tn = in_g / gsz;
  int m0 = tm * G9_BM, n0 = tn * G9_BN;

  int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = lane_id();
  const int wn = wave;

  f32x4 acc[8][NJ];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 ra0[8], rb0[NJ], rb1[NJ], rb2[NJ];
  s16x4 la0[8], ha0[8];

  const int nh = p.K >> 5;
  uint32_t aoff[NIA];
  dma_setup<AKS, G9_BM, 4>(p.lda, m0, p.M, wave, aoff);
  const long pk_nh = nh;
  const char* pk_base = (const char*)p.B + ((long)((n0 + wn * WN) >> 4) * pk_nh) * 1024;
  const uint32_t pk_lane = (uint32_t)lane * 16u;
  const uint32_t smem_lds = (uint32_t)(uintptr_t)LDS_PTR(char, smem);
  const uint32_t lbA = AKS ? ks32_lane_base<G9_BM>(0) : 0u;
#define DMA_A(H) do { const int hh_ = min((H), nh - 1); dma_full<AKS, G9_BM, 4>(p.A, p.lda, hh_, smem + ((H) % NSTA) * A_SUB, wave, aoff); } while (0)
#define LOAD_B(S, H) do { const int hh_ = min((H), nh - 1);                                                          \
    _Pragma("unroll") for (int j = 0; j < NJ; ++j) g9_frag_packed(pk_base + ((long)j * pk_nh + hh_) * 1024, pk_lane, rb##S[j]); } while (0)
#define LOAD_A(H) do { const char* b_ = smem + ((H) % NSTA) * A_SUB; uint32_t ub_ = smem_lds + ((H) % NSTA) * A_SUB;      \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                                \
      if (AKS) frag_ks32_asm<G9_BM>(lbA + ub_, i, la0[i], ha0[i]);                                                 \
      else ra0[i] = frag_kc32(b_, i * 16); } } while (0)
#define MFMAS(S) do { G9_PRIO_MFMA(1);                                                               \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                                \
      bf16x8 fa_ = AKS ? join_halves(la0[i], ha0[i]) : ra0[i];                                                     \
      _Pragma("unroll") for (int j = 0; j < NJ; ++j) acc[i][j] = MFMA16(rb##S[j], fa_, acc[i][j]); }               \
    G9_PRIO_MFMA(0); } while (0)
// step H on B set SC, loading B(H + 2) into set SN (= the set of H - 1, consumed one step ago).  (A second A-fragment set -- the reads of H + 1 under the
// MFMAs of H -- was tried: 240 + registers, 12 - 144 bytes of scratch per lane; the two co-resident workgroups are what hides this wave's LDS latency.)
#define STEP(H, SC, SN) do {                                                                                        \
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                                                \
    G9_FENCE(); __builtin_amdgcn_s_barrier(); G9_FENCE();                                                           \
    G9_PRIO_MEM(1);                                                                                                 \
    LOAD_A(H);                                                                                                      \
    G9_FENCE(); LOAD_B(SN, (H) + 2); DMA_A((H) + 3); G9_FENCE();                                                    \
    G9_PRIO_MEM(0);                                                                                                 \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); G9_FENCE();                                                  \
    MFMAS(SC); } while (0)
  // virtual steps -3 .. -1: A(0) | B(0), A(1) | B(1), A(2)
  DMA_A(0); LOAD_B(0, 0); DMA_A(1); LOAD_B(1, 1); DMA_A(2);
  G9_FENCE();
  int h = 0;
#if G9_PIPE
  if constexpr (!AKS) {
    // ROLLING half-set software pipeline (k-contiguous A): the eight A fragments live in ONE register set whose halves turn over half a step apart --
    // rows 4 - 7 of step h are read under the MFMAs of rows 0 - 3, rows 0 - 3 of step h + 1 (whose registers those MFMAs just released) under the MFMAs of
    // rows 4 - 7, behind the step's one barrier; the B loads of h + 2 and the A half-stage h + 3 go out between the last two MFMA groups.  A wave never waits
    // for its own fragment reads with the matrix pipe empty.  Waits: A(h + 1) before the barrier (6 youngest in flight), B(h + 1) at the step's end (8).
#define LOAD_A_HALF(H, I0) do { const char* b_ = smem + ((H) % NSTA) * A_SUB;                                           \
      _Pragma("unroll") for (int i = (I0); i < (I0) + 4; ++i) ra0[i] = frag_kc32(b_, i * 16); } while (0)
#define MF_ROWS(S, I0, I1) do { _Pragma("unroll") for (int i = (I0); i < (I1); ++i) {                                  \
      _Pragma("unroll") for (int j = 0; j < NJ; ++j) acc[i][j] = MFMA16(rb##S[j], ra0[i], acc[i][j]); } } while (0)
#define PSTEP(H, SC, SN) do {                                                                                       \
      LOAD_A_HALF(H, 4); G9_FENCE();                                                                                \
      MF_ROWS(SC, 0, 4); G9_FENCE();                                                                                \
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                                              \
      G9_FENCE(); __builtin_amdgcn_s_barrier(); G9_FENCE();                                                         \
      LOAD_A_HALF((H) + 1, 0); G9_FENCE();                                                                          \
      MF_ROWS(SC, 4, 6); G9_FENCE();                                                                                \
      LOAD_B(SN, (H) + 2); DMA_A((H) + 3); G9_FENCE();                                                              \
      MF_ROWS(SC, 6, 8); G9_FENCE();                                                                                \
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); G9_FENCE(); } while (0)
    // (B three half-steps ahead -- a fourth register set -- measured the same: the B loads' latency is not what this loop waits for)
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");          // A(0), B(0) have landed
    G9_FENCE(); __builtin_amdgcn_s_barrier(); G9_FENCE();
    LOAD_A_HALF(0, 0); G9_FENCE();
#pragma unroll 1
    for (; h + 3 <= nh; h += 3) { PSTEP(h, 0, 2); PSTEP(h + 1, 1, 0); PSTEP(h + 2, 2, 1); }
    if (h < nh) { PSTEP(h, 0, 2); ++h; if (h < nh) { PSTEP(h, 1, 0); } }
#undef LOAD_A_HALF
#undef MF_ROWS
#undef PSTEP
  } else
#endif
  {
#pragma unroll 1
    for (; h + 3 <= nh; h += 3) { STEP(h, 0, 2); STEP(h + 1, 1, 0); STEP(h + 2, 2, 1); }
    if (h < nh) { STEP(h, 0, 2); ++h; if (h < nh) { STEP(h, 1, 0); } }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the clamped look-ahead loads and DMAs past the end of K
  G9_FENCE(); __builtin_amdgcn_s_barrier(); G9_FENCE();
#undef DMA_A
#undef LOAD_B
#undef LOAD_A
#undef MFMAS
#undef STEP

  float gate = 1.f;
  if (p.gate) gate = tanhf(bf2f(*p.gate));
  constexpr int ESTR = WN * 4, UNITS = WN / 4;
  char* er = smem + wave * (64 * ESTR);
  bool fast = ((p.ldc | p.ldres | p.ldaux | p.ldpre) & 7) == 0;
#define EPI_STAGE(PASS) do {                                                                                      \
    _Pragma("unroll") for (int i2 = 0; i2 < 4; ++i2)                                                               \
      _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                                           \
        int row = i2 * 16 + (lane & 15), u = j * 4 + (lane >> 4);                                                  \
        *(f32x4*)(er + row * ESTR + ((u ^ (row & (UNITS - 1))) << 4)) = acc[(PASS) * 4 + i2][j];                   \
      }                                                                                                            \
    __builtin_amdgcn_s_waitcnt(0xc07f); } while (0)
  const int em = m0, en = n0 + wn * WN;
  const int kind = EPI >= 0 ? EPI : epi_kind(p, fast);
  EpiPre<WN> pre0, pre1;
  bf16x8 biasv = epi_bias<WN>(p, lane, en, kind);
  epi_fetch<WN>(p, lane, em, en, kind, pre0);
  EPI_STAGE(0);
  epi_fetch<WN>(p, lane, em + 64, en, kind, pre1);
  if (kind != EK_GENERIC) epi_inputs_ready();
  if (EPI >= 0) epi_pass_fixed<WN, EPI < 0 ? 0 : EPI>(p, er, lane, em, en, gate, pre0, biasv);
  else epi_pass_kind<WN, 64, false>(p, er, lane, em, en, gate, fast, kind, pre0, biasv);
  __builtin_amdgcn_s_waitcnt(0xc07f);
  EPI_STAGE(1);
  if (EPI >= 0) epi_pass_fixed<WN, EPI < 0 ? 0 : EPI>(p, er, lane, em + 64, en, gate, pre1, biasv);
  else epi_pass_kind<WN, 64, false>(p, er, lane, em + 64, en, gate, fast, kind, pre1, biasv);
#undef EPI_STAGE
}

template <bool AKS, int EPI = -1>
static void launch9pk(const Gemm2Params& p, hipStream_t s) {
  static bool attr_set = false;
  constexpr size_t lds = 4 * 64 * 64 * 4;          // 64 KiB: the epilogue's four wave-private [64][64] f32 regions; the A ring (4 x 8 KiB) lives inside
  auto kern = gemm9pk_bf16_kernel<AKS, EPI>;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; }
  hipLaunchKernelGGL(kern, dim3(p.nbm * p.nbn), dim3(256), lds, s, p);
}

template <bool AKS, bool BKS, int EPI = -1>
static void launch9(const Gemm2Params& p, hipStream_t s) {
  static bool attr_set = false;
  constexpr size_t lds = G9_NST * (G9_BM * 64 + G9_BN * 64);          // 72 KiB; the epilogue's 4 x 16 KiB staging fits inside
  auto kern = gemm9_bf16_kernel<AKS, BKS, EPI>;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; }
  hipLaunchKernelGGL(kern, dim3(p.nbm * p.nbn), dim3(256), lds, s, p);
}

// Returns 1 if launched, 0 if the form is not served (rotary epilogue, packed B: the caller falls back).
extern "C" int unimp_gemm9_launch(const unimp_gemm_desc* d, void* stream) {
  if (d->rope_rot) return 0;
  if (d->b_kstrided == 2 && ((d->K & 31) || d->K < 96)) return 0;
  Gemm2Params p;
  p.A = (const bf16*)d->A; p.B = (const bf16*)d->B; p.C = d->C;
  p.M = d->M; p.N = d->N; p.K = d->K; p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc;
  p.bias = (const bf16*)d->bias; p.res = (const bf16*)d->res; p.ldres = d->ldres;
  p.aux = (const bf16*)d->aux; p.ldaux = d->ldaux; p.pre = (bf16*)d->pre; p.ldpre = d->ldpre;
  p.gate = (const bf16*)d->gate; p.alpha = d->alpha; p.act = d->act; p.dact = d->dact;
  p.out_f32 = d->out_f32; p.accumulate = d->accumulate; p.pre_deriv = d->pre_deriv; GEMM2_FILL_ROPE(p, d); p.ksplit = 0; p.gm = 0;
  p.nbm = (d->M + G9_BM - 1) / G9_BM;
  p.nbn = (d->N + G9_BN - 1) / G9_BN;
  hipStream_t s = (hipStream_t)stream;
  const int a = d->a_kstrided, b = d->b_kstrided;
  if (b == 2) {                           // pre-packed B image: B bypasses the LDS
    const int ek = a ? -1 : epi_kind_host(p);
    switch (ek) {
      case EK_PLAIN: launch9pk<false, EK_PLAIN>(p, s); return 1;
      case EK_ACT:   launch9pk<false, EK_ACT>(p, s); return 1;
      case EK_GELU2: launch9pk<false, EK_GELU2>(p, s); return 1;
      case EK_RES:   launch9pk<false, EK_RES>(p, s); return 1;
      case EK_AUX:   launch9pk<false, EK_AUX>(p, s); return 1;
      default: break;
    }
    if (a) launch9pk<true>(p, s); else launch9pk<false>(p, s);
    return 1;
  }
  if (!a) {                               // fixed-kind instantiations for a k-contiguous A (every forward and dX GEMM)
    const int ek = epi_kind_host(p);
#define L9F(K_) do { if (b) launch9<false, true, K_>(p, s); else launch9<false, false, K_>(p, s); return 1; } while (0)
    switch (ek) {
      case EK_PLAIN: L9F(EK_PLAIN);
      case EK_ACT:   L9F(EK_ACT);
      case EK_GELU2: L9F(EK_GELU2);
      case EK_RES:   L9F(EK_RES);
      case EK_AUX:   if (b) { launch9<false, true, EK_AUX>(p, s); return 1; } break;
      default: break;
    }
#undef L9F
  }
  if (!a && !b) launch9<false, false>(p, s); else if (!a && b) launch9<false, true>(p, s);
  else if (a && b) launch9<true, true>(p, s); else launch9<true, false>(p, s);
  return 1;
}
