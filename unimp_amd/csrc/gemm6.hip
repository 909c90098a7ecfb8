// Persistent form of the ping-pong GEMM (gemm3.hip, 256 x 256 tiles): one workgroup per CU loops over its tiles, and the
// LDS-DMA prologue of the NEXT tile (half-stages 0..2) is issued before the epilogue of the current one, so the cold-miss
// latency of a tile's first operands, and the gap in which the dispatcher replaces a finished workgroup, are hidden under
// the epilogue's stores.  Why it matters: s_memrealtime stamps (-DG3_STAMP) put 2.3-3.9 us of prologue and ~1.4 us of
// dispatch gap on every tile -- against 25 us of main loop at K = 1024 (the ViT's GEMMs) and 60 us at K = 2560.
// The ring keeps slots 0..2 for the incoming half-stages, so the epilogue stages the accumulators through slot 3 plus the
// 32 KiB the ring leaves free (160 KiB LDS in total): four 32-row sub-passes of 8 KiB per wave instead of two 64-row ones.
// Main loop, LDS images, DMA addressing: gemm3.hip / gemm_half.h.  Epilogue kinds: gemm_tile.h.
#include "gemm_half.h"

#ifdef G3X            // second build of this file with other schedule switches, under its own symbols (Makefile: gemm6x.o)
#define gemm6_bf16_kernel gemm6x_bf16_kernel
#define unimp_gemm6_launch unimp_gemm6x_launch
#define launch6 launch6x
#endif

#define G3_BM 256
#define G3_NST 4
#ifdef G3_NO_PRIO
#define G3_PRIO(x) do {} while (0)
#else
#define G3_PRIO(x) __builtin_amdgcn_s_setprio(x)
#endif
#define G3_FENCE() do { __builtin_amdgcn_sched_barrier(0); asm volatile("" ::: "memory"); } while (0)
#define G3_BARRIER() do { G3_FENCE(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); G3_FENCE(); } while (0)


template <bool AKS, bool BKS, bool ROPE = false>
__global__ __launch_bounds__(512, 2) void gemm6_bf16_kernel(Gemm2Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BN = 256, NJ = BN / 64, WN = BN / 4;
  constexpr int A_SUB = G3_BM * 64, B_SUB = BN * 64, SUB = A_SUB + B_SUB;
  constexpr int NEW = G3_BM / 128 + BN / 128;          // LDS-DMA instructions a wave issues per half-step
  constexpr int GM = 4;
  constexpr int PD = G3_NST - 1;                        // prefetch distance in half-steps

  const int ntiles = p.nbm * p.nbn;
  const int per_group = GM * p.nbn;
  int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = lane_id();
  int wm = wave >> 2, wn = wave & 3;                    // wm doubles as the ping-pong group
  const int nh = (p.K + 31) >> 5;

  uint32_t aoff[G3_BM / 128], boff[BN / 128];
  int m0, n0;
  // tile t of this workgroup -> (m0, n0): blocks with equal blockIdx % 8 share an XCD and walk one contiguous range of
  // logical ids (xcd_remap), 4-row groups of tiles inside it -- the mapping of gemm3.hip, 256 tiles at a time
#define TILE(T) do { int id_ = xcd_remap((T), ntiles); int grp_ = id_ / per_group, first_m = grp_ * GM;                \
    int gsz = min(p.nbm - first_m, GM), in_g = id_ - grp_ * per_group;                                                 \
    m0 = (first_m + in_g % gsz) * G3_BM; n0 = (in_g / gsz) * BN;                                                       \
    dma_setup<AKS, G3_BM>(p.lda, m0, p.M, wave, aoff); dma_setup<BKS, BN>(p.ldb, n0, p.N, wave, boff); } while (0)
#define DMA(H) do { char* b_ = smem + ((H) % G3_NST) * SUB;                                                             \
    dma_issue<AKS, G3_BM>(p.A, p.lda, (H), p.K, b_, wave, aoff);                                                   \
    dma_issue<BKS, BN>(p.B, p.ldb, (H), p.K, b_ + A_SUB, wave, boff); } while (0)
  const uint32_t smem_lds = (uint32_t)(uintptr_t)LDS_PTR(char, smem);
  const uint32_t lbA = AKS ? ks32_lane_base<G3_BM>(wm * 128) : 0u, lbB = BKS ? ks32_lane_base<BN>(wn * WN) : 0u;
#define LOADF(S, H) do { const char* b_ = smem + ((H) % G3_NST) * SUB;                                                  \
    uint32_t ub_ = smem_lds + ((H) % G3_NST) * SUB;                                                                     \
    _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                                               \
      if (BKS) frag_ks32_asm<BN>(lbB + ub_ + A_SUB, j, lb##S[j], hb##S[j]);                                        \
      else rb##S[j] = frag_kc32(b_ + A_SUB, wn * WN + j * 16); }                                                   \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                                \
      if (AKS) frag_ks32_asm<G3_BM>(lbA + ub_, i, la##S[i], ha##S[i]);                                             \
      else ra##S[i] = frag_kc32(b_, wm * 128 + i * 16); } } while (0)
#define MFMAS(S) do { G3_PRIO(1);                                                                                  \
    bf16x8 fb_[NJ];                                                                                                \
    _Pragma("unroll") for (int j = 0; j < NJ; ++j) fb_[j] = BKS ? join_halves(lb##S[j], hb##S[j]) : rb##S[j];      \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                                \
      bf16x8 fa_ = AKS ? join_halves(la##S[i], ha##S[i]) : ra##S[i];                                               \
      _Pragma("unroll") for (int j = 0; j < NJ; ++j) acc[i][j] = MFMA16(fb_[j], fa_, acc[i][j]); }                 \
    G3_PRIO(0); } while (0)
#ifndef G3_ONESET
// one half-step: L phase (prefetch h+3, fragments of h+1 -> RN), barrier, C phase (MFMA on RC), barrier
#define HALF_STEP(H, SC, SN) do {                                                                                  \
    if ((H) + PD < nh) DMA((H) + PD);                                                                              \
    if ((H) + 1 < nh) LOADF(SN, (H) + 1);                                                                          \
    /* half-step H+2 must have landed; the (PD-2) younger ones may stay in flight (fewer near the end of K) */     \
    if ((H) + PD < nh) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((PD - 2) * NEW) : "memory");                      \
    else if ((H) + PD - 1 < nh && PD >= 4) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((PD - 3) * NEW) : "memory");  \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                          \
    G3_BARRIER();                                                                                                  \
    MFMAS(SC);                                                                                                     \
    G3_BARRIER(); } while (0)
#else
// one fragment register set (gemm3.hip, G3_ONESET): L(h) reads the fragments of h itself, reads before the DMA issue, h+2 / h+3 in flight
#define HALF_STEP(H, SC, SN) do {                                                                                  \
    LOADF(0, (H));                                                                                                 \
    if ((H) + PD < nh) { DMA((H) + PD); asm volatile("s_waitcnt vmcnt(%0)" :: "n"((PD - 1) * NEW) : "memory"); }   \
    else if ((H) + PD - 1 < nh) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((PD - 2) * NEW) : "memory");             \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                          \
    G3_BARRIER();                                                                                                  \
    MFMAS(0);                                                                                                      \
    G3_BARRIER(); } while (0)
#endif


  int t = blockIdx.x;
  TILE(t);
  for (int h0 = 0; h0 < PD && h0 < nh; ++h0) DMA(h0);
  for (;;) {
    // accumulators and fragment registers are scoped to one tile: declared outside the loop the compiler has to assume the
    // fragments live across the back-edge (through the whole epilogue) and spills
    f32x4 acc[8][NJ];
#ifndef G3_ONESET
    bf16x8 ra0[8], rb0[NJ], ra1[8], rb1[NJ];                  // k-contiguous operands: whole fragments
    s16x4 la0[8], ha0[8], lb0[NJ], hb0[NJ], la1[8], ha1[8], lb1[NJ], hb1[NJ];   // k-strided operands: two tr halves
#else
    bf16x8 ra0[8], rb0[NJ];
    s16x4 la0[8], ha0[8], lb0[NJ], hb0[NJ];
#endif
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this tile's first half-stages (and the previous tile's stores)
    G3_BARRIER();                                           // ... of every wave; everybody has left the previous epilogue
#ifndef G3_ONESET
    LOADF(0, 0);
#endif
    if (wm == 1) G3_BARRIER();                            // group B runs one phase behind group A
    {
      int h = 0;
#if defined(G3_ONESET) && !defined(G3_NO_PEEL)
      // round 5 (gemm3.hip): steady-state half-steps without their per-half-step decisions -- no scalar branch in the L phase; same bits
#define DMA_F(H) do { char* b_ = smem + ((H) % G3_NST) * SUB;                                                        \
        dma_full<AKS, G3_BM>(p.A, p.lda, (H), b_, wave, aoff); dma_full<BKS, BN>(p.B, p.ldb, (H), b_ + A_SUB, wave, boff); } while (0)
#pragma unroll 1
      for (; h + PD + 2 < nh; h += 2) {
        LOADF(0, h); DMA_F(h + PD); asm volatile("s_waitcnt vmcnt(%0)" :: "n"((PD - 1) * NEW) : "memory");
        G3_BARRIER(); MFMAS(0); G3_BARRIER();
        LOADF(0, h + 1); DMA_F(h + 1 + PD); asm volatile("s_waitcnt vmcnt(%0)" :: "n"((PD - 1) * NEW) : "memory");
        G3_BARRIER(); MFMAS(0); G3_BARRIER();
      }
#undef DMA_F
#endif
      for (; h < nh; h += 2) {
        HALF_STEP(h, 0, 1);
        if (h + 1 < nh) HALF_STEP(h + 1, 1, 0);
      }
    }
    if (wm == 0) G3_BARRIER();                            // equalise the barrier count; all LDS reads are complete

    // ---- epilogue of this tile, with the next tile's prologue issued as soon as the epilogue's own inputs have landed
    const int em = m0 + wm * 128, en = n0 + wn * WN;
    t += gridDim.x;
    const bool more = t < ntiles;
    constexpr int ESTR = WN * 4, UNITS = WN / 4;
    int el = lane;                                          // laundered: keeps the epilogue's per-lane address math inside the tile
    asm volatile("" : "+v"(el));                            // loop (hoisted out of it, it is live through the main loop and spills)
    char* er = smem + PD * SUB + wave * (32 * ESTR);        // slot 3 + the spare 32 KiB: 8 KiB per wave
    float gate = 1.f;
    if (p.gate) gate = tanhf(bf2f(*p.gate));
    bool fast = ((p.ldc | p.ldres | p.ldaux | p.ldpre) & 7) == 0;      // N may be ragged: only its last 8-column group is element-wise
#define EPI_STAGE(SP) do {                                                                                         \
    _Pragma("unroll") for (int i2 = 0; i2 < 2; ++i2)                                                               \
      _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                                           \
        int row = i2 * 16 + (el & 15), u = j * 4 + (el >> 4);                                                  \
        *(f32x4*)(er + row * ESTR + ((u ^ (row & (UNITS - 1))) << 4)) = acc[(SP) * 2 + i2][j];                     \
      }                                                                                                            \
    __builtin_amdgcn_s_waitcnt(0xc07f); } while (0)
    const int kind = ROPE ? EK_ROPE : epi_kind(p, fast);
    EpiPre<WN, 32> q0, q1, q2, q3;
    bf16x8 biasv = epi_bias<WN>(p, el, en, kind);
    epi_fetch<WN, 32>(p, el, em, en, kind, q0);
    epi_fetch<WN, 32>(p, el, em + 32, en, kind, q1);
    EPI_STAGE(0);
    epi_fetch<WN, 32>(p, el, em + 64, en, kind, q2);
    epi_fetch<WN, 32>(p, el, em + 96, en, kind, q3);
    if (kind == EK_AUX || kind == EK_RES || (kind != EK_GENERIC && p.bias)) epi_inputs_ready();
    // the epilogue's inputs are in registers; now the next tile's first half-stages go out and land under the stores below
    // (issued before the inputs, the single wait above would also wait for these cold misses)
    if (more) {
      TILE(t);
      for (int h0 = 0; h0 < PD && h0 < nh; ++h0) DMA(h0);
    }
    epi_pass_kind<WN, 32, ROPE>(p, er, el, em, en, gate, fast, kind, q0, biasv);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    EPI_STAGE(1);
    epi_pass_kind<WN, 32, ROPE>(p, er, el, em + 32, en, gate, fast, kind, q1, biasv);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    EPI_STAGE(2);
    epi_pass_kind<WN, 32, ROPE>(p, er, el, em + 64, en, gate, fast, kind, q2, biasv);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    EPI_STAGE(3);
    epi_pass_kind<WN, 32, ROPE>(p, er, el, em + 96, en, gate, fast, kind, q3, biasv);
    __builtin_amdgcn_s_waitcnt(0xc07f);
#undef EPI_STAGE
    if (!more) break;
  }
#undef DMA
#undef LOADF
#undef MFMAS
#undef HALF_STEP
#undef TILE
}

template <bool AKS, bool BKS, bool ROPE = false>
static void launch6(const Gemm2Params& p, hipStream_t s) {
  static bool attr_set = false;
  static int ncu = 0;
  constexpr size_t lds = G3_NST * (G3_BM * 64 + 256 * 64) + 32768;
  auto kern = gemm6_bf16_kernel<AKS, BKS, ROPE>;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; }
  if (!ncu) { int dev = 0; hipDeviceProp_t pr; (void)hipGetDevice(&dev); (void)hipGetDeviceProperties(&pr, dev); ncu = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256; }
  int ntiles = p.nbm * p.nbn;
  hipLaunchKernelGGL(kern, dim3(ntiles < ncu ? ntiles : ncu), dim3(512), lds, s, p);
}

extern "C" int unimp_gemm6_launch(const unimp_gemm_desc* d, void* stream) {
  Gemm2Params p;
  p.A = (const bf16*)d->A; p.B = (const bf16*)d->B; p.C = d->C;
  p.M = d->M; p.N = d->N; p.K = d->K; p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc;
  p.bias = (const bf16*)d->bias; p.res = (const bf16*)d->res; p.ldres = d->ldres;
  p.aux = (const bf16*)d->aux; p.ldaux = d->ldaux; p.pre = (bf16*)d->pre; p.ldpre = d->ldpre;
  p.gate = (const bf16*)d->gate; p.alpha = d->alpha; p.act = d->act; p.dact = d->dact;
  p.out_f32 = d->out_f32; p.accumulate = d->accumulate; p.pre_deriv = d->pre_deriv; GEMM2_FILL_ROPE(p, d); p.ksplit = 0; p.gm = 0;
  p.nbm = (d->M + G3_BM - 1) / G3_BM;
  p.nbn = (d->N + 255) / 256;
  hipStream_t s = (hipStream_t)stream;
  int a = d->a_kstrided, b = d->b_kstrided;
  if (p.rope_rot) { if (b == 1) launch6<false, true, true>(p, s); else launch6<false, false, true>(p, s); return 1; }   // host-validated: k-contiguous A
  if (!a && !b) launch6<false, false>(p, s); else if (!a && b) launch6<false, true>(p, s);
  else if (a && b) launch6<true, true>(p, s); else launch6<true, false>(p, s);
  return 1;
}
