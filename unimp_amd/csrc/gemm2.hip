// Large-tile bf16 MFMA GEMM for gfx950: 256 x BN x 64 tiles (BN = 256 or 128), 512 threads = 8 waves (2 x 4),
// one workgroup per CU (up to 128 KiB of the 160 KiB LDS), operands staged global -> LDS by LDS-DMA
// (global_load_lds_dwordx4, 1 KiB per wave-instruction, no VGPR round trip), double-buffered, one barrier per K-tile:
// the next tile's DMA is in flight while the current tile's 64 (BN=256) MFMAs per wave run.
//
// LDS-DMA writes lane-linear (wave-uniform base + lane*16), so the bank-conflict swizzles of common.h are applied on
// the per-lane SOURCE address (cdna guide rule 21: linear destination + swizzled source + the same swizzle on the read).
// Out-of-range chunks (M / N / K edges) are sourced from a 16-byte zero constant, so no predication exists anywhere.
// Same three operand forms as gemm.hip: k-contiguous images are read with ds_read_b128, k-strided images with the
// transposing ds_read_b64_tr_b16, so forward / dX / dW all run on this kernel without transposed copies.
//
// Epilogue: accumulators go through LDS (the operand buffers are dead by then) and come back row-major, so every
// lane handles 8 consecutive columns of one row: 16-byte loads of bias / aux / residual and 16-byte bf16 stores,
// 128 contiguous bytes per row segment, in a rolled loop (small code, no runtime-indexed accumulator arrays).
#include "gemm_tile.h"

// ---- images for a tile of ROWS rows (m or n) x 64 k ---------------------------------------------------------
// KC: [ROWS][64] bf16, 128-B rows, 16-B chunk index XOR ((row>>1)&7)            (kc_off of common.h)
// KS: [64][ROWS] bf16, 2*ROWS-B k-rows, 32-B granule index XOR ks_h(krow)
template <int ROWS>
__device__ __forceinline__ int ks2_off(int krow, int col) {
  return krow * (ROWS * 2) + ((((col >> 4) ^ ks_h(krow)) << 5) | ((col & 15) << 1));
}
template <int ROWS>
__device__ __forceinline__ bf16x8 frag_ks2(const char* tile, int r0, int kk) {
  int l = lane_id();
  int g = l >> 4, q = (l >> 2) & 3, p = l & 3;
  int kr = kk * 32 + g * 8 + q;
  int col = r0 + 4 * p;
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, tile + ks2_off<ROWS>(kr, col)));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, tile + ks2_off<ROWS>(kr + 4, col)));
  union { struct { s16x4 a, b; } s; bf16x8 v; } u;
  u.s.a = lo; u.s.b = hi;
  return u.v;
}

template <bool KS, int ROWS>
__device__ __forceinline__ void stage(const bf16* __restrict__ X, long ld, int r0, int k0, int R, int K, char* img) {
  constexpr int NI = ROWS / 64;                 // wave-instructions per wave (8 waves, 1 KiB each)
  int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = lane_id();
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    int slot0 = (wave * NI + i) * 64;           // 16-B slot index of lane 0
    const bf16* src;
    if (!KS) {
      int row = (slot0 >> 3) + (l >> 3), s = l & 7;
      int c = s ^ ((row >> 1) & 7);
      int gr = r0 + row, gk = k0 + c * 8;
      src = (gr < R && gk < K) ? X + (long)gr * ld + gk : (const bf16*)g_zero16;
    } else {
      constexpr int SPR = ROWS / 8;             // slots per k-row
      int krow = (slot0 + l) / SPR, s = (slot0 + l) % SPR;
      int col = (((s >> 1) ^ ks_h(krow)) << 4) | ((s & 1) << 3);
      int gk = k0 + krow, gr = r0 + col;
      src = (gk < K && gr < R) ? X + (long)gk * ld + gr : (const bf16*)g_zero16;
    }
    glds16(src, img + slot0 * 16);
  }
}

#define G2_BM 256
#define G2_BK 64

template <bool AKS, bool BKS, int BN>
__global__ __launch_bounds__(512, 2) void gemm2_bf16_kernel(Gemm2Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NJ = BN / 64;                      // 16-col MFMA tiles per wave along n (wave tile 128 x BN/4)
  constexpr int A_BYTES = G2_BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int WN = BN / 4;                       // wave tile width

  int nwg = p.nbm * p.nbn;
  int id = xcd_remap(blockIdx.x, nwg);
  constexpr int GM = 4;
  int per_group = GM * p.nbn;
  int grp = id / per_group;
  int first_m = grp * GM;
  int gsz = min(p.nbm - first_m, GM);
  int in_g = id - grp * per_group;
  int tm = first_m + in_g % gsz, tn = in_g / gsz;
  int m0 = tm * G2_BM, n0 = tn * BN;

  int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = lane_id();
  int wm = wave >> 2, wn = wave & 3;

  f32x4 acc[8][NJ];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Main loop, software-pipelined by HALF a K-step with two register sets of fragments (P, Q):
  //   half 0 of step k : MFMA(P = frags(k, kk0))  ||  ds_read frags(k, kk1) -> Q                (from buffer k&1)
  //   -- s_waitcnt vmcnt(0) [DMA of stage k+1, issued one step ago]; barrier --
  //   half 1 of step k : issue DMA of stage k+2 into buffer k&1 (now dead)
  //                      MFMA(Q)                  ||  ds_read frags(k+1, kk0) -> P               (from buffer (k+1)&1)
  // LDS read latency is always covered by 32 (BN=256) MFMAs and the DMA is in flight for a whole K-step.
  int nk = (p.K + G2_BK - 1) / G2_BK;
  bf16x8 pa[8], pb[NJ], qa[8], qb[NJ];
#define LOAD_FRAGS(FA, FB, BUF, KK)                                                                                   \
  do {                                                                                                                \
    _Pragma("unroll") for (int j = 0; j < NJ; ++j)                                                                    \
      FB[j] = BKS ? frag_ks2<BN>((BUF) + A_BYTES, wn * WN + j * 16, KK) : frag_kc((BUF) + A_BYTES, wn * WN + j * 16, KK); \
    _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                                     \
      FA[i] = AKS ? frag_ks2<G2_BM>((BUF), wm * 128 + i * 16, KK) : frag_kc((BUF), wm * 128 + i * 16, KK);            \
  } while (0)
#define MFMA_ALL(FA, FB)                                                                                              \
  do {                                                                                                                \
    _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                                     \
      _Pragma("unroll") for (int j = 0; j < NJ; ++j) acc[i][j] = MFMA16(FB[j], FA[i], acc[i][j]);                     \
  } while (0)

  stage<AKS, G2_BM>(p.A, p.lda, m0, 0, p.M, p.K, smem);
  stage<BKS, BN>(p.B, p.ldb, n0, 0, p.N, p.K, smem + A_BYTES);
  if (nk > 1) {
    stage<AKS, G2_BM>(p.A, p.lda, m0, G2_BK, p.M, p.K, smem + STAGE);
    stage<BKS, BN>(p.B, p.ldb, n0, G2_BK, p.N, p.K, smem + STAGE + A_BYTES);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  LOAD_FRAGS(pa, pb, smem, 0);

  for (int kt = 0; kt < nk; ++kt) {
    char* cur = smem + (kt & 1) * STAGE;
    char* nxt = smem + ((kt + 1) & 1) * STAGE;
    LOAD_FRAGS(qa, qb, cur, 1);
    MFMA_ALL(pa, pb);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's share of stage kt+1 has landed
    __syncthreads();                                      // everyone is done reading `cur`; stage kt+1 visible to all
    if (kt + 2 < nk) {
      stage<AKS, G2_BM>(p.A, p.lda, m0, (kt + 2) * G2_BK, p.M, p.K, cur);
      stage<BKS, BN>(p.B, p.ldb, n0, (kt + 2) * G2_BK, p.N, p.K, cur + A_BYTES);
    }
    if (kt + 1 < nk) LOAD_FRAGS(pa, pb, nxt, 0);
    MFMA_ALL(qa, qb);
  }
#undef LOAD_FRAGS
#undef MFMA_ALL
  __syncthreads();                                        // all waves done with the operand buffers

  // ---- epilogue through LDS: wave-private region [64 rows][WN cols] f32, 16-B units XOR-swizzled by row
  constexpr int ESTR = WN * 4, UNITS = WN / 4;
  char* er = smem + wave * (64 * ESTR);
  float gate = 1.f;
  if (p.gate) gate = tanhf(bf2f(*p.gate));
  bool fast = ((p.N & 7) == 0) && (((p.ldc | p.ldres | p.ldaux | p.ldpre) & 7) == 0);
  constexpr int LPR = WN / 8;                       // lanes per row on the way out
  constexpr int RPI = 64 / LPR;                     // rows per wave-instruction
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    // lane holds acc[i][j][r] = C[row i*16 + (lane&15)][col j*16 + (lane>>4)*4 + r]
#pragma unroll
    for (int i2 = 0; i2 < 4; ++i2)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        int row = i2 * 16 + (lane & 15), u = j * 4 + (lane >> 4);
        *(f32x4*)(er + row * ESTR + ((u ^ (row & (UNITS - 1))) << 4)) = acc[pass * 4 + i2][j];
      }
    __builtin_amdgcn_s_waitcnt(0xc07f);             // lgkmcnt(0): the wave's own LDS writes have landed
    for (int it = 0; it < 64 / RPI; ++it) {
      int row = it * RPI + lane / LPR, cg = lane % LPR;
      int m = m0 + wm * 128 + pass * 64 + row, n = n0 + wn * WN + cg * 8;
      int sw = row & (UNITS - 1);
      f32x4 x0 = *(const f32x4*)(er + row * ESTR + (((2 * cg) ^ sw) << 4));
      f32x4 x1 = *(const f32x4*)(er + row * ESTR + (((2 * cg + 1) ^ sw) << 4));
      if (m < p.M && n < p.N) {
        float v[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
        if (fast) epi8<true>(p, v, m, n, gate); else epi8<false>(p, v, m, n, gate);
      }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);             // reads done before the next pass overwrites the region
  }
}

template <bool AKS, bool BKS, int BN>
static void launch2(const Gemm2Params& p, hipStream_t s) {
  static bool attr_set = false;
  constexpr size_t lds = 2 * (G2_BM * 128 + BN * 128);
  auto kern = gemm2_bf16_kernel<AKS, BKS, BN>;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; }
  hipLaunchKernelGGL(kern, dim3(p.nbm * p.nbn), dim3(512), lds, s, p);
}

// bn: 256 or 128 = tile width.  Returns 1 if launched.
extern "C" int unimp_gemm2_launch(const unimp_gemm_desc* d, int bn, void* stream) {
  Gemm2Params p;
  p.A = (const bf16*)d->A; p.B = (const bf16*)d->B; p.C = d->C;
  p.M = d->M; p.N = d->N; p.K = d->K; p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc;
  p.bias = (const bf16*)d->bias; p.res = (const bf16*)d->res; p.ldres = d->ldres;
  p.aux = (const bf16*)d->aux; p.ldaux = d->ldaux; p.pre = (bf16*)d->pre; p.ldpre = d->ldpre;
  p.gate = (const bf16*)d->gate; p.alpha = d->alpha; p.act = d->act; p.dact = d->dact;
  p.out_f32 = d->out_f32; p.accumulate = d->accumulate; p.pre_deriv = d->pre_deriv; GEMM2_FILL_ROPE(p, d); p.ksplit = 0; p.gm = 0;
  p.nbm = (d->M + G2_BM - 1) / G2_BM;
  p.nbn = (d->N + bn - 1) / bn;
  hipStream_t s = (hipStream_t)stream;
  int a = d->a_kstrided, b = d->b_kstrided;
#define L2(AK, BK_) do { if (bn == 256) launch2<AK, BK_, 256>(p, s); else launch2<AK, BK_, 128>(p, s); } while (0)
  if (!a && !b) L2(false, false); else if (!a && b) L2(false, true); else if (a && b) L2(true, true); else L2(true, false);
#undef L2
  return 1;
}
