// bf16 MFMA GEMM with fused epilogues for gfx950 (CDNA4, wave64).
//
//   C[M,N] = epilogue( alpha * sum_k A(m,k) * B(n,k) )
//
// Operand storage (each operand independently):
//   KC ("k-contiguous"): X(r,k) = X[r*ld + k]     -- activations [M,K], nn.Linear weights [N,K]
//   KS ("k-strided")   : X(r,k) = X[k*ld + r]     -- dX: W stored [N_out=k][K_in=r]; dW: dY,X stored [tokens=k][feat=r]
// so forward  y = x W^T      is (KC,KC),
//    dX      dx = dy W       is (KC,KS),
//    dW      dW = dy^T x     is (KS,KS);  no transposed copies are ever materialised.
//
// Tile 128x128x64, 256 threads = 4 waves (2x2), 64x64 per wave as 4x4 v_mfma_f32_16x16x32_bf16.
// Global->register->LDS staging (issue loads before the MFMA phase, write after: T14), two LDS
// stages, one barrier per K-tile.  KC tiles are read with ds_read_b128 from an XOR-swizzled image,
// KS tiles with ds_read_b64_tr_b16 (hardware transpose) from a granule-swizzled image; both
// conflict-free (see common.h).  The MFMA is issued with the operands swapped (D = B_frag x A_frag)
// so each lane ends up with 4 CONSECUTIVE n of one m: 8-byte bf16 / 16-byte f32 epilogue accesses.
// Block ids are remapped XCD-contiguously and then swept in GROUP_M-row groups for L2 reuse.
#include <stdlib.h>
#include "common.h"
#include <cstdlib>
#include "unimp_hip.h"

struct GemmParams {
  const bf16* A; const bf16* B; void* C;
  int M, N, K;
  long lda, ldb, ldc;
  const bf16* bias;                 // [N] or null
  const bf16* res;  long ldres;     // [M,N] or null: added last
  const bf16* aux;  long ldaux;     // [M,N] or null: v *= act'(aux)   (dact)
  bf16* pre;        long ldpre;     // [M,N] or null: receives v before act/gate/res
  const bf16* gate;                 // device scalar or null: v *= tanh(*gate)
  float alpha;
  int act, dact, out_f32, accumulate, pre_deriv;
  int nbm, nbn;
  int ksplit;                       // > 0: blockIdx.y handles k in [y*ksplit, (y+1)*ksplit) and writes f32 slab y of C
  const bf16* ln_gamma; const bf16* ln_beta; float ln_eps;   // skinny2 only: A rows are layer-normalised on the fly (ln_gamma != null)
};

#define BM 128
#define BN 128
#define BK 64
#define GROUP_M 8
#define STAGE_BYTES 32768          // A tile 16 KiB + B tile 16 KiB

template <bool KS>
__device__ __forceinline__ void g2r(const bf16* __restrict__ X, long ld, int r0, int k0, int R, int K,
                                    u32x4 (&v)[4]) {
  int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int q = tid + 256 * i;
    if (!KS) {
      int row = q >> 3, c = q & 7;
      int gr = r0 + row, gk = k0 + c * 8;
      if (gr < R && gk < K) v[i] = *(const u32x4*)(X + (long)gr * ld + gk);
      else v[i] = u32x4{0, 0, 0, 0};
    } else {
      int kr = q >> 4, c = q & 15;
      int gk = k0 + kr, gr = r0 + c * 8;
      if (gk < K && gr < R) v[i] = *(const u32x4*)(X + (long)gk * ld + gr);
      else v[i] = u32x4{0, 0, 0, 0};
    }
  }
}

template <bool KS>
__device__ __forceinline__ void r2s(char* tile, const u32x4 (&v)[4]) {
  int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int q = tid + 256 * i;
    int off;
    if (!KS) { int row = q >> 3, c = q & 7; off = kc_off(row, c); }
    else     { int kr = q >> 4, c = q & 15; off = ks_off(kr, c * 8); }
    *(u32x4*)(tile + off) = v[i];
  }
}

// one 16x16 accumulator tile: 4 consecutive n of row m per lane.  FAST: N % 4 == 0 and every ld % 4 == 0, so the
// 4 columns are all valid and 8-byte (bf16) / 16-byte (f32) aligned; otherwise per-element with bounds checks.
template <bool FAST>
__device__ __forceinline__ void epi_tile(const GemmParams& p, f32x4 a, int m, int n, float gate) {
  if (m >= p.M || n >= p.N) return;
  int nv = FAST ? 4 : min(4, p.N - n);
  float v[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = mul_rn(a[r], p.alpha);
  if (p.bias) {
    if (FAST) { bf16x4 b = *(const bf16x4*)(p.bias + n);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = add_rn(v[r], bf2f(b[r])); }
    else { for (int r = 0; r < nv; ++r) v[r] = add_rn(v[r], bf2f(p.bias[n + r])); }
  }
  if (p.pre && p.pre_deriv) {
    float dv[4];
    act_fwd_deriv_n<4>(p.act, v, dv);
    if (p.pre_deriv == 2) {                     // derivative as uint8 (common.h DERIV_U8): ldpre counts bytes
      uint8_t* d8 = (uint8_t*)p.pre + (long)m * p.ldpre + n;
      uint32_t w = deriv_u8_pack4(dv[0], dv[1], dv[2], dv[3]);
      if (FAST) *(uint32_t*)d8 = w;
      else { for (int r = 0; r < nv; ++r) d8[r] = (uint8_t)((w >> (8 * r)) & 0xffu); }
    } else {
    bf16* d = p.pre + (long)m * p.ldpre + n;
    if (FAST) { bf16x4 o = {f2bf(dv[0]), f2bf(dv[1]), f2bf(dv[2]), f2bf(dv[3])}; *(bf16x4*)d = o; }
    else { for (int r = 0; r < nv; ++r) d[r] = f2bf(dv[r]); }
    }
  } else {
    if (p.pre) {
      bf16* d = p.pre + (long)m * p.ldpre + n;
      if (FAST) { bf16x4 o = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])}; *(bf16x4*)d = o; }
      else { for (int r = 0; r < nv; ++r) d[r] = f2bf(v[r]); }
    }
    if (p.act) act_fwd_n<4>(p.act, v);
  }
  if (p.aux && p.dact == ACT_DERIV_U8) {
    const uint8_t* s8 = (const uint8_t*)p.aux + (long)m * p.ldaux + n;
    if (FAST) { uint32_t w = *(const uint32_t*)s8;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] *= deriv_u8_get(w, r); }
    else { for (int r = 0; r < nv; ++r) v[r] *= deriv_u8_get(s8[r], 0); }
  } else if (p.aux) {
    const bf16* s = p.aux + (long)m * p.ldaux + n;
    if (FAST) { bf16x4 x = *(const bf16x4*)s;
      float xf[4] = {bf2f(x[0]), bf2f(x[1]), bf2f(x[2]), bf2f(x[3])};
      act_bwd_mul_n<4>(p.dact, v, xf); }
    else { for (int r = 0; r < nv; ++r) v[r] *= act_bwd(p.dact, bf2f(s[r])); }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = mul_rn(v[r], gate);
  if (p.res) {
    const bf16* s = p.res + (long)m * p.ldres + n;
    if (FAST) { bf16x4 x = *(const bf16x4*)s;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = add_rn(v[r], bf2f(x[r])); }
    else { for (int r = 0; r < nv; ++r) v[r] = add_rn(v[r], bf2f(s[r])); }
  }
  if (p.out_f32) {
    float* d = (float*)p.C + (long)m * p.ldc + n;
    if (FAST) {
      f32x4 o = {v[0], v[1], v[2], v[3]};
      if (p.accumulate) o += *(const f32x4*)d;
      *(f32x4*)d = o;
    } else { for (int r = 0; r < nv; ++r) d[r] = p.accumulate ? d[r] + v[r] : v[r]; }
  } else {
    bf16* d = (bf16*)p.C + (long)m * p.ldc + n;
    if (FAST) {
      if (p.accumulate) { bf16x4 c = *(const bf16x4*)d;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += bf2f(c[r]); }
      bf16x4 o = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
      *(bf16x4*)d = o;
    } else { for (int r = 0; r < nv; ++r) d[r] = f2bf(p.accumulate ? bf2f(d[r]) + v[r] : v[r]); }
  }
}

template <bool AKS, bool BKS>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];

  // ---- block -> tile (XCD-contiguous, grouped along M)
  int nwg = p.nbm * p.nbn;
  int id = xcd_remap(blockIdx.x, nwg);
  int per_group = GROUP_M * p.nbn;
  int grp = id / per_group;
  int first_m = grp * GROUP_M;
  int gsz = min(p.nbm - first_m, GROUP_M);
  int in_g = id - grp * per_group;
  int tm = first_m + in_g % gsz, tn = in_g / gsz;
  int m0 = tm * BM, n0 = tn * BN;

  int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int wm = wave >> 1, wn = wave & 1;

  if (p.ksplit > 0) {               // split-K: this block reduces one K slice into its own f32 slab (summed by splitk_reduce)
    int k_off = blockIdx.y * p.ksplit;
    p.A += AKS ? (long)k_off * p.lda : (long)k_off;
    p.B += BKS ? (long)k_off * p.ldb : (long)k_off;
    p.K = min(p.K - k_off, p.ksplit);
    p.C = (float*)p.C + (long)blockIdx.y * p.M * p.ldc;
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 ra[4], rb[4];
  int nk = (p.K + BK - 1) / BK;

  g2r<AKS>(p.A, p.lda, m0, 0, p.M, p.K, ra);
  g2r<BKS>(p.B, p.ldb, n0, 0, p.N, p.K, rb);
  r2s<AKS>(smem, ra);
  r2s<BKS>(smem + 16384, rb);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    char* cur = smem + (kt & 1) * STAGE_BYTES;
    char* nxt = smem + ((kt + 1) & 1) * STAGE_BYTES;
    bool more = kt + 1 < nk;
    if (more) {
      g2r<AKS>(p.A, p.lda, m0, (kt + 1) * BK, p.M, p.K, ra);
      g2r<BKS>(p.B, p.ldb, n0, (kt + 1) * BK, p.N, p.K, rb);
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        fa[i] = AKS ? frag_ks(cur, wm * 64 + i * 16, kk) : frag_kc(cur, wm * 64 + i * 16, kk);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        fb[j] = BKS ? frag_ks(cur + 16384, wn * 64 + j * 16, kk) : frag_kc(cur + 16384, wn * 64 + j * 16, kk);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = MFMA16(fb[j], fa[i], acc[i][j]);   // D[n][m]
    }
    if (more) {
      r2s<AKS>(nxt, ra);
      r2s<BKS>(nxt + 16384, rb);
    }
    __syncthreads();
  }

  // ---- epilogue: lane holds C[m = .. + (lane&15)][n = .. + (lane>>4)*4 + r], r = 0..3
  // Every acc[i][j] is named with literal indices (a runtime-indexed accumulator array is demoted to scratch
  // and then spilled on every K-step: cdna guide rule 20).
  float gate = 1.f;
  if (p.gate) gate = tanhf(bf2f(*p.gate));
  bool fast = ((p.N & 3) == 0) && (((p.ldc | p.ldres | p.ldaux | p.ldpre) & 3) == 0);
  int mb = m0 + wm * 64 + (lane & 15), nb = n0 + wn * 64 + (lane >> 4) * 4;
#define EPI_TILE_true(i, j) epi_tile<true>(p, acc[i][j], mb + (i) * 16, nb + (j) * 16, gate);
#define EPI_TILE_false(i, j) epi_tile<false>(p, acc[i][j], mb + (i) * 16, nb + (j) * 16, gate);
#define EPI_TILE_EPI_PLAIN_(i, j) EPI_PLAIN(i, j)
#define EPI_TILE(F, i, j) EPI_TILE_##F(i, j)
#define EPI_ROW(F, i) EPI_TILE(F, i, 0) EPI_TILE(F, i, 1) EPI_TILE(F, i, 2) EPI_TILE(F, i, 3)
#define EPI_ALL(F) EPI_ROW(F, 0) EPI_ROW(F, 1) EPI_ROW(F, 2) EPI_ROW(F, 3)
  // the plain form (alpha only: split-K slabs, weight gradients) has its own compact code: the 32 inlined copies of the fully
  // general epi_tile are 110 KiB, and running through them costs more instruction-cache misses than the stores cost cycles
  bool plain = fast && !p.bias && !p.act && !p.pre && !p.aux && !p.res && !p.gate && !p.accumulate;
#define EPI_PLAIN(i, j) { int m = mb + (i) * 16, n = nb + (j) * 16;                                                   \
    if (m < p.M && n < p.N) { f32x4 v = acc[i][j] * p.alpha;                                                          \
      if (p.out_f32) *(f32x4*)((float*)p.C + (long)m * p.ldc + n) = v;                                                \
      else { bf16x4 o = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])}; *(bf16x4*)((bf16*)p.C + (long)m * p.ldc + n) = o; } } }
  if (plain) { EPI_ALL(EPI_PLAIN_) }
  else if (fast) { EPI_ALL(true) } else { EPI_ALL(false) }
#undef EPI_PLAIN
#undef EPI_ALL
#undef EPI_ROW
#undef EPI_TILE
#undef EPI_TILE_true
#undef EPI_TILE_false
#undef EPI_TILE_EPI_PLAIN_
}

// ---- skinny GEMM for decoding (M <= 64 rows: K beams x one new token).  Weight-bandwidth bound: every element of W is
// used once, so W streams global -> registers directly in MFMA operand layout (no LDS); the few X rows come out of L2.
// One block = NR tiles of 16 rows of W; its NW waves take interleaved 64-k chunks (4 adjacent 128-B lines per W row and iteration)
// and meet in LDS.  Inside a chunk lane (r, g) owns k = 16g .. 16g+15 of row r -- 32 contiguous bytes -- and feeds them to two
// MFMAs; X uses the same k assignment, and a contraction does not care in which order k is visited.
// Round 3: NR > 1 -- an X fragment fetched from L2 serves NR weight tiles.  With one tile per block every block re-read all of X
// (as many L2 -> CU bytes as W itself at 10 rows, 2.5x as many at 40: the 4-users-per-call decode ran its GEMMs at 1.5 TB/s) --
// and 16 waves per block at <= 16 rows (+10-23 % weight bandwidth: tools/bench_skinny.py, profiles/r03_decode_timings.txt).
template <int MB, int NW, int NR, int U>
__global__ __launch_bounds__(64 * NW) void gemm_skinny_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char sk_smem[];
  f32x4* red = (f32x4*)sk_smem;                       // [NW][MB][NR][64]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int g = lane >> 4, r = lane & 15;
  const int n0 = blockIdx.x * (16 * NR);
  const bf16* wp[NR];
#pragma unroll
  for (int t = 0; t < NR; ++t) wp[t] = p.B + (long)min(n0 + 16 * t + r, p.N - 1) * p.ldb + g * 16;
  const bf16* xp[MB];
#pragma unroll
  for (int i = 0; i < MB; ++i) xp[i] = p.A + (long)min(i * 16 + r, p.M - 1) * p.lda + g * 16;
  f32x4 acc[MB][NR];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int t = 0; t < NR; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nchunk = p.K >> 6;
  int c = w;
  for (; c + NW * (U - 1) < nchunk; c += NW * U) {
    u32x4 wv[U][NR][2];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int t = 0; t < NR; ++t) {
        const bf16* q = wp[t] + (long)(c + NW * u) * 64;
        wv[u][t][0] = __builtin_nontemporal_load((const u32x4*)q);
        wv[u][t][1] = __builtin_nontemporal_load((const u32x4*)(q + 8));
      }
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int i = 0; i < MB; ++i) {
        const bf16* q = xp[i] + (long)(c + NW * u) * 64;
        u32x4 x0 = *(const u32x4*)q, x1 = *(const u32x4*)(q + 8);
#pragma unroll
        for (int t = 0; t < NR; ++t) {
          acc[i][t] = MFMA16(wv[u][t][0], x0, acc[i][t]);
          acc[i][t] = MFMA16(wv[u][t][1], x1, acc[i][t]);
        }
      }
    }
  }
  for (; c < nchunk; c += NW) {
#pragma unroll
    for (int i = 0; i < MB; ++i) {
      const bf16* qx = xp[i] + (long)c * 64;
      u32x4 x0 = *(const u32x4*)qx, x1 = *(const u32x4*)(qx + 8);
#pragma unroll
      for (int t = 0; t < NR; ++t) {
        const bf16* q = wp[t] + (long)c * 64;
        acc[i][t] = MFMA16(*(const u32x4*)q, x0, acc[i][t]);
        acc[i][t] = MFMA16(*(const u32x4*)(q + 8), x1, acc[i][t]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int t = 0; t < NR; ++t) red[((w * MB + i) * NR + t) * 64 + lane] = acc[i][t];
  __syncthreads();
  float gate = 1.f;
  if (p.gate) gate = tanhf(bf2f(*p.gate));
  bool fast = ((p.N & 3) == 0) && (((p.ldc | p.ldres | p.ldaux | p.ldpre) & 3) == 0);
  for (int it = w; it < MB * NR; it += NW) {           // waves share the (row group, weight tile) pairs; fixed summation order over the waves
    const int i = it / NR, t = it % NR;
    f32x4 a = red[((0 * MB + i) * NR + t) * 64 + lane];
#pragma unroll
    for (int ww = 1; ww < NW; ++ww) a += red[((ww * MB + i) * NR + t) * 64 + lane];
    const int nn = n0 + 16 * t + g * 4;
    if (n0 + 16 * t >= p.N) continue;
    if (fast) epi_tile<true>(p, a, i * 16 + r, nn, gate);
    else epi_tile<false>(p, a, i * 16 + r, nn, gate);
  }
}

// ---- skinny2 (round 6): decode rows M <= 16 -- one token per row of K <= 16 beams (eval_rec.py:100-110) or one greedy row
// (eval_img_gen.py:102-111) -- with K <= 4096.  Round 3's blocking (one workgroup per 16 weight rows, its waves split K in 64-k chunks,
// partial sums meet in LDS) with two changes that the decode step's trace asked for (profiles/r06_decode_step_k1_round5_launches.csv: 506 launches
// of >= 4.5 us each -- the floor of a graph node -- of which 97 are LayerNorms on one row):
//   * EVERYTHING a wave needs goes out before anything is waited for: its CW <= 4 chunks of the weight rows (32 contiguous bytes per lane and
//     chunk) and the matching activation fragments -- one memory round trip per workgroup where the round-3 loop (not unrolled at K = 2560:
//     2.5 chunks per wave) made two or three dependent ones;
//   * the LayerNorm in front of the projection is computed HERE (ln_gamma / ln_beta / ln_eps): a wave sums its own chunks of the M rows, the
//     16 waves' partial sums meet in LDS (fixed order; mean first, then the centred squares: two barriers under the weights' flight), every
//     lane normalises the fragments it holds -- (x - mean) * rstd * gamma + beta in fp32, rounded to bf16: the values unimp_layernorm_fwd stores
//     up to the summation order of the statistics -- and the LayerNorm is no launch of its own.
// Lane (r, g) holds k = 16 g .. 16 g + 15 of weight row r (operand map of the round-3 kernel); accumulator lane (r, g) = activation row r,
// weight rows 4 g .. 4 g + 3.  Deterministic: fixed summation order over chunks and waves.
// (A PERSISTENT form -- one workgroup per CU walking its tiles with a ring of loads in flight -- was built first and measured slower:
// profiles/r06_negative_results_decode_and_mx.txt.)
// weight loads use the DEFAULT cache policy: nontemporal loads measured 10 - 25 % slower on every shape (profiles/r06_negative_results_decode_and_mx.txt)
__device__ __forceinline__ u32x4 sk2_ldw(const bf16* q) { return *(const u32x4*)q; }

// a workgroup barrier that waits for THIS wave's LDS operations only: __syncthreads() also drains vmcnt -- here that is the 10 KB of weights in flight, i.e. the
// LayerNorm's statistics would start AFTER the weights have landed instead of under their flight
#define SK2_LDS_BARRIER() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)

// the epilogue shared by the decode-row kernels: the waves' partial tiles meet in LDS, wave 0 sums them in wave order and stores
// R: weight rows per workgroup (<= 16; the tile's columns beyond R are not stored)
template <int NW>
__device__ __forceinline__ void sk2_finish(const GemmParams& p, f32x4 (&red)[NW * 64], f32x4 acc, int w, int lane, int n0, int R) {
  const int g = lane >> 4, r = lane & 15;
  red[w * 64 + lane] = acc;
  __syncthreads();
  if (w == 0) {
    f32x4 a = red[lane];
#pragma unroll 4
    for (int ww = 1; ww < NW; ++ww) a += red[ww * 64 + lane];
    if (n0 < p.N) {
      float gate = 1.f;
      if (p.gate) gate = tanhf(bf2f(*p.gate));
      GemmParams q = p;
      q.N = min(p.N, n0 + R);                       // epi_tile's column bound
      const bool fast = ((q.N & 3) == 0) && ((R & 3) == 0) && (((p.ldc | p.ldres | p.ldaux | p.ldpre) & 3) == 0);
      if (fast) epi_tile<true>(q, a, r, n0 + g * 4, gate);
      else epi_tile<false>(q, a, r, n0 + g * 4, gate);
    }
  }
}

// ROWS PER WORKGROUP (R <= 16, kernel argument): a CU streams weights at about 1 / 256 of the chip's HBM rate whatever it runs, so the launch takes as
// long as the CU with the most rows.  With 16-row tiles N = 2560 is 160 workgroups (96 CUs idle, 16.2 us for 52 MB where N = 10 240 took 12.4) and
// N = 10 240 is 640 (2.5 per CU: 48 rows on the busiest CU where 40 is the mean).  The host picks R so that ceil(ceil(N / R) / 256) * R is smallest:
// 10 rows x 256 workgroups for N = 2560, 15 x 512 for N = 7680, 4 x 128 for N = 512 (below two 16-row workgroups per CU only: skinny2_rows).  Lanes r >= R of the MFMA tile re-read row R - 1 (the
// same addresses in the same instruction: no traffic) and their columns are not stored.
// (8-row tiles whose lanes r >= 8 hold the other k-half of rows r - 8 -- twice the workgroups, one MFMA per chunk, M <= 8 only -- were built first:
// 16.2 us where the 16-row tiles took 17.2, i.e. the workgroup count was not what mattered; profiles/r06_negative_results_decode_and_mx.txt.)
template <int NW, int CW>
__global__ __launch_bounds__(64 * NW) void gemm_skinny2_kernel(GemmParams p, int R) {
  __shared__ f32x4 red[NW * 64];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, r = lane & 15;
  const int n0 = blockIdx.x * R, nchunk = p.K >> 6;
  // lane (r, g) holds k = 8 g .. 8 g + 7 of each 32-k half of the chunk: one load instruction of the wave covers a contiguous 64-byte half line of each
  // of its 16 rows (the round-3 map -- 16 consecutive k per lane -- made every instruction touch both halves of every line; with streaming (nt) loads
  // that measured as twice the requests: profiles/r06_negative_results_decode_and_mx.txt)
  const bf16* wp = p.B + (long)min(n0 + min(r, R - 1), p.N - 1) * p.ldb + g * 8;
  const bf16* xp = p.A + (long)min(r, p.M - 1) * p.lda + g * 8;
  u32x4 wv[CW][2];
  bf16x8 xv[CW][2];
  // STRAIGHT-LINE issue (no branch between the first load and the last, so that hipcc's wait counts stay exact): a wave whose chunks lie beyond K
  // loads the last real chunk again and zeroes the activation fragment with a select
#pragma unroll
  for (int u = 0; u < CW; ++u) {
    const long c = min(w + NW * u, nchunk - 1);
#pragma unroll
    for (int h = 0; h < 2; ++h) { wv[u][h] = sk2_ldw(wp + c * 64 + h * 32); xv[u][h] = *(const bf16x8*)(xp + c * 64 + h * 32); }
  }
  __builtin_amdgcn_sched_barrier(0);          // every load of the wave goes out before its first wait (left alone hipcc folds the late loads into the MFMA chain)
  const bf16x8 z8 = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
  {                                                     // only the last chunk of a wave can lie beyond K (host: CW = ceil(nchunk / NW))
    const bool live = w + NW * (CW - 1) < nchunk;
    xv[CW - 1][0] = live ? xv[CW - 1][0] : z8; xv[CW - 1][1] = live ? xv[CW - 1][1] : z8;
  }
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < CW; ++u)
#pragma unroll
    for (int h = 0; h < 2; ++h) acc = MFMA16(wv[u][h], xv[u][h], acc);
  sk2_finish<NW>(p, red, acc, w, lane, n0, R);
}

// ---- the same with the LayerNorm of the rows fused in front (unimp_gemm_desc.ln_gamma): K = 64 NW CW exactly (512 ... 2560 with 8 waves, 3072 / 4096 with 16)
// The workgroup normalises the M REAL rows once, cooperatively, and the waves read their MFMA fragments of the result from LDS.  A UNIT is one 512-element
// segment of one row (one 16-byte piece per lane); wave w owns units w, w + NW, ... (UM of them at most; template, so that the loads are straight-line
// code).  The first form of this fusion normalised in the MFMA layout -- every lane the 16 k-values x CW chunks of "its" row, rows beyond M included:
// 16 K element-normalisations per workgroup whatever M is, about 1000 VALU instructions per wave, 640 workgroups each: + 4 us on a 14 us launch at M = 1.
// Here the work is M K per workgroup (M = 1: one piece per lane in five waves).
// ISSUE ORDER matters: a wave's loads return in order.  The unit pieces and gamma / beta go out FIRST (L2 hits), the weights behind them: statistics,
// barriers and normalisation run under the weights' flight.  No branch between the first load and the last (hipcc's wait counts go to vmcnt(0) at
// control-flow joins): units beyond the last are clamped to it and recomputed (same values to the same LDS addresses).
template <int CTRL> __device__ __forceinline__ float sk2_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float sk2_wave_sum(float v) {       // fixed order; the total in every lane (through SGPRs)
  v += sk2_dpp<0xB1>(v);            // quad_perm [1, 0, 3, 2]
  v += sk2_dpp<0x4E>(v);            // quad_perm [2, 3, 0, 1]
  v += sk2_dpp<0x141>(v);           // row_half_mirror: the other quad of the eight
  v += sk2_dpp<0x140>(v);           // row_mirror: the other eight of the row -> all 16 lanes of a row hold the row's sum
  const int vi = __builtin_bit_cast(int, v);
  const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 0)), b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 16));
  const float c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 32)), d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 48));
  return (a + b) + (c + d);
}

// NT = 2: the workgroup serves TWO tiles of R weight rows with one normalisation (the second tile's weights are issued when the units are dead, i.e.
// behind the normalisation, and fly under the first tile's MFMAs): at M = 10 the normalisation is 7 units per wave, about 1 000 VALU instructions in
// each of 640 workgroups -- as much VALU time as the launch has -- and every workgroup re-reads the M rows; 2 x 10 rows x 512 workgroups is also the
// balanced split of N = 10 240 over 256 CUs (40 rows each; 16-row tiles: 48 on the busiest).
template <int NW, int CW, int UM, int NT>
__global__ __launch_bounds__(64 * NW) void gemm_skinny2_ln_kernel(GemmParams p, int R) {
  constexpr int S = NW * CW / 8, K = 64 * NW * CW, LDX = K + 8;         // segments per row; the row pitch in LDS shifts consecutive rows by four banks
  __shared__ f32x4 red[NT][NW * 64];
  __shared__ float stat[2][16][8];
  extern __shared__ __attribute__((aligned(16))) char sk_smem[];
  bf16* gbs = (bf16*)sk_smem;                 // gamma [K], beta [K]
  bf16* xs = gbs + 2 * K;                     // the normalised rows [M][LDX]
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, r = lane & 15;
  const int n0 = blockIdx.x * (NT * R), nunits = p.M * S;
  const bf16* wp = p.B + (long)min(n0 + min(r, R - 1), p.N - 1) * p.ldb + g * 8;
  bf16x8 xu[UM];
  int urow[UM], useg[UM];
#pragma unroll
  for (int i = 0; i < UM; ++i) {
    const int u = min(w + NW * i, nunits - 1);
    urow[i] = u / S; useg[i] = u - urow[i] * S;
    xu[i] = *(const bf16x8*)(p.A + (long)urow[i] * p.lda + useg[i] * 512 + lane * 8);
  }
  const int gbi = min((int)threadIdx.x, K / 8 - 1) * 8;                // K / 8 pieces <= 64 NW threads
  bf16x8 gmr = *(const bf16x8*)(p.ln_gamma + gbi);
  bf16x8 btr = *(const bf16x8*)((p.ln_beta ? p.ln_beta : p.ln_gamma) + gbi);
  __builtin_amdgcn_sched_barrier(0);          // the loads above stay above the weight loads
  u32x4 wv[NT][CW][2];
#pragma unroll
  for (int u = 0; u < CW; ++u) {
    const long c = w + NW * u;
    wv[0][u][0] = sk2_ldw(wp + c * 64);
    wv[0][u][1] = sk2_ldw(wp + c * 64 + 32);
  }
  __builtin_amdgcn_sched_barrier(0);
  const bf16x8 z8 = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
  if (!p.ln_beta) btr = z8;
  *(bf16x8*)(gbs + gbi) = gmr; *(bf16x8*)(gbs + K + gbi) = btr;
  const float inv_d = 1.f / (float)K;
#pragma unroll
  for (int i = 0; i < UM; ++i) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += bf2f(xu[i][j]);
    s = sk2_wave_sum(s);
    if (lane == 0) stat[0][urow[i]][useg[i]] = s;
  }
  SK2_LDS_BARRIER();
  float mu[UM];
#pragma unroll
  for (int i = 0; i < UM; ++i) {
    float tot = 0.f;
#pragma unroll
    for (int sg = 0; sg < S; ++sg) tot += stat[0][urow[i]][sg];
    mu[i] = tot * inv_d;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float d = bf2f(xu[i][j]) - mu[i]; q += d * d; }
    q = sk2_wave_sum(q);
    if (lane == 0) stat[1][urow[i]][useg[i]] = q;
  }
  SK2_LDS_BARRIER();
#pragma unroll
  for (int i = 0; i < UM; ++i) {
    float qt = 0.f;
#pragma unroll
    for (int sg = 0; sg < S; ++sg) qt += stat[1][urow[i]][sg];
    const float rs = rsqrtf(qt * inv_d + p.ln_eps);
    const int ko = useg[i] * 512 + lane * 8;
    const bf16x8 gm = *(const bf16x8*)(gbs + ko), bt = *(const bf16x8*)(gbs + K + ko);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = f2bf((bf2f(xu[i][j]) - mu[i]) * rs * bf2f(gm[j]) + bf2f(bt[j]));
    *(bf16x8*)(xs + urow[i] * LDX + ko) = o;
  }
  if (NT > 1) {                                 // the second tile's weights: the units are dead, the registers are free
    const bf16* wp1 = p.B + (long)min(n0 + R + min(r, R - 1), p.N - 1) * p.ldb + g * 8;
#pragma unroll
    for (int u = 0; u < CW; ++u) {
      const long c = w + NW * u;
      wv[NT - 1][u][0] = sk2_ldw(wp1 + c * 64);
      wv[NT - 1][u][1] = sk2_ldw(wp1 + c * 64 + 32);
    }
  }
  SK2_LDS_BARRIER();
  const bf16* xr = xs + min(r, p.M - 1) * LDX + g * 8;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < CW; ++u) {
      const int c = w + NW * u;
      const bf16x8 x0 = *(const bf16x8*)(xr + c * 64), x1 = *(const bf16x8*)(xr + c * 64 + 32);
      acc = MFMA16(wv[t][u][0], x0, acc);
      acc = MFMA16(wv[t][u][1], x1, acc);
    }
    red[t][w * 64 + lane] = acc;
  }
  __syncthreads();
  if (w < NT && n0 + w * R < p.N) {             // wave t finishes tile t
    f32x4 a = red[w][lane];
#pragma unroll 4
    for (int ww = 1; ww < NW; ++ww) a += red[w][ww * 64 + lane];
    const int nt0 = n0 + w * R;
    float gate = 1.f;
    if (p.gate) gate = tanhf(bf2f(*p.gate));
    GemmParams q = p;
    q.N = min(p.N, nt0 + R);                      // epi_tile's column bound
    const bool fast = ((q.N & 3) == 0) && ((R & 3) == 0) && (((p.ldc | p.ldres | p.ldaux | p.ldpre) & 3) == 0);
    if (fast) epi_tile<true>(q, a, r, nt0 + g * 4, gate);
    else epi_tile<false>(q, a, r, nt0 + g * 4, gate);
  }
}

// the same for a LONG contraction (K > 4096, no LayerNorm: the down-projections, K = 10 240 / 16 384): 16 waves, rounds of CW chunks per wave with the
// NEXT round's weights and activation fragments issued before the current round's MFMAs (two register sets, no barrier inside the loop) -- the
// round-3 loop drained its loads between iterations (four dependent memory round trips per workgroup at K = 10 240: 20 us for 52 MB).
template <int CW>
__global__ __launch_bounds__(1024) void gemm_skinny2_long_kernel(GemmParams p, int R) {
  constexpr int NW = 16;
  __shared__ f32x4 red[NW * 64];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, r = lane & 15;
  const int n0 = blockIdx.x * R, nchunk = p.K >> 6;
  const bf16* wp = p.B + (long)min(n0 + min(r, R - 1), p.N - 1) * p.ldb + g * 8;
  const bf16* xp = p.A + (long)min(r, p.M - 1) * p.lda + g * 8;
  const int nrounds = ((nchunk + NW - 1) / NW + CW - 1) / CW;
  u32x4 wb[2][CW][2];
  bf16x8 xb[2][CW][2];
  auto load = [&](u32x4 (&wv)[CW][2], bf16x8 (&xv)[CW][2], int rd) {
#pragma unroll
    for (int u = 0; u < CW; ++u) {
      const int c = w + NW * (rd * CW + u);
      if (c < nchunk) {                    // wave-uniform
        wv[u][0] = sk2_ldw(wp + (long)c * 64);
        wv[u][1] = sk2_ldw(wp + (long)c * 64 + 32);
        xv[u][0] = *(const bf16x8*)(xp + (long)c * 64); xv[u][1] = *(const bf16x8*)(xp + (long)c * 64 + 32);
      } else {
        wv[u][0] = u32x4{0, 0, 0, 0}; wv[u][1] = u32x4{0, 0, 0, 0};
        xv[u][0] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; xv[u][1] = xv[u][0];
      }
    }
  };
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  auto mm = [&](const u32x4 (&wv)[CW][2], const bf16x8 (&xv)[CW][2]) {
#pragma unroll
    for (int u = 0; u < CW; ++u) { acc = MFMA16(wv[u][0], xv[u][0], acc); acc = MFMA16(wv[u][1], xv[u][1], acc); }
  };
  load(wb[0], xb[0], 0);
  for (int rd = 0; rd < nrounds; rd += 2) {
    if (rd + 1 < nrounds) load(wb[1], xb[1], rd + 1);
    mm(wb[0], xb[0]);
    if (rd + 2 < nrounds) load(wb[0], xb[0], rd + 2);
    if (rd + 1 < nrounds) mm(wb[1], xb[1]);
  }
  sk2_finish<NW>(p, red, acc, w, lane, n0, R);
}

// weight rows per workgroup: the busiest CU's rows = ceil(workgroups / 256) * rows per workgroup, smallest over R = 4 .. 16 (x NT tiles where the kernel
// has a two-tile form); ties: the larger workgroup
static int g_skinny2_rows = -1;
static int skinny2_rows(int N, int max_nt, int* nt_out) {
  if (nt_out) *nt_out = 1;
  if (g_skinny2_rows < 0) { const char* e = getenv("UNIMP_SKINNY2_ROWS"); g_skinny2_rows = e ? atoi(e) : 0; }
  if (g_skinny2_rows >= 1 && g_skinny2_rows <= 16) return g_skinny2_rows;          // A/B knob: 16 = the fixed tiles, one per workgroup
  // from two 16-row workgroups per CU on the balance no longer decides with ONE tile per workgroup: measured, R = 10 x 1024 workgroups at N = 10 240 was
  // 14.1 us where 16 x 640 took 13.0 (18.2 against 15.7 at M = 10: every workgroup re-reads the M rows), and the head (4629 tiles) lost a quarter.
  // Two tiles per workgroup keep the count down: searched up to 2048 tiles.
  const int tiles16 = (N + 15) / 16;
  if (tiles16 >= (max_nt > 1 ? 2048 : 512)) return 16;
  int best = 16, best_nt = 1; long cost = -1;
  for (int nt = 1; nt <= max_nt; ++nt)
    for (int R = 16; R >= 4; --R) {
      if (nt == 1 && tiles16 >= 512 && R < 16) continue;         // the one-tile rule stops at 512 tiles (above)
      const long wgs = (N + nt * R - 1) / (nt * R), c = ((wgs + 255) / 256) * nt * R;
      // ties: two tiles where they are allowed (the caller allows them where the normalisation weighs) and the tiles stay at least 8 rows high
      if (cost < 0 || c < cost || (c == cost && nt > best_nt && R >= 8)) { cost = c; best = R; best_nt = nt; }
    }
  if (nt_out) *nt_out = best_nt;
  return best;
}
static int skinny2_ln_lds(int M, int K) { return (2 * K + M * (K + 8)) * 2; }
static bool skinny2_ln_ok(int M, int K) {
  if (M < 1 || M > 16 || (K & 511) || K < 512 || K > 4096 || K == 3584) return false;
  if (K > 2560 && (M * (K >> 9) + 15) / 16 > 7) return false;        // 16 waves of 128 registers: seven units per wave at most (K = 4096, M = 15, 16)
  return skinny2_ln_lds(M, K) + 2 * 16 * 1024 + 1024 + 1024 <= 160 * 1024;        // + the two reduction tiles of the two-tile form
}
static bool skinny2_ok(const unimp_gemm_desc* d) {
  // M <= 16 decode rows, k-contiguous operands, whole 64-k chunks
  if (d->M > 16 || d->a_kstrided || d->b_kstrided || (d->K & 63) || d->K < 64) return false;
  return !d->ln_gamma || skinny2_ln_ok(d->M, d->K);
}

template <int NW, int CW, int NT>
static void launch_skinny2_ln_nt(const unimp_gemm_desc* d, GemmParams& p, hipStream_t s, int R) {
  using kern_t = void (*)(GemmParams, int);
  const int um = (d->M * (NW * CW / 8) + NW - 1) / NW;           // units per wave
  const int b = um <= 1 ? 0 : um <= 2 ? 1 : um <= 4 ? 2 : um <= 7 ? 3 : 4;
  static const kern_t kerns[5] = {gemm_skinny2_ln_kernel<NW, CW, 1, NT>, gemm_skinny2_ln_kernel<NW, CW, 2, NT>, gemm_skinny2_ln_kernel<NW, CW, 4, NT>,
                                  gemm_skinny2_ln_kernel<NW, CW, 7, NT>, gemm_skinny2_ln_kernel<NW, CW, NW == 16 ? 7 : 10, NT>};   // 16 waves: never more than 7 (skinny2_ln_ok)
  static bool attr[5] = {false, false, false, false, false};
  if (!attr[b]) { (void)hipFuncSetAttribute((const void*)kerns[b], hipFuncAttributeMaxDynamicSharedMemorySize, 142 * 1024 - (NT - 1) * NW * 1024); attr[b] = true; }
  hipLaunchKernelGGL(kerns[b], dim3((d->N + NT * R - 1) / (NT * R)), dim3(64 * NW), skinny2_ln_lds(d->M, d->K), s, p, R);
}
template <int NW, int CW, bool TWO>
static void launch_skinny2_ln(const unimp_gemm_desc* d, GemmParams& p, hipStream_t s) {
  int nt = 1;
  static int max_nt = -1;
  if (max_nt < 0) { const char* e = getenv("UNIMP_SKINNY2_NT"); max_nt = e ? atoi(e) : 2; }          // A/B knob: 1 = one tile per workgroup
  // two tiles only where the normalisation weighs (M >= 4 rows): at M = 1 it is one unit per wave and the two-tile form measured 1 % slower in the
  // greedy token-step (2.476 | 2.453 ms, alternating) while K = 10 gained 4 % (3.55 | 3.71)
  const int R = skinny2_rows(d->N, TWO && max_nt >= 2 && d->M >= 4 ? 2 : 1, &nt);
  if (TWO && nt == 2) launch_skinny2_ln_nt<NW, CW, TWO ? 2 : 1>(d, p, s, R);
  else launch_skinny2_ln_nt<NW, CW, 1>(d, p, s, R);
}

static void launch_skinny2(const unimp_gemm_desc* d, GemmParams& p, void* stream) {
  const int nchunk = d->K >> 6;
  hipStream_t s = (hipStream_t)stream;
  const int R = skinny2_rows(d->N, 1, nullptr);
  dim3 grid((d->N + R - 1) / R);
  if (d->ln_gamma) {
    switch (d->K) {
      case 512: launch_skinny2_ln<8, 1, false>(d, p, s); break;
      case 1024: launch_skinny2_ln<8, 2, false>(d, p, s); break;
      case 1536: launch_skinny2_ln<8, 3, false>(d, p, s); break;
      case 2048: launch_skinny2_ln<8, 4, false>(d, p, s); break;
      case 2560: launch_skinny2_ln<8, 5, true>(d, p, s); break;          // two-tile forms for the cfg2 / cfg5 widths
      case 3072: launch_skinny2_ln<16, 3, false>(d, p, s); break;
      default: launch_skinny2_ln<16, 4, true>(d, p, s); break;           // 4096
    }
    return;
  }
  if (d->K > 4096) { hipLaunchKernelGGL((gemm_skinny2_long_kernel<3>), grid, dim3(1024), 0, s, p, R); return; }
  // K <= 2560: 8 waves with up to five chunks each (two workgroups per CU at <= 128 registers); beyond: 16 waves with three or four
  const int nw = nchunk <= 40 ? 8 : 16, cw = (nchunk + nw - 1) / nw;
#define SK2_GO(NW_, CW_) hipLaunchKernelGGL((gemm_skinny2_kernel<NW_, CW_>), grid, dim3(64 * NW_), 0, s, p, R)
  if (nw == 8) { switch (cw) { case 1: SK2_GO(8, 1); break; case 2: SK2_GO(8, 2); break; case 3: SK2_GO(8, 3); break; case 4: SK2_GO(8, 4); break; default: SK2_GO(8, 5); break; } }
  else if (cw <= 3) SK2_GO(16, 3); else SK2_GO(16, 4);
#undef SK2_GO
}

static bool skinny_ok(const unimp_gemm_desc* d) {
  return d->M <= 64 && !d->a_kstrided && !d->b_kstrided && (d->K & 63) == 0;
}

static int g_skinny2 = -1;
static int skinny2_on() {
  if (g_skinny2 < 0) { const char* e = getenv("UNIMP_SKINNY2"); g_skinny2 = e ? (atoi(e) != 0) : 1; }
  return g_skinny2;
}
// the split of N weight rows over workgroups the decode-row GEMM takes (tests, tools): rows per tile; *tiles_per_workgroup = 1 or 2 (max_tiles = 2: the
// fused-LayerNorm forms of K = 2560 / 4096 at M >= 4)
extern "C" int unimp_gemm_skinny_rows(int N, int max_tiles, int* tiles_per_workgroup) { return skinny2_rows(N, max_tiles >= 2 ? 2 : 1, tiles_per_workgroup); }
extern "C" int unimp_gemm_set_skinny2(int on) { int old = skinny2_on(); g_skinny2 = on != 0; return old; }
extern "C" int unimp_gemm_skinny_ln_ok(int M, int K) {      // may a decode GEMM of M rows and depth K take its LayerNorm fused (unimp_gemm_desc.ln_gamma)?
  unimp_gemm_desc d = {};
  d.M = M; d.K = K; d.ln_gamma = (const void*)1;
  return skinny2_on() && skinny2_ok(&d);
}

static void launch_skinny(const unimp_gemm_desc* d, void* stream) {
  GemmParams p;
  p.A = (const bf16*)d->A; p.B = (const bf16*)d->B; p.C = d->C;
  p.M = d->M; p.N = d->N; p.K = d->K; p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc;
  p.bias = (const bf16*)d->bias; p.res = (const bf16*)d->res; p.ldres = d->ldres;
  p.aux = (const bf16*)d->aux; p.ldaux = d->ldaux; p.pre = (bf16*)d->pre; p.ldpre = d->ldpre;
  p.gate = (const bf16*)d->gate; p.alpha = d->alpha; p.act = d->act; p.dact = d->dact;
  p.out_f32 = d->out_f32; p.accumulate = d->accumulate; p.pre_deriv = d->pre_deriv;
  p.nbm = 1; p.nbn = (d->N + 15) / 16; p.ksplit = 0;
  p.ln_gamma = (const bf16*)d->ln_gamma; p.ln_beta = (const bf16*)d->ln_beta; p.ln_eps = d->ln_eps;
  if (skinny2_on() && skinny2_ok(d)) { launch_skinny2(d, p, stream); return; }
  hipStream_t s = (hipStream_t)stream;
  // Configuration by shape, from tools/bench_skinny.py on MI355X (profiles/r03_skinny_gemm_configs.txt; UNIMP_SKINNY_CFG = "nw,nr"
  // overrides for A/B):
  //   rows <= 16: 16 waves split K sixteen ways, ONE weight tile per block (two tiles halve the blocks in flight and lose 10-60 %)
  //   more rows:  8 waves; two weight tiles per block once that still leaves >= 160 blocks (N >= 5120) -- X fragments are re-read per
  //               block: with 40 rows (4 users x 10 beams) one tile per block moved 2.5x as many X bytes as W bytes; 7680 x 2560:
  //               24.8 -> 17.7 us, the 74 053-row head 208 -> 146 us; four tiles lose again (too few blocks)
  static int f_nw = -1, f_nr = 0;
  if (f_nw < 0) { const char* e = getenv("UNIMP_SKINNY_CFG"); f_nw = 0; if (e) sscanf(e, "%d,%d", &f_nw, &f_nr); }
  const int mb = (d->M + 15) / 16;
  int nw = mb == 1 ? 16 : 8;
  if (d->K < 1024) nw = mb == 1 ? 8 : 4;                                             // short K: fewer waves have a chunk each
  int nr = (mb >= 2 && p.nbn >= 320) ? 2 : 1;
  if (f_nw) { nw = f_nw; nr = f_nr ? f_nr : nr; }
  // (measured and dropped: 8 chunks in flight per wave on the 160-block K = 10240 down-projection -- 25.0 vs 22.9 us)
  dim3 grid((p.nbn + nr - 1) / nr);
#define SK_GO(MB_, NW_, NR_, U_) do {                                                                                              \
    auto kern = gemm_skinny_kernel<MB_, NW_, NR_, U_>;                                                                            \
    size_t lds = (size_t)NW_ * MB_ * NR_ * 64 * sizeof(f32x4);                                                                    \
    static bool attr = false;                                                                                                     \
    if (!attr) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; } \
    hipLaunchKernelGGL(kern, grid, dim3(64 * NW_), lds, s, p); } while (0)
#define SK_NR(MB_, NW_) do { if (nr >= 4) SK_GO(MB_, NW_, 4, 2); else if (nr == 2) SK_GO(MB_, NW_, 2, 2); else SK_GO(MB_, NW_, 1, 4); } while (0)
#define SK_NW(MB_) do { if (nw >= 16) { if (nr >= 2) SK_GO(MB_, 16, 2, 2); else SK_GO(MB_, 16, 1, 4); }                             \
                        else if (nw >= 8) SK_NR(MB_, 8); else SK_NR(MB_, 4); } while (0)
  switch (mb) {
    case 1: SK_NW(1); break;
    case 2: if (nw > 8) nw = 8; SK_NW(2); break;
    case 3: if (nw > 8) nw = 8; SK_NW(3); break;
    default: if (nw > 8) nw = 8; SK_NW(4); break;
  }
#undef SK_NW
#undef SK_NR
#undef SK_GO
}

extern "C" int unimp_gemm2_launch(const unimp_gemm_desc* d, int bn, void* stream);   // gemm2.hip: 256-row tiles, LDS-DMA, 2-stage
extern "C" int unimp_gemm3_launch(const unimp_gemm_desc* d, int bn, void* stream);   // gemm3.hip: 256-row tiles, LDS-DMA, ping-pong
extern "C" int unimp_gemm3_launch_splitk(const unimp_gemm_desc* d, int bn, int splits, float* slabs, void* stream);
extern "C" int unimp_gemm4_launch(const unimp_gemm_desc* d, void* stream);           // gemm4.hip: 256 x 256 tiles, one wave per SIMD
extern "C" int unimp_gemm5_launch(const unimp_gemm_desc* d, void* stream);           // gemm5.hip: 8 self-interleaving waves, one barrier per half-stage
extern "C" int unimp_gemm6_launch(const unimp_gemm_desc* d, void* stream);           // gemm6.hip: persistent ping-pong, next tile's prologue under the epilogue
extern "C" int unimp_gemm3x_launch(const unimp_gemm_desc* d, int bn, void* stream);  // gemm3.hip built with G3_ONESET: one fragment register set
extern "C" int unimp_gemm6x_launch(const unimp_gemm_desc* d, void* stream);          // gemm6.hip built with G3_ONESET
extern "C" int unimp_gemm3a_launch(const unimp_gemm_desc* d, int bn, void* stream);  // gemm3.hip built with G3_ONESET + G3_AFULL: A staged in whole 128-byte rows
extern "C" int unimp_gemm3b_launch(const unimp_gemm_desc* d, int bn, void* stream);  // A/B build of gemm3a (G3_FUSE)
extern "C" int unimp_gemm7_launch(const unimp_gemm_desc* d, void* stream);           // gemm7.hip: one wave per SIMD, hand-ordered two-set main loop, 64-k stages
extern "C" int unimp_gemm7o_launch(const unimp_gemm_desc* d, void* stream);          // ... its first schedule (A/B partner)
extern "C" int unimp_gemm7p_launch(const unimp_gemm_desc* d, void* stream);          // ... with the L2 prefetch of the panels' shares
extern "C" int unimp_gemm9_launch(const unimp_gemm_desc* d, void* stream);           // gemm9.hip: 128 x 256 tiles, 4 waves, TWO workgroups per CU

static int check_operand(const void* p, long ld, int ks, int rows) {
  if (((uintptr_t)p & 15) != 0) return UNIMP_ERR_ALIGN;
  if ((ld & 7) != 0) return UNIMP_ERR_ALIGN;
  if (ks && ld < ((rows + 7) & ~7)) return UNIMP_ERR_SHAPE;
  return 0;
}

static int validate(const unimp_gemm_desc* d) {
  if (!d || !d->A || !d->B || !d->C) return unimp_set_error(UNIMP_ERR_ARG, "gemm: null pointer");
  if (d->M <= 0 || d->N <= 0 || d->K <= 0) return unimp_set_error(UNIMP_ERR_SHAPE, "gemm: empty shape");
  int e;
  if ((e = check_operand(d->A, d->lda, d->a_kstrided, d->M)))
    return unimp_set_error(e, "gemm: operand base must be 16-B aligned, ld %% 8 == 0, k-strided ld >= roundup8(rows)");
  if (d->b_kstrided == 2) {                 // pre-packed B image (unimp_pack_b_bf16): no ld, 16-byte aligned
    if ((uintptr_t)d->B & 15) return unimp_set_error(UNIMP_ERR_ALIGN, "gemm: packed B must be 16-B aligned");
  } else if ((e = check_operand(d->B, d->ldb, d->b_kstrided, d->N)))
    return unimp_set_error(e, "gemm: operand base must be 16-B aligned, ld %% 8 == 0, k-strided ld >= roundup8(rows)");
  if (!d->a_kstrided && d->lda < ((d->K + 7) & ~7)) return unimp_set_error(UNIMP_ERR_SHAPE, "gemm: lda < roundup8(K)");
  if (!d->b_kstrided && d->ldb < ((d->K + 7) & ~7)) return unimp_set_error(UNIMP_ERR_SHAPE, "gemm: ldb < roundup8(K)");
  // the 8-bit stored derivative is served by the specialised epilogue kinds only (gemm_tile.h epi_kind): the MLP blocks' two forms
  const bool pre_u8 = d->pre && d->pre_deriv == 2, aux_u8 = d->aux && d->dact == ACT_DERIV_U8;
  if (pre_u8 || aux_u8) {
    if ((d->N & 7) || (d->ldc & 7) || d->accumulate || d->res || (pre_u8 && ((d->ldpre & 7) || !d->act || d->aux || d->gate)) ||
        (aux_u8 && ((d->ldaux & 7) || d->act || d->pre || d->bias)))
      return unimp_set_error(UNIMP_ERR_UNSUPPORTED, "gemm: the uint8 derivative needs N, ld % 8 == 0 and one of the two forms: "
                             "[bias +] act with the derivative as second output, or plain [x gate] times the stored derivative");
  }
  if (d->dact == ACT_DERIV_U8 && !d->aux) return unimp_set_error(UNIMP_ERR_ARG, "gemm: dact = 6 without an aux operand");
  return 0;
}

// ---- B operand pre-packed for the ping-pong kernels (frozen weights): out[(nt * nh + h) * 64 + lane][j] = B[n = 16 nt + (lane & 15)]
// [k = 32 h + 8 (lane >> 4) + j], zero beyond (N, K); nt < ceil(N / 16) rounded up to a multiple of 16 tiles (a 256-column block),
// nh = ceil(K / 32).  B[n][k] = X[n * ld + k] (k-contiguous source) or X[k * ld + n] (k-strided source).
__global__ void pack_b_kernel(const bf16* __restrict__ X, long ld, int N, int K, int ks, bf16* __restrict__ out, long ntiles, long nh) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ntiles * nh * 64) return;
  int lane = i & 63; long t = i >> 6;
  long h = t % nh, nt = t / nh;
  int n = (int)(nt * 16 + (lane & 15)), k0 = (int)(h * 32 + 8 * (lane >> 4));
  bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
  if (n < N) {
    if (!ks && k0 + 8 <= K) v = *(const bf16x8*)(X + (long)n * ld + k0);
    else
#pragma unroll
      for (int j = 0; j < 8; ++j) if (k0 + j < K) v[j] = ks ? X[(long)(k0 + j) * ld + n] : X[(long)n * ld + k0 + j];
  }
  *(bf16x8*)(out + i * 8) = v;
}

extern "C" int64_t unimp_pack_b_bytes(int N, int K) {
  long ntiles = (((long)N + 255) / 256) * 16, nh = ((long)K + 31) / 32;
  return ntiles * nh * 1024;
}

extern "C" int unimp_pack_b_bf16(const void* X, int64_t ld, int N, int K, int kstrided, void* out, void* stream) {
  if (!X || !out) return unimp_set_error(UNIMP_ERR_ARG, "pack_b: null pointer");
  if (N <= 0 || K <= 0) return unimp_set_error(UNIMP_ERR_SHAPE, "pack_b: empty shape");
  int e = check_operand(X, ld, kstrided, N);
  if (e || ((uintptr_t)out & 15)) return unimp_set_error(UNIMP_ERR_ALIGN, "pack_b: 16-B aligned pointers, ld %% 8 == 0");
  long ntiles = (((long)N + 255) / 256) * 16, nh = ((long)K + 31) / 32;
  long total = ntiles * nh * 64;
  hipLaunchKernelGGL(pack_b_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16*)X, (long)ld, N, K, kstrided,
                     (bf16*)out, ntiles, nh);
  return unimp_check_launch("pack_b");
}

static void launch_v1(const unimp_gemm_desc* d, void* stream, int splits = 1, float* slabs = nullptr) {
  GemmParams p;
  p.A = (const bf16*)d->A; p.B = (const bf16*)d->B; p.C = d->C;
  p.M = d->M; p.N = d->N; p.K = d->K; p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc;
  p.bias = (const bf16*)d->bias; p.res = (const bf16*)d->res; p.ldres = d->ldres;
  p.aux = (const bf16*)d->aux; p.ldaux = d->ldaux; p.pre = (bf16*)d->pre; p.ldpre = d->ldpre;
  p.gate = (const bf16*)d->gate; p.alpha = d->alpha; p.act = d->act; p.dact = d->dact;
  p.out_f32 = d->out_f32; p.accumulate = d->accumulate; p.pre_deriv = d->pre_deriv;
  p.nbm = (d->M + BM - 1) / BM; p.nbn = (d->N + BN - 1) / BN;
  p.ksplit = 0;
  if (splits > 1) {
    p.ksplit = ((d->K + splits - 1) / splits + 63) & ~63;
    p.C = slabs; p.ldc = d->N; p.out_f32 = 1; p.accumulate = 0; p.alpha = 1.f;
    p.bias = nullptr; p.res = nullptr; p.aux = nullptr; p.pre = nullptr; p.gate = nullptr; p.act = 0; p.dact = 0; p.pre_deriv = 0;
  }
  dim3 grid(p.nbm * p.nbn, splits > 1 ? (d->K + p.ksplit - 1) / p.ksplit : 1), block(256);
  size_t lds = 2 * STAGE_BYTES;
  hipStream_t s = (hipStream_t)stream;
  if (!d->a_kstrided && !d->b_kstrided) hipLaunchKernelGGL((gemm_bf16_kernel<false, false>), grid, block, lds, s, p);
  else if (!d->a_kstrided && d->b_kstrided) hipLaunchKernelGGL((gemm_bf16_kernel<false, true>), grid, block, lds, s, p);
  else if (d->a_kstrided && d->b_kstrided) hipLaunchKernelGGL((gemm_bf16_kernel<true, true>), grid, block, lds, s, p);
  else hipLaunchKernelGGL((gemm_bf16_kernel<true, false>), grid, block, lds, s, p);
}

// default choice when the caller does not autotune: the ping-pong kernel once there are enough rows for 256-row tiles,
// tile width by round quantisation over the 256 CUs; the 128x128 kernel otherwise.
static int auto_variant(const unimp_gemm_desc* d) {
  if (d->b_kstrided == 2) return d->N >= 256 ? UNIMP_GEMM_PP256 : UNIMP_GEMM_PP128;
  if (skinny_ok(d)) return UNIMP_GEMM_SKINNY;
  if (d->M < 1024 || d->N < 128 || d->K < 128) return UNIMP_GEMM_V1;
  long nbm = (d->M + 255) / 256;
  long t256 = nbm * ((d->N + 255) / 256), t128 = nbm * ((d->N + 127) / 128);
  auto eff = [](long tiles) { long rounds = (tiles + 255) / 256; return (double)tiles / (double)(rounds * 256); };
  bool wide = d->N >= 256 && eff(t256) >= eff(t128) * 0.88;
  // round 5: the whole-row-A one-set builds with the peeled K loop (forms they do not serve fall through to the generic one-set kernel inside)
  return wide ? UNIMP_GEMM_PP256A : UNIMP_GEMM_PP128A;
}

extern "C" int unimp_gemm_bf16_variant(const unimp_gemm_desc* d, int variant, void* stream) {
  int e = validate(d);
  if (e) return e;
  if (variant == UNIMP_GEMM_AUTO) variant = auto_variant(d);
  if (d->ln_gamma && variant != UNIMP_GEMM_SKINNY)
    return unimp_set_error(UNIMP_ERR_UNSUPPORTED, "gemm: a fused LayerNorm of the A rows (ln_gamma) is served by the decode-row kernel only (variant skinny, M <= 16)");
  if (d->b_kstrided == 2 && variant != UNIMP_GEMM_PP256 && variant != UNIMP_GEMM_PP128 && variant != UNIMP_GEMM_DW)
    return unimp_set_error(UNIMP_ERR_UNSUPPORTED, "gemm: a pre-packed B operand is served by the ping-pong kernels (variants pp256 / pp128) and dw only");
  if (((d->pre && d->pre_deriv == 2) || d->dact == ACT_DERIV_U8) &&
      (variant == UNIMP_GEMM_DMA256 || variant == UNIMP_GEMM_DMA128 || variant == UNIMP_GEMM_W4 || variant == UNIMP_GEMM_SKINNY))   // (w4x has the kinds)
    return unimp_set_error(UNIMP_ERR_UNSUPPORTED, "gemm: the uint8 derivative is served by variants v1 / pp256 / pp128 / w8 / pp256p (kernels with the specialised epilogue kinds)");
  if (d->rope_rot) {
    if (d->rope_rot < 0 || (d->rope_rot & 7) || d->rope_hd <= 0 || (d->rope_hd & 7) || d->rope_rot > d->rope_hd || d->rope_L <= 0 ||
        d->rope_period <= 0 || d->rope_period % d->rope_hd || d->rope_span % d->rope_hd || d->rope_span > d->rope_period || !(d->rope_log2_base > 0.f))
      return unimp_set_error(UNIMP_ERR_ARG, "gemm: rotary epilogue needs rot % 8 == 0 <= hd, hd % 8 == 0, period / span multiples of hd, L > 0, base > 1");
    if (d->M >= (1 << 24)) return unimp_set_error(UNIMP_ERR_SHAPE, "gemm: rotary epilogue needs M < 2^24");
    bool kern = (variant == UNIMP_GEMM_PP256 || variant == UNIMP_GEMM_PP256P || variant == UNIMP_GEMM_PP256X || variant == UNIMP_GEMM_PP256PX || variant == UNIMP_GEMM_PP256A || variant == UNIMP_GEMM_PP256B ||
                 variant == UNIMP_GEMM_W4X || variant == UNIMP_GEMM_W4X_S1 || variant == UNIMP_GEMM_W4X_PF) && !d->a_kstrided && d->b_kstrided != 2;
    if (!kern || d->res || d->aux || d->pre || d->act || d->dact || d->accumulate || d->gate || d->out_f32 || (d->N & 7) || (d->ldc & 7) || d->M < 256 || d->N < 128)
      return unimp_set_error(UNIMP_ERR_UNSUPPORTED, "gemm: the rotary epilogue is served by variants pp256 / pp256p (and their one-set forms) with a k-contiguous A operand, an unpacked B operand and a plain (alpha, bias) bf16 epilogue, N % 8 == 0, ldc % 8 == 0");
  }
  switch (variant) {
    case UNIMP_GEMM_V1: launch_v1(d, stream); break;
    case UNIMP_GEMM_DMA256: unimp_gemm2_launch(d, 256, stream); break;
    case UNIMP_GEMM_DMA128: unimp_gemm2_launch(d, 128, stream); break;
    case UNIMP_GEMM_PP256: unimp_gemm3_launch(d, 256, stream); break;
    case UNIMP_GEMM_PP128: unimp_gemm3_launch(d, 128, stream); break;
    case UNIMP_GEMM_W4: unimp_gemm4_launch(d, stream); break;
    case UNIMP_GEMM_W8: unimp_gemm5_launch(d, stream); break;
    case UNIMP_GEMM_PP256P: unimp_gemm6_launch(d, stream); break;
    case UNIMP_GEMM_PP256X: unimp_gemm3x_launch(d, 256, stream); break;
    case UNIMP_GEMM_PP128X: unimp_gemm3x_launch(d, 128, stream); break;
    case UNIMP_GEMM_PP256PX: unimp_gemm6x_launch(d, stream); break;
    case UNIMP_GEMM_PP256A: unimp_gemm3a_launch(d, 256, stream); break;
    case UNIMP_GEMM_PP128A: unimp_gemm3a_launch(d, 128, stream); break;
    case UNIMP_GEMM_PP256B: unimp_gemm3b_launch(d, 256, stream); break;
    case UNIMP_GEMM_PP128B: unimp_gemm3b_launch(d, 128, stream); break;
    case UNIMP_GEMM_W4X: case UNIMP_GEMM_W4X_S1: case UNIMP_GEMM_W4X_PF: {
      int ok = variant == UNIMP_GEMM_W4X ? unimp_gemm7_launch(d, stream) : variant == UNIMP_GEMM_W4X_S1 ? unimp_gemm7o_launch(d, stream) : unimp_gemm7p_launch(d, stream);
      if (!ok) return unimp_set_error(UNIMP_ERR_UNSUPPORTED, "gemm: variant w4x serves a k-contiguous A operand, an unpacked B operand, K % 64 == 0, K >= 128, operands below 4 GiB");
      break; }
    case UNIMP_GEMM_DW:
      if (!unimp_gemm9_launch(d, stream)) return unimp_set_error(UNIMP_ERR_UNSUPPORTED, "gemm: variant dw has no rotary epilogue; its packed-B form needs K % 32 == 0, K >= 96");
      break;
    case UNIMP_GEMM_SKINNY:
      if (!skinny_ok(d)) return unimp_set_error(UNIMP_ERR_UNSUPPORTED, "gemm: the skinny kernel needs M <= 64, k-contiguous operands, K %% 64 == 0");
      if (d->ln_gamma && !(skinny2_on() && skinny2_ok(d)))
        return unimp_set_error(UNIMP_ERR_UNSUPPORTED, "gemm: a fused LayerNorm (ln_gamma) needs M <= 16 and K = 512 ... 3072 or 4096 in whole 512s (unimp_gemm_skinny_ln_ok)");
      launch_skinny(d, stream); break;
    default: return unimp_set_error(UNIMP_ERR_ARG, "gemm: unknown variant");
  }
  return unimp_check_launch("gemm");
}

// ---- split-K for weight-gradient GEMMs whose output is far smaller than the chip (e.g. dW of a 512 x 2560 projection
// with K = all tokens): K slices -> f32 slabs [S][M][N] (plain stores), then one ordered reduction pass (reproducible,
// no atomics) that applies alpha * tanh(gate) and writes bf16 / f32.
__global__ void splitk_reduce_kernel(const float* __restrict__ slabs, int S, long MN, int N, void* __restrict__ C, long ldc,
                                     int out_f32, float alpha, const bf16* __restrict__ gate, int accumulate) {
  long i4 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i4 >= MN) return;
  float g = alpha * (gate ? tanhf(bf2f(*gate)) : 1.f);
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < S; ++s) a += *(const f32x4*)(slabs + (long)s * MN + i4);
  a *= g;
  long m = i4 / N; int n = (int)(i4 - m * N);
  if (out_f32) {
    float* d = (float*)C + m * ldc + n;
    if (accumulate) a += *(const f32x4*)d;
    *(f32x4*)d = a;
  } else {
    bf16* d = (bf16*)C + m * ldc + n;
    if (accumulate) { bf16x4 c = *(const bf16x4*)d; a += f32x4{bf2f(c[0]), bf2f(c[1]), bf2f(c[2]), bf2f(c[3])}; }
    bf16x4 o = {f2bf(a[0]), f2bf(a[1]), f2bf(a[2]), f2bf(a[3])};
    *(bf16x4*)d = o;
  }
}

extern "C" int unimp_gemm_bf16_splitk(const unimp_gemm_desc* d, int splits, float* slabs, void* stream) {
  int e = validate(d);
  if (e) return e;
  if (splits < 2 || !slabs) return unimp_set_error(UNIMP_ERR_ARG, "gemm_splitk: need splits >= 2 and a slab workspace");
  if (d->bias || d->res || d->aux || d->pre || d->act || d->dact || d->rope_rot)
    return unimp_set_error(UNIMP_ERR_UNSUPPORTED, "gemm_splitk: only alpha / gate (/ accumulate) epilogues");
  if ((d->N & 3) || (d->ldc & 3)) return unimp_set_error(UNIMP_ERR_SHAPE, "gemm_splitk: N and ldc must be multiples of 4");
  // 256 x 256 ping-pong tiles once the output holds a few of them (twice the 128 x 128 kernel's rate); 128 x 128 tiles otherwise
  if (d->M >= 256 && d->N >= 256) unimp_gemm3_launch_splitk(d, 256, splits, slabs, stream);
  else launch_v1(d, stream, splits, slabs);
  int ks = ((d->K + splits - 1) / splits + 63) & ~63;
  int S = (d->K + ks - 1) / ks;
  long MN = (long)d->M * d->N;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((MN / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, slabs, S, MN, d->N,
                     d->C, (long)d->ldc, d->out_f32, d->alpha, (const bf16*)d->gate, d->accumulate);
  return unimp_check_launch("gemm_splitk");
}

extern "C" int unimp_gemm_bf16(const unimp_gemm_desc* d, void* stream) {
  return unimp_gemm_bf16_variant(d, UNIMP_GEMM_AUTO, stream);
}
