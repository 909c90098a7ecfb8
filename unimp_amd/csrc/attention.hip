// Flash-style attention forward / backward on v_mfma_f32_16x16x32_bf16 for gfx950 (wave64).
//
// One code path serves every attention in the Flamingo step: ViT (non-causal, S=257, hd 64), the causal LM
// (hd 80 / 64 / 128, key-padding via kv_len), the Perceiver resampler (64 latents x 320 keys) and the gated
// cross-attention (each text token attends only the 64 latents of its own image: MASK_SEGMENT -- the
// (L x T*64) masked score matrix of the reference is never formed, only the overlapping 64-key tiles run).
//
// Orientation trick (cdna guide §3 "accumulator tile as the next MFMA's operand"): scores are produced
// TRANSPOSED, S^T = K·Q^T, so a lane holds 16 keys of ONE query row (q = lane&15).  Row max / row sum are
// in-lane + two shuffles, and the probabilities already sit in the register layout the next MFMA wants as
// its B operand (contraction over keys) if V is fetched in the matching permuted key order
//     slot (g = lane>>4, j) of k-step ks  <->  key 16*(2ks + (j>>2)) + 4g + (j&3),
// which is two ds_read_b64_tr_b16 of 4 consecutive V rows each.  P never touches LDS.
// The backward uses the same trick for dQ (contraction over keys) and, with S = Q·K^T un-transposed, for
// dK / dV (contraction over queries, P and dS stay in registers, Q / dO fetched transposed from LDS).
// dQ and dK/dV are separate kernels: 7 instead of 5 MFMA products, but no atomics and bit-reproducible.
#include <stdlib.h>
#include "common.h"
#include "unimp_hip.h"

#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f
#define EXP2(x) __builtin_amdgcn_exp2f(x)

#include "attention_params.h"
// LDS row pitch of the register-staged Q / K / V / dO tiles = row bytes + ATTN_PAD.  The transposed fragment reads (lfrag_tr_perm:
// ds_read_b64_tr_b16, 32 lanes = 8 rows x 8-byte pieces per LDS cycle) are conflict-free when the pitch in dwords is 8 mod 16 -- the 8 rows
// then start 8 banks apart: row bytes rounded up to 32 mod 64 gives 40 / 56 / 72 dwords at head dims 64 / 96 / 128.  With +16 (rounds 1-2) the pitches 36 / 52 / 68
// put rows (0,5), (1,6), (2,7) on overlapping banks: PMC showed 18 % of the dK/dV kernel's CU cycles as LDS bank conflicts.  The row reads
// (ds_read_b128, 16 rows per group) stay conflict-free at both pitches.
#ifdef ATTN_PITCH_R2      // A/B: the pitch of rounds 1-2
__host__ __device__ constexpr int attn_pitch(int row_bytes) { return row_bytes + 16; }
#else                     // smallest pitch >= row_bytes with pitch % 64 == 32 (dwords: 8 mod 16); rows of 160 bytes (head dim 80 as V) need no pad.
// Head dim 64 keeps +16: at 160 bytes the dK/dV kernel loses its eighth workgroup per CU (21.5 instead of 19.9 KiB of LDS) and measured
// 7-14 % SLOWER (ViT / Perceiver / cross-attention shapes); head dims 80 and 128 gain 2.4 / 3.3 % on the backward (tools/gpu/r3q.sh)
__host__ __device__ constexpr int attn_pitch(int row_bytes) { return row_bytes == 128 ? 144 : row_bytes + (32 - row_bytes % 64 + 64) % 64; }
#endif

// key range [lo, hi) attended by query row `qr` of batch b
__device__ __forceinline__ void key_range(const AttnP& p, int b, int qr, int& lo, int& hi) {
  lo = 0; hi = 0;
  if (qr >= p.Sq) return;
  int kvl = p.kv_len ? p.kv_len[b] : p.Sk;
  if (p.mask_mode == UNIMP_MASK_NONE) { hi = kvl; }
  else if (p.mask_mode == UNIMP_MASK_CAUSAL) { hi = min(qr + 1, kvl); }
  else { int t = p.seg[(long)b * p.SqS + qr]; if (t > 0) { lo = (t - 1) * p.seg_len; hi = min(t * p.seg_len, p.Sk); } }
}

// the same with the batch row's key count already in a register: inside a tile loop the p.kv_len[b] of key_range() is a scalar load +
// s_waitcnt lgkmcnt(0) per iteration (the compiler cannot hoist it past the loop's stores)
__device__ __forceinline__ void key_range_kvl(const AttnP& p, int b, int qr, int kvl, int& lo, int& hi) {
  lo = 0; hi = 0;
  if (qr >= p.Sq) return;
  if (p.mask_mode == UNIMP_MASK_NONE) { hi = kvl; }
  else if (p.mask_mode == UNIMP_MASK_CAUSAL) { hi = min(qr + 1, kvl); }
  else { int t = p.seg[(long)b * p.SqS + qr]; if (t > 0) { lo = (t - 1) * p.seg_len; hi = min(t * p.seg_len, p.Sk); } }
}

// cooperative [rows x DPAD] bf16 tile load into LDS with row stride STR bytes; zero fill outside (nrows, D)
template <int ROWS, int DPAD, int STR>
__device__ __forceinline__ void load_tile(char* lds, const bf16* __restrict__ base, long row_stride, int row0, int nrows, int D) {
  constexpr int CPR = DPAD / 8;
  for (int q = threadIdx.x; q < ROWS * CPR; q += 256) {
    int r = q / CPR, c = q - r * CPR;
    u32x4 val = u32x4{0, 0, 0, 0};
    if (row0 + r < nrows && c * 8 < D) val = *(const u32x4*)(base + (long)(row0 + r) * row_stride + c * 8);
    *(u32x4*)(lds + r * STR + c * 16) = val;
  }
}

// KC fragment: lane holds X[r0 + (l&15)][32*ks + 8*(l>>4) + j]
template <int STR>
__device__ __forceinline__ bf16x8 lfrag_kc(const char* tile, int r0, int ks) {
  int l = lane_id();
  return *(const bf16x8*)(tile + (r0 + (l & 15)) * STR + (ks * 32 + (l >> 4) * 8) * 2);
}
// transposed fragment in the permuted contraction order: lane (g,i=l&15) gets
//   X[rbase + 4g + {0..3}][c0 + i]  ++  X[rbase + 16 + 4g + {0..3}][c0 + i]
template <int STR>
__device__ __forceinline__ bf16x8 lfrag_tr_perm(const char* tile, int rbase, int c0) {
  int l = lane_id();
  int g = l >> 4, qq = (l >> 2) & 3, pp = l & 3;
  const char* a = tile + (rbase + 4 * g + qq) * STR + (c0 + 4 * pp) * 2;
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, a));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, a + 16 * STR));
  union { struct { s16x4 a, b; } s; bf16x8 v; } u;
  u.s.a = lo; u.s.b = hi;
  return u.v;
}

__device__ __forceinline__ bf16x8 gfrag(const bf16* __restrict__ base, long row_stride, int row, int nrows, int ks, int D) {
  int l = lane_id();
  int d = ks * 32 + (l >> 4) * 8;
  bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
  if (row < nrows && d < D) return *(const bf16x8*)(base + (long)row * row_stride + d);
  return z;
}

__device__ __forceinline__ bf16x8 pack8(const f32x4& a, const f32x4& b) {
  bf16x8 r = {f2bf(a[0]), f2bf(a[1]), f2bf(a[2]), f2bf(a[3]), f2bf(b[0]), f2bf(b[1]), f2bf(b[2]), f2bf(b[3])};
  return r;
}

// block-uniform tile range of keys needed by query rows [q_first, q_last]
__device__ __forceinline__ void block_key_tiles(const AttnP& p, int b, int q_first, int q_last, int& kt_lo, int& kt_hi) {
  q_last = min(q_last, p.Sq - 1);
  int kvl = p.kv_len ? p.kv_len[b] : p.Sk;
  if (p.mask_mode == UNIMP_MASK_NONE) { kt_lo = 0; kt_hi = (kvl + 63) >> 6; }
  else if (p.mask_mode == UNIMP_MASK_CAUSAL) { kt_lo = 0; kt_hi = (min(q_last + 1, kvl) + 63) >> 6; }
  else {
    int t0 = p.seg[(long)b * p.SqS + q_first], t1 = p.seg[(long)b * p.SqS + q_last];
    if (t1 == 0) { kt_lo = 0; kt_hi = 0; }
    else { kt_lo = (max(t0 - 1, 0) * p.seg_len) >> 6; kt_hi = (min(t1 * p.seg_len, p.Sk) + 63) >> 6; }
  }
}

// Register-staged tile copy, split so the global loads of tile t+1 fly under the MFMAs of tile t (T14), with all the
// index / bounds arithmetic hoisted out of the tile loop:
//   init(): per-thread byte offsets of its 16-byte chunks inside a tile (global and LDS side), computed once;
//   g2r(tile_base, rows_left): the steady state is `scalar tile base + per-lane constant`; rows beyond the tensor are
//        CLAMPED to the last valid row (they only meet masked probabilities, never garbage: 0 * finite = 0);
//   r2s(lds): plain 16-byte LDS stores.  Pad columns d in [D, DPAD) are never written: the caller zero-fills LDS once.
template <int ROWS, int DPAD, int STR>
struct TileCopy {
  static constexpr int CPR = DPAD / 8, N = (ROWS * CPR + 255) / 256;
  uint32_t goff[N], loff[N];
  int row[N];
  u32x4 v[N];
  long stride_b;
  __device__ __forceinline__ void init(long row_stride, int D) {
    stride_b = row_stride * 2;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      int q = threadIdx.x + 256 * i;
      int r = q / CPR, c = q - r * CPR;
      bool ok = q < ROWS * CPR && c * 8 < D;
      row[i] = ok ? r : -1;
      goff[i] = (uint32_t)(r * stride_b + c * 16);
      loff[i] = (uint32_t)(r * STR + c * 16);
    }
  }
  __device__ __forceinline__ void g2r(const bf16* __restrict__ tile_base, int rows_left) {
    const char* tb = (const char*)tile_base;
    if (rows_left >= ROWS) {
#pragma unroll
      for (int i = 0; i < N; ++i) if (row[i] >= 0) v[i] = *(const u32x4*)(tb + goff[i]);
    } else {
#pragma unroll
      for (int i = 0; i < N; ++i)
        if (row[i] >= 0) { int over = max(row[i] - (rows_left - 1), 0); v[i] = *(const u32x4*)(tb + goff[i] - over * stride_b); }
    }
  }
  __device__ __forceinline__ void r2s(char* lds) const {
#pragma unroll
    for (int i = 0; i < N; ++i) if (row[i] >= 0) *(u32x4*)(lds + loff[i]) = v[i];
  }
};

__device__ __forceinline__ void lds_zero(char* smem, int bytes) {
  for (int i = threadIdx.x * 16; i < bytes; i += 256 * 16) *(u32x4*)(smem + i) = u32x4{0, 0, 0, 0};
  __syncthreads();
}

// ------------------------------------------------------------------------------------------- forward
// block = 4 waves x 32 query rows (two 16-row MFMA blocks per wave share every K / V fragment read); 64-key tiles,
// two LDS stages, one barrier per tile.
template <int DQK, int DV, bool ALIBI>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnP p) {
  constexpr int KSTR = attn_pitch(DQK * 2), VSTR = attn_pitch(DV * 2), NKS = DQK / 32, ND = DV / 16;
  constexpr int STAGE = 64 * KSTR + 64 * VSTR;
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
  int b = blockIdx.z, h = blockIdx.y;
  if (!attn_varlen(p, b) || (int)blockIdx.x * 128 >= p.Sq) return;
  int wave = threadIdx.x >> 6, l = lane_id(), g = l >> 4;
  int q0 = blockIdx.x * 128 + wave * 32;
  const bf16* qb = p.q + b * p.q_bs + h * p.q_hs;
  const bf16* kb = p.k + b * p.k_bs + h * p.k_hs;
  const bf16* vb = p.v + b * p.v_bs + h * p.v_hs;
  bf16x8 qf[2][NKS];
  int lo[2], hi[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    int qr = q0 + u * 16 + (l & 15);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) qf[u][ks] = gfrag(qb, p.q_ss, qr, p.Sq, ks, p.D);
    key_range(p, b, qr, lo[u], hi[u]);
  }
  int kt_lo, kt_hi;
  block_key_tiles(p, b, blockIdx.x * 128, blockIdx.x * 128 + 127, kt_lo, kt_hi);
  float sc2 = p.scale * LOG2E;
  const float ab = ALIBI ? p.alibi[h] / p.scale : 0.f;
  float m[2] = {-INFINITY, -INFINITY}, lsum[2] = {0.f, 0.f};
  f32x4 o[2][ND];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int nd = 0; nd < ND; ++nd) o[u][nd] = f32x4{0.f, 0.f, 0.f, 0.f};

  TileCopy<64, DQK, KSTR> rk; TileCopy<64, DV, VSTR> rv;
  rk.init(p.k_ss, p.D); rv.init(p.v_ss, p.D);
  if (DQK != DV) lds_zero(smem, 2 * STAGE);              // pad columns [D, DQK) of the K images must read as zeros
  if (kt_lo < kt_hi) {
    rk.g2r(kb + (long)kt_lo * 64 * p.k_ss, p.Sk - kt_lo * 64);
    rv.g2r(vb + (long)kt_lo * 64 * p.v_ss, p.Sk - kt_lo * 64);
    rk.r2s(smem);
    rv.r2s(smem + 64 * KSTR);
  }
  __syncthreads();
  for (int kt = kt_lo; kt < kt_hi; ++kt) {
    char* ks_t = smem + ((kt - kt_lo) & 1) * STAGE;
    char* vs_t = ks_t + 64 * KSTR;
    char* nx = smem + ((kt - kt_lo + 1) & 1) * STAGE;
    bool more = kt + 1 < kt_hi;
    if (more) {
      rk.g2r(kb + (long)(kt + 1) * 64 * p.k_ss, p.Sk - (kt + 1) * 64);
      rv.g2r(vb + (long)(kt + 1) * 64 * p.v_ss, p.Sk - (kt + 1) * 64);
    }
    if (q0 < p.Sq) {        // wave-uniform: a wave whose rows all lie beyond Sq (S = 257 -> 384 padded rows) only helps with the loads
    f32x4 s[2][4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      s[0][nt] = f32x4{0.f, 0.f, 0.f, 0.f}; s[1][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        bf16x8 kf = lfrag_kc<KSTR>(ks_t, nt * 16, ks);
        s[0][nt] = MFMA16(kf, qf[0][ks], s[0][nt]);
        s[1][nt] = MFMA16(kf, qf[1][ks], s[1][nt]);
      }
    }
    if (ALIBI) {                        // ALiBi (its own kernel instantiation: the plain kernels keep their registers)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float kb = ab * (float)(kt * 64 + nt * 16 + g * 4 + r);
          s[0][nt][r] += kb; s[1][nt][r] += kb;
        }
    }
    bf16x8 pf[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float mloc = -INFINITY;
      bool full = __all(kt * 64 >= lo[u] && kt * 64 + 64 <= hi[u]);      // interior tile: no masking work
      if (full) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) mloc = fmaxf(mloc, s[u][nt][r]);
      } else {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            int key = kt * 64 + nt * 16 + g * 4 + r;
            float v = (key >= lo[u] && key < hi[u]) ? s[u][nt][r] : -INFINITY;
            s[u][nt][r] = v;
            mloc = fmaxf(mloc, v);
          }
      }
      mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
      mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
      float mnew = fmaxf(m[u], mloc * sc2);                 // running max in the scaled (log2) domain; sc2 > 0
      float muse = (mnew == -INFINITY) ? 0.f : mnew;
      float alpha = EXP2(m[u] - muse);
      float rs = 0.f;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { float e = EXP2(fmaf(s[u][nt][r], sc2, -muse)); s[u][nt][r] = e; rs += e; }
      rs += __shfl_xor(rs, 16, 64);
      rs += __shfl_xor(rs, 32, 64);
      lsum[u] = lsum[u] * alpha + rs;
      m[u] = mnew;
      pf[u][0] = pack8(s[u][0], s[u][1]); pf[u][1] = pack8(s[u][2], s[u][3]);
#pragma unroll
      for (int nd = 0; nd < ND; ++nd) o[u][nd] *= alpha;
    }
#pragma unroll
    for (int nd = 0; nd < ND; ++nd)
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        bf16x8 vf = lfrag_tr_perm<VSTR>(vs_t, k2 * 32, nd * 16);
        o[0][nd] = MFMA16(vf, pf[0][k2], o[0][nd]);
        o[1][nd] = MFMA16(vf, pf[1][k2], o[1][nd]);
      }
    }
    if (more) {
      rk.r2s(nx);
      rv.r2s(nx + 64 * KSTR);
    }
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    int qr = q0 + u * 16 + (l & 15);
    if (qr < p.Sq) {
      float inv = lsum[u] > 0.f ? 1.f / lsum[u] : 0.f;
      bf16* ob = p.o + b * p.o_bs + (long)qr * p.o_ss + h * p.o_hs;
#pragma unroll
      for (int nd = 0; nd < ND; ++nd) {
        bf16x4 w = {f2bf(o[u][nd][0] * inv), f2bf(o[u][nd][1] * inv), f2bf(o[u][nd][2] * inv), f2bf(o[u][nd][3] * inv)};
        *(bf16x4*)(ob + nd * 16 + g * 4) = w;
      }
      if (g == 0 && p.lse) p.lse[((long)b * p.H + h) * p.SqS + qr] = lsum[u] > 0.f ? (m[u] + log2f(lsum[u])) * LN2 : -INFINITY;
    }
  }
}

// ------------------------------------------------------------------------------------------- delta = rowsum(dO * O)
__global__ void attn_delta_kernel(AttnP p) {
  // 16 lanes per (b, h, q) row: lane c reads the 16-byte chunk c of O and dO (coalesced 160..256-byte rows), then a
  // 16-lane butterfly; 4 rows per wave-instruction.
  long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  int c = threadIdx.x & 15;
  long n = (long)p.B * p.H * p.Sq;
  float acc = 0.f;
  if (i < n) {
    int h = i % p.H; long t = i / p.H; int qr = t % p.Sq; int b = t / p.Sq;          // heads fastest: adjacent rows are adjacent in memory
    if (c * 8 < p.D) {
      bf16x8 a = *(const bf16x8*)(p.o + b * p.o_bs + (long)qr * p.o_ss + h * p.o_hs + c * 8);
      bf16x8 d = *(const bf16x8*)(p.d_o + b * p.do_bs + (long)qr * p.do_ss + h * p.do_hs + c * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc += bf2f(a[j]) * bf2f(d[j]);
    }
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (i < n && c == 0) {
    int h = i % p.H; long t = i / p.H; int qr = t % p.Sq; int b = t / p.Sq;
    p.delta[((long)b * p.H + h) * p.SqS + qr] = acc;
  }
}

// ------------------------------------------------------------------------------------------- dQ
// same geometry as the forward: 4 waves x 32 query rows, 64-key tiles double-buffered; K and V tiles share one image
// layout (row reads for S and dP, transposed K reads for dQ).
template <int DQK, int DV, bool ALIBI>
__global__ __launch_bounds__(256) void attn_dq_kernel(AttnP p) {
  constexpr int STR = attn_pitch(DQK * 2), NKS = DQK / 32, ND = DV / 16;
  constexpr int STAGE = 2 * 64 * STR;
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
  int b = blockIdx.z, h = blockIdx.y;
  int wave = threadIdx.x >> 6, l = lane_id(), g = l >> 4;
  int q0 = blockIdx.x * 128 + wave * 32;
  const bf16* qb = p.q + b * p.q_bs + h * p.q_hs;
  const bf16* kb = p.k + b * p.k_bs + h * p.k_hs;
  const bf16* vb = p.v + b * p.v_bs + h * p.v_hs;
  const bf16* dob = p.d_o + b * p.do_bs + h * p.do_hs;
  bf16x8 qf[2][NKS], dof[2][NKS];
  int lo[2], hi[2];
  float lse2[2], dl[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    int qr = q0 + u * 16 + (l & 15);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) { qf[u][ks] = gfrag(qb, p.q_ss, qr, p.Sq, ks, p.D); dof[u][ks] = gfrag(dob, p.do_ss, qr, p.Sq, ks, p.D); }
    key_range(p, b, qr, lo[u], hi[u]);
    long sidx = ((long)b * p.H + h) * p.SqS + min(qr, p.Sq - 1);
    lse2[u] = p.lse[sidx] * LOG2E; dl[u] = p.delta[sidx];
  }
  int kt_lo, kt_hi;
  block_key_tiles(p, b, blockIdx.x * 128, blockIdx.x * 128 + 127, kt_lo, kt_hi);
  float sc2 = p.scale * LOG2E;
  const float ab = ALIBI ? p.alibi[h] / p.scale : 0.f;
  f32x4 dq[2][ND];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int nd = 0; nd < ND; ++nd) dq[u][nd] = f32x4{0.f, 0.f, 0.f, 0.f};

  TileCopy<64, DQK, STR> rk, rv;
  rk.init(p.k_ss, p.D); rv.init(p.v_ss, p.D);
  if (DQK != DV) lds_zero(smem, 2 * STAGE);
  if (kt_lo < kt_hi) {
    rk.g2r(kb + (long)kt_lo * 64 * p.k_ss, p.Sk - kt_lo * 64);
    rv.g2r(vb + (long)kt_lo * 64 * p.v_ss, p.Sk - kt_lo * 64);
    rk.r2s(smem);
    rv.r2s(smem + 64 * STR);
  }
  __syncthreads();
  for (int kt = kt_lo; kt < kt_hi; ++kt) {
    char* ks_t = smem + ((kt - kt_lo) & 1) * STAGE;
    char* vs_t = ks_t + 64 * STR;
    char* nx = smem + ((kt - kt_lo + 1) & 1) * STAGE;
    bool more = kt + 1 < kt_hi;
    if (more) {
      rk.g2r(kb + (long)(kt + 1) * 64 * p.k_ss, p.Sk - (kt + 1) * 64);
      rv.g2r(vb + (long)(kt + 1) * 64 * p.v_ss, p.Sk - (kt + 1) * 64);
    }
    if (q0 < p.Sq) {
    f32x4 s[2][4], dp[2][4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
      for (int u = 0; u < 2; ++u) { s[u][nt] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[u][nt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        bf16x8 kf = lfrag_kc<STR>(ks_t, nt * 16, ks), vf = lfrag_kc<STR>(vs_t, nt * 16, ks);
#pragma unroll
        for (int u = 0; u < 2; ++u) { s[u][nt] = MFMA16(kf, qf[u][ks], s[u][nt]); dp[u][nt] = MFMA16(vf, dof[u][ks], dp[u][nt]); }
      }
    }
    if (ALIBI) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float kb = ab * (float)(kt * 64 + nt * 16 + g * 4 + r);
          s[0][nt][r] += kb; s[1][nt][r] += kb;
        }
    }
    bf16x8 dsf[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      bool full = __all(kt * 64 >= lo[u] && kt * 64 + 64 <= hi[u]);      // interior tile: no masking work
      if (full) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float pr = EXP2(fmaf(s[u][nt][r], sc2, -lse2[u]));
            s[u][nt][r] = pr * (dp[u][nt][r] - dl[u]) * p.scale;
          }
      } else {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            int key = kt * 64 + nt * 16 + g * 4 + r;
            float pr = (key >= lo[u] && key < hi[u]) ? EXP2(fmaf(s[u][nt][r], sc2, -lse2[u])) : 0.f;
            s[u][nt][r] = pr * (dp[u][nt][r] - dl[u]) * p.scale;
          }
      }
      dsf[u][0] = pack8(s[u][0], s[u][1]); dsf[u][1] = pack8(s[u][2], s[u][3]);
    }
#pragma unroll
    for (int nd = 0; nd < ND; ++nd)
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        bf16x8 kt_f = lfrag_tr_perm<STR>(ks_t, k2 * 32, nd * 16);
        dq[0][nd] = MFMA16(kt_f, dsf[0][k2], dq[0][nd]);
        dq[1][nd] = MFMA16(kt_f, dsf[1][k2], dq[1][nd]);
      }
    }
    if (more) {
      rk.r2s(nx);
      rv.r2s(nx + 64 * STR);
    }
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    int qr = q0 + u * 16 + (l & 15);
    if (qr < p.Sq) {
      bf16* ob = p.dq + b * p.dq_bs + (long)qr * p.dq_ss + h * p.dq_hs;
#pragma unroll
      for (int nd = 0; nd < ND; ++nd) {
        bf16x4 w = {f2bf(dq[u][nd][0]), f2bf(dq[u][nd][1]), f2bf(dq[u][nd][2]), f2bf(dq[u][nd][3])};
        *(bf16x4*)(ob + nd * 16 + g * 4) = w;
      }
    }
  }
}

#ifdef ATTN_STAMP     // debug build (tools/stamp_attn.py): where a dK/dV block's time goes -- wave 0's s_memtime cycles per loop segment, summed over its tiles
__device__ unsigned long long attn_stamps[16384 * 12];
extern "C" int unimp_debug_attn_stamps(void* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(attn_stamps), sizeof(attn_stamps)); }
#define AT_NOW() __builtin_amdgcn_s_memtime()
#define AT_SEG(K_) do { unsigned long long t_ = AT_NOW(); at_acc[K_] += t_ - at_last; at_last = t_; } while (0)
#else
#define AT_SEG(K_) do {} while (0)
#endif
// ------------------------------------------------------------------------------------------- dK, dV
// block = 64 * KU keys: each wave owns 16 * KU keys (KU 16-key blocks whose K / V fragments live in registers) and sweeps the
// query tiles (32 rows, double-buffered Q / dO images + per-row lse / delta / key range); P and dS stay in registers.
template <int DQK, int DV, bool ALIBI, int KU = 2>
__global__ __launch_bounds__(256) void attn_dkv_kernel(AttnP p, int nx) {
#ifdef ATTN_STAMP
  unsigned long long at_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, at_t0 = __builtin_amdgcn_s_memrealtime(), at_c0 = AT_NOW(), at_last = 0;
#endif
  constexpr int KPW = 16 * KU, KPB = 4 * KPW;               // keys per wave (KU blocks of 16) and per workgroup
  constexpr int STR = attn_pitch(DQK * 2), NKS = DQK / 32, ND = DV / 16;
  constexpr int STAGE = 2 * 32 * STR + 32 * 16 + 16;
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
  // 1-D launch decoded XCD-aware (attention2.hip a2_decode): the key blocks of one (batch, head) share an XCD's L2, first
  // key block (the one that sees the most query tiles under the causal mask) first
  int id_ = xcd_remap(blockIdx.x, nx * p.H * p.B);
  int kblk = id_ % nx, t_ = id_ / nx;
  int h = t_ % p.H, b = t_ / p.H;
  if (!attn_varlen(p, b) || kblk * KPB >= p.Sk) return;       // packed rows: the grid covers the longest sequence
  int wave = threadIdx.x >> 6, l = lane_id(), g = l >> 4;
  int key0 = kblk * KPB + wave * KPW;
  const bf16* qb = p.q + b * p.q_bs + h * p.q_hs;
  const bf16* kb = p.k + b * p.k_bs + h * p.k_hs;
  const bf16* vb = p.v + b * p.v_bs + h * p.v_hs;
  const bf16* dob = p.d_o + b * p.do_bs + h * p.do_hs;
  bf16x8 kf[KU][NKS], vf[KU][NKS];
  f32x4 dk[KU][ND], dv[KU][ND];
#pragma unroll
  for (int u = 0; u < KU; ++u) {
    int key = key0 + u * 16 + (l & 15);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) { kf[u][ks] = gfrag(kb, p.k_ss, key, p.Sk, ks, p.D); vf[u][ks] = gfrag(vb, p.v_ss, key, p.Sk, ks, p.D); }
#pragma unroll
    for (int nd = 0; nd < ND; ++nd) { dk[u][nd] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[u][nd] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  }
  float sc2 = p.scale * LOG2E;
  const float ab = ALIBI ? p.alibi[h] / p.scale : 0.f;
  int nqt = (p.Sq + 31) >> 5;
  int kfirst = kblk * KPB, klast = kblk * KPB + KPB - 1;
  // block-uniform list of query tiles that can see this key block: [qt_a, qt_b)
  int qt_a = 0, qt_b = nqt;
  if (p.mask_mode == UNIMP_MASK_CAUSAL) qt_a = kfirst >> 5;
  else if (p.mask_mode == UNIMP_MASK_SEGMENT) {
    // media_time is non-decreasing along the sequence: first / last tile whose rows can attend [kfirst, klast]
    while (qt_a < nqt) { int tb = p.seg[(long)b * p.SqS + min(qt_a * 32 + 31, p.Sq - 1)]; if (tb * p.seg_len > kfirst) break; ++qt_a; }
    while (qt_b > qt_a) { int ta = p.seg[(long)b * p.SqS + (qt_b - 1) * 32]; if ((max(ta, 1) - 1) * p.seg_len <= klast) break; --qt_b; }
  }
  TileCopy<32, DQK, STR> rq, rdo;
  rq.init(p.q_ss, p.D); rdo.init(p.do_ss, p.D);
  if (DQK != DV) lds_zero(smem, 2 * STAGE);
  // per-row lse / delta / key range of a query tile (first 32 threads): the global LOADS are issued with the tile's Q / dO loads at the
  // top of an iteration and fly under its MFMAs; only the LDS stores wait at the bottom.  (As one step at the bottom, the loads were issued
  // and waited for after the MFMAs: one exposed memory latency per query tile, in a kernel whose tile is ~0.4 us of arithmetic.)
  // Head dim 64 keeps the late form: the four values live across the MFMAs cost it its fourth wave per SIMD (128 -> 136 registers).
  constexpr bool EARLY_AUX = DQK != 64;
  const int kvl_b = p.kv_len ? p.kv_len[b] : p.Sk;          // once per block, not once per query tile (stamps: the scalar load + wait was
                                                            // inside the largest segment of the loop)
  struct Aux { float lse, dl; int lo, hi; };               // values of ONE iteration (not loop-carried: that would keep them live everywhere)
  auto aux_load = [&](int qt) {
    Aux a;
    a.lse = 0.f; a.dl = 0.f; a.lo = 0; a.hi = 0;
    if (threadIdx.x < 32) {
      int qr = qt * 32 + threadIdx.x;
      long sidx = ((long)b * p.H + h) * p.SqS + min(qr, p.Sq - 1);
      a.lse = p.lse[sidx]; a.dl = p.delta[sidx];
      key_range_kvl(p, b, qr, kvl_b, a.lo, a.hi);
    }
    return a;
  };
  auto aux_store = [&](char* st, const Aux& a) {
    if (threadIdx.x < 32) {
      float* st_lse = (float*)(st + 2 * 32 * STR);
      st_lse[threadIdx.x] = a.lse * LOG2E;
      st_lse[32 + threadIdx.x] = a.dl;
      ((int*)st_lse)[64 + threadIdx.x] = a.lo; ((int*)st_lse)[96 + threadIdx.x] = a.hi;
      // all 32 rows of this query tile see every key of the block -> the waves skip the per-element mask
      unsigned long long okm = __ballot(a.lo <= kfirst && a.hi > klast);
      if (threadIdx.x == 0) ((int*)st_lse)[128] = (okm & 0xffffffffull) == 0xffffffffull;
    }
  };
  if (qt_a < qt_b) {
    rq.g2r(qb + (long)qt_a * 32 * p.q_ss, p.Sq - qt_a * 32);
    rdo.g2r(dob + (long)qt_a * 32 * p.do_ss, p.Sq - qt_a * 32);
    Aux a0 = aux_load(qt_a);
    rq.r2s(smem);
    rdo.r2s(smem + 32 * STR);
    aux_store(smem, a0);
  }
  __syncthreads();
#ifdef ATTN_STAMP
  at_last = AT_NOW();
  const unsigned long long at_c1 = at_last;
#endif
  for (int qt = qt_a; qt < qt_b; ++qt) {
    char* st = smem + ((qt - qt_a) & 1) * STAGE;
    char* nx = smem + ((qt - qt_a + 1) & 1) * STAGE;
    const char* qs_t = st; const char* dos_t = st + 32 * STR;
    const float* st_lse = (const float*)(st + 2 * 32 * STR);
    const float* st_dl = st_lse + 32;
    const int* st_lo = (const int*)(st_lse + 64); const int* st_hi = st_lo + 32;
    bool more = qt + 1 < qt_b;
    if (more) {
      rq.g2r(qb + (long)(qt + 1) * 32 * p.q_ss, p.Sq - (qt + 1) * 32);
      rdo.g2r(dob + (long)(qt + 1) * 32 * p.do_ss, p.Sq - (qt + 1) * 32);
    }
    Aux an;
    if (EARLY_AUX && more) an = aux_load(qt + 1);
    AT_SEG(0);                                               // next tile's global loads issued
    if (key0 < p.Sk) {
    f32x4 s[KU][2], dp[KU][2];              // [key block u][query block qb2]: lane holds S[q = 16*qb2 + 4g + r][key = l&15]
#pragma unroll
    for (int qb2 = 0; qb2 < 2; ++qb2) {
#pragma unroll
      for (int u = 0; u < KU; ++u) { s[u][qb2] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[u][qb2] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        bf16x8 qf_ = lfrag_kc<STR>(qs_t, qb2 * 16, ks), dof_ = lfrag_kc<STR>(dos_t, qb2 * 16, ks);
#pragma unroll
        for (int u = 0; u < KU; ++u) { s[u][qb2] = MFMA16(qf_, kf[u][ks], s[u][qb2]); dp[u][qb2] = MFMA16(dof_, vf[u][ks], dp[u][qb2]); }
      }
    }
    bf16x8 pf[KU], dsf[KU];
    float lse_r[2][4], dl_r[2][4];
#pragma unroll
    for (int qb2 = 0; qb2 < 2; ++qb2) {
      f32x4 a = *(const f32x4*)(st_lse + qb2 * 16 + g * 4), d = *(const f32x4*)(st_dl + qb2 * 16 + g * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) { lse_r[qb2][r] = a[r]; dl_r[qb2][r] = d[r]; }
    }
    bool all_visible = st_lo[64] != 0;        // word 128 of the aux block
#ifdef ATTN_STAMP
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); asm volatile("" :: "v"(s[0][0][0]), "v"(dp[0][0][0]));
#endif
    AT_SEG(1);                                               // S, dP products done (their fragment reads + 12 MFMAs), row constants read
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      int key = key0 + u * 16 + (l & 15);
      if (ALIBI) {
        float kb = ab * (float)key;
#pragma unroll
        for (int qb2 = 0; qb2 < 2; ++qb2)
#pragma unroll
          for (int r = 0; r < 4; ++r) s[u][qb2][r] += kb;
      }
      if (all_visible && key < p.Sk) {
#pragma unroll
        for (int qb2 = 0; qb2 < 2; ++qb2)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float pr = EXP2(fmaf(s[u][qb2][r], sc2, -lse_r[qb2][r]));
            s[u][qb2][r] = pr;
            dp[u][qb2][r] = pr * (dp[u][qb2][r] - dl_r[qb2][r]) * p.scale;
          }
      } else {
#pragma unroll
        for (int qb2 = 0; qb2 < 2; ++qb2)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            int ql = qb2 * 16 + g * 4 + r;
            float pr = (key >= st_lo[ql] && key < st_hi[ql]) ? EXP2(fmaf(s[u][qb2][r], sc2, -lse_r[qb2][r])) : 0.f;
            s[u][qb2][r] = pr;
            dp[u][qb2][r] = pr * (dp[u][qb2][r] - dl_r[qb2][r]) * p.scale;
          }
      }
      pf[u] = pack8(s[u][0], s[u][1]); dsf[u] = pack8(dp[u][0], dp[u][1]);
    }
#ifdef ATTN_STAMP
    asm volatile("" :: "v"(pf[0]), "v"(dsf[0]));
#endif
    AT_SEG(2);                                               // exponentials, dS, packing
#pragma unroll
    for (int nd = 0; nd < ND; ++nd) {
      bf16x8 dot_f = lfrag_tr_perm<STR>(dos_t, 0, nd * 16), qt_f = lfrag_tr_perm<STR>(qs_t, 0, nd * 16);
#pragma unroll
      for (int u = 0; u < KU; ++u) { dv[u][nd] = MFMA16(dot_f, pf[u], dv[u][nd]); dk[u][nd] = MFMA16(qt_f, dsf[u], dk[u][nd]); }
    }
    }
#ifdef ATTN_STAMP
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); asm volatile("" :: "v"(dv[0][0][0]), "v"(dk[0][0][0]));
#endif
    AT_SEG(3);                                               // dV, dK products (transposed fragment reads + 10 MFMAs)
    if (more) {
      rq.r2s(nx);
      rdo.r2s(nx + 32 * STR);
      if (!EARLY_AUX) an = aux_load(qt + 1);
      aux_store(nx, an);
    }
#ifdef ATTN_STAMP
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
    AT_SEG(4);                                               // wait for the next tile's loads + LDS stores
    __syncthreads();
    AT_SEG(5);                                               // barrier
  }
#ifdef ATTN_STAMP
  const unsigned long long at_c2 = AT_NOW();
#endif
  // epilogue through a wave-private LDS region (the loop ended with a barrier): accumulator layout (lane: key = l & 15 of
  // block u, 4 consecutive d) -> whole 16-byte chunks of consecutive key rows; 8-byte stores straight from the accumulators
  // touched 16 different rows per instruction (store-issue bound: ~80 us of this kernel at the LM shape)
  constexpr int EP = DV * 2 + 16, ECPR = DV / 8;
  static_assert(4 * KPW * EP <= 2 * STAGE, "epilogue staging fits");
  char* ew = smem + wave * (KPW * EP);
  const bool wide = !((p.dk_bs | p.dk_ss | p.dk_hs | p.dv_bs | p.dv_ss | p.dv_hs) & 7) && !(((uintptr_t)p.dk | (uintptr_t)p.dv) & 15);
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    bf16* gb = (which ? p.dv + b * p.dv_bs + h * p.dv_hs : p.dk + b * p.dk_bs + h * p.dk_hs);
    long gs = which ? p.dv_ss : p.dk_ss;
    if (wide) {
#pragma unroll
      for (int u = 0; u < KU; ++u)
#pragma unroll
        for (int nd = 0; nd < ND; ++nd) {
          const f32x4& a = which ? dv[u][nd] : dk[u][nd];
          bf16x4 w = {f2bf(a[0]), f2bf(a[1]), f2bf(a[2]), f2bf(a[3])};
          *(bf16x4*)(ew + (u * 16 + (l & 15)) * EP + (nd * 16 + g * 4) * 2) = w;
        }
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
      constexpr int NCH = (KPW * ECPR + 63) / 64;
      if (which == 0 && p.rope_step != 0.f) {                // adjacent-pair layout: no tables, no partner chunk
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
          int id = l + 64 * i;
          int r = id / ECPR, c = id - r * ECPR;
          if (id < KPW * ECPR && key0 + r < p.Sk)
            *(u32x4*)(gb + (long)(key0 + r) * gs + c * 8) = attn_rope_inv_adjacent(ew + r * EP, c, p.rope_half, (float)(key0 + r), p.rope_step);
        }
      } else if (which == 0 && p.rope_cos) {                  // dk leaves rotated back (attention_params.h): all table loads first
        AttnRopeChunk ch[NCH];
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
          int id = min(l + 64 * i, KPW * ECPR - 1);
          int r = id / ECPR, c = id - r * ECPR;
          long pos = min(key0 + r, p.Sk - 1);
          attn_rope_inv_load(ch[i], ew + r * EP, c, p.rope_half, p.rope_cos + pos * p.rope_half, p.rope_sin + pos * p.rope_half);
        }
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
          int id = l + 64 * i;
          int r = id / ECPR, c = id - r * ECPR;
          u32x4 v = attn_rope_inv_apply(ch[i]);
          if (id < KPW * ECPR && key0 + r < p.Sk) *(u32x4*)(gb + (long)(key0 + r) * gs + c * 8) = v;
        }
      } else {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
          int id = l + 64 * i;
          int r = id / ECPR, c = id - r * ECPR;
          if (id < KPW * ECPR && key0 + r < p.Sk) *(u32x4*)(gb + (long)(key0 + r) * gs + c * 8) = *(const u32x4*)(ew + r * EP + c * 16);
        }
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
    } else {
#pragma unroll
      for (int u = 0; u < KU; ++u) {
        int key = key0 + u * 16 + (l & 15);
        if (key < p.Sk)
#pragma unroll
          for (int nd = 0; nd < ND; ++nd) {
            const f32x4& a = which ? dv[u][nd] : dk[u][nd];
            bf16x4 w = {f2bf(a[0]), f2bf(a[1]), f2bf(a[2]), f2bf(a[3])};
            *(bf16x4*)(gb + (long)key * gs + nd * 16 + g * 4) = w;
          }
      }
    }
  }
#ifdef ATTN_STAMP
  if (threadIdx.x == 0 && blockIdx.x < 16384) {
    unsigned long long* o = attn_stamps + (long)blockIdx.x * 12;
    for (int i = 0; i < 6; ++i) o[i] = at_acc[i];
    o[6] = at_c1 - at_c0; o[7] = at_c2 - at_c1; o[8] = AT_NOW() - at_c2; o[9] = qt_b - qt_a;
    o[10] = at_t0; o[11] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}

// ------------------------------------------------------------------------------------------- host
static int fill(AttnP& p, const unimp_attn_desc* d, bool bwd) {
  p = AttnP{};
  if (!d || !d->q || !d->k || !d->v || !d->o) return unimp_set_error(UNIMP_ERR_ARG, "attn: null pointer");
  if (d->D != 64 && d->D != 80 && d->D != 128) return unimp_set_error(UNIMP_ERR_UNSUPPORTED, "attn: head dim must be 64, 80 or 128");
  if (d->mask_mode == UNIMP_MASK_SEGMENT && (!d->seg || d->seg_len <= 0)) return unimp_set_error(UNIMP_ERR_ARG, "attn: segment mask needs seg/seg_len");
  if (d->mask_mode == UNIMP_MASK_CAUSAL && d->Sq != d->Sk) return unimp_set_error(UNIMP_ERR_SHAPE, "attn: causal needs Sq == Sk");
  int64_t strides[] = {d->q_bs, d->q_ss, d->q_hs, d->k_bs, d->k_ss, d->k_hs, d->v_bs, d->v_ss, d->v_hs, d->o_bs, d->o_ss, d->o_hs};
  for (int64_t s : strides) if (s & 7) return unimp_set_error(UNIMP_ERR_ALIGN, "attn: strides must be multiples of 8 elements");
  const void* ptrs[] = {d->q, d->k, d->v, d->o};
  for (const void* q : ptrs) if ((uintptr_t)q & 15) return unimp_set_error(UNIMP_ERR_ALIGN, "attn: pointers must be 16-B aligned");
  p.q = (const bf16*)d->q; p.k = (const bf16*)d->k; p.v = (const bf16*)d->v; p.o = (bf16*)d->o; p.lse = d->lse;
  p.q_bs = d->q_bs; p.q_ss = d->q_ss; p.q_hs = d->q_hs; p.k_bs = d->k_bs; p.k_ss = d->k_ss; p.k_hs = d->k_hs;
  p.v_bs = d->v_bs; p.v_ss = d->v_ss; p.v_hs = d->v_hs; p.o_bs = d->o_bs; p.o_ss = d->o_ss; p.o_hs = d->o_hs;
  p.B = d->B; p.H = d->H; p.Sq = d->Sq; p.Sk = d->Sk; p.D = d->D; p.scale = d->scale; p.mask_mode = d->mask_mode;
  p.kv_len = d->kv_len; p.seg = d->seg; p.seg_len = d->seg_len;
  p.q_off = d->q_row_off; p.q_len = d->q_len; p.k_off = d->k_row_off; p.SqS = d->Sq;
  if (p.q_off && !p.q_len) return unimp_set_error(UNIMP_ERR_ARG, "attn: q_row_off needs q_len");
  if (p.k_off && !p.kv_len) return unimp_set_error(UNIMP_ERR_ARG, "attn: k_row_off needs kv_len");
  p.alibi = d->alibi_slopes;
  p.d_o = nullptr; p.dq = p.dk = p.dv = nullptr; p.delta = nullptr;
  p.rope_cos = p.rope_sin = nullptr; p.rope_half = 0; p.rope_step = 0.f;
  if (bwd && d->rope_log2_base != 0.f) {
    if (d->rope_cos || d->rope_sin || d->rope_half <= 0 || (d->rope_half & 3) || 2 * d->rope_half > d->D || !(d->rope_log2_base > 0.f))
      return unimp_set_error(UNIMP_ERR_ARG, "attn_bwd: adjacent-pair rope needs rope_half % 4 == 0, 2 * rope_half <= D, log2(base) > 0 and no tables");
    p.rope_half = d->rope_half; p.rope_step = d->rope_log2_base / (float)d->rope_half;
  } else if (bwd && (d->rope_cos || d->rope_sin)) {
    if (!d->rope_cos || !d->rope_sin || d->rope_half <= 0 || (d->rope_half & 7) || 2 * d->rope_half > d->D)
      return unimp_set_error(UNIMP_ERR_ARG, "attn_bwd: rope needs cos and sin tables, rope_half % 8 == 0, 2 * rope_half <= D");
    if (((uintptr_t)d->rope_cos | (uintptr_t)d->rope_sin) & 15) return unimp_set_error(UNIMP_ERR_ALIGN, "attn_bwd: rope tables must be 16-B aligned");
    p.rope_cos = d->rope_cos; p.rope_sin = d->rope_sin; p.rope_half = d->rope_half;
  }
  if (bwd) {
    if (!d->d_o || !d->dq || !d->dk || !d->dv || !d->delta || !d->lse) return unimp_set_error(UNIMP_ERR_ARG, "attn_bwd: null pointer");
    int64_t s2[] = {d->do_bs, d->do_ss, d->do_hs, d->dq_bs, d->dq_ss, d->dq_hs, d->dk_bs, d->dk_ss, d->dk_hs, d->dv_bs, d->dv_ss, d->dv_hs};
    for (int64_t s : s2) if (s & 3) return unimp_set_error(UNIMP_ERR_ALIGN, "attn_bwd: strides must be multiples of 4 elements");
    if ((d->do_bs | d->do_ss | d->do_hs) & 7) return unimp_set_error(UNIMP_ERR_ALIGN, "attn_bwd: dO strides must be multiples of 8");
    p.d_o = (const bf16*)d->d_o; p.dq = (bf16*)d->dq; p.dk = (bf16*)d->dk; p.dv = (bf16*)d->dv; p.delta = d->delta;
    p.do_bs = d->do_bs; p.do_ss = d->do_ss; p.do_hs = d->do_hs; p.dq_bs = d->dq_bs; p.dq_ss = d->dq_ss; p.dq_hs = d->dq_hs;
    p.dk_bs = d->dk_bs; p.dk_ss = d->dk_ss; p.dk_hs = d->dk_hs; p.dv_bs = d->dv_bs; p.dv_ss = d->dv_ss; p.dv_hs = d->dv_hs;
  }
  return 0;
}

// kernel generation knob: 2 (default) = attention2.hip where it has a kernel, 1 = the first-generation kernels of this file
// everywhere (A/B measurements, tests run both).  Initial value from UNIMP_ATTN_GEN.
static int g_attn_gen = -1;
static int attn_generation() {
  if (g_attn_gen < 0) { const char* e = getenv("UNIMP_ATTN_GEN"); g_attn_gen = e ? atoi(e) : 2; }
  return g_attn_gen;
}
extern "C" int unimp_attn_set_generation(int gen) { int old = attn_generation(); g_attn_gen = gen; return old; }
extern "C" int unimp_attn_get_generation(void) { return attn_generation(); }
static int g_attn_dkv3 = -1;
static int attn_dkv3_on() {
  if (g_attn_dkv3 < 0) { const char* e = getenv("UNIMP_DKV3"); int v = e ? atoi(e) : 1; g_attn_dkv3 = v < 0 ? 0 : (v > 2 ? 2 : v); }
  return g_attn_dkv3;
}
static int g_attn_last_dkv = 0;
extern "C" int unimp_attn_last_dkv(void) { return g_attn_last_dkv; }
extern "C" int unimp_attn_set_dkv3(int mode) { int old = attn_dkv3_on(); g_attn_dkv3 = mode < 0 ? 0 : (mode > 2 ? 2 : mode); return old; }

extern "C" int unimp_attn_fwd(const unimp_attn_desc* d, void* stream) {
  AttnP p;
  int e = fill(p, d, false);
  if (e) return e;
  if (p.B <= 0 || p.H <= 0 || p.Sq <= 0) return UNIMP_OK;
  if (attn_generation() >= 2) return unimp_attn_fwd2_dispatch(p, stream);
  dim3 grid((p.Sq + 127) / 128, p.H, p.B), block(256);
  hipStream_t s = (hipStream_t)stream;
#define FWD(A_) do { if (p.D == 64) hipLaunchKernelGGL((attn_fwd_kernel<64, 64, A_>), grid, block, 0, s, p);       \
    else if (p.D == 80) hipLaunchKernelGGL((attn_fwd_kernel<96, 80, A_>), grid, block, 0, s, p);                     \
    else hipLaunchKernelGGL((attn_fwd_kernel<128, 128, A_>), grid, block, 0, s, p); } while (0)
  if (p.alibi) FWD(true); else FWD(false);
#undef FWD
  return unimp_check_launch("attn_fwd");
}

extern "C" int unimp_attn_bwd(const unimp_attn_desc* d, void* stream) {
  AttnP p;
  int e = fill(p, d, true);
  if (e) return e;
  if (p.B <= 0 || p.H <= 0 || p.Sq <= 0) return UNIMP_OK;
  hipStream_t s = (hipStream_t)stream;
  long n = (long)p.B * p.H * p.Sq;
  // generation 2 (default): second-generation dQ kernel + first-generation dK/dV kernel (measured faster at 2 waves per SIMD:
  // profiles/ r02 attention notes); generation 3: both second generation.  The second-generation kernels store 16-byte chunks:
  // output strides must be multiples of 8 elements.
  int gen = attn_generation();
  // generation 2 (default) hands the forms attention3.hip serves (head dim 80, causal / no mask, 32-row multiples: the language
  // model's self-attention) to its 64-keys-per-wave dK/dV kernel; generation 4 = 2 without it (the round-4 default; A/B and tests);
  // UNIMP_DKV3 / unimp_attn_set_dkv3: 0 = never (what generation 4 selects), 1 (default) = where it serves the form AND the launch has a
  // (batch, head) pair per CU (below that the first generation's many small workgroups win), 2 = wherever it serves the form (tests)
  bool al16 = !(((p.dq_bs | p.dq_ss | p.dq_hs | p.dk_bs | p.dk_ss | p.dk_hs | p.dv_bs | p.dv_ss | p.dv_hs) & 7) ||
                (((uintptr_t)p.dq | (uintptr_t)p.dk | (uintptr_t)p.dv) & 15)) && p.Sq >= 4;
  int which2 = (gen >= 2 && al16) ? (gen == 3 ? 3 : 1) : 0;
  if ((p.q_off || p.k_off) && !(which2 & 1))
    return unimp_set_error(UNIMP_ERR_UNSUPPORTED, "attn_bwd: packed rows need kernel generation >= 2 and 16-byte aligned dq / dk / dv views");
  if ((p.rope_cos || p.rope_step != 0.f) && !(which2 & 1))
    return unimp_set_error(UNIMP_ERR_UNSUPPORTED, "attn_bwd: fused rope needs kernel generation >= 2 and 16-byte aligned dq / dk / dv views");
  // delta = rowsum(dO * O): the second-generation dQ kernel computes and publishes it itself
  if (!(which2 & 1)) hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((n * 16 + 255) / 256)), dim3(256), 0, s, p);
  if (which2) { int e2 = unimp_attn_bwd2_dispatch(p, which2, stream); if (e2) return e2; }
  if (gen == 2 && which2 == 1 && !(d->flags & UNIMP_ATTN_NO_PERSISTENT) && attn_dkv3_on() && unimp_attn_dkv3_eligible(&p) && (attn_dkv3_on() == 2 || unimp_attn_dkv3_preferred(&p))) {
    g_attn_last_dkv = 3;
    return unimp_attn_dkv3_launch(p, stream);
  }
  g_attn_last_dkv = (which2 & 2) ? 2 : 1;
  // dK/dV: 16 keys per wave (KU = 1, 64-key workgroups; default) or 32 (KU = 2, 128-key workgroups; UNIMP_DKV_KU=2).  With 32 the
  // kernel held 248 registers at head dim 80 (dK^T, dV^T accumulators and the K / V fragments of two 16-key blocks): two waves
  // per SIMD, 55 % of wave time in waits.  With 16 it holds 166: a third wave per SIMD (hd 64: four; hd 128: two instead of one
  // and no spill into the AGPRs) for twice the Q / dO fragment reads per product, on an LDS pipe that was 14 % busy: LM shape
  // backward 0.767 -> 0.704 ms per layer, gated cross-attention -27 %, MPT hd 128 -21 %.
  static const int ku = [] { const char* e = getenv("UNIMP_DKV_KU"); int v = e ? atoi(e) : 1; return v == 2 ? 2 : 1; }();
  const int nkb = (p.Sk + 64 * ku - 1) / (64 * ku);
  dim3 gq((p.Sq + 127) / 128, p.H, p.B), gk(nkb * p.H * p.B), block(256);
#define DKV(DQ_, DV_, A_) do { if (ku == 1) hipLaunchKernelGGL((attn_dkv_kernel<DQ_, DV_, A_, 1>), gk, block, 0, s, p, nkb);          \
                               else hipLaunchKernelGGL((attn_dkv_kernel<DQ_, DV_, A_, 2>), gk, block, 0, s, p, nkb); } while (0)
#define BWD(A_) do {                                                                                                   \
    if (p.D == 64) { if (!(which2 & 1)) hipLaunchKernelGGL((attn_dq_kernel<64, 64, A_>), gq, block, 0, s, p); if (!(which2 & 2)) DKV(64, 64, A_); }            \
    else if (p.D == 80) { if (!(which2 & 1)) hipLaunchKernelGGL((attn_dq_kernel<96, 80, A_>), gq, block, 0, s, p); if (!(which2 & 2)) DKV(96, 80, A_); }       \
    else { if (!(which2 & 1)) hipLaunchKernelGGL((attn_dq_kernel<128, 128, A_>), gq, block, 0, s, p); if (!(which2 & 2)) DKV(128, 128, A_); } } while (0)
  if (p.alibi) BWD(true); else BWD(false);
#undef BWD
#undef DKV
  return unimp_check_launch("attn_bwd");
}
