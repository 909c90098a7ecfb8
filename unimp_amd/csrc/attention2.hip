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
#include "attention2_common.h"

// TAIL (host: no mask, no kv_len, no ALiBi, Sk % 64 == 1 -- the ViT's 256 patches + CLS): the tile loop runs over the Sk / 64 FULL
// tiles and the last key SEEDS the online-softmax state instead of costing a fifth tile with one live column: m = q . k_last,
// l = 1, O = v_last (p = exp2(0) = 1 exactly, so the seed row enters O unrounded).  The dot product is a 64-term fp32 FMA chain
// on the same bf16 products the matrix pipe forms.
template <int D, int NW, bool ALIBI, bool TAIL = false>
__global__ __launch_bounds__(64 * NW) void attn_fwd2_kernel(AttnP p, int row_base, int nx) {
  using C = A2Cfg<D>;
  constexpr int CPR = C::CPR, PK = C::PK, PV = C::PV, KS = C::KS, ND = C::ND, NI = C::NI, STAGE = C::STAGE;
  constexpr int NT = (NI + NW - 1) / NW;                     // DMA instructions per wave and tile
  __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE];
  int bx, h, b;
  a2_decode(nx, p.H, p.B, bx, h, b);
  if (!attn_varlen(p, b)) return;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = lane_id(), hi5 = l >> 5, ql = l & 31;
  const int qblk0 = row_base + bx * (32 * NW);               // first query row of the block
  if (qblk0 >= p.Sq) return;                                 // packed rows: the grid covers the longest sequence
  const int q0 = qblk0 + wave * 32;
  const bf16* qb = p.q + b * p.q_bs + h * p.q_hs;
  const char* kb = (const char*)(p.k + b * p.k_bs + h * p.k_hs);
  const char* vb = (const char*)(p.v + b * p.v_bs + h * p.v_hs);

  // ---- block-uniform key tile range
  int kt_lo = 0, kt_hi = 0;
  {
    int q_last = min(qblk0 + 32 * NW - 1, p.Sq - 1);
    int kvl = p.kv_len ? p.kv_len[b] : p.Sk;
    if (TAIL) kt_hi = kvl >> 6;
    else if (p.mask_mode == UNIMP_MASK_NONE) kt_hi = (kvl + 63) >> 6;
    else if (p.mask_mode == UNIMP_MASK_CAUSAL) kt_hi = (min(q_last + 1, kvl) + 63) >> 6;
    else {
      int t0 = p.seg[(long)b * p.SqS + qblk0], t1 = p.seg[(long)b * p.SqS + q_last];
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
  bf16x8 kx[KS]; bf16x4 vx[ND][4];
  if (TAIL) {                                                 // the last key / value row, the lane's own dims (half-wave-uniform addresses)
    const bf16* kr = (const bf16*)(kb + (long)(p.Sk - 1) * k_sb);
    const bf16* vr = (const bf16*)(vb + (long)(p.Sk - 1) * v_sb);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) kx[ks] = *(const bf16x8*)(kr + ks * 16 + hi5 * 8);
#pragma unroll
    for (int nd = 0; nd < ND; ++nd)
#pragma unroll
      for (int g = 0; g < 4; ++g) vx[nd][g] = *(const bf16x4*)(vr + min(32 * nd + 8 * g + 4 * hi5, D - 4));
  }

  if (kt_lo < kt_hi) dma_tile(kt_lo, 0);
  if (TAIL) {
    float sx = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) sx = fmaf(bf2f(qf[ks][j]), bf2f(kx[ks][j]), sx);
    sx += __shfl_xor(sx, 32, 64);
    if (hi > 0) {                                             // rows past Sq keep the empty state
      m = sx * sc2;
      lsum = hi5 == 0 ? 1.f : 0.f;
#pragma unroll
      for (int nd = 0; nd < ND; ++nd)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int e = 0; e < 4; ++e) o[nd][4 * g + e] = bf2f(vx[nd][g][e]);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) a2_pin(qf[ks]);
  a2_pin(lo); a2_pin(hi);
  if (TAIL) {
    a2_pin(m); a2_pin(lsum);
#pragma unroll
    for (int nd = 0; nd < ND; ++nd) a2_pin(o[nd]);
  }
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
      // row maximum of the lane's 32 scores as 16 v_max3_f32 (asm: fmaxf on MFMA outputs makes the compiler canonicalise every
      // operand first -- 49 v_max_f32 per tile instead of 16; the scores are finite or -inf, never signalling)
      float mloc = a2_max3(s0[0], s1[0], s0[1]);
      mloc = a2_max3(mloc, s1[1], s0[2]);
#pragma unroll
      for (int r = 2; r < 15; ++r) mloc = a2_max3(mloc, s1[r], s0[r + 1]);
      mloc = a2_max3(mloc, s1[15], s1[15]);
      mloc = a2_max3(mloc, __shfl_xor(mloc, 32, 64), mloc);
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

  // ---- epilogue: O^T[d][q] -> O[q][d] through LDS (the tile loop ended with a barrier: the stages are free)
  {
    float ltot = lsum + __shfl_xor(lsum, 32, 64);
    int qr = q0 + ql;
    float inv = ltot > 0.f ? 1.f / ltot : 0.f;
    if (qr < p.Sq && hi5 == 0 && p.lse) p.lse[((long)b * p.H + h) * p.SqS + qr] = ltot > 0.f ? (m + log2f(ltot)) * LN2 : -INFINITY;
    static_assert(NW * 32 * (D * 2 + 16) <= 2 * STAGE, "epilogue staging fits the tile stages");
    // per-row scale: inv differs per lane row, so scale while writing (mul is per lane = per row here)
    a2_store_rows<D, ND>(smem + wave * (32 * (D * 2 + 16)), o, inv, p.o + b * p.o_bs + h * p.o_hs, p.o_ss, q0, p.Sq);
  }
}

// =========================================================================================== backward, second generation
// Same building blocks as the forward: 32x32x16 MFMA, 64-row tiles staged by LDS-DMA into conflict-free images (a tile that
// is read both by rows -- ds_read_b128, odd pitch -- and transposed -- ds_read_b64_tr_b16, pitch with (4 P) % 64 in {16, 48} --
// is staged twice: the two pitch rules exclude each other, and a second DMA of a 10-KiB tile that sits in L2 is cheaper than a
// conflicted read on every MFMA), P / dS stay in registers as the next product's B operand, no atomics (dQ and dK/dV are
// separate kernels over disjoint outputs: bit-reproducible).  The score blocks are processed 32 keys (dQ) / 32 queries
// (dK/dV) at a time so that S and dP take 32 registers together and the dQ kernel runs at 3 waves per SIMD.

// DMA plan: the instructions of one tile are numbered over the images in order; instruction i is issued by wave i % NW.
template <int NIMG> struct A2Img { int first[NIMG + 1]; int pitch[NIMG]; int lds[NIMG]; };

// ------------------------------------------------------------------------------------------- dQ
// wave = 32 query rows (Q, dO fragments, lse, delta in registers); per 64-key tile: images K-row, V-row.
template <int D, bool ALIBI>
__global__ __launch_bounds__(256) void attn_dq2_kernel(AttnP p, int nx) {
  using C = A2Cfg<D>;
  constexpr int CPR = C::CPR, PK = C::PK, KS = C::KS, ND = C::ND;
  // Two images per tile, K-row and V-row (odd pitch PK: conflict-free ds_read_b128 row reads for S and dP).  The dQ product
  // reads K TRANSPOSED from the same row image: at the odd pitch the ds_read_b64_tr_b16 half-waves overlap on a few banks
  // (~1.5x on a third of the LDS reads of a kernel whose LDS pipe is 5-7 % busy), but without a third, transposed-pitch image
  // a stage is 22.5 KiB instead of 35: three workgroups per CU instead of two and a third less DMA -- measured -18 % on this
  // kernel (LM shape 0.34 -> 0.28 ms, MPT hd 128 -18 % on the whole backward).
  constexpr int NW = 4, NI = PK + PK, NT = (NI + NW - 1) / NW;
  constexpr int OFF_KT = 0, OFF_V = 64 * PK * 16, STAGE = OFF_V + 64 * PK * 16, PT = PK;
  __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE];
  int bx, h, b;
  a2_decode(nx, p.H, p.B, bx, h, b);
  if (!attn_varlen(p, b)) return;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = lane_id(), hi5 = l >> 5, ql = l & 31;
  const int qblk0 = bx * 128, q0 = qblk0 + wave * 32;
  if (qblk0 >= p.Sq) return;                                 // packed rows: the grid covers the longest sequence
  const bf16* qb = p.q + b * p.q_bs + h * p.q_hs;
  const bf16* dob = p.d_o + b * p.do_bs + h * p.do_hs;
  const char* kb = (const char*)(p.k + b * p.k_bs + h * p.k_hs);
  const char* vb = (const char*)(p.v + b * p.v_bs + h * p.v_hs);

  int kt_lo = 0, kt_hi = 0;
  {
    int q_last = min(qblk0 + 127, p.Sq - 1);
    int kvl = p.kv_len ? p.kv_len[b] : p.Sk;
    if (p.mask_mode == UNIMP_MASK_NONE) kt_hi = (kvl + 63) >> 6;
    else if (p.mask_mode == UNIMP_MASK_CAUSAL) kt_hi = (min(q_last + 1, kvl) + 63) >> 6;
    else {
      int t0 = p.seg[(long)b * p.SqS + qblk0], t1 = p.seg[(long)b * p.SqS + q_last];
      if (t1 > 0) { kt_lo = (max(t0 - 1, 0) * p.seg_len) >> 6; kt_hi = (min(t1 * p.seg_len, p.Sk) + 63) >> 6; }
    }
  }
  int d_row[NT]; uint32_t d_col[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    int i = wave + NW * t;
    int img = i < PK ? 0 : 2;
    int ii = i - (img == 0 ? 0 : PK);
    int pitch = PK;
    int s_ = 64 * ii + l;
    int row = s_ / pitch, cs = s_ - row * pitch;
    d_row[t] = row;
    d_col[t] = (uint32_t)(min(cs, CPR - 1) * 16);
  }
  const uint32_t smem_lds = (uint32_t)(uintptr_t)LDS_PTR(char, smem);
  const long k_sb = p.k_ss * 2, v_sb = p.v_ss * 2;
  auto dma_tile = [&](int kt, int stage) {
    int rows_left = p.Sk - kt * 64;
    const char* kt_base = kb + (long)kt * 64 * k_sb;
    const char* vt_base = vb + (long)kt * 64 * v_sb;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      int i = wave + NW * t;
      if (i < NI) {
        int img = i < PK ? 0 : 2;
        int ii = i - (img == 0 ? 0 : PK);
        int rc = min(d_row[t], rows_left - 1);
        uint32_t off = (uint32_t)(rc * (img == 2 ? v_sb : k_sb)) + d_col[t];
        uint32_t dst = smem_lds + stage * STAGE + (img == 0 ? 0 : OFF_V) + ii * 1024;
        a2_glds(img == 2 ? vt_base : kt_base, off, __builtin_amdgcn_readfirstlane(dst));
      }
    }
  };

  // the first tile's DMA goes out BEFORE the wave's own Q / dO / O row loads: the delta row sums below wait for those loads, and a
  // DMA issued after them paid a second, serial memory latency before the first tile (a block sweeps only 2-8 tiles at L = 512)
  if (kt_lo < kt_hi) dma_tile(kt_lo, 0);
  bf16x8 qf[KS], dof[KS];
  int lo, hi;
  float lse2, dl;
  {
    int qr = q0 + ql, qc = min(qr, p.Sq - 1);
    const bf16* qrow = qb + (long)qc * p.q_ss;
    const bf16* dorow = dob + (long)qc * p.do_ss;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { qf[ks] = *(const bf16x8*)(qrow + ks * 16 + hi5 * 8); dof[ks] = *(const bf16x8*)(dorow + ks * 16 + hi5 * 8); }
    a2_key_range(p, b, qr, lo, hi);
    long sidx = ((long)b * p.H + h) * p.SqS + qc;
    lse2 = p.lse[sidx] * LOG2E;
    // delta = rowsum(dO * O), computed here (the wave holds its dO rows anyway) and published for the dK/dV kernel that
    // follows on the stream: the separate delta pass over O and dO is gone
    const bf16* orow = p.o + b * p.o_bs + h * p.o_hs + (long)qc * p.o_ss;
    float acc = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bf16x8 of = *(const bf16x8*)(orow + ks * 16 + hi5 * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc += bf2f(of[j]) * bf2f(dof[ks][j]);
    }
    dl = acc + __shfl_xor(acc, 32, 64);
    if (hi5 == 0 && qr < p.Sq) p.delta[sidx] = dl;
  }
  int wlo = lo, whi = hi;
#pragma unroll
  for (int o_ = 32; o_ > 0; o_ >>= 1) { wlo = min(wlo, __shfl_xor(wlo, o_, 64)); whi = max(whi, __shfl_xor(whi, o_, 64)); }
  wlo = __builtin_amdgcn_readfirstlane(wlo); whi = __builtin_amdgcn_readfirstlane(whi);
  const float sc2 = p.scale * LOG2E;
  const float ab = ALIBI ? p.alibi[h] / p.scale : 0.f;
  f32x16 dq[ND];
#pragma unroll
  for (int nd = 0; nd < ND; ++nd)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[nd][r] = 0.f;

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) { a2_pin(qf[ks]); a2_pin(dof[ks]); }
  a2_pin(lo); a2_pin(hi); a2_pin(lse2); a2_pin(dl);
  __syncthreads();
  const int k_rd = (ql * PK + hi5) * 16;
  const int g16 = (l >> 4) & 1, i16 = l & 15;
  const int t_rd = ((4 * hi5 + (i16 >> 2)) * PT) * 16 + (16 * g16 + 4 * (i16 & 3)) * 2;

  for (int kt = kt_lo; kt < kt_hi; ++kt) {
    const int st = (kt - kt_lo) & 1;
    const char* sb = smem + st * STAGE;
    if (kt + 1 < kt_hi) dma_tile(kt + 1, st ^ 1);
    if (q0 < p.Sq && kt * 64 < whi && kt * 64 + 64 > wlo) {
      bool full = __all(kt * 64 >= lo && kt * 64 + 64 <= hi);
#pragma unroll
      for (int kb2 = 0; kb2 < 2; ++kb2) {                    // 32-key block of the tile
        const int kbase = kt * 64 + kb2 * 32;
        if (kbase >= whi || kbase + 32 <= wlo) continue;     // wave-uniform: nobody in this wave sees these 32 keys
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          bf16x8 kf = *(const bf16x8*)(sb + k_rd + kb2 * 32 * PK * 16 + ks * 32);
          bf16x8 vf = *(const bf16x8*)(sb + OFF_V + k_rd + kb2 * 32 * PK * 16 + ks * 32);
          s = MFMA32(kf, qf[ks], s);
          dp = MFMA32(vf, dof[ks], dp);
        }
        const int key0 = kbase + 4 * hi5;
        bf16x8 dsf[2];
        // two copies of the element loop behind a wave-uniform BRANCH: with the mask as a select inside one loop it compiled to 4
        // compare / select instructions per element on every block (64 of the 140 vector instructions of a 32-key block), although
        // only the blocks on the diagonal or at a sequence end need it; each copy still consumes s[r] / dp[r] as it goes (registers)
        if (full) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float sv = ALIBI ? s[r] + ab * (float)(key0 + 8 * (r >> 2) + (r & 3)) : s[r];
            float pr = EXP2(fmaf(sv, sc2, -lse2));
            dsf[r >> 3][r & 7] = f2bf(pr * (dp[r] - dl) * p.scale);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            int key = key0 + 8 * (r >> 2) + (r & 3);
            float sv = ALIBI ? s[r] + ab * (float)key : s[r];
            float pr = EXP2(fmaf(sv, sc2, -lse2));
            pr = (key >= lo && key < hi) ? pr : 0.f;
            dsf[r >> 3][r & 7] = f2bf(pr * (dp[r] - dl) * p.scale);
          }
        }
#pragma unroll
        for (int nd = 0; nd < ND; ++nd)
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            const char* a = sb + OFF_KT + t_rd + (kb2 * 32 + half * 16) * PT * 16 + nd * 64;
            s16x4 tlo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, a));
            s16x4 thi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, a + 8 * PT * 16));
            union { struct { s16x4 a, b; } s; bf16x8 v; } u;
            u.s.a = tlo; u.s.b = thi;
            dq[nd] = MFMA32(u.v, dsf[half], dq[nd]);
          }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  a2_store_rows<D, ND>(smem + wave * (32 * (D * 2 + 16)), dq, 1.f, p.dq + b * p.dq_bs + h * p.dq_hs, p.dq_ss, q0, p.Sq,
                       p.rope_cos, p.rope_sin, p.rope_half, p.rope_step);
}

// ------------------------------------------------------------------------------------------- dK, dV
// block = 128 keys, wave = 32 keys (K, V fragments as B operands in registers, dK^T / dV^T accumulators [d][key]); sweeps the
// query tiles of 32 rows.  Per tile five LDS images, ALL by LDS-DMA (no compiler-visible load inside the loop, so the counted
// vmcnt below is exact): Q-row, Q-tr, dO-row, dO-tr and a 384-byte row block {lse[32], delta[32], seg[32]}.  Three stages,
// tiles fetched TWO ahead: at 2 waves per SIMD (the accumulators take 192 registers) one tile of compute does not cover the
// latency of a first-touch fetch.  Every wave issues exactly NT instructions per tile (the last ones are dummies into a
// scratch slot) so that `s_waitcnt vmcnt(NT)` means "everything but the newest tile has landed" on every wave.
template <int D, bool ALIBI>
__global__ __launch_bounds__(256, D == 128 ? 1 : 2) void attn_dkv2_kernel(AttnP p, int nx, int dbg) {
  using C = A2Cfg<D>;
  constexpr int CPR = C::CPR, PK = C::PK, PV = C::PV, KS = C::KS, ND = C::ND;
  constexpr int NW = 4, NIR = (32 * PK + 63) / 64, NIT = PV / 2, NI = 2 * (NIR + NIT) + 1, NT = (NI + NW - 1) / NW;
  constexpr int OFF_QT = NIR * 1024, OFF_DO = OFF_QT + NIT * 1024, OFF_DOT = OFF_DO + NIR * 1024, OFF_AUX = OFF_DOT + NIT * 1024;
  constexpr int STAGE = OFF_AUX + 1024, NST = 3, OFF_SCRATCH = NST * STAGE;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  int kblk, h, b;
  a2_decode(nx, p.H, p.B, kblk, h, b);
  kblk = nx - 1 - kblk;                                      // causal: the FIRST key block sees the most query tiles
  if (!attn_varlen(p, b) || kblk * 128 >= p.Sk) return;      // packed rows: the grid covers the longest sequence
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = lane_id(), hi5 = l >> 5, kl = l & 31;
  const int key0 = kblk * 128 + wave * 32;
  const char* qb = (const char*)(p.q + b * p.q_bs + h * p.q_hs);
  const char* dob = (const char*)(p.d_o + b * p.do_bs + h * p.do_hs);
  const bf16* kbp = p.k + b * p.k_bs + h * p.k_hs;
  const bf16* vbp = p.v + b * p.v_bs + h * p.v_hs;
  const int kfirst = kblk * 128, klast = min(kblk * 128 + 127, p.Sk - 1);
  const int kvl_b = p.kv_len ? p.kv_len[b] : p.Sk;
  const int mode = p.mask_mode;

  const int nqt = (p.Sq + 31) >> 5;
  int qt_a = 0, qt_b = nqt;
  if (mode == UNIMP_MASK_CAUSAL) qt_a = kfirst >> 5;
  else if (mode == UNIMP_MASK_SEGMENT) {
    while (qt_a < nqt) { int tb = p.seg[(long)b * p.SqS + min(qt_a * 32 + 31, p.Sq - 1)]; if (tb * p.seg_len > kfirst) break; ++qt_a; }
    while (qt_b > qt_a) { int ta = p.seg[(long)b * p.SqS + (qt_b - 1) * 32]; if ((max(ta, 1) - 1) * p.seg_len <= klast) break; --qt_b; }
  }

  // DMA plan: instruction i = wave + 4 t.  Images 0..3 = Q-row, Q-tr, dO-row, dO-tr; instruction NI - 1 = the row block;
  // i >= NI: dummy.  Per lane and t one packed register: row (bits 0..7) | byte column (bits 8..).
  uint32_t d_plan[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    int i = wave + NW * t;
    int img = i < NIR ? 0 : (i < NIR + NIT ? 1 : (i < 2 * NIR + NIT ? 2 : 3));
    int ii = i - (img == 0 ? 0 : (img == 1 ? NIR : (img == 2 ? NIR + NIT : 2 * NIR + NIT)));
    int pitch = (img & 1) ? PV : PK;
    int s_ = 64 * ii + l;
    int row = min(s_ / pitch, 31), cs = s_ - (s_ / pitch) * pitch;
    d_plan[t] = (uint32_t)row | ((uint32_t)(min(cs, CPR - 1) * 16) << 8);
  }
  const uint32_t smem_lds = (uint32_t)(uintptr_t)LDS_PTR(char, smem);
  const long q_sb = p.q_ss * 2, do_sb = p.do_ss * 2;
  const char* lse_b = (const char*)(p.lse + ((long)b * p.H + h) * p.SqS);
  const char* dl_b = (const char*)(p.delta + ((long)b * p.H + h) * p.SqS);
  const char* seg_b = mode == UNIMP_MASK_SEGMENT ? (const char*)(p.seg + (long)b * p.SqS) : lse_b;
  auto dma_tile = [&](int qt, int stage) {
    int rows_left = p.Sq - qt * 32;
    const char* q_base = qb + (long)qt * 32 * q_sb;
    const char* do_base = dob + (long)qt * 32 * do_sb;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      int i = wave + NW * t;                                  // wave-uniform
      if (i < NI - 1) {
        int img = i < NIR ? 0 : (i < NIR + NIT ? 1 : (i < 2 * NIR + NIT ? 2 : 3));
        int ii = i - (img == 0 ? 0 : (img == 1 ? NIR : (img == 2 ? NIR + NIT : 2 * NIR + NIT)));
        int rc = min((int)(d_plan[t] & 255u), rows_left - 1);
        uint32_t off = (uint32_t)(rc * (img >= 2 ? do_sb : q_sb)) + (d_plan[t] >> 8);
        uint32_t dst = smem_lds + stage * STAGE + (img == 0 ? 0 : (img == 1 ? OFF_QT : (img == 2 ? OFF_DO : OFF_DOT))) + ii * 1024;
        a2_glds(img >= 2 ? do_base : q_base, off, __builtin_amdgcn_readfirstlane(dst));
      } else if (i == NI - 1) {
        // row block: lanes 0-7 lse, 8-15 delta, 16-23 seg (4 rows of 4 bytes per lane); chunk starts clamped into the row
        int j = l & 7, arr = (l >> 3) & 3;
        int r0 = max(min(qt * 32 + 4 * j, p.Sq - 4), 0);
        const char* base = arr == 0 ? lse_b : (arr == 1 ? dl_b : seg_b);          // one instruction, three source arrays
        a2_glds_v(base + (long)r0 * 4, __builtin_amdgcn_readfirstlane(smem_lds + stage * STAGE + OFF_AUX));
      } else {
        a2_glds(q_base, 0u, __builtin_amdgcn_readfirstlane(smem_lds + OFF_SCRATCH));
      }
    }
  };

  bf16x8 kf[KS], vf[KS];
  if (!(dbg & 16)) {
    int key = min(key0 + kl, p.Sk - 1);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      kf[ks] = *(const bf16x8*)(kbp + (long)key * p.k_ss + ks * 16 + hi5 * 8);
      vf[ks] = *(const bf16x8*)(vbp + (long)key * p.v_ss + ks * 16 + hi5 * 8);
    }
  } else {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { kf[ks] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; vf[ks] = kf[ks]; }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) { a2_pin(kf[ks]); a2_pin(vf[ks]); }
  f32x16 dk[ND], dv[ND];
#pragma unroll
  for (int nd = 0; nd < ND; ++nd)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[nd][r] = 0.f; dv[nd][r] = 0.f; }
  const float sc2 = p.scale * LOG2E;
  const float ab = ALIBI ? p.alibi[h] / p.scale : 0.f;
  const int mykey = key0 + kl;

  if (qt_a < qt_b && !(dbg & 32)) dma_tile(qt_a, 0);
  if (qt_a + 1 < qt_b && !(dbg & 32)) { dma_tile(qt_a + 1, 1); asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NT) : "memory"); }
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int r_rd = (kl * PK + hi5) * 16;                                                          // row image: row = l & 31 (a query row here)
  const int g16 = (l >> 4) & 1, i16 = l & 15;
  const int t_rd = ((4 * hi5 + (i16 >> 2)) * PV) * 16 + (16 * g16 + 4 * (i16 & 3)) * 2;
  // key range of query row qr, from the staged seg value (segment mode) or arithmetic
  auto range_of = [&](int qr, int segv, int& lo_, int& hi_) {
    lo_ = 0; hi_ = 0;
    if (qr < p.Sq) {
      if (mode == UNIMP_MASK_NONE) hi_ = kvl_b;
      else if (mode == UNIMP_MASK_CAUSAL) hi_ = min(qr + 1, kvl_b);
      else if (segv > 0) { lo_ = (segv - 1) * p.seg_len; hi_ = min(segv * p.seg_len, p.Sk); }
    }
  };

  int stage = 0;
  for (int qt = qt_a; qt < qt_b; ++qt) {
    const char* sb = smem + stage * STAGE;
    if (qt + 2 < qt_b && !(dbg & 2)) { int s2 = stage + 2; if (s2 >= NST) s2 -= NST; dma_tile(qt + 2, s2); }
    if (key0 < p.Sk && !(dbg & 1)) {
      const float* ax = (const float*)(sb + OFF_AUX);
      const int* axi = (const int*)ax;
      // lane's own query row of the tile (row kl): does the wave's key range meet it at all / cover it fully?
      int lo_l, hi_l;
      {
        int qr = qt * 32 + kl;
        int c0 = qt * 32 + (kl & ~3), r0 = max(min(c0, p.Sq - 4), 0);
        int idx = (kl & ~3) + (min(qr, p.Sq - 1) - r0);
        range_of(qr, axi[64 + idx], lo_l, hi_l);
      }
      bool none = __all(!(lo_l < key0 + 32 && hi_l > key0));
      if (!none) {
        bool full = __all(lo_l <= key0 && hi_l >= key0 + 32) && key0 + 32 <= p.Sk;
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          bf16x8 qf_ = *(const bf16x8*)(sb + r_rd + ks * 32);
          bf16x8 dof_ = *(const bf16x8*)(sb + OFF_DO + r_rd + ks * 32);
          s = MFMA32(qf_, kf[ks], s);
          dp = MFMA32(dof_, vf[ks], dp);
        }
        // lane holds key = mykey and query rows 8 g + 4 hi5 + e (g = r >> 2, e = r & 3) of the tile
        bf16x8 pf[2], dsf[2];
        const float kbias = ALIBI ? ab * (float)mykey : 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int e0 = 8 * g + 4 * hi5;                       // first of 4 consecutive rows: one 16-byte chunk of each array
          int r0 = max(min(qt * 32 + e0, p.Sq - 4), 0);
          int sh = qt * 32 + e0 - r0;                           // > 0 only in the last, ragged tile: rows e map to chunk index e + sh (clamped)
          f32x4 l4 = *(const f32x4*)(ax + e0), d4 = *(const f32x4*)(ax + 32 + e0);
          typedef __attribute__((ext_vector_type(4))) int i32x4;
          i32x4 s4 = *(const i32x4*)(axi + 64 + e0);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            int r = 4 * g + e;
            int ei = min(e + sh, 3);
            float lse_e = sh == 0 ? l4[e] : (ei == 0 ? l4[0] : ei == 1 ? l4[1] : ei == 2 ? l4[2] : l4[3]);
            float dl_e = sh == 0 ? d4[e] : (ei == 0 ? d4[0] : ei == 1 ? d4[1] : ei == 2 ? d4[2] : d4[3]);
            float pr = EXP2(fmaf(s[r] + kbias, sc2, -lse_e * LOG2E));
            if (!full) {
              int sg = sh == 0 ? s4[e] : (ei == 0 ? s4[0] : ei == 1 ? s4[1] : ei == 2 ? s4[2] : s4[3]);
              int lo_, hi_;
              range_of(qt * 32 + e0 + e, sg, lo_, hi_);
              pr = (mykey >= lo_ && mykey < hi_) ? pr : 0.f;
            }
            pf[r >> 3][r & 7] = f2bf(pr);
            dsf[r >> 3][r & 7] = f2bf(pr * (dp[r] - dl_e) * p.scale);
          }
        }
#pragma unroll
        for (int nd = 0; nd < ND; ++nd)
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            const char* a = sb + t_rd + half * 16 * PV * 16 + nd * 64;
            union { struct { s16x4 a, b; } s; bf16x8 v; } uq, ud;
            uq.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, a + OFF_QT));
            uq.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, a + OFF_QT + 8 * PV * 16));
            ud.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, a + OFF_DOT));
            ud.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, a + OFF_DOT + 8 * PV * 16));
            dv[nd] = MFMA32(ud.v, pf[half], dv[nd]);
            dk[nd] = MFMA32(uq.v, dsf[half], dk[nd]);
          }
      }
    }
    // tile qt + 1 must have landed; tile qt + 2 (just issued) may stay in flight
    if (qt + 2 < qt_b && !(dbg & 2)) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NT) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!(dbg & 4)) __syncthreads();
    if (++stage == NST) stage = 0;
  }
  if (!(dbg & 8)) {
    char* ew = smem + wave * (32 * (D * 2 + 16));
    a2_store_rows<D, ND>(ew, dk, 1.f, p.dk + b * p.dk_bs + h * p.dk_hs, p.dk_ss, key0, p.Sk, p.rope_cos, p.rope_sin, p.rope_half, p.rope_step);
    __builtin_amdgcn_wave_barrier();
    a2_store_rows<D, ND>(ew, dv, 1.f, p.dv + b * p.dv_bs + h * p.dv_hs, p.dv_ss, key0, p.Sk);
  }
}

template <int D, bool ALIBI>
static void launch_dkv2(const AttnP& p, int nk, hipStream_t s) {
  using C = A2Cfg<D>;
  constexpr int NIR = (32 * C::PK + 63) / 64, NIT = C::PV / 2;
  constexpr size_t lds = 3 * ((2 * (NIR + NIT) + 1) * 1024) + 1024;
  static bool attr_set = false;
  auto kern = attn_dkv2_kernel<D, ALIBI>;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; }
  static const int dbg = [] { const char* e = getenv("UNIMP_A2_DBG"); return e ? atoi(e) : 0; }();
  hipLaunchKernelGGL(kern, dim3(nk * p.H * p.B), dim3(256), lds, s, p, nk, dbg);
}

// which = 1: dQ kernel, 2: dK/dV kernel, 3: both
int unimp_attn_bwd2_dispatch(const AttnP& p, int which, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  int nq = (p.Sq + 127) / 128, nk = (p.Sk + 127) / 128;
  dim3 gq(nq * p.H * p.B), block(256);
#define BWD2(D_, A_) do { if (which & 1) hipLaunchKernelGGL((attn_dq2_kernel<D_, A_>), gq, block, 0, s, p, nq); if (which & 2) launch_dkv2<D_, A_>(p, nk, s); } while (0)
#define BWD2A(A_) do { if (p.D == 64) BWD2(64, A_); else if (p.D == 80) BWD2(80, A_); else BWD2(128, A_); } while (0)
  if (p.alibi) BWD2A(true); else BWD2A(false);
#undef BWD2A
#undef BWD2
  return unimp_check_launch("attn_bwd2");
}

// UNIMP_ATTN_VIT / unimp_attn_set_vit_tail: -1 = not read yet, 0 = the general forward path for S = 257, 1 = the seeded form (default)
static int g_attn_vit_tail = -1;
extern "C" int unimp_attn_set_vit_tail(int on) {
  if (g_attn_vit_tail < 0) { const char* e = getenv("UNIMP_ATTN_VIT"); g_attn_vit_tail = (!e || atoi(e) != 0) ? 1 : 0; }
  int old = g_attn_vit_tail;
  g_attn_vit_tail = on ? 1 : 0;
  return old;
}

template <int D, bool ALIBI>
static void launch_fwd2(const AttnP& p, hipStream_t s) {
  // a wave owns 32 query rows; 4 or 5 waves per block, whichever wastes fewer wave slots: S = 257 (the ViT: 256 patches + CLS)
  // is 9 waves of rows = 2 blocks of 5 instead of 3 blocks of 4 with a third nearly empty one
  int w = (p.Sq + 31) / 32;
  int b4 = (w + 3) / 4, b5 = (w + 4) / 5;
  if constexpr (D == 64 && !ALIBI) {
    // the ViT (S = 257, every key visible): the last key seeds the softmax state (TAIL: 4 tiles instead of 5) and the 9 waves of
    // query rows go out as 3 blocks of 3 (no idle tenth wave slot).  Measured at 512 images x 16 heads (profiles/r04_vit_attention_ab.txt):
    // 446 us before; TAIL with 5 / 9 / 3 waves per block 405 / 413 / 361 us.  UNIMP_ATTN_VIT=0: the general path.
    if (g_attn_vit_tail < 0) { const char* e = getenv("UNIMP_ATTN_VIT"); g_attn_vit_tail = (!e || atoi(e) != 0) ? 1 : 0; }
    const bool vit = g_attn_vit_tail != 0;
    if (vit && p.mask_mode == UNIMP_MASK_NONE && !p.kv_len && !p.k_off && !p.q_off && (p.Sk & 63) == 1 && p.Sk > 64 && w == 9) {
      hipLaunchKernelGGL((attn_fwd2_kernel<D, 3, false, true>), dim3(3 * p.H * p.B), dim3(192), 0, s, p, 0, 3);
      return;
    }
  }
  if (b5 * 5 < b4 * 4) hipLaunchKernelGGL((attn_fwd2_kernel<D, 5, ALIBI>), dim3(b5 * p.H * p.B), dim3(320), 0, s, p, 0, b5);
  else hipLaunchKernelGGL((attn_fwd2_kernel<D, 4, ALIBI>), dim3(b4 * p.H * p.B), dim3(256), 0, s, p, 0, b4);
}

int unimp_attn_fwd2_dispatch(const AttnP& p, void* stream) {
  hipStream_t s = (hipStream_t)stream;
#define FWD2(A_) do { if (p.D == 64) launch_fwd2<64, A_>(p, s); else if (p.D == 80) launch_fwd2<80, A_>(p, s); else launch_fwd2<128, A_>(p, s); } while (0)
  if (p.alibi) FWD2(true); else FWD2(false);
#undef FWD2
  return unimp_check_launch("attn_fwd2");
}
