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
// LDS-DMA with the lanes chosen by a scalar mask (the compiler sees no control flow): a piece that only has to be COUNTED moves 16 bytes
__device__ __forceinline__ void a3_glds_v_masked(const void* vaddr, uint32_t lds_dst, unsigned long long mask) {
  unsigned long long save;
  asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %3\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b64 exec, %0"
               : "=&s"(save) : "v"(vaddr), "s"(lds_dst), "s"(mask) : "memory", "m0");
}
// the value lives in the accumulator half of the register file from here on (its only readers are MFMA operands)
template <typename T> __device__ __forceinline__ void a3_pin_acc(T& v) { asm volatile("" : "+a"(v)); }

// Epilogue of this kernel: a wave's transposed accumulator tile X^T[d][row] (lane: row = l & 31, d = 32 nd + 8 g + 4 hi5 + e) -> 32 global rows of D bf16
// through the wave's staging area, as whole 16-byte chunks (a2_store_rows' scheme), specialised to what this kernel can meet: all 32 rows exist
// (Sk is a multiple of 32), the 320 chunks are exactly five per lane, the rotation is none (ROT 0), the adjacent-pair form over part of the row (1) or
// over the whole row (2: rotary_pct = 1, the LM's) -- fixed at compile time.  With one wave per SIMD every instruction of an item's drain is exposed:
// the general function's three run-time rotation forms, bounds branches and 64-bit row products were ~3 000 instructions per item.
// Same arithmetic as attn_rope_inv_adjacent (attention_params.h) on the same staged bf16 values: no bit changes.
template <int D, int ND, int ROT>
__device__ __forceinline__ void a3_store_rows(char* lds_wave, const f32x16 (&acc)[ND], float mul, bf16* __restrict__ gbase, uint32_t row_stride, int row0,
                                              int rope_half, float rope_step, int l) {
  constexpr int PITCH = D * 2 + 16, CPR = D / 8, NCH = 32 * CPR / 64;
  static_assert(32 * CPR % 64 == 0, "whole chunks per lane");
  const int hi5 = l >> 5, rl = l & 31;
#pragma unroll
  for (int nd = 0; nd < ND; ++nd)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      if (32 * nd + 8 * g + 4 < D) {                          // d0 = 32 nd + 8 g + 4 hi5 < D for both halves (D is a multiple of 8)
        const int d0 = 32 * nd + 8 * g + 4 * hi5;
        bf16x4 w = {f2bf(acc[nd][4 * g] * mul), f2bf(acc[nd][4 * g + 1] * mul), f2bf(acc[nd][4 * g + 2] * mul), f2bf(acc[nd][4 * g + 3] * mul)};
        *(bf16x4*)(lds_wave + rl * PITCH + d0 * 2) = w;
      }
  __builtin_amdgcn_s_waitcnt(0xc07f);                         // lgkmcnt(0): the wave's own LDS writes have landed (wave-private region)
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int id = l + 64 * i;
    const int r = id / CPR, c = id - r * CPR;
    bf16x8 x = *(const bf16x8*)(lds_wave + r * PITCH + c * 16);
    if (ROT != 0) {
      const float pos = (float)(row0 + r);
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float turns = __builtin_amdgcn_fractf(pos * (__builtin_amdgcn_exp2f(-(float)(4 * c + j) * rope_step) * 0.15915494309189535f));
        float co = __builtin_amdgcn_cosf(turns), si = __builtin_amdgcn_sinf(turns);
        float x1 = bf2f(x[j]), x2 = bf2f(x[j + 4]);
        o[j] = f2bf(x1 * co + x2 * si);
        o[j + 4] = f2bf(x2 * co - x1 * si);
      }
      if (ROT == 2) x = o;
      else {                                                  // chunks beyond the rotated part pass through: a select, not a branch
        const bool rot = c * 4 < rope_half;
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = rot ? o[j] : x[j];
      }
    }
    union { bf16x8 b; u32x4 u; } cv; cv.b = x;
    // row and row pitch below 2^24 elements, the slice below 4 GiB (unimp_attn_dkv3_eligible): one 24-bit multiply per chunk
    *(u32x4*)((char*)gbase + (size_t)(__umul24((uint32_t)(row0 + r), row_stride) + (uint32_t)c * 8u) * 2u) = cv.u;
  }
}

// LDS of the workgroup (4 waves): four tile stages (Q image + dO image each, row pitch 12 chunks); per wave the staging of its two
// key blocks' K and V rows -- four 32-row images at the odd pitch of 11 chunks = 22 DMA pieces; per wave the epilogue staging.
constexpr int A3_PT = 12, A3_IMG = 32 * A3_PT * 16, A3_STAGE = 2 * A3_IMG, A3_KV0 = 4 * A3_STAGE;
constexpr int A3_KVW = 22 * 1024, A3_EP0 = A3_KV0 + 4 * A3_KVW, A3_EPW = 6144;      // 128 rows of a wave, six per 1-KiB piece
constexpr size_t A3_LDS = A3_EP0 + 4 * A3_EPW;

// One workgroup per CU, each wave alone on its SIMD (512 registers): nothing is co-resident to hide an item's start-up (K / V rows
// in, first tiles) or drain (dK / dV out) -- 55 % of the first version's run time (tools/bench_attn3_parts.py).  So the workgroups
// are persistent and an item's memory phases ride inside its neighbours' tile loops:
//   * workgroup g walks the (batch, head) pairs g, g + G, ... and inside a pair the key super-blocks first to last (a later
//     super-block re-reads Q / dO tiles the earlier one pulled through the XCD's L2); with fewer pairs than workgroups (item_mode)
//     the (pair, super-block) items are dealt out one by one instead;
//   * the NEXT item's K / V rows go into the wave's staging during the first eight iterations of the current item's loop (three
//     pieces per iteration), its first three tiles take the place of the fetches "past the end" of the current item's last three
//     iterations: when an item starts, its operands are in LDS;
//   * all of that is confirmed landed BEFORE the epilogue issues its stores (stores count in vmcnt and complete out of order with
//     loads: a counted wait behind them would wait for their acknowledgements) and the next item's first two iterations need no
//     wait at all -- by its third the stores have drained.
template <int D, int ROT>
__global__ __launch_bounds__(256, 1) void attn_dkv3_kernel(AttnP p, int nx, int item_mode, int dbg) {
  static_assert(D == 80, "image layout below: 10 chunks per row in a pitch of 12");
  constexpr int CPR = D / 8, PT = A3_PT, KS = D / 16, ND = (D + 31) / 32, NW = 4;
  constexpr int IMG = A3_IMG, NJ = IMG / 1024, OFF_DO = IMG, STAGE = A3_STAGE;
  constexpr int NT = (2 * NJ) / NW;                           // tile DMA instructions per wave and tile
  constexpr int NKJ = A3_KVW / 1024;                          // K / V staging of a wave: 128 rows in NKJ pieces of six
  constexpr int KVI = 3, KVN = 8;                             // K / V pieces per iteration, iterations that carry them
  static_assert((2 * NJ) % NW == 0 && A3_KVW % 1024 == 0 && KVI * KVN >= NKJ, "pieces divide over the waves; the staging is whole pieces");
  constexpr int WKEYS = 64 * NW;                              // keys per workgroup
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = lane_id(), hi5 = l >> 5, kl = l & 31;
  const bool causal = p.mask_mode == UNIMP_MASK_CAUSAL;
  const uint32_t q_sb = (uint32_t)(p.q_ss * 2), do_sb = (uint32_t)(p.do_ss * 2);
  const uint32_t smem_lds = (uint32_t)(uintptr_t)LDS_PTR(char, smem);
  const float sc2 = p.scale * LOG2E;
  // fragment addresses inside a tile image.  Row fragments (ds_read_b128, lane: row kl, chunk 2 ks + hi5): chunks hi5, 4 + hi5,
  // 8 + hi5 (ks 0, 2, 4) share the swizzled low bits, chunks 2 + hi5, 6 + hi5 (ks 1, 3) the other pair.
  const int kq = (kl >> 2) & 3;
  const int r_a = (kl * PT + (hi5 ^ kq)) * 16, r_b = (kl * PT + ((hi5 ^ kq) ^ 2)) * 16;
  // transposed fragments (two ds_read_b64_tr_b16: rows 4 hi5 + (i16 >> 2) and + 8 of a 16-row group; 32 g16 + 8 (i16 & 3) bytes
  // into a 64-byte group of columns): a row's swizzle key is 2 half + hi5
  const int g16 = (l >> 4) & 1, i16 = l & 15;
  const int trow = 4 * hi5 + (i16 >> 2), tc = 2 * g16 + ((i16 & 3) >> 1);
  const int t_0 = (trow * PT + (tc ^ hi5)) * 16 + 8 * (i16 & 1);
  const int t_1 = ((trow + 8) * PT + (tc ^ (2 + hi5))) * 16 + 8 * (i16 & 1);
  char* const kvw = smem + A3_KV0 + wave * A3_KVW;            // this wave's K / V staging: K blocks a, b, V blocks a, b
  char* const epw = smem + A3_EP0 + wave * A3_EPW;            // this wave's epilogue staging

  // Tile DMA plan: instruction i = wave + NW t (t < NT) moves piece i % NJ of image i / NJ (0: Q, 1: dO).  Slot s = 64 piece + lane
  // of an image holds row s / PT, chunk (s % PT) ^ ((row >> 2) & 3).  Chunks 10, 11 of a row are padding -- and chunk 10 of the Q
  // image's rows 0-7 / 8-15 carries the tile's lse / delta (four rows of 4 bytes per chunk): no separate row-block instruction,
  // every wave issues exactly NT per tile.  Per lane and t one 32-bit byte offset from the tile's first row (a lane of the lse /
  // delta chunks: from the tile's first lse / delta entry) -- the same for every tile of every item.
  uint32_t voff[NT];
  int lane_arr0 = 0;                                          // this lane in the wave's first piece: 0 = a Q chunk, 1 = an lse chunk, 2 = a delta chunk
  static_assert(NW <= NJ && 3 <= NW, "instruction t = 0 of every wave is a Q piece, and Q pieces 0 - 2 are t = 0 pieces");
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    int i = wave + NW * t;
    int img = i >= NJ, j = i - img * NJ;
    int s_ = 64 * j + l;
    int r = s_ / PT, cp = s_ - r * PT;
    int c = cp ^ ((r >> 2) & 3);
    const bool aux = !img && c == CPR && r < 16;
    if (t == 0 && aux) lane_arr0 = r < 8 ? 1 : 2;
    voff[t] = aux ? (uint32_t)(16 * (r & 7)) : (uint32_t)r * (img ? do_sb : q_sb) + (uint32_t)min(c, CPR - 1) * 16;
  }

  // an item = (pair, key super-block); round rnd of this workgroup
  struct Item { bool valid; int h, b, kbase, kvl, qt_a; const char *qb, *dob, *kbp, *vbp, *lse_b, *dl_b; };
  const int npair = p.H * p.B, G = gridDim.x, qt_b = p.Sq >> 5;
  auto item_of = [&](int rnd) {
    Item it;
    int pair, kblk;
    if (item_mode) { const int id = blockIdx.x + rnd * G; kblk = id / npair; pair = id - kblk * npair; it.valid = kblk < nx; }
    else { pair = (rnd / nx) * G + blockIdx.x; kblk = rnd % nx; it.valid = pair < npair; }
    if (!it.valid) { pair = 0; kblk = 0; }                    // pointers stay inside the tensors: fetches past the last item are harmless
    it.h = pair % p.H; it.b = pair / p.H;
    it.kbase = kblk * WKEYS;
    it.kvl = p.kv_len ? min(p.kv_len[it.b], p.Sk) : p.Sk;
    it.qt_a = causal ? it.kbase >> 5 : 0;
    if (it.kbase >= it.kvl) it.qt_a = qt_b;                   // every key of the super-block is padding: no tiles, dk = dv = 0
    it.qb = (const char*)(p.q + it.b * p.q_bs + it.h * p.q_hs);
    it.dob = (const char*)(p.d_o + it.b * p.do_bs + it.h * p.do_hs);
    it.kbp = (const char*)(p.k + it.b * p.k_bs + it.h * p.k_hs);
    it.vbp = (const char*)(p.v + it.b * p.v_bs + it.h * p.v_hs);
    it.lse_b = (const char*)(p.lse + ((long)it.b * p.H + it.h) * p.SqS);
    it.dl_b = (const char*)(p.delta + ((long)it.b * p.H + it.h) * p.SqS);
    return it;
  };
  // K / V rows of an item's two blocks of this wave -> staging pieces [j0, j0 + n).  The wave stages 128 rows R = 32 image + row (images:
  // K block a, K block b, V block a, V block b); piece j holds rows 6 j .. 6 j + 5, lane l the chunk l % 10 of row 6 j + l / 10 (lanes 60-63:
  // the next row's first chunks again, never read): whole 160-byte row segments per ten lanes (the per-lane 16-byte gathers of the
  // first version took 7 us per item) and an address a lane forms from two constants of its own.  Rows clamped to Sk; pieces past the
  // last: one lane of it again (uniform instruction counts without the traffic).
  const int kv_ri = l / 10, kv_c16 = (l - 10 * kv_ri) * 16;
  const uint32_t k_rb = (uint32_t)(p.k_ss * 2), v_rb = (uint32_t)(p.v_ss * 2);
  auto dma_kv = [&](const Item& it, int j0, int n) {
    const unsigned long long dv_ = (unsigned long long)(it.vbp - it.kbp);
#pragma unroll 3
    for (int jj = 0; jj < n; ++jj) {
      const unsigned long long live = j0 + jj < NKJ ? ~0ull : 1ull;
      const int j = min(j0 + jj, NKJ - 1);
      const int R = min(6 * j + kv_ri, 127);
      const uint32_t row = (uint32_t)min(it.kbase + 32 * wave + ((R & 32) << 2) + (R & 31), p.Sk - 1);
      const uint32_t mv = 0u - (uint32_t)(R >= 64);           // V rows: bitwise selection of the (scalar) distance, no select of pointers
      const unsigned long long sel = (((unsigned long long)(mv & (uint32_t)(dv_ >> 32))) << 32) | (mv & (uint32_t)dv_);
      const uint32_t rb = R >= 64 ? v_rb : k_rb;
      // row and row pitch are below 2^24, a (batch, head) slice below 4 GiB (unimp_attn_dkv3_eligible): one 24-bit multiply
      a3_glds_v_masked(it.kbp + sel + (__umul24(row, rb) + (uint32_t)kv_c16), __builtin_amdgcn_readfirstlane(smem_lds + A3_KV0 + wave * A3_KVW + j * 1024), live);
    }
  };
  // tile qt of item a -- past its last tile: the next tiles of item b from its first on -- -> a stage (branch-free: the selects are
  // scalar; tile indices clamped into the tensor: an item without tiles fetches harmlessly)
  auto dma_tile = [&](const Item& a, const Item& b_, int qt, int stage) {
    const bool own = qt < qt_b;
    const int qc = max(min(own ? qt : b_.qt_a + (qt - qt_b), qt_b - 1), 0);
    const char* q_t = (own ? a.qb : b_.qb) + (long)qc * 32 * q_sb;
    const char* do_t = (own ? a.dob : b_.dob) + (long)qc * 32 * do_sb;
    const char* lse_t = (own ? a.lse_b : b_.lse_b) + (long)qc * 128;
    const char* dl_t = (own ? a.dl_b : b_.dl_b) + (long)qc * 128;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      int i = wave + NW * t;                                  // wave-uniform
      int img = i >= NJ, j = i - img * NJ;
      uint32_t dst = smem_lds + stage * STAGE + img * IMG + j * 1024;
      if (t == 0) {
        // the wave's first piece is a Q piece (i = wave < NJ), and only Q pieces 0 - 2 carry lse / delta lanes: this one takes the per-lane
        // base -- bitwise selection of the (scalar) distances, no select of pointers (hipcc turns that into divergent branches) -- the
        // others the scalar base + 32-bit lane offset form.  No run-time choice of form anywhere.
        const unsigned long long dL = (unsigned long long)(lse_t - q_t), dD = (unsigned long long)(dl_t - q_t);
        const uint32_t mL = 0u - (uint32_t)(lane_arr0 == 1), mD = 0u - (uint32_t)(lane_arr0 == 2);
        const uint32_t lo = (mL & (uint32_t)dL) | (mD & (uint32_t)dD), hi = (mL & (uint32_t)(dL >> 32)) | (mD & (uint32_t)(dD >> 32));
        a2_glds_v(q_t + ((((unsigned long long)hi) << 32) | lo) + voff[t], __builtin_amdgcn_readfirstlane(dst));
      } else a2_glds(img ? do_t : q_t, voff[t], __builtin_amdgcn_readfirstlane(dst));
    }
  };

  // first item of this workgroup: everything it needs, confirmed
  Item cur = item_of(0);
  if (!cur.valid) return;
  int stage = 0;
  dma_kv(cur, 0, NKJ);
  dma_tile(cur, cur, cur.qt_a, 0); dma_tile(cur, cur, cur.qt_a + 1, 1); dma_tile(cur, cur, cur.qt_a + 2, 2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int rnd = 0; cur.valid; ++rnd) {
    const Item nxt = item_of(rnd + 1);
    const int h = cur.h, b = cur.b, kvl = cur.kvl, qt_a = cur.qt_a;
    const int key0a = cur.kbase + 32 * wave, key0b = key0a + 32 * NW;  // the wave's two 32-key blocks

    // K / V fragments (B operands, all k-steps) out of the staging, which is then free for the next item's rows
    bf16x8 kf[2][KS], vf[2][KS];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        // staged row R = 32 image + kl sits in piece R / 6 at row R % 6 (a 2-way bank conflict on these twenty reads per item)
        const int Rk = 32 * kb + kl, Rv = 64 + 32 * kb + kl;
        kf[kb][ks] = *(const bf16x8*)(kvw + (Rk / 6) * 1024 + (Rk % 6) * 160 + hi5 * 16 + ks * 32);
        vf[kb][ks] = *(const bf16x8*)(kvw + (Rv / 6) * 1024 + (Rv % 6) * 160 + hi5 * 16 + ks * 32);
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
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
      float nl[16];                                           // scalars, not a register tuple: only the MFMA's C operand (ndl) must be contiguous
      f32x16 ndl;
  #pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 a = *(const f32x4*)(ax + (2 * g * PT + (CPR ^ (g >> 1))) * 16);
        f32x4 d = *(const f32x4*)(ax + ((8 + 2 * g) * PT + (CPR ^ ((2 + (g >> 1)) & 3))) * 16);
  #pragma unroll
        for (int e = 0; e < 4; ++e) { nl[4 * g + e] = a[e] * -LOG2E; ndl[4 * g + e] = -d[e]; }
      }
      // lane holds key = mykey and query rows q0 + 8 g + 4 hi5 + e (r = 4 g + e); element visible <=> qmin <= row (and the key is real)
      float nlm0[16], nlm1[16];
      auto masked = [&](int mykey, float (&m)[16]) {
        const int qmin = causal ? mykey : 0;
        const int a0 = q0 + 4 * hi5 - qmin;
        const unsigned rng = mykey < kvl ? (unsigned)(p.Sq - qmin) : 0u;
  #pragma unroll
        for (int r = 0; r < 16; ++r) m[r] = (unsigned)(a0 + 8 * (r >> 2) + (r & 3)) < rng ? nl[r] : -1e30f;
      };
      if (V0 == 2) masked(mykey_a, nlm0);
      if (V1 == 2) masked(mykey_b, nlm1);
      const float* nl0 = V0 == 2 ? nlm0 : nl;
      const float* nl1 = V1 == 2 ? nlm1 : nl;
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

    // The tile loop, one straight-line loop per visibility state of the wave's two blocks.  A block's state over the tiles of a
    // sequence only moves none -> mixed -> all (causal: the diagonal passes once; keys beyond kv_len stay invisible), so the pairs
    // are visited in an order that is a chain of the product order and every loop below runs at most once.  (One loop with a
    // switch over the states made the structurizer merge the 192 accumulator registers through copies after every tile.)
    // Iteration it of the item: the next item's K / V pieces 3 it .. 3 it + 2 (from the eighth iteration on: one lane of the last piece again --
    // every iteration issues the same count), then the tile three ahead -- past this item's last
    // tile: the next item's first tiles -- then the arithmetic, then the wait for tile + 1: newer than its pieces are the pieces of
    // the last two iterations.  The first two iterations wait for nothing (tiles 1 and 2 were confirmed before the item began).
    int qt = qt_a;
    auto run = [&](auto v0c, auto v1c) {
      constexpr int V0 = decltype(v0c)::value, V1 = decltype(v1c)::value;
      // the state can only change at the tiles where a block's diagonal starts and ends (causal) -- between them it is not re-evaluated
      // (the kernel is issue-bound: two visibility tests per iteration were 5 % of its scalar instructions)
      while (qt < qt_b && vis(key0a, qt * 32) == V0 && vis(key0b, qt * 32) == V1) {
        int qe = qt_b;
        if (causal) {
          const int ta = key0a >> 5, tb = key0b >> 5;
          for (int c : {ta, ta + 1, tb, tb + 1}) if (c > qt && c < qe) qe = c;
        }
        for (; qt < qe; ++qt) {
        const int it = qt - qt_a;
        dma_kv(nxt, KVI * it, KVI);
        dma_tile(cur, nxt, qt + 3, (stage + 3) & 3);
        if ((V0 || V1) && !(dbg & 1)) tile(v0c, v1c, stage, qt * 32);                       // (measurement switch UNIMP_A3_DBG: 1 = no tile arithmetic)
        if (it >= 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NT + 2 * KVI) : "memory");
        __syncthreads();
        stage = (stage + 1) & 3;
        }
      }
    };
    run(A3V<0>{}, A3V<0>{});
    run(A3V<2>{}, A3V<0>{}); run(A3V<0>{}, A3V<2>{});
    run(A3V<1>{}, A3V<0>{}); run(A3V<2>{}, A3V<2>{}); run(A3V<0>{}, A3V<1>{});
    run(A3V<1>{}, A3V<2>{}); run(A3V<2>{}, A3V<1>{});
    run(A3V<1>{}, A3V<1>{});

    // what the loop did not get to (an item of fewer than eight / three tiles): the rest of the next item's K / V rows and first tiles
    const int n_it = qt_b - qt_a;
    if (n_it < KVN) dma_kv(nxt, KVI * n_it, NKJ - KVI * n_it);
    for (int j = 0; j < 3 - n_it; ++j) dma_tile(nxt, nxt, nxt.qt_a + j, (stage + j) & 3);       // iteration i fetched the next item's tile 3 - n_it + i
    // ... all of it landed before the first store goes out; the barrier publishes every wave's tile pieces
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // epilogue through the wave's own staging area.  The lane id is made opaque per item: hipcc otherwise hoists the epilogue's per-lane
    // constants (row / chunk indices, store addresses, the twenty rotary frequencies of the dk rotation) out of the item loop, has no
    // registers to keep them across the tile loops and reloads each from scratch behind its own vmcnt(0) -- with one wave per SIMD that
    // was 2.8 us per dk block, 97 us per launch with the rotation (profiles/r05_attention3_parts.txt); recomputing them is ~100 instructions
    int lz = l;
    asm volatile("" : "+v"(lz));
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const int key0 = kb ? key0b : key0a;
      if (key0 < p.Sk && !(dbg & 2)) {                        // (2 = no result stores)
        a3_store_rows<D, ND, ROT>(epw, dk[kb], p.scale, p.dk + b * p.dk_bs + h * p.dk_hs, (uint32_t)p.dk_ss, key0, p.rope_half, p.rope_step, lz);
        __builtin_amdgcn_wave_barrier();
        a3_store_rows<D, ND, 0>(epw, dv[kb], 1.f, p.dv + b * p.dv_bs + h * p.dv_hs, (uint32_t)p.dv_ss, key0, 0, 0.f, lz);
        __builtin_amdgcn_wave_barrier();
      }
    }
    cur = nxt;
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
  if (p->rope_cos || p->rope_sin) return 0;                   // the table form of the fused rotation stays on the earlier kernels
  if (p->dk_ss < 0 || p->dv_ss < 0 || p->dk_ss >= (1 << 24) || p->dv_ss >= (1 << 24) || (long)p->Sk * p->dk_ss * 2 >= (1ll << 32) || (long)p->Sk * p->dv_ss * 2 >= (1ll << 32)) return 0;
  if (p->Sk >= (1 << 24) || p->k_ss * 2 >= (1 << 24) || p->v_ss * 2 >= (1 << 24) || p->k_ss < 0 || p->v_ss < 0) return 0;      // 24-bit row x pitch products
  if ((long)p->Sk * p->k_ss * 2 >= (1ll << 32) || (long)p->Sk * p->v_ss * 2 >= (1ll << 32)) return 0;
  return 1;
}

static int a3_ncu() {
  static int ncu = 0;
  if (!ncu) {
    int dev = 0; hipDeviceProp_t prop;
    ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  return ncu;
}

// ... and where it is the faster choice: one persistent workgroup per CU needs a (batch, head) pair per CU to keep the chip busy and
// to carry an item's memory phases inside its neighbour's loop.  Measured at the LM's shape (profiles/r05_attention3_ab.txt, backward
// = dQ + dK/dV, this kernel vs the first generation): 3 x 32 pairs 69 vs 58 us, 6 x 32: 96 vs 94, 8 x 32: 106 vs 117, 64 x 32: 723 vs 840.
extern "C" int unimp_attn_dkv3_preferred(const AttnP* p) { return p->H * p->B >= a3_ncu(); }

int unimp_attn_dkv3_launch(const AttnP& p, void* stream) {
  constexpr int D = 80;
  static_assert(32 * (D * 2 + 16) <= A3_EPW, "epilogue staging fits");
  static_assert(A3_LDS <= 160 * 1024, "one workgroup per CU: 160 KiB of LDS");
  // rotation form of the dk epilogue, fixed per instantiation: 0 none, 1 adjacent pairs over part of the row, 2 over the whole row
  const int rot = p.rope_step == 0.f ? 0 : (2 * p.rope_half == D ? 2 : 1);
  auto kern = rot == 0 ? attn_dkv3_kernel<D, 0> : (rot == 1 ? attn_dkv3_kernel<D, 1> : attn_dkv3_kernel<D, 2>);
  static bool attr_set[3] = {false, false, false};
  if (!attr_set[rot]) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)A3_LDS); attr_set[rot] = true; }
  const int ncu = a3_ncu();
  const int nk = (p.Sk + 255) / 256, npair = p.H * p.B;
  const int item_mode = npair < ncu;
  const int grid = item_mode ? (npair * nk < ncu ? npair * nk : ncu) : ncu;
  static const int dbg = [] { const char* e = getenv("UNIMP_A3_DBG"); return e ? atoi(e) : 0; }();       // measurement switches (kernel comments)
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), A3_LDS, (hipStream_t)stream, p, nk, item_mode, dbg);
  return unimp_check_launch("attn_dkv3");
}
