// Kernel-side parameter block shared by the attention kernels (attention.hip, attention2.hip); filled from the C-ABI
// descriptor unimp_attn_desc (include/unimp_hip.h) by attention.hip::fill().
#pragma once
#include "common.h"

struct AttnP {
  const bf16* q; const bf16* k; const bf16* v; bf16* o; float* lse;
  long q_bs, q_ss, q_hs, k_bs, k_ss, k_hs, v_bs, v_ss, v_hs, o_bs, o_ss, o_hs;
  int B, H, Sq, Sk, D;
  float scale;
  int mask_mode;
  const int* kv_len; const int* seg; int seg_len;
  const bf16* d_o; bf16* dq; bf16* dk; bf16* dv; float* delta;
  long do_bs, do_ss, do_hs, dq_bs, dq_ss, dq_hs, dk_bs, dk_ss, dk_hs, dv_bs, dv_ss, dv_hs;
  const float* alibi;        // per-head slopes or null: raw score += slope / scale * key  (so that score * scale gains slope * key)
  const float* rope_cos; const float* rope_sin; int rope_half;     // backward: transpose half-split rotation of dq / dk rows (null: none)
  float rope_step;           // != 0: adjacent-pair layout, cos / sin computed in the epilogue (2 log2(base) / (2 rope_half)); tables unused
  // packed rows (unimp_attn_desc.q_row_off / k_row_off): the query-side rows of sequence b (q, o, dO, dq, seg) are rows q_off[b] ..
  // q_off[b] + q_len[b] - 1 of ONE [rows][H][D] buffer, the key-side rows (k, v, dk, dv) rows k_off[b] .. + kv_len[b] - 1.
  // attn_varlen() rebases the pointers of the by-value parameter block for the block's batch index and sets Sq / Sk to the
  // sequence's own counts; every kernel then addresses "its" sequence exactly as in the padded layout.
  const int* q_off; const int* q_len; const int* k_off;
  int SqS;                   // rows per (batch, head) of lse / delta [B][H][SqS] and of an unpacked seg [B][SqS]: the descriptor's Sq
};

// call once per block, right after the batch index is known; false: this sequence has no row for the block to work on
__device__ __forceinline__ bool attn_varlen(AttnP& p, int b) {
  if (p.q_off) {
    const long o = p.q_off[b];
    p.q += o * p.q_ss - (long)b * p.q_bs;
    p.o += o * p.o_ss - (long)b * p.o_bs;
    p.d_o += o * p.do_ss - (long)b * p.do_bs;            // forward: strides 0, pointer null and never used
    p.dq += o * p.dq_ss - (long)b * p.dq_bs;
    if (p.seg) p.seg += o - (long)b * p.SqS;             // seg is indexed b * SqS + row: lands on packed row o + row
    p.Sq = p.q_len[b];
    if (p.Sq <= 0) return false;
  }
  if (p.k_off) {
    const long o = p.k_off[b];
    p.k += o * p.k_ss - (long)b * p.k_bs;
    p.v += o * p.v_ss - (long)b * p.v_bs;
    p.dk += o * p.dk_ss - (long)b * p.dk_bs;
    p.dv += o * p.dv_ss - (long)b * p.dv_bs;
    p.Sk = p.kv_len[b];
    if (p.Sk <= 0) return false;
  }
  return true;
}

// Adjacent-pair form (the layout the GEMM's rotary epilogue writes, gemm_tile.h): chunk c holds the pairs (j, j + 4), j < 4, at
// frequencies 4c + j; the transpose rotation needs neither a partner chunk nor a table.
__device__ __forceinline__ u32x4 attn_rope_inv_adjacent(const char* row, int c, int half, float pos, float step) {
  bf16x8 x = *(const bf16x8*)(row + c * 16);
  if (c * 4 < half) {
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float turns = __builtin_amdgcn_fractf(pos * (__builtin_amdgcn_exp2f(-(float)(4 * c + j) * step) * 0.15915494309189535f));
      float co = __builtin_amdgcn_cosf(turns), si = __builtin_amdgcn_sinf(turns);
      float x1 = bf2f(x[j]), x2 = bf2f(x[j + 4]);
      o[j] = f2bf(x1 * co + x2 * si);
      o[j + 4] = f2bf(x2 * co - x1 * si);
    }
    x = o;
  }
  union { bf16x8 b; u32x4 u; } cv; cv.b = x;
  return cv.u;
}

// Inverse (transpose) half-split rotation of the 16-byte chunks of gradient rows that are staged in LDS as bf16 [D]:
// chunk c holds dims 8c .. 8c + 7; its partner (dims +- half) is chunk c +- half / 8 of the same row.  Same arithmetic
// as rope_kernel<8>(inverse) on the stored bf16 values (elementwise.hip), so fusing it here changes no bits.
// Two phases so that a lane's table loads for ALL its chunks are in flight together (a load inside the per-chunk bounds
// branch gets its own s_waitcnt vmcnt(0): one L2 round trip per chunk, +30 us on a 450 us kernel).
struct AttnRopeChunk { bf16x8 x, y; f32x4 c0, c1, s0, s1; int mode; };       // mode 0: pass through, 1: first half, 2: second half

__device__ __forceinline__ void attn_rope_inv_load(AttnRopeChunk& k, const char* row, int c, int half, const float* __restrict__ cosr,
                                                   const float* __restrict__ sinr) {
  const int hc = half >> 3;
  k.mode = c < hc ? 1 : (c < 2 * hc ? 2 : 0);
  const int pc = k.mode == 1 ? c + hc : (k.mode == 2 ? c - hc : c);
  const int t = (k.mode == 2 ? c - hc : (k.mode == 1 ? c : 0)) * 8;          // mode 0 reads the row's first table entries (unused)
  k.x = *(const bf16x8*)(row + c * 16);
  k.y = *(const bf16x8*)(row + pc * 16);
  k.c0 = *(const f32x4*)(cosr + t); k.c1 = *(const f32x4*)(cosr + t + 4);
  k.s0 = *(const f32x4*)(sinr + t); k.s1 = *(const f32x4*)(sinr + t + 4);
}

__device__ __forceinline__ u32x4 attn_rope_inv_apply(const AttnRopeChunk& k) {
  bf16x8 o = k.x;
  if (k.mode) {
    const bool first = k.mode == 1;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float co = j < 4 ? k.c0[j & 3] : k.c1[j & 3], si = -(j < 4 ? k.s0[j & 3] : k.s1[j & 3]);
      float x1 = bf2f(first ? k.x[j] : k.y[j]), x2 = bf2f(first ? k.y[j] : k.x[j]);
      o[j] = first ? f2bf(x1 * co - x2 * si) : f2bf(x2 * co + x1 * si);
    }
  }
  union { bf16x8 b; u32x4 u; } cv; cv.b = o;
  return cv.u;
}

int unimp_attn_fwd2_dispatch(const AttnP& p, void* stream);      // attention2.hip
int unimp_attn_bwd2_dispatch(const AttnP& p, int which, void* stream);      // attention2.hip (after the delta kernel): 1 dQ, 2 dK/dV
extern "C" int unimp_attn_dkv3_eligible(const AttnP* p);         // attention3.hip: forms its dK/dV kernel serves
extern "C" int unimp_attn_dkv3_preferred(const AttnP* p);        // ... and sizes at which it is the faster choice (a pair per CU)
int unimp_attn_dkv3_launch(const AttnP& p, void* stream);        // attention3.hip (after the dQ kernel, which publishes delta)
