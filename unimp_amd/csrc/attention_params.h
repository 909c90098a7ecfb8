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
};

// Inverse (transpose) half-split rotation of one 16-byte chunk of a gradient row that is staged in LDS as bf16 [D]:
// chunk c holds dims 8c .. 8c + 7; its partner (dims +- half) is chunk c +- half / 8 of the same row.  Same arithmetic
// as rope_kernel<8>(inverse) on the stored bf16 values (elementwise.hip), so fusing it here changes no bits.
__device__ __forceinline__ u32x4 attn_rope_inv_chunk(const char* row, int c, int half, const float* __restrict__ cosr,
                                                     const float* __restrict__ sinr) {
  const int hc = half >> 3;
  bf16x8 x = *(const bf16x8*)(row + c * 16);
  if (c < 2 * hc) {
    const bool first = c < hc;
    bf16x8 y = *(const bf16x8*)(row + (first ? c + hc : c - hc) * 16);
    const int t = (first ? c : c - hc) * 8;
    f32x4 c0 = *(const f32x4*)(cosr + t), c1 = *(const f32x4*)(cosr + t + 4);
    f32x4 s0 = *(const f32x4*)(sinr + t), s1 = *(const f32x4*)(sinr + t + 4);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float co = j < 4 ? c0[j & 3] : c1[j & 3], si = -(j < 4 ? s0[j & 3] : s1[j & 3]);
      float x1 = bf2f(first ? x[j] : y[j]), x2 = bf2f(first ? y[j] : x[j]);
      o[j] = first ? f2bf(x1 * co - x2 * si) : f2bf(x2 * co + x1 * si);
    }
    x = o;
  }
  union { bf16x8 b; u32x4 u; } cv; cv.b = x;
  return cv.u;
}

int unimp_attn_fwd2_dispatch(const AttnP& p, void* stream);      // attention2.hip
int unimp_attn_bwd2_dispatch(const AttnP& p, int which, void* stream);      // attention2.hip (after the delta kernel): 1 dQ, 2 dK/dV
