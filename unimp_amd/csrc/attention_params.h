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
};

int unimp_attn_fwd2_dispatch(const AttnP& p, void* stream);      // attention2.hip
int unimp_attn_bwd2_dispatch(const AttnP& p, int which, void* stream);      // attention2.hip (after the delta kernel): 1 dQ, 2 dK/dV
