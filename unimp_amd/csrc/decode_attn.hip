// Split-key decode attention (F1: Flamingo.generate with a KV cache -- eval_rec.py:100-110 and, at 256-600 new tokens,
// eval_exp.py:103-113 / eval_img_gen.py:102-111): ONE query row per (cache row, head) against every cached key.
//
// HBM-bound: the step reads the row's K and V once (2 * kv_len * D * 2 bytes per (row, head)) and does one FMA per byte, so
// there is no MFMA tile to fill -- the training kernel (one 128-query block per (row, head) walking the keys tile by tile) left
// all but R * H workgroups idle and paid its fixed per-tile cost on every 64 keys.  Here the KEYS are split: block
// (split, head, row) owns a contiguous chunk of CH keys (4 waves x CH / 4), every wave keeps its whole sub-range of K and V
// loads in flight (16-byte loads, G lanes per key row: consecutive lanes read consecutive bytes of one key), reduces q.k inside
// the lane group with DPP butterflies, runs the online softmax per lane-group slot and accumulates p * v for its 8 dims; the
// slots of a block are merged through LDS into one partial (m, l, o[D]) in fp32, and a second tiny kernel merges the splits of a
// (row, head) in a FIXED order (no atomics: bit-reproducible, graph-replayable -- the grid depends on the cache CAPACITY only,
// blocks beyond the row's kv_len publish an empty partial).
// SHARED PREFIX (beam search): the K beams of a prompt hold IDENTICAL K / V for the prompt's positions (the prompt is prefilled once
// and its cache rows repeated per beam; a beam reorder only moves the generated tail), so with `group` rows per prompt and
// `shared_len[prompt]` the prefix keys are read ONCE per prompt -- from the group's first row -- by attn_decode_prefix, whose four
// waves each take a quarter of the group's queries over the whole 128-key chunk; the per-row tails go through attn_decode_partial
// from `shared_len` on; the merge kernel folds prefix and tail partials.  At K = 10 beams and a 469-token prompt that is 10x less
// K / V traffic on 90 % of the keys.
// Scores live in the log2 domain: s = (scale * log2 e) * q.k + (slope * log2 e) * key_index (ALiBi, key-position form: the
// per-row constant of transformers' slope * (j - (L - 1)) cancels in the softmax).
#include "common.h"
#include "unimp_hip.h"

struct DecP {
  const bf16* q; const bf16* k; const bf16* v; bf16* o;
  long q_bs, q_hs, k_bs, k_ss, k_hs, v_bs, v_ss, v_hs, o_bs, o_hs;
  int B, H, Sk, D;
  float scale_log2;
  const int* kv_len;
  const float* alibi;
  float* ws;             // [B][H][nslots][D + 2]: m (log2 domain), l, o[D];  nslots = splits (plain) or 2 * splits (grouped: prefix | tail)
  int splits, chunk;     // keys per block
  int group;             // rows per prompt that share a prefix (1: none)
  const int* shared_len; // [B / group] prefix length per prompt (grouped only)
  int nslots, slot0;     // partial slots per (row, head); first slot this launch writes
};

template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ void widen8(u32x4 r, float (&f)[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) { f[2 * i] = __uint_as_float(r[i] << 16); f[2 * i + 1] = __uint_as_float(r[i] & 0xffff0000u); }
}

// G lanes per key (8: D <= 64, 16: D <= 128); U key-groups of loads in flight per round
template <int G, int U>
__global__ __launch_bounds__(256) void attn_decode_partial(DecP p) {
  constexpr int KPL = 64 / G;                       // keys per wave-wide load
  __shared__ float sm[4 * KPL * (128 + 2)];
  const int split = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int slot = lane / G, c = lane % G;          // c: 16-byte chunk of the key row
  const bool act = c * 8 < p.D;
  const int n = p.kv_len ? min(p.kv_len[b], p.Sk) : p.Sk;
  const int ks = p.group > 1 ? min(p.shared_len[b / p.group], n) : 0;       // grouped: the prefix below ks belongs to attn_decode_prefix
  const int k0 = max(split * p.chunk + wave * (p.chunk >> 2), ks), k1 = min(split * p.chunk + (wave + 1) * (p.chunk >> 2), n);

  float qf[8];
  {
    u32x4 qr = u32x4{0u, 0u, 0u, 0u};
    if (act) qr = *(const u32x4*)(p.q + (long)b * p.q_bs + (long)h * p.q_hs + c * 8);
    widen8(qr, qf);
#pragma unroll
    for (int i = 0; i < 8; ++i) qf[i] *= p.scale_log2;
  }
  const float slope = p.alibi ? p.alibi[h] * 1.4426950408889634f : 0.f;
  const int ca = act ? c * 8 : 0;                   // idle lanes of a group (D = 80: chunks 10..15) re-read chunk 0, never beyond the row
  const bf16* kb = p.k + (long)b * p.k_bs + (long)h * p.k_hs + ca;
  const bf16* vb = p.v + (long)b * p.v_bs + (long)h * p.v_hs + ca;

  float m = -INFINITY, l = 0.f, acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;

  for (int kk = k0; kk < k1; kk += KPL * U) {
    u32x4 kr[U], vr[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {                   // all loads of the round go out before the first use
      int key = kk + u * KPL + slot;
      int kc = key < k1 ? key : k0;                 // clamped, masked below: no divergent load
      kr[u] = *(const u32x4*)(kb + (long)kc * p.k_ss);
      vr[u] = *(const u32x4*)(vb + (long)kc * p.v_ss);
    }
    float s[U];
    float mx = m;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float kf[8];
      widen8(kr[u], kf);
      float d = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) d = fmaf(qf[i], kf[i], d);
      if (!act) d = 0.f;
      d = group_sum<G>(d);
      int key = kk + u * KPL + slot;
      s[u] = key < k1 ? d + slope * (float)key : -INFINITY;
      mx = fmaxf(mx, s[u]);
    }
    if (mx > m) {                                   // slot-uniform inside a lane group; rare after the first rounds
      float r = m == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(m - mx);
      l *= r;
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] *= r;
      m = mx;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float pe = s[u] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(s[u] - m);
      l += pe;
      float vf[8];
      widen8(vr[u], vf);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = fmaf(pe, vf[i], acc[i]);
    }
  }
  // slot partials -> LDS: [wave * KPL + slot][D + 2]
  const int DS = p.D + 2;
  float* mine = sm + (wave * KPL + slot) * DS;
  if (act) {
#pragma unroll
    for (int i = 0; i < 8; ++i) mine[2 + c * 8 + i] = acc[i];
    if (c == 0) { mine[0] = m; mine[1] = l; }
  }
  __syncthreads();
  // merge the 4 * KPL slots in a fixed order, one thread per dim
  const int d = threadIdx.x;
  if (d < p.D) {
    float M = -INFINITY;
    for (int j = 0; j < 4 * KPL; ++j) M = fmaxf(M, sm[j * DS]);
    float L = 0.f, O = 0.f;
    for (int j = 0; j < 4 * KPL; ++j) {
      float mj = sm[j * DS];
      float w = mj == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mj - M);
      L = fmaf(w, sm[j * DS + 1], L);
      O = fmaf(w, sm[j * DS + 2 + d], O);
    }
    if (p.nslots == 1) {
      p.o[(long)b * p.o_bs + (long)h * p.o_hs + d] = f2bf(L > 0.f ? O / L : 0.f);
    } else {
      float* w = p.ws + (((long)b * p.H + h) * p.nslots + p.slot0 + split) * DS;
      w[2 + d] = O;
      if (d == 0) { w[0] = M; w[1] = L; }
    }
  }
}

// Shared-prefix pass: block (split, head, prompt).  All four waves walk the chunk's keys [k0, k1) of the group's FIRST row; wave w owns
// the queries w, w + 4, ... of the group (NQ per wave at most), so a key is fetched from HBM once per prompt (the other three waves
// find it in L1 / L2).  Per query the same slot arithmetic as attn_decode_partial; one partial per (row, head, split).
template <int G, int U, int NQ>
__global__ __launch_bounds__(256) void attn_decode_prefix(DecP p) {
  constexpr int KPL = 64 / G;
  __shared__ float sm[4 * NQ * KPL * (128 + 2)];
  const int split = blockIdx.x, h = blockIdx.y, gi = blockIdx.z;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int slot = lane / G, c = lane % G;
  const bool act = c * 8 < p.D;
  const int b0 = gi * p.group;
  const int n = min(p.shared_len[gi], p.Sk);
  const int k0 = split * p.chunk, k1 = min(k0 + p.chunk, n);
  const int ca = act ? c * 8 : 0;
  float qf[NQ][8], m[NQ], l[NQ], acc[NQ][8];
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    const int qi = wave + 4 * j;
    u32x4 qr = u32x4{0u, 0u, 0u, 0u};
    if (act && qi < p.group) qr = *(const u32x4*)(p.q + (long)(b0 + qi) * p.q_bs + (long)h * p.q_hs + c * 8);
    widen8(qr, qf[j]);
    m[j] = -INFINITY; l[j] = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { qf[j][i] *= p.scale_log2; acc[j][i] = 0.f; }
  }
  const float slope = p.alibi ? p.alibi[h] * 1.4426950408889634f : 0.f;
  const bf16* kb = p.k + (long)b0 * p.k_bs + (long)h * p.k_hs + ca;
  const bf16* vb = p.v + (long)b0 * p.v_bs + (long)h * p.v_hs + ca;
  for (int kk = k0; kk < k1; kk += KPL * U) {
    u32x4 kr[U], vr[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int key = kk + u * KPL + slot;
      int kc = key < k1 ? key : k0;
      kr[u] = *(const u32x4*)(kb + (long)kc * p.k_ss);
      vr[u] = *(const u32x4*)(vb + (long)kc * p.v_ss);
    }
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      float s[U];
      float mx = m[j];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float kf[8];
        widen8(kr[u], kf);
        float d = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) d = fmaf(qf[j][i], kf[i], d);
        if (!act) d = 0.f;
        d = group_sum<G>(d);
        int key = kk + u * KPL + slot;
        s[u] = key < k1 ? d + slope * (float)key : -INFINITY;
        mx = fmaxf(mx, s[u]);
      }
      if (mx > m[j]) {
        float r = m[j] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(m[j] - mx);
        l[j] *= r;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[j][i] *= r;
        m[j] = mx;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float pe = s[u] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(s[u] - m[j]);
        l[j] += pe;
        float vf[8];
        widen8(vr[u], vf);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[j][i] = fmaf(pe, vf[i], acc[j][i]);
      }
    }
  }
  // slot partials -> LDS: [(wave * NQ + j) * KPL + slot][D + 2]; merged per (wave, j) by that wave's own lanes
  const int DS = p.D + 2;
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    float* mine = sm + ((wave * NQ + j) * KPL + slot) * DS;
    if (act) {
#pragma unroll
      for (int i = 0; i < 8; ++i) mine[2 + c * 8 + i] = acc[j][i];
      if (c == 0) { mine[0] = m[j]; mine[1] = l[j]; }
    }
  }
  __syncthreads();
  for (int j = 0; j < NQ; ++j) {
    const int qi = wave + 4 * j;
    if (qi >= p.group) break;
    const float* base = sm + (wave * NQ + j) * KPL * DS;
    for (int d = lane; d < p.D; d += 64) {
      float M = -INFINITY;
      for (int t = 0; t < KPL; ++t) M = fmaxf(M, base[t * DS]);
      float L = 0.f, O = 0.f;
      for (int t = 0; t < KPL; ++t) {
        float mt = base[t * DS];
        float w = mt == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mt - M);
        L = fmaf(w, base[t * DS + 1], L);
        O = fmaf(w, base[t * DS + 2 + d], O);
      }
      float* w = p.ws + (((long)(b0 + qi) * p.H + h) * p.nslots + p.slot0 + split) * DS;
      w[2 + d] = O;
      if (d == 0) { w[0] = M; w[1] = L; }
    }
  }
}

__global__ __launch_bounds__(128) void attn_decode_merge(DecP p) {
  const int h = blockIdx.x, b = blockIdx.y, d = threadIdx.x;
  if (d >= p.D) return;
  const int DS = p.D + 2;
  const float* w = p.ws + ((long)b * p.H + h) * p.nslots * DS;
  float M = -INFINITY;
  for (int s = 0; s < p.nslots; ++s) M = fmaxf(M, w[s * DS]);
  float L = 0.f, O = 0.f;
  for (int s = 0; s < p.nslots; ++s) {
    float ms = w[s * DS];
    float e = ms == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(ms - M);
    L = fmaf(e, w[s * DS + 1], L);
    O = fmaf(e, w[s * DS + 2 + d], O);
  }
  p.o[(long)b * p.o_bs + (long)h * p.o_hs + d] = f2bf(L > 0.f ? O / L : 0.f);
}

// Keys per workgroup: FIXED, so that the partition of a row's keys -- hence every partial sum and the merge order -- depends on
// the key index only, never on the batch size, the head count or the cache capacity: a row decoded alone and the same row inside
// a batch of prompts (or after the cache grew) get the same bits (chunks beyond kv_len merge with weight 0, exactly).
#define DEC_CHUNK 128
extern "C" int unimp_attn_decode_splits(int B, int H, int Sk) {
  (void)B; (void)H;
  int s = (Sk + DEC_CHUNK - 1) / DEC_CHUNK;
  return s < 1 ? 1 : s;
}

extern "C" int unimp_attn_decode_grouped(const unimp_attn_desc* d, float* workspace, int splits, int group, const int* shared_len, void* stream) {
  if (!d || !d->q || !d->k || !d->v || !d->o) return unimp_set_error(UNIMP_ERR_ARG, "attn_decode: null pointer");
  if (d->Sq != 1) return unimp_set_error(UNIMP_ERR_SHAPE, "attn_decode: one query row per (batch row, head) (Sq == 1)");
  if (d->q_row_off || d->k_row_off) return unimp_set_error(UNIMP_ERR_UNSUPPORTED, "attn_decode: packed rows are a training-path layout");
  if (d->D % 8 || d->D > 128 || d->D < 8) return unimp_set_error(UNIMP_ERR_SHAPE, "attn_decode: head dim must be a multiple of 8, <= 128");
  if (d->mask_mode != UNIMP_MASK_NONE) return unimp_set_error(UNIMP_ERR_UNSUPPORTED, "attn_decode: kv_len masking only (the new token attends every cached key)");
  if (d->B <= 0 || d->H <= 0) return UNIMP_OK;
  if (d->Sk <= 0) return unimp_set_error(UNIMP_ERR_SHAPE, "attn_decode: Sk <= 0");
  int64_t st[] = {d->q_bs, d->q_hs, d->k_bs, d->k_ss, d->k_hs, d->v_bs, d->v_ss, d->v_hs};
  for (int64_t s : st) if (s & 7) return unimp_set_error(UNIMP_ERR_ALIGN, "attn_decode: q / k / v strides must be multiples of 8 elements");
  if (((uintptr_t)d->q | (uintptr_t)d->k | (uintptr_t)d->v) & 15) return unimp_set_error(UNIMP_ERR_ALIGN, "attn_decode: q / k / v must be 16-byte aligned");
  if (splits != unimp_attn_decode_splits(d->B, d->H, d->Sk)) return unimp_set_error(UNIMP_ERR_ARG, "attn_decode: splits must be unimp_attn_decode_splits(B, H, Sk)");
  const bool grouped = group > 1;
  if (grouped && (!shared_len || d->B % group || group > 16)) return unimp_set_error(UNIMP_ERR_ARG, "attn_decode: grouped form needs shared_len, B % group == 0 and group <= 16");
  if ((splits > 1 || grouped) && !workspace) return unimp_set_error(UNIMP_ERR_ARG, "attn_decode: needs a workspace of B*H*nslots*(D+2) floats (nslots = splits, or 2*splits grouped)");
  DecP p;
  p.q = (const bf16*)d->q; p.k = (const bf16*)d->k; p.v = (const bf16*)d->v; p.o = (bf16*)d->o;
  p.q_bs = d->q_bs; p.q_hs = d->q_hs; p.k_bs = d->k_bs; p.k_ss = d->k_ss; p.k_hs = d->k_hs;
  p.v_bs = d->v_bs; p.v_ss = d->v_ss; p.v_hs = d->v_hs; p.o_bs = d->o_bs; p.o_hs = d->o_hs;
  p.B = d->B; p.H = d->H; p.Sk = d->Sk; p.D = d->D;
  p.scale_log2 = d->scale * 1.4426950408889634f;
  p.kv_len = (const int*)d->kv_len; p.alibi = (const float*)d->alibi_slopes;
  p.ws = workspace; p.splits = splits;
  p.chunk = DEC_CHUNK;
  p.group = grouped ? group : 1; p.shared_len = grouped ? shared_len : nullptr;
  p.nslots = grouped ? 2 * splits : splits; p.slot0 = 0;
  hipStream_t s = (hipStream_t)stream;
  if (grouped) {                                   // prefix partials -> slots [0, splits), tails -> [splits, 2 splits)
    dim3 gp(splits, d->H, d->B / group);
    const int nq = (group + 3) / 4;                // queries per wave
#define PREFIX(G_, U_) do { if (nq <= 1) hipLaunchKernelGGL((attn_decode_prefix<G_, U_, 1>), gp, dim3(256), 0, s, p);          \
      else if (nq == 2) hipLaunchKernelGGL((attn_decode_prefix<G_, U_, 2>), gp, dim3(256), 0, s, p);                        \
      else if (nq == 3) hipLaunchKernelGGL((attn_decode_prefix<G_, U_, 3>), gp, dim3(256), 0, s, p);                        \
      else hipLaunchKernelGGL((attn_decode_prefix<G_, U_, 4>), gp, dim3(256), 0, s, p); } while (0)
    if (d->D <= 64) PREFIX(8, 4); else PREFIX(16, 4);
#undef PREFIX
    p.slot0 = splits;
  }
  dim3 grid(splits, d->H, d->B);
  if (d->D <= 64) hipLaunchKernelGGL((attn_decode_partial<8, 4>), grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL((attn_decode_partial<16, 8>), grid, dim3(256), 0, s, p);
  if (p.nslots > 1) hipLaunchKernelGGL(attn_decode_merge, dim3(d->H, d->B), dim3(128), 0, s, p);
  return unimp_check_launch("attn_decode");
}

extern "C" int unimp_attn_decode(const unimp_attn_desc* d, float* workspace, int splits, void* stream) {
  return unimp_attn_decode_grouped(d, workspace, splits, 1, nullptr, stream);
}
