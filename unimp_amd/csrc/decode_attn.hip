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

// ------------------------------------------------------------------------------------------- the whole attention of a decode step in ONE launch
// A cached decode step is bound by its launches (a graph node costs about 5 us, the K / V bytes of a 500-key row about 1): rotating the new q / k and
// appending k / v (decode_rope_append_kernel), the split-key partials and their merge were three nodes per layer.  Here block (split, head, row) of
// attn_decode_partial's grid does all of it: it rotates the row's new q (its lanes' 16-byte chunks; same arithmetic, rounded to bf16 like the stored
// form); the block whose chunk holds slot pos also rotates the new k, takes that key's K / V from its registers instead of the cache (the slot is
// being written by this very block) and writes them to the cache; the partial (m, l, o[D]) goes to the workspace and the LAST block of a (row, head)
// to arrive merges the partials in slot order -- same chunks, same orders: the same bits as the three launches.
// The hand-over between workgroups (which sit on different XCDs, each with its own L2) uses agent-scope accesses on exactly the words that cross:
// partials are written with sc1 stores (write-through), the arrival counter is an agent-scope atomic, the merging block reads with sc1 loads.  A
// __threadfence() per block instead writes back and invalidates the WHOLE L2 of its XCD -- measured 16.5 -> 295 us per launch
// (profiles/r06_negative_results_decode_and_mx.txt).
// (A first form gave each (row, head) ONE workgroup of 16 waves and no cross-block step at all: 11.8 us per launch at K = 1 where the three launches
// took 15.1, and 25.8 at K = 10 -- a CU keeps too few misses in flight to stream 166 KB of strided 160-byte key rows; the keys have to be spread.)
struct DecStepP {
  const bf16* qkv; long row_stride, head_stride; int q_off, k_off, v_off;
  int rows, H, D, half;                   // half = rot / 2 (0: no rotation)
  const float* cs; const float* sn;       // fp32 [rows][half]: the table row of every row's position
  bf16* kc; bf16* vc; long c_row, c_slot, c_head;
  const int64_t* pos_idx;                 // [rows] slot of the new token = keys visible - 1
  float scale_log2; const float* alibi;
  bf16* o; long o_bs, o_hs;
  float* ws; unsigned* arrived;           // [rows][H][nslots][D + 2]; [rows][H] arrival counters (zero between launches)
  int splits, chunk, nslots;              // nslots = splits; grouped: psplits prefix slots (pchunk keys each) + splits tail slots
  int group; const int* shared_len;       // beam search: rows per prompt, int32 [rows / group] shared prefix length
  int psplits, pchunk;
  int alias;                              // beam groups by ADDRESS: the keys below shared_len[g] are read from the group's first row by every row (no prefix workgroups)
};

// grouped: which slots of row b hold a partial in this launch -- the prefix chunks below its group's shared length and the tail chunks that meet
// [shared length, pos]; every workgroup of the row derives the same count from shared_len / pos_idx, so empty workgroups neither publish nor arrive
struct DecSlots { int pv, t_lo, nv; };           // prefix slots [0, pv), tail slots psplits + t_lo ..., nv in all
__device__ __forceinline__ DecSlots dec_slots(const DecStepP& p, int b) {
  DecSlots v;
  if (p.group <= 1 || p.alias) { v.pv = 0; v.t_lo = 0; v.nv = p.splits; return v; }
  const int pos = (int)p.pos_idx[b], ks = min(p.shared_len[b / p.group], pos);
  v.pv = (ks + p.pchunk - 1) / p.pchunk;
  v.t_lo = ks / p.chunk;
  v.nv = v.pv + pos / p.chunk - v.t_lo + 1;
  return v;
}
__device__ __forceinline__ int dec_slot_of(const DecStepP& p, const DecSlots& v, int i) {      // i-th valid slot -> workspace slot
  return p.group <= 1 || p.alias ? i : (i < v.pv ? i : p.psplits + v.t_lo + (i - v.pv));
}

// chunk c (8 elements) of the rotated vector at `vec` (half-split rope: pairs (i, i + half) for i < half; elements >= 2 half pass through)
__device__ __forceinline__ bf16x8 rope_chunk8(const bf16* vec, int c, int half, const float* cr, const float* sr) {
  const int P = half >> 3;
  const bool lo = c < P, rotd = c < 2 * P;
  const int cc = rotd ? (lo ? c : c - P) : 0;
  const bf16x8 own = *(const bf16x8*)(vec + c * 8);
  if (half == 0) return own;                                       // uniform
  const bf16x8 par = *(const bf16x8*)(vec + (rotd ? (lo ? c + P : c - P) : c) * 8);
  const f32x4 c0 = *(const f32x4*)(cr + cc * 8), c1 = *(const f32x4*)(cr + cc * 8 + 4), s0 = *(const f32x4*)(sr + cc * 8), s1 = *(const f32x4*)(sr + cc * 8 + 4);
  bf16x8 out;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float co = j < 4 ? c0[j & 3] : c1[j & 3], si = j < 4 ? s0[j & 3] : s1[j & 3];
    const float x1 = lo ? bf2f(own[j]) : bf2f(par[j]), x2 = lo ? bf2f(par[j]) : bf2f(own[j]);
    const bf16 a = f2bf(x1 * co - x2 * si), b = f2bf(x2 * co + x1 * si);
    out[j] = rotd ? (lo ? a : b) : own[j];
  }
  return out;
}

__device__ __forceinline__ void st_agent(float* q, float v) { __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_agent(const float* q) { return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// an agent-scope load the compiler does not serialise: the caller waits once for a batch (sc1_wait names the registers, so no use moves above the wait)
// (no memory clobber on the load itself: with one, everything the address needs from LDS is re-read in front of EVERY load -- 64 loads of a merge batch
// became 64 LDS round trips; volatile keeps the loads behind the arrival atomic and in order, the wait below carries the clobber)
__device__ __forceinline__ float ld_sc1(const float* q) { float v; asm volatile("global_load_dword %0, %1, off sc1" : "=v"(v) : "v"(q)); return v; }
__device__ __forceinline__ void sc1_wait(float (&v)[8]) {
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) :: "memory");
}

// the partials of (row b, head h) -> its output row: attn_decode_merge's arithmetic, eight slots at a time with all their loads in flight before one wait
// (relaxed atomic loads are not reordered by hipcc: one memory round trip per load -- about twenty in a row for five slots)
__device__ __forceinline__ void dec_merge_row(const DecStepP& p, int b, int h, int d) {
  const int DS = p.D + 2;
  const float* wsp = p.ws + ((long)b * p.H + h) * p.nslots * DS;
  const DecSlots sl = dec_slots(p, b);
  float MM = -INFINITY;
  for (int s0 = 0; s0 < sl.nv; s0 += 8) {
    float ms[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) ms[j] = ld_sc1(wsp + dec_slot_of(p, sl, min(s0 + j, sl.nv - 1)) * DS);
    sc1_wait(ms);
#pragma unroll
    for (int j = 0; j < 8; ++j) MM = fmaxf(MM, ms[j]);
  }
  float LL = 0.f, OO = 0.f;
  for (int s0 = 0; s0 < sl.nv; s0 += 8) {
    float ms[8], ls[8], os[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float* q = wsp + dec_slot_of(p, sl, min(s0 + j, sl.nv - 1)) * DS;
      ms[j] = ld_sc1(q); ls[j] = ld_sc1(q + 1); os[j] = ld_sc1(q + 2 + d);
    }
    sc1_wait(ms); sc1_wait(ls); sc1_wait(os);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (s0 + j < sl.nv) {
        const float e = ms[j] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(ms[j] - MM);
        LL = fmaf(e, ls[j], LL);
        OO = fmaf(e, os[j], OO);
      }
    }
  }
  p.o[(long)b * p.o_bs + (long)h * p.o_hs + d] = f2bf(LL > 0.f ? OO / LL : 0.f);
}

// this workgroup's partial of (b, h) is in the workspace (every thread waited for its stores): count it; true for the LAST of the nslots to arrive
__device__ __forceinline__ bool dec_arrive(const DecStepP& p, int b, int h) {
  unsigned* cnt = p.arrived + (long)b * p.H + h;
  const bool last = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)dec_slots(p, b).nv - 1;
  if (last) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
  return last;
}

// grouped: merge every row of the group that this workgroup completed (flags[j]), all 256 threads together, in TWO memory round trips whatever the
// number of rows: (1) the (m, l) of every valid slot of those rows -> LDS, softmax weights per slot there; (2) the o values: a thread owns up to four
// (row, dim) pairs and has all their slots' loads in flight before one wait.  (Row after row through dec_merge_row the last prefix workgroup of a head --
// it completes ALL rows of its group at once -- spent 4 x 7 us here.)  sm: >= 2 * 16 * DEC_MAXNV floats, free at this point.
#define DEC_MAXNV 64
__device__ __forceinline__ void dec_merge_group(const DecStepP& p, int b0, int h, const unsigned* flags, float* sm) {
  const int DS = p.D + 2, tid = threadIdx.x;
  float* gm = sm; float* gl = sm + 16 * DEC_MAXNV;
  // the rows' slot lists ONCE, in LDS: dec_slots() reads pos_idx / shared_len from memory, and behind the asm memory clobbers of the sc1 loads hipcc
  // reloads them at every call -- a dozen dependent round trips per merge, 40 us per launch
  __shared__ int s_pv[16], s_tlo[16], s_nv[16];
  if (tid < 16) {
    DecSlots sl; sl.pv = 0; sl.t_lo = 0; sl.nv = 0;
    if (tid < p.group && flags[tid]) sl = dec_slots(p, b0 + tid);
    s_pv[tid] = sl.pv; s_tlo[tid] = sl.t_lo; s_nv[tid] = sl.nv;       // nv = 0: the row is not ours
  }
  __syncthreads();
  {
    float mv[4], lv[4];
    int jj[4], pv[4], tl[4], nn[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { jj[u] = (tid + 256 * u) / DEC_MAXNV; pv[u] = s_pv[jj[u]]; tl[u] = s_tlo[jj[u]]; nn[u] = s_nv[jj[u]]; }      // rows beyond the group: nv = 0
#pragma unroll
    for (int u = 0; u < 4; ++u) {                    // 16 rows x 64 slots = 1024 (row, slot) pairs over 256 threads
      const int i = (tid + 256 * u) % DEC_MAXNV;
      mv[u] = -INFINITY; lv[u] = 0.f;
      if (i < nn[u]) {
        const float* q = p.ws + (((long)(b0 + jj[u]) * p.H + h) * p.nslots + (i < pv[u] ? i : p.psplits + tl[u] + (i - pv[u]))) * DS;
        mv[u] = ld_sc1(q); lv[u] = ld_sc1(q + 1);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(mv[0]), "+v"(mv[1]), "+v"(mv[2]), "+v"(mv[3]), "+v"(lv[0]), "+v"(lv[1]), "+v"(lv[2]), "+v"(lv[3]) :: "memory");
#pragma unroll
    for (int u = 0; u < 4; ++u) { gm[tid + 256 * u] = mv[u]; gl[tid + 256 * u] = lv[u]; }
  }
  __syncthreads();
  if (tid < 16 && s_nv[tid] > 0) {                   // softmax weights of the row's slots (slot order), their sum in gl[row][0]
    const int nv = s_nv[tid];
    float* m = gm + tid * DEC_MAXNV; float* l = gl + tid * DEC_MAXNV;
    float MM = -INFINITY;
    for (int i = 0; i < nv; ++i) MM = fmaxf(MM, m[i]);
    float LL = 0.f;
    for (int i = 0; i < nv; ++i) { const float e = m[i] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(m[i] - MM); m[i] = e; LL = fmaf(e, l[i], LL); }
    l[0] = LL;
  }
  __syncthreads();
  // the o values: loads without a branch around them (a load inside `if (slot < nv)` costs twenty instructions of exec-mask handling: 64 of them per
  // batch were 3 us of issue) -- slots beyond the row's count re-read its last one and are dropped by a select
  constexpr int SB = 24;
  float os[4][SB];
  for (int tb = 0; tb < p.group * p.D; tb += 1024)   // (row, dim) pairs, 1024 at a time
  for (int i0 = 0; i0 < DEC_MAXNV; i0 += SB) {       // SB slots x four (row, dim) pairs in flight
    int pv[4], tl[4], nn[4], jr[4];
    const float* bs[4];
    bool any = false;
#pragma unroll
    for (int u = 0; u < 4; ++u) {                    // everything the addresses need, out of LDS before the first load
      const int t = tb + tid + 256 * u, j = min(t / p.D, p.group - 1), d = t - (t / p.D) * p.D;      // pairs beyond the group: its last row, dropped (nn = 0)
      jr[u] = j; pv[u] = s_pv[j]; tl[u] = s_tlo[j]; nn[u] = t / p.D < p.group ? s_nv[j] : 0;
      bs[u] = p.ws + ((long)(b0 + j) * p.H + h) * p.nslots * DS + 2 + d;
      any |= i0 < nn[u];
    }
    if (!__syncthreads_or(any)) break;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int i = 0; i < SB; ++i) {
        const int ii = max(min(i0 + i, nn[u] - 1), 0);
        os[u][i] = ld_sc1(bs[u] + (ii < pv[u] ? ii : p.psplits + tl[u] + (ii - pv[u])) * DS);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(os[u][0]), "+v"(os[u][1]), "+v"(os[u][2]), "+v"(os[u][3]), "+v"(os[u][4]), "+v"(os[u][5]), "+v"(os[u][6]), "+v"(os[u][7]),
                   "+v"(os[u][8]), "+v"(os[u][9]), "+v"(os[u][10]), "+v"(os[u][11]), "+v"(os[u][12]), "+v"(os[u][13]), "+v"(os[u][14]), "+v"(os[u][15]),
                   "+v"(os[u][16]), "+v"(os[u][17]), "+v"(os[u][18]), "+v"(os[u][19]), "+v"(os[u][20]), "+v"(os[u][21]), "+v"(os[u][22]), "+v"(os[u][23]) :: "memory");
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = tb + tid + 256 * u, j = jr[u], d = t - (t / p.D) * p.D, nv = nn[u];
      if (nv == 0 || i0 >= nv) continue;
      float* accp = gl + 16 * DEC_MAXNV + t;         // running o of the pair across slot batches (third LDS plane: 16 rows x D <= 2048 floats)
      float OO = i0 ? *accp : 0.f;
#pragma unroll
      for (int i = 0; i < SB; ++i) OO = fmaf(i0 + i < nv ? gm[j * DEC_MAXNV + i0 + i] : 0.f, i0 + i < nv ? os[u][i] : 0.f, OO);
      if (i0 + SB < nv) *accp = OO;
      else { const float LL = gl[j * DEC_MAXNV]; p.o[(long)(b0 + j) * p.o_bs + (long)h * p.o_hs + d] = f2bf(LL > 0.f ? OO / LL : 0.f); }
    }
  }
}

// GROUPED (beam search): the shared prefix as workgroups (prefix chunk, head, rows + group index) of the same launch: the keys below shared_len[g] are
// read once per prompt, from the group's first row, for all its queries (wave w: queries w, w + 4, ...: NQ per wave, rotated here) in chunks of
// pchunk = 32 keys -- ONE round of loads per workgroup, as short as a tail workgroup (with attn_decode_prefix's 128-key chunks a wave walked eight
// rounds for three queries and the launch took 40 us where the ungrouped one took 22); a partial slot per (row, prefix chunk).  Its arrival counts for
// each of the group's rows; whatever rows it completes it merges, 256 / D rows at a time.
template <int G, int U, int NQ>
__device__ __forceinline__ void dec_step_prefix(const DecStepP& p, float* sm, unsigned* last_s) {
  constexpr int KPL = 64 / G;
  const int split = blockIdx.x % p.psplits, h = blockIdx.y, gi = blockIdx.x / p.psplits;      // grouped grid: x = [prefix (chunk, group) pairs | tail (split, row) pairs], y = head
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int slot = lane / G, c = lane % G;
  const bool act = c * 8 < p.D;
  const int ca = act ? c : 0;
  const int b0 = gi * p.group;
  const int n = min(p.shared_len[gi], (int)p.pos_idx[b0]);
  const int k0 = split * p.pchunk, k1 = min(k0 + p.pchunk, n);
  if (k0 >= n) return;                              // beyond the prefix: no slot, no arrival (dec_slots)
  float qf[NQ][8], m[NQ], l[NQ], acc[NQ][8];
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    const int qi = min(wave + 4 * j, p.group - 1);                   // idle query slots redo the last row, never stored
    const bf16* base = p.qkv + (long)(b0 + qi) * p.row_stride + (long)h * p.head_stride;
    const bf16x8 qn = rope_chunk8(base + p.q_off, ca, p.half, p.cs + (long)(b0 + qi) * p.half, p.sn + (long)(b0 + qi) * p.half);
    m[j] = -INFINITY; l[j] = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { qf[j][i] = bf2f(qn[i]) * p.scale_log2; acc[j][i] = 0.f; }
  }
  const float slope = p.alibi ? p.alibi[h] * 1.4426950408889634f : 0.f;
  const bf16* kb = p.kc + (long)b0 * p.c_row + (long)h * p.c_head + ca * 8;
  const bf16* vb = p.vc + (long)b0 * p.c_row + (long)h * p.c_head + ca * 8;
  for (int kk = k0; kk < k1; kk += KPL * U) {
    u32x4 kr[U], vr[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int key = kk + u * KPL + slot;
      const int kc_ = key < k1 ? key : k0;
      kr[u] = *(const u32x4*)(kb + (long)kc_ * p.c_slot);
      vr[u] = *(const u32x4*)(vb + (long)kc_ * p.c_slot);
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
        const int key = kk + u * KPL + slot;
        s[u] = key < k1 ? d + slope * (float)key : -INFINITY;
        mx = fmaxf(mx, s[u]);
      }
      if (mx > m[j]) {
        const float r = m[j] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(m[j] - mx);
        l[j] *= r;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[j][i] *= r;
        m[j] = mx;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float pe = s[u] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(s[u] - m[j]);
        l[j] += pe;
        float vf[8];
        widen8(vr[u], vf);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[j][i] = fmaf(pe, vf[i], acc[j][i]);
      }
    }
  }
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
  for (int j = 0; j < NQ; ++j) {                    // attn_decode_prefix's merge of a query's KPL slots, by the wave that owns the query
    const int qi = wave + 4 * j;
    if (qi >= p.group) break;
    const float* base = sm + (wave * NQ + j) * KPL * DS;
    float* w = p.ws + (((long)(b0 + qi) * p.H + h) * p.nslots + split) * DS;
    for (int d = lane; d < p.D; d += 64) {
      float M = -INFINITY;
      for (int t = 0; t < KPL; ++t) M = fmaxf(M, base[t * DS]);
      float L = 0.f, O = 0.f;
      for (int t = 0; t < KPL; ++t) {
        const float mt = base[t * DS];
        const float e = mt == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mt - M);
        L = fmaf(e, base[t * DS + 1], L);
        O = fmaf(e, base[t * DS + 2 + d], O);
      }
      st_agent(w + 2 + d, O);
      if (d == 0) { st_agent(w, M); st_agent(w + 1, L); }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if ((int)threadIdx.x < p.group) last_s[threadIdx.x] = dec_arrive(p, b0 + threadIdx.x, h);
  __syncthreads();
  bool any = false;
  for (int j = 0; j < p.group; ++j) any |= last_s[j] != 0;
  if (any) dec_merge_group(p, b0, h, last_s, sm);
}

template <int G, int U, int UP, int NQ>          // U / UP: keys-in-flight factor of the tail / prefix rounds (attn_decode_partial's / attn_decode_prefix's); NQ = 0: no groups
__global__ __launch_bounds__(256) void attn_decode_step_kernel(DecStepP p) {
  constexpr int KPL = 64 / G;
  constexpr int SM_F = 4 * (NQ > 1 ? NQ : 1) * KPL * (128 + 2), SM_G = 2 * 16 * DEC_MAXNV + 16 * 128;      // slot partials | the group merge's planes
  __shared__ float sm[NQ > 0 && SM_G > SM_F ? SM_G : SM_F];
  __shared__ unsigned last_s[16];
  // grouped grid: x = [prefix (chunk, group) pairs, the long ones first | tail (split, row) pairs], y = head -- no workgroup without a possible key
  // range (a (max(psplits, splits), heads, rows + groups) grid started 5984 workgroups of which 800 had keys: 40 us per launch)
  const int npre = NQ > 0 ? p.psplits * (p.rows / p.group) : 0;
  if (NQ > 0 && (int)blockIdx.x < npre) { dec_step_prefix<G, UP, NQ ? NQ : 1>(p, sm, last_s); return; }
  const int xt = NQ > 0 ? blockIdx.x - npre : 0;
  const int split = NQ > 0 ? xt % p.splits : blockIdx.x, h = blockIdx.y, b = NQ > 0 ? xt / p.splits : blockIdx.z;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int slot = lane / G, c = lane % G;
  const bool act = c * 8 < p.D;
  const int ca = act ? c : 0;                       // idle lanes of a group (D = 80: chunks 10 .. 15) redo chunk 0, never beyond the row
  const int pos = (int)p.pos_idx[b], n = pos + 1;
  const int ks = NQ > 0 ? min(p.shared_len[b / p.group], pos) : 0;        // grouped: the keys below ks belong to the prefix workgroups
  if (NQ > 0 && (split >= p.splits || (split + 1) * p.chunk <= ks || split * p.chunk > pos)) return;      // no key of [ks, pos] here: no slot, no arrival
  const int k0 = max(split * p.chunk + wave * (p.chunk >> 2), ks), k1 = min(split * p.chunk + (wave + 1) * (p.chunk >> 2), n);
  const bf16* base = p.qkv + (long)b * p.row_stride + (long)h * p.head_stride;
  const float* cr = p.cs + (long)b * p.half; const float* sr = p.sn + (long)b * p.half;
  const bf16x8 qn = rope_chunk8(base + p.q_off, ca, p.half, cr, sr);
  const long cslot = (long)b * p.c_row + (long)h * p.c_head + ca * 8;
  const bool mine = pos >= k0 && pos < k1;          // wave-uniform: this wave's keys include the new one
  u32x4 kn = u32x4{0u, 0u, 0u, 0u}, vn = kn;
  if (mine) {
    kn = __builtin_bit_cast(u32x4, rope_chunk8(base + p.k_off, ca, p.half, cr, sr));
    vn = *(const u32x4*)(base + p.v_off + ca * 8);
    if (slot == 0 && act) { *(u32x4*)(p.kc + cslot + (long)pos * p.c_slot) = kn; *(u32x4*)(p.vc + cslot + (long)pos * p.c_slot) = vn; }
  }
  float qf[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) qf[i] = bf2f(qn[i]) * p.scale_log2;
  const float slope = p.alibi ? p.alibi[h] * 1.4426950408889634f : 0.f;
  const bf16* kb = p.kc + cslot; const bf16* vb = p.vc + cslot;
  // beam groups BY ADDRESS (p.alias): the K rows of a prompt hold identical K / V below shared_len[g], so every row reads those keys from the group's
  // FIRST row -- the ten copies of a 469-token prompt (48 of the 53 MB a layer's decode attention reads at K = 10) become one set of lines that L2 and
  // the memory-side cache serve ten times.  Same keys, same values, same partition: the bits of the ungrouped form.  No extra workgroup, slot or merge.
  const int ka = (NQ == 0 && p.alias) ? min(p.shared_len[b / p.group], pos) : 0;
  const long to_first = (NQ == 0 && p.alias) ? -(long)(b % p.group) * p.c_row : 0;
  // (An XCD-aware workgroup order -- the rows of a (split, head) pair 8 apart in linear id, i.e. on ONE XCD whose L2 then fetches their shared lines
  // once -- measured 3.434 ms per K = 10 token-step against 3.413 with the plain (split, head, row) grid, alternating three times: the memory-side cache
  // already serves the repeats.  Removed.)

  float m = -INFINITY, l = 0.f, acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  for (int kk = k0; kk < k1; kk += KPL * U) {
    u32x4 kr[U], vr[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {                   // all loads of the round go out before the first use
      const int key = kk + u * KPL + slot;
      const int kc_ = key < k1 && key != pos ? key : k0 < pos ? k0 : 0;      // clamped, masked below; never the slot being written
      const long off = (long)kc_ * p.c_slot + (kc_ < ka ? to_first : 0);
      kr[u] = *(const u32x4*)(kb + off);
      vr[u] = *(const u32x4*)(vb + off);
    }
    if (mine) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool isnew = kk + u * KPL + slot == pos;
        kr[u] = isnew ? kn : kr[u];
        vr[u] = isnew ? vn : vr[u];
      }
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
      const int key = kk + u * KPL + slot;
      s[u] = key < k1 ? d + slope * (float)key : -INFINITY;
      mx = fmaxf(mx, s[u]);
    }
    if (mx > m) {
      const float r = m == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(m - mx);
      l *= r;
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] *= r;
      m = mx;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float pe = s[u] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(s[u] - m);
      l += pe;
      float vf[8];
      widen8(vr[u], vf);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = fmaf(pe, vf[i], acc[i]);
    }
  }
  // slot partials -> LDS: [wave * KPL + slot][D + 2]; merged in a fixed order, one thread per dim (attn_decode_partial's arithmetic)
  const int DS = p.D + 2;
  float* mn = sm + (wave * KPL + slot) * DS;
  if (act) {
#pragma unroll
    for (int i = 0; i < 8; ++i) mn[2 + c * 8 + i] = acc[i];
    if (c == 0) { mn[0] = m; mn[1] = l; }
  }
  __syncthreads();
  const int d = threadIdx.x;
  float M = -INFINITY, L = 0.f, O = 0.f;
  if (d < p.D) {
    for (int j = 0; j < 4 * KPL; ++j) M = fmaxf(M, sm[j * DS]);
    for (int j = 0; j < 4 * KPL; ++j) {
      const float mj = sm[j * DS];
      const float w = mj == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mj - M);
      L = fmaf(w, sm[j * DS + 1], L);
      O = fmaf(w, sm[j * DS + 2 + d], O);
    }
  }
  if (p.nslots == 1) {
    if (d < p.D) p.o[(long)b * p.o_bs + (long)h * p.o_hs + d] = f2bf(L > 0.f ? O / L : 0.f);
    return;
  }
  float* wsp = p.ws + (((long)b * p.H + h) * p.nslots + (NQ > 0 ? p.psplits : 0) + split) * DS;     // grouped: tails behind the prefix slots
  if (d < p.D) {
    st_agent(wsp + 2 + d, O);
    if (d == 0) { st_agent(wsp, M); st_agent(wsp + 1, L); }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this thread's partial has reached the coherence point
  __syncthreads();
  if (NQ > 0) {                                     // grouped: the two-round-trip merge (a row has 15 - 20 slots)
    if ((int)threadIdx.x < 16) last_s[threadIdx.x] = 0;
    __syncthreads();
    if (threadIdx.x == 0) last_s[b % p.group] = dec_arrive(p, b, h);
    __syncthreads();
    if (last_s[b % p.group]) dec_merge_group(p, b - b % p.group, h, last_s, sm);
    return;
  }
  if (threadIdx.x == 0) last_s[0] = dec_arrive(p, b, h);
  __syncthreads();
  if (last_s[0] && d < p.D) dec_merge_row(p, b, h, d);
}

#define DEC_PCHUNK 32          // prefix keys per workgroup of the grouped one-launch form: one round of loads (64 / G * U keys: 32 for both head-dim classes)
extern "C" int unimp_attn_decode_step_slots(int rows, int heads, int capacity, int group) {       // partial slots per (row, head): sizes the workspace (either group mode)
  const int ps = (capacity + DEC_PCHUNK - 1) / DEC_PCHUNK, sp = unimp_attn_decode_splits(rows, heads, capacity);
  return sp + (group > 1 && ps + sp <= DEC_MAXNV ? ps : 0);
}
extern "C" int unimp_attn_decode_step(const unimp_decode_step_desc* d, void* stream) {
  if (!d || !d->qkv || !d->kcache || !d->vcache || !d->pos_idx || !d->out) return unimp_set_error(UNIMP_ERR_ARG, "attn_decode_step: null pointer");
  if (d->rot > 0 && (!d->cos_rows || !d->sin_rows)) return unimp_set_error(UNIMP_ERR_ARG, "attn_decode_step: rot > 0 needs the cos / sin rows");
  if (d->rows <= 0 || d->heads <= 0) return UNIMP_OK;
  const int half = d->rot / 2;
  if (d->hd % 8 || d->hd > 128 || d->hd < 8 || d->rot < 0 || d->rot > d->hd || (half & 7))
    return unimp_set_error(UNIMP_ERR_SHAPE, "attn_decode_step: head dim a multiple of 8, <= 128; rot / 2 a multiple of 8, rot <= head dim");
  if (((d->row_stride | d->head_stride | d->c_row_stride | d->c_slot_stride | d->c_head_stride | d->o_row_stride | d->o_head_stride) & 7) ||
      ((d->q_off | d->k_off | d->v_off) & 7) || (((uintptr_t)d->qkv | (uintptr_t)d->kcache | (uintptr_t)d->vcache) & 15) ||
      (d->rot > 0 && (((uintptr_t)d->cos_rows | (uintptr_t)d->sin_rows) & 15)))
    return unimp_set_error(UNIMP_ERR_ALIGN, "attn_decode_step: strides and offsets must be multiples of 8 elements, pointers 16-byte aligned");
  if (d->capacity <= 0) return unimp_set_error(UNIMP_ERR_SHAPE, "attn_decode_step: capacity <= 0");
  const int splits = unimp_attn_decode_splits(d->rows, d->heads, d->capacity);
  // beam groups: by address (default; group_mode 0) or by prefix workgroups (group_mode 1: UNIMP_DECODE_STEP_GROUPED=2 in the package; needs the merge to hold the slots)
  const bool alias = d->group > 1 && d->group_mode == 0;
  const bool grouped = d->group > 1 && !alias && (d->capacity + DEC_PCHUNK - 1) / DEC_PCHUNK + splits <= DEC_MAXNV;
  if (d->group > 1 && (!d->shared_len || d->rows % d->group || d->group > 16)) return unimp_set_error(UNIMP_ERR_ARG, "attn_decode_step: grouped form needs shared_len, rows % group == 0 and group <= 16");
  const int psplits = grouped ? (d->capacity + DEC_PCHUNK - 1) / DEC_PCHUNK : 0;
  const int nslots = psplits + splits;
  if (nslots > 1 && (!d->workspace || !d->arrived)) return unimp_set_error(UNIMP_ERR_ARG, "attn_decode_step: needs the workspace (unimp_attn_decode_step_slots(...) * rows*heads*(hd+2) floats) and the zeroed arrival counters (rows*heads)");
  DecStepP p;
  p.qkv = (const bf16*)d->qkv; p.row_stride = d->row_stride; p.head_stride = d->head_stride; p.q_off = d->q_off; p.k_off = d->k_off; p.v_off = d->v_off;
  p.rows = d->rows; p.H = d->heads; p.D = d->hd; p.half = half; p.cs = d->cos_rows; p.sn = d->sin_rows;
  p.kc = (bf16*)d->kcache; p.vc = (bf16*)d->vcache; p.c_row = d->c_row_stride; p.c_slot = d->c_slot_stride; p.c_head = d->c_head_stride;
  p.pos_idx = d->pos_idx; p.scale_log2 = d->scale * 1.4426950408889634f; p.alibi = d->alibi_slopes;
  p.o = (bf16*)d->out; p.o_bs = d->o_row_stride; p.o_hs = d->o_head_stride;
  p.ws = d->workspace; p.arrived = (unsigned*)d->arrived; p.splits = splits; p.chunk = DEC_CHUNK; p.nslots = nslots;
  p.group = grouped || alias ? d->group : 1; p.shared_len = grouped || alias ? d->shared_len : nullptr; p.psplits = psplits; p.pchunk = DEC_PCHUNK;
  p.alias = alias ? 1 : 0;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(splits, d->heads, d->rows);
  if (grouped) grid = dim3(psplits * (d->rows / d->group) + splits * d->rows, d->heads, 1);          // prefix (chunk, group) pairs, then tail (split, row) pairs
  const int nq = grouped ? (d->group + 3) / 4 : 0;                                    // queries per wave of a prefix workgroup
#define STEP(G_, U_) do { switch (nq) {                                                                                  \
    case 0: hipLaunchKernelGGL((attn_decode_step_kernel<G_, U_, U_, 0>), grid, dim3(256), 0, s, p); break;                 \
    case 1: hipLaunchKernelGGL((attn_decode_step_kernel<G_, U_, U_, 1>), grid, dim3(256), 0, s, p); break;                 \
    case 2: hipLaunchKernelGGL((attn_decode_step_kernel<G_, U_, U_, 2>), grid, dim3(256), 0, s, p); break;                 \
    case 3: hipLaunchKernelGGL((attn_decode_step_kernel<G_, U_, U_, 3>), grid, dim3(256), 0, s, p); break;                 \
    default: hipLaunchKernelGGL((attn_decode_step_kernel<G_, U_, U_, 4>), grid, dim3(256), 0, s, p); break; } } while (0)
  if (d->hd <= 64) STEP(8, 4); else STEP(16, 8);
#undef STEP
  return unimp_check_launch("attn_decode_step");
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
