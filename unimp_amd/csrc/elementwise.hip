// HBM-bound helpers: RoPE, embedding gather / scatter-add, ViT patchify + token assembly, adds, casts,
// SwiGLU, dot, periodic row broadcast / reduction.  All 16-byte vectorised where the layout allows.
#include <algorithm>
#include "common.h"
#include "unimp_hip.h"

#define GRID1D(n, per) dim3((unsigned)std::min<long>(((n) + (per) - 1) / (per), 65535L * 16))

// ------------------------------------------------------------------------------------------- RoPE (half-split)
// One block per group of rows; the (head, vector, chunk) decomposition of a thread's items is the same in every row, so it
// is done ONCE per thread (32-bit arithmetic) and the row loop only adds the row base and looks the position up -- the
// first version decomposed a flat 64-bit index per item (four 64-bit divisions: 4.2 TB/s; this one streams).
template <int CH>
__global__ __launch_bounds__(256) void rope_kernel(bf16* __restrict__ x, long row_stride, long head_stride, int rows, int L, int heads,
                                                   int half, int nvec, int off0, int off1, const float* __restrict__ cs,
                                                   const float* __restrict__ sn, int inverse, const int* __restrict__ pos_tab) {
  constexpr int MAXI = 4;                                   // items per thread and row (a row has heads * nvec * half / CH items)
  const int cpr = half / CH, per_head = nvec * cpr, items = heads * per_head;
  int eoff[MAXI], toff[MAXI];
#pragma unroll
  for (int k = 0; k < MAXI; ++k) {
    int t = threadIdx.x + 256 * k;
    int h = t / per_head, rem = t - h * per_head;
    int vsel = rem / cpr, c = rem - vsel * cpr;
    eoff[k] = t < items ? (int)(h * head_stride + (vsel ? off1 : off0) + c * CH) : -1;
    toff[k] = c * CH;
  }
  const float sgn = inverse ? -1.f : 1.f;
  for (int r = blockIdx.x; r < rows; r += gridDim.x) {
    int pos = pos_tab ? pos_tab[r] : r % L;
    bf16* xr = x + (long)r * row_stride;
    const float* cr = cs + (long)pos * half;
    const float* sr = sn + (long)pos * half;
    for (int k0 = 0; k0 * 256 < items; k0 += MAXI) {       // rows with more than MAXI * 256 items: further passes (not the LM shapes)
#pragma unroll
      for (int k = 0; k < MAXI; ++k) {
        int eo = eoff[k], to = toff[k];
        if (k0 > 0) {                                       // recompute for the later passes
          int t = threadIdx.x + 256 * (k0 + k);
          int h = t / per_head, rem = t - h * per_head;
          int vsel = rem / cpr, c = rem - vsel * cpr;
          eo = t < items ? (int)(h * head_stride + (vsel ? off1 : off0) + c * CH) : -1;
          to = c * CH;
        }
        if (eo < 0) continue;
        bf16* p = xr + eo;
        if (CH == 8) {
          bf16x8 a = *(bf16x8*)p, b = *(bf16x8*)(p + half), oa, ob;
          f32x4 c0 = *(const f32x4*)(cr + to), c1 = *(const f32x4*)(cr + to + 4);
          f32x4 s0 = *(const f32x4*)(sr + to), s1 = *(const f32x4*)(sr + to + 4);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float co = j < 4 ? c0[j & 3] : c1[j & 3], si = sgn * (j < 4 ? s0[j & 3] : s1[j & 3]);
            float x1 = bf2f(a[j]), x2 = bf2f(b[j]);
            oa[j] = f2bf(x1 * co - x2 * si);
            ob[j] = f2bf(x2 * co + x1 * si);
          }
          *(bf16x8*)p = oa; *(bf16x8*)(p + half) = ob;
        } else {
          float co = cr[to], si = sgn * sr[to];
          float x1 = bf2f(p[0]), x2 = bf2f(p[half]);
          p[0] = f2bf(x1 * co - x2 * si);
          p[half] = f2bf(x2 * co + x1 * si);
        }
      }
    }
  }
}

// Streaming form for the vectorisable case: a block takes G rows at a time and spreads their G * items 16-byte items evenly over its
// 256 threads (NS slots per thread, the last one partly empty), and all 6 * NS loads of a thread - both halves of the pair, cos
// and sin - are issued before the first one is waited for (a lane without an item reads item 0 of the first row instead of
// branching: with a branch per item the compiler waited for every item's loads separately, 4.4 TB/s).
template <int NS>
__global__ __launch_bounds__(256) void rope_rows_kernel(bf16* __restrict__ x, long row_stride, long head_stride, int rows, int L, int heads,
                                                        int half, int nvec, int off0, int off1, const float* __restrict__ cs,
                                                        const float* __restrict__ sn, int inverse, int G, const int* __restrict__ pos_tab) {
  const int cpr = half >> 3, per_head = nvec * cpr, items = heads * per_head;
  int eoff[NS], toff[NS], rg[NS];
  bool has[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    int t = threadIdx.x + 256 * k;
    has[k] = t < G * items;
    int tt = has[k] ? t : 0;
    rg[k] = tt / items;
    int it = tt - rg[k] * items;
    int h = it / per_head, rem = it - h * per_head;
    int vsel = rem / cpr, c = rem - vsel * cpr;
    eoff[k] = (int)(h * head_stride + (vsel ? off1 : off0) + c * 8);
    toff[k] = c * 8;
  }
  const float sgn = inverse ? -1.f : 1.f;
  for (int r0 = blockIdx.x * G; r0 < rows; r0 += gridDim.x * G) {
    bf16x8 a[NS], b[NS];
    f32x4 c0[NS], c1[NS], s0[NS], s1[NS];
    bf16* p[NS];
    bool ok[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      int r = r0 + rg[k];
      ok[k] = has[k] && r < rows;
      if (!ok[k]) r = r0;
      int pos = pos_tab ? pos_tab[r] : r % L;
      p[k] = x + (long)r * row_stride + eoff[k];
      const float* cr = cs + (long)pos * half + toff[k];
      const float* sr = sn + (long)pos * half + toff[k];
      a[k] = *(const bf16x8*)p[k]; b[k] = *(const bf16x8*)(p[k] + half);
      c0[k] = *(const f32x4*)cr; c1[k] = *(const f32x4*)(cr + 4);
      s0[k] = *(const f32x4*)sr; s1[k] = *(const f32x4*)(sr + 4);
    }
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      bf16x8 oa, ob;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float co = j < 4 ? c0[k][j & 3] : c1[k][j & 3], si = sgn * (j < 4 ? s0[k][j & 3] : s1[k][j & 3]);
        float x1 = bf2f(a[k][j]), x2 = bf2f(b[k][j]);
        oa[j] = f2bf(x1 * co - x2 * si);
        ob[j] = f2bf(x2 * co + x1 * si);
      }
      if (ok[k]) { *(bf16x8*)p[k] = oa; *(bf16x8*)(p[k] + half) = ob; }
    }
  }
}

static int rope_launch(void* x, int64_t row_stride, int64_t head_stride, int rows, int L, int heads, int rot,
                       int nvec, int vec_off0, int vec_off1, const float* cos_t, const float* sin_t, int inverse, const int* pos_tab,
                       void* stream) {
  if (!x || !cos_t || !sin_t) return unimp_set_error(UNIMP_ERR_ARG, "rope: null pointer");
  if (rot <= 0 || (rot & 1) || nvec < 1 || nvec > 2) return unimp_set_error(UNIMP_ERR_SHAPE, "rope: bad rot/nvec");
  if (rows <= 0) return UNIMP_OK;
  int half = rot / 2;
  bool vec = (half % 8 == 0) && (row_stride % 8 == 0) && (head_stride % 8 == 0) && (vec_off0 % 8 == 0) && (vec_off1 % 8 == 0) &&
             (((uintptr_t)x & 15) == 0);
  hipStream_t s = (hipStream_t)stream;
  // cos / sin rows are read as float4: the table's row length (half) is a multiple of 8 in the vector form, so every chunk is
  // 16-byte aligned when the table itself is (torch allocations are)
  if ((long)heads * head_stride + rot > (1L << 30)) return unimp_set_error(UNIMP_ERR_SHAPE, "rope: row too long");
  int grid = rows < 16384 ? rows : 16384;
  const bool vec16 = vec && (((uintptr_t)cos_t | (uintptr_t)sin_t) & 15) == 0;
  const int items = vec16 ? heads * nvec * (half / 8) : 0;
  if (vec16 && items <= 6 * 256) {
    // rows per block pass: the smallest G in {1, 2, 4, 8} that fills the threads' slots best (G * items close below a multiple of 256)
    int G = 1, best_waste = 1 << 30;
    for (int g = 1; g <= 8; g *= 2) {
      int ns = (g * items + 255) / 256;
      if (ns > 6 || g > rows) break;
      int waste = (ns * 256 - g * items) * (8 / g);           // idle slots per 8 rows
      if (waste < best_waste) { best_waste = waste; G = g; }
    }
    int ns = (G * items + 255) / 256;
    int g2 = (rows + G - 1) / G; if (g2 > 16384) g2 = 16384;
#define ROPE_NS(N) case N: hipLaunchKernelGGL((rope_rows_kernel<N>), dim3(g2), dim3(256), 0, s, (bf16*)x, (long)row_stride, (long)head_stride, rows, L, \
                       heads, half, nvec, vec_off0, vec_off1, cos_t, sin_t, inverse, G, pos_tab); break
    switch (ns) { ROPE_NS(1); ROPE_NS(2); ROPE_NS(3); ROPE_NS(4); ROPE_NS(5); ROPE_NS(6); default: break; }
#undef ROPE_NS
    return unimp_check_launch("rope");
  }
  if (vec16)
    hipLaunchKernelGGL((rope_kernel<8>), dim3(grid), dim3(256), 0, s, (bf16*)x, (long)row_stride, (long)head_stride, rows, L,
                       heads, half, nvec, vec_off0, vec_off1, cos_t, sin_t, inverse, pos_tab);
  else
    hipLaunchKernelGGL((rope_kernel<1>), dim3(grid), dim3(256), 0, s, (bf16*)x, (long)row_stride, (long)head_stride, rows, L,
                       heads, half, nvec, vec_off0, vec_off1, cos_t, sin_t, inverse, pos_tab);
  return unimp_check_launch("rope");
}

extern "C" int unimp_rope_halfsplit(void* x, int64_t row_stride, int64_t head_stride, int rows, int L, int heads, int rot,
                                    int nvec, int vec_off0, int vec_off1, const float* cos_t, const float* sin_t, int inverse,
                                    void* stream) {
  return rope_launch(x, row_stride, head_stride, rows, L, heads, rot, nvec, vec_off0, vec_off1, cos_t, sin_t, inverse, nullptr, stream);
}
extern "C" int unimp_rope_halfsplit_pos(void* x, int64_t row_stride, int64_t head_stride, int rows, const int32_t* pos, int heads, int rot,
                                        int nvec, int vec_off0, int vec_off1, const float* cos_t, const float* sin_t, int inverse,
                                        void* stream) {
  if (!pos) return unimp_set_error(UNIMP_ERR_ARG, "rope: null position table");
  return rope_launch(x, row_stride, head_stride, rows, 1, heads, rot, nvec, vec_off0, vec_off1, cos_t, sin_t, inverse, pos, stream);
}

// ------------------------------------------------------------------------------------------- decode step: rotate q / k, append k / v
// One launch per layer and decode step instead of three (rope_rows_kernel + two index_put_ of torch): item = (row, head, 8-element chunk).
// Chunks [0, P) rotate the q pair (c, c + P) in place; [P, 2P) rotate the k pair and write it to qkv AND to the cache slot pos_idx[row];
// [2P, 2P + T) copy the un-rotated tail chunks of k to the cache; the last hd / 8 chunks copy v.  P = rot / 16, T = (hd - rot) / 8.
// Same arithmetic per element as rope_rows_kernel (x1 * cos - x2 * sin, x2 * cos + x1 * sin in fp32, rounded to bf16): same bits.
__global__ __launch_bounds__(256) void decode_rope_append_kernel(bf16* __restrict__ qkv, long row_stride, long head_stride, int rows, int heads, int hd,
                                                                 int q_off, int k_off, int v_off, int half, const float* __restrict__ cs,
                                                                 const float* __restrict__ sn, bf16* __restrict__ kc, bf16* __restrict__ vc,
                                                                 long c_row, long c_slot, long c_head, const int64_t* __restrict__ pos_idx) {
  const int P = half >> 3, T = (hd - 2 * half) >> 3, V = hd >> 3, per = 2 * P + T + V;
  const long total = (long)rows * heads * per;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % per); long rh = i / per;
    int h = (int)(rh % heads); int r = (int)(rh / heads);
    bf16* base = qkv + (long)r * row_stride + (long)h * head_stride;
    const long slot = (long)r * c_row + (long)pos_idx[r] * c_slot + (long)h * c_head;
    if (c < 2 * P) {
      const bool isk = c >= P;
      const int cc = isk ? c - P : c;
      bf16* p = base + (isk ? k_off : q_off) + cc * 8;
      bf16x8 a = *(const bf16x8*)p, b = *(const bf16x8*)(p + half);
      const float* cr = cs + (long)r * half + cc * 8;
      const float* sr = sn + (long)r * half + cc * 8;
      f32x4 c0 = *(const f32x4*)cr, c1 = *(const f32x4*)(cr + 4), s0 = *(const f32x4*)sr, s1 = *(const f32x4*)(sr + 4);
      bf16x8 oa, ob;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float co = j < 4 ? c0[j & 3] : c1[j & 3], si = j < 4 ? s0[j & 3] : s1[j & 3];
        float x1 = bf2f(a[j]), x2 = bf2f(b[j]);
        oa[j] = f2bf(x1 * co - x2 * si);
        ob[j] = f2bf(x2 * co + x1 * si);
      }
      *(bf16x8*)p = oa; *(bf16x8*)(p + half) = ob;
      if (isk) { *(bf16x8*)(kc + slot + cc * 8) = oa; *(bf16x8*)(kc + slot + half + cc * 8) = ob; }
    } else if (c < 2 * P + T) {
      const int cc = c - 2 * P;
      *(bf16x8*)(kc + slot + 2 * half + cc * 8) = *(const bf16x8*)(base + k_off + 2 * half + cc * 8);
    } else {
      const int cc = c - 2 * P - T;
      *(bf16x8*)(vc + slot + cc * 8) = *(const bf16x8*)(base + v_off + cc * 8);
    }
  }
}

extern "C" int unimp_decode_rope_append(void* qkv, int64_t row_stride, int64_t head_stride, int rows, int heads, int hd, int q_off, int k_off,
                                        int v_off, int rot, const float* cos_rows, const float* sin_rows, void* kcache, void* vcache,
                                        int64_t c_row_stride, int64_t c_slot_stride, int64_t c_head_stride, const int64_t* pos_idx, void* stream) {
  if (!qkv || !kcache || !vcache || !pos_idx || (rot > 0 && (!cos_rows || !sin_rows))) return unimp_set_error(UNIMP_ERR_ARG, "decode_rope_append: null pointer");
  if (rows <= 0 || heads <= 0) return UNIMP_OK;
  const int half = rot / 2;
  if (rot < 0 || rot > hd || (hd & 7) || (half & 7) || ((row_stride | head_stride | c_row_stride | c_slot_stride | c_head_stride) & 7) ||
      ((q_off | k_off | v_off) & 7) || (((uintptr_t)qkv | (uintptr_t)kcache | (uintptr_t)vcache) & 15) ||
      (rot > 0 && (((uintptr_t)cos_rows | (uintptr_t)sin_rows) & 15)))
    return unimp_set_error(UNIMP_ERR_ALIGN, "decode_rope_append: hd, rot / 2, strides and offsets must be multiples of 8 elements, pointers 16-byte aligned");
  long total = (long)rows * heads * (2 * (half >> 3) + ((hd - rot) >> 3) + (hd >> 3));
  hipLaunchKernelGGL(decode_rope_append_kernel, GRID1D(total, 256), dim3(256), 0, (hipStream_t)stream, (bf16*)qkv, (long)row_stride, (long)head_stride,
                     rows, heads, hd, q_off, k_off, v_off, half, cos_rows, sin_rows, (bf16*)kcache, (bf16*)vcache, (long)c_row_stride,
                     (long)c_slot_stride, (long)c_head_stride, pos_idx);
  return unimp_check_launch("decode_rope_append");
}

// ------------------------------------------------------------------------------------------- beam search: the cache rows follow their hypotheses
// transformers' beam search reorders past_key_values with index_select after every step (GenerationMixin._reorder_cache; eval_rec.py:100-110 runs
// K = 10 beams).  The prompt's K / V are shared by a prompt's beams, so only the GENERATED tail moves: block (slot, plane = layer x {k, v}, group) permutes
// the K rows of one cache slot in place -- every thread reads its 16-byte chunk of all K source rows, then writes them -- and slots the decode has not
// reached yet (>= pos_idx of the group's first row) return at once.  One launch and one pass over the bytes where tail.copy_(tail.index_select(...))
// was two launches and two passes over the WHOLE reserved tail (237 us of a 4.6 ms step at K = 10).
template <int KMAX>
__global__ __launch_bounds__(256) void kv_reorder_beams_kernel(bf16* __restrict__ kv, long s_plane, long s_row, long s_slot, int chunks, int K,
                                                               const int64_t* __restrict__ src_local, const int32_t* __restrict__ slot0,
                                                               const int64_t* __restrict__ pos_idx) {
  const int grp = blockIdx.z, plane = blockIdx.y;
  const int slot = slot0[grp] + blockIdx.x;
  if (slot >= (int)pos_idx[(long)grp * K]) return;                     // nothing generated there yet
  bf16* base = kv + (long)plane * s_plane + (long)grp * K * s_row + (long)slot * s_slot;
  for (int c = threadIdx.x; c < chunks; c += blockDim.x) {
    u32x4 v[KMAX];
#pragma unroll
    for (int j = 0; j < KMAX; ++j) if (j < K) v[j] = *(const u32x4*)(base + src_local[(long)grp * K + j] * s_row + c * 8);
#pragma unroll
    for (int j = 0; j < KMAX; ++j) if (j < K) *(u32x4*)(base + (long)j * s_row + c * 8) = v[j];
  }
}

extern "C" int unimp_kv_reorder_beams(void* kv, int64_t s_plane, int n_planes, int64_t s_row, int64_t s_slot, int row_elems, int K, int n_groups,
                                      const int64_t* src_local, const int32_t* slot0, const int64_t* pos_idx, int max_new, void* stream) {
  if (!kv || !src_local || !slot0 || !pos_idx) return unimp_set_error(UNIMP_ERR_ARG, "kv_reorder_beams: null pointer");
  if (n_planes <= 0 || n_groups <= 0 || max_new <= 0 || K <= 1) return UNIMP_OK;
  if (K > 16 || (row_elems & 7) || ((s_plane | s_row | s_slot) & 7) || ((uintptr_t)kv & 15))
    return unimp_set_error(UNIMP_ERR_SHAPE, "kv_reorder_beams: at most 16 beams, row length and strides multiples of 8 elements, 16-byte aligned cache");
  dim3 grid(max_new, n_planes, n_groups);
  hipStream_t s = (hipStream_t)stream;
  if (K <= 4) hipLaunchKernelGGL((kv_reorder_beams_kernel<4>), grid, dim3(256), 0, s, (bf16*)kv, (long)s_plane, (long)s_row, (long)s_slot, row_elems >> 3, K, src_local, slot0, pos_idx);
  else if (K <= 8) hipLaunchKernelGGL((kv_reorder_beams_kernel<8>), grid, dim3(256), 0, s, (bf16*)kv, (long)s_plane, (long)s_row, (long)s_slot, row_elems >> 3, K, src_local, slot0, pos_idx);
  else hipLaunchKernelGGL((kv_reorder_beams_kernel<16>), grid, dim3(256), 0, s, (bf16*)kv, (long)s_plane, (long)s_row, (long)s_slot, row_elems >> 3, K, src_local, slot0, pos_idx);
  return unimp_check_launch("kv_reorder_beams");
}

// ------------------------------------------------------------------------------------------- decode: pull the NEXT weights into the memory-side cache
// A decode step is a chain of weight-streaming GEMMs, each a graph node: between two of them HBM idles for the launch, the first-byte latency and the
// tail (about 4 of 12 us at 52 MB).  This kernel, launched on a SECOND stream beside GEMM i, reads the weights of GEMM i + 1 and throws them away: the
// 256 MB Infinity Cache (memory side: every read allocates) then holds them when GEMM i + 1 starts.  `blocks` small workgroups; sixteen 16-byte loads per
// lane in flight.
__global__ __launch_bounds__(256) void prefetch_kernel(const u32x4* __restrict__ p, long n16, unsigned* sink) {
  const long stride = (long)gridDim.x * 256;
  u32x4 acc = u32x4{0u, 0u, 0u, 0u};
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride * 8) {
    u32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const long j = i + u * stride; v[u] = p[j < n16 ? j : i]; }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc ^= v[u];
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x9e3779b9u && sink) *sink = 1u;      // keeps the loads alive; practically never true
}
extern "C" int unimp_prefetch(const void* p, int64_t bytes, int blocks, void* sink, void* stream) {
  if (!p || bytes < 16) return UNIMP_OK;
  if ((uintptr_t)p & 15) return unimp_set_error(UNIMP_ERR_ALIGN, "prefetch: 16-byte aligned pointer");
  if (blocks <= 0) blocks = 256;
  hipLaunchKernelGGL(prefetch_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const u32x4*)p, (long)(bytes >> 4), (unsigned*)sink);
  return unimp_check_launch("prefetch");
}

// ------------------------------------------------------------------------------------------- beam search: log_softmax + beam scores + top-2K in two launches
// transformers' BeamSearchScorer step (GenerationMixin.beam_search; eval_rec.py:100-110, K = 10): next_token_scores = log_softmax(logits) + beam_scores,
// viewed as [prompts, K * V], torch.topk(2 K).  As torch ops that is a float copy, log_softmax, an add and a sort-based top-k over 740 530 elements:
// 233 us per token-step where the decode step itself takes 3.5 ms.  Here: pass A, block (slice, row) reads its slice of the row's logits ONCE -- row
// maximum and sum of exponentials of the slice (online), and the slice's C largest logits (a row's order does not depend on its normaliser) by C
// rounds of a block-wide argmax; pass B, one block per prompt: the rows' normalisers from the slice partials (slice order), the candidates' scores
// ((x - max) - log sum) + beam score as torch computes them, and the C best of the K * slices * C candidates, sorted, ties to the smaller flat index.
#define BTK_SLICES 16
#define BTK_MAXC 32
__device__ __forceinline__ void btk_argmax(float& v, int& i) {               // wave-wide (value, index) maximum, ties to the smaller index; result in every lane
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float v2 = __shfl_xor(v, o, 64); const int i2 = __shfl_xor(i, o, 64);
    if (v2 > v || (v2 == v && i2 < i)) { v = v2; i = i2; }
  }
}
template <typename T, int PER>
__global__ __launch_bounds__(256) void beam_topk_a(const T* __restrict__ logits, long ld, int V, int C, float* __restrict__ part_ms, float* __restrict__ cand_v,
                                                   int* __restrict__ cand_i) {
  __shared__ float sv[4]; __shared__ int si[4]; __shared__ float red[8];
  const int slice = blockIdx.x, row = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int per = (V + BTK_SLICES - 1) / BTK_SLICES, s0 = slice * per, s1 = min(s0 + per, V);
  const T* x = logits + (long)row * ld;
  float v[PER];
  float m = -INFINITY;
#pragma unroll
  for (int k = 0; k < PER; ++k) { const int i = s0 + tid + 256 * k; v[k] = i < s1 ? (float)x[i] : -INFINITY; m = fmaxf(m, v[k]); }
  m = wave_max(m);
  if (lane == 0) red[w] = m;
  __syncthreads();
  const float M = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < PER; ++k) sum += v[k] == -INFINITY ? 0.f : __expf(v[k] - M);
  sum = wave_sum(sum);
  if (lane == 0) red[4 + w] = sum;
  __syncthreads();
  if (tid == 0) { float* o = part_ms + ((long)row * BTK_SLICES + slice) * 2; o[0] = M; o[1] = (red[4] + red[5]) + (red[6] + red[7]); }
  for (int c = 0; c < C; ++c) {
    float bv = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
    for (int k = 0; k < PER; ++k) { const int i = s0 + tid + 256 * k; if (v[k] > bv) { bv = v[k]; bi = i; } }      // ascending i inside a thread: the first maximum wins
    btk_argmax(bv, bi);
    if (lane == 0) { sv[w] = bv; si[w] = bi; }
    __syncthreads();
    float gv = sv[0]; int gi = si[0];
#pragma unroll
    for (int q = 1; q < 4; ++q) if (sv[q] > gv || (sv[q] == gv && si[q] < gi)) { gv = sv[q]; gi = si[q]; }
    if (tid == 0) { const long o = ((long)row * BTK_SLICES + slice) * BTK_MAXC + c; cand_v[o] = gv; cand_i[o] = gi; }
#pragma unroll
    for (int k = 0; k < PER; ++k) if (s0 + tid + 256 * k == gi) v[k] = -INFINITY;
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void beam_topk_b(const float* __restrict__ part_ms, const float* __restrict__ cand_v, const int* __restrict__ cand_i,
                                                   const float* __restrict__ beam_scores, int K, int V, int C, float* __restrict__ out_s, int64_t* __restrict__ out_i) {
  __shared__ float rowM[16], rowL[16];
  __shared__ float sv[4]; __shared__ long si[4];
  const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid < K) {
    const float* pm = part_ms + (long)(g * K + tid) * BTK_SLICES * 2;
    float M = -INFINITY;
    for (int s_ = 0; s_ < BTK_SLICES; ++s_) M = fmaxf(M, pm[2 * s_]);
    float S = 0.f;
    for (int s_ = 0; s_ < BTK_SLICES; ++s_) S += pm[2 * s_ + 1] * __expf(pm[2 * s_] - M);
    rowM[tid] = M; rowL[tid] = logf(S);
  }
  __syncthreads();
  constexpr int PER = 16 * BTK_SLICES * BTK_MAXC / 256;           // candidates per thread at most (16 rows)
  float v[PER]; long id[PER];
  const int n = K * BTK_SLICES * C;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int t = tid + 256 * k;
    v[k] = -INFINITY; id[k] = 0x7fffffffffffffffL;
    if (t < n) {
      const int r = t / (BTK_SLICES * C), rem = t - r * (BTK_SLICES * C), sl = rem / C, c = rem - sl * C;
      const long o = ((long)(g * K + r) * BTK_SLICES + sl) * BTK_MAXC + c;
      const float x = cand_v[o];
      if (x != -INFINITY) { v[k] = ((x - rowM[r]) - rowL[r]) + beam_scores[g * K + r]; id[k] = (long)r * V + cand_i[o]; }
    }
  }
  for (int c = 0; c < C; ++c) {
    float bv = -INFINITY; long bi = 0x7fffffffffffffffL;
#pragma unroll
    for (int k = 0; k < PER; ++k) if (v[k] > bv || (v[k] == bv && id[k] < bi)) { bv = v[k]; bi = id[k]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float v2 = __shfl_xor(bv, o, 64); const long i2 = __shfl_xor(bi, o, 64);
      if (v2 > bv || (v2 == bv && i2 < bi)) { bv = v2; bi = i2; }
    }
    if (lane == 0) { sv[w] = bv; si[w] = bi; }
    __syncthreads();
    float gv = sv[0]; long gi = si[0];
#pragma unroll
    for (int q = 1; q < 4; ++q) if (sv[q] > gv || (sv[q] == gv && si[q] < gi)) { gv = sv[q]; gi = si[q]; }
    if (tid == 0) { out_s[(long)g * C + c] = gv; out_i[(long)g * C + c] = gi; }
#pragma unroll
    for (int k = 0; k < PER; ++k) if (id[k] == gi) v[k] = -INFINITY;
    __syncthreads();
  }
}
extern "C" int64_t unimp_beam_topk_scratch(int rows) { return (int64_t)rows * BTK_SLICES * (2 + 2 * BTK_MAXC) * 4; }      // bytes
extern "C" int unimp_beam_topk(const void* logits, int logits_f32, int64_t ld, int rows, int V, int K, int C, const float* beam_scores, void* scratch,
                               float* out_scores, int64_t* out_idx, void* stream) {
  if (!logits || !beam_scores || !scratch || !out_scores || !out_idx) return unimp_set_error(UNIMP_ERR_ARG, "beam_topk: null pointer");
  if (rows <= 0 || K <= 0 || rows % K || K > 16 || C <= 0 || C > BTK_MAXC || V <= 0) return unimp_set_error(UNIMP_ERR_SHAPE, "beam_topk: rows a multiple of K <= 16, 1 <= C <= 32");
  const int per = (V + BTK_SLICES - 1) / BTK_SLICES;
  if (per > 256 * 24 || per < C) return unimp_set_error(UNIMP_ERR_UNSUPPORTED, "beam_topk: vocabulary between 16 C and 98 304");
  float* part = (float*)scratch; float* cv = part + (long)rows * BTK_SLICES * 2; int* ci = (int*)(cv + (long)rows * BTK_SLICES * BTK_MAXC);
  hipStream_t s = (hipStream_t)stream;
  dim3 ga(BTK_SLICES, rows);
#define BTK_A(T_, P_) hipLaunchKernelGGL((beam_topk_a<T_, P_>), ga, dim3(256), 0, s, (const T_*)logits, (long)ld, V, C, part, cv, ci)
  const int pt = (per + 255) / 256;
  if (logits_f32) { if (pt <= 8) BTK_A(float, 8); else if (pt <= 16) BTK_A(float, 16); else BTK_A(float, 24); }
  else { if (pt <= 8) BTK_A(bf16, 8); else if (pt <= 16) BTK_A(bf16, 16); else BTK_A(bf16, 24); }
#undef BTK_A
  hipLaunchKernelGGL(beam_topk_b, dim3(rows / K), dim3(256), 0, s, (const float*)part, (const float*)cv, (const int*)ci, beam_scores, K, V, C, out_scores, out_idx);
  return unimp_check_launch("beam_topk");
}

// ------------------------------------------------------------------------------------------- embedding
__global__ void embedding_fwd_kernel(const int64_t* __restrict__ ids, const bf16* __restrict__ W, long ldw,
                                     const int64_t* __restrict__ pos, const bf16* __restrict__ P, long ldp,
                                     bf16* __restrict__ out, long ldo, int rows, int D, int vocab) {
  int cpr = D >> 3;
  long total = (long)rows * cpr;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int c = i % cpr; long r = i / cpr;
    long id = ids[r];
    bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (id >= 0 && id < vocab) v = *(const bf16x8*)(W + id * ldw + c * 8);
    if (P) {
      bf16x8 q = *(const bf16x8*)(P + pos[r] * ldp + c * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = f2bf(bf2f(v[j]) + bf2f(q[j]));
    }
    *(bf16x8*)(out + r * ldo + c * 8) = v;
  }
}

__global__ void embedding_bwd_kernel(const int64_t* __restrict__ ids, const bf16* __restrict__ dout, long lddo,
                                     float* __restrict__ dW, long lddw, int rows, int D, int vocab) {
  int cpr = D >> 2;   // 4 elements per thread: one wave-instruction adds 256 contiguous bytes of one row
  long total = (long)rows * cpr;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int c = i % cpr; long r = i / cpr;
    long id = ids[r];
    if (id < 0 || id >= vocab) continue;
    bf16x4 v = *(const bf16x4*)(dout + r * lddo + c * 4);
    float* d = dW + id * lddw + c * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float f = bf2f(v[j]);
      if (f != 0.f) atomicAdd(d + j, f);
    }
  }
}

// the same WITHOUT atomics, bit-reproducible: the rows arrive sorted by id (stable: equal ids in row order; perm[t] = the row at sorted position t).
// (With fp32 atomicAdd the order of the adds into a hot row -- the pad token, <image> -- changes from launch to launch: 3 of 20 launches of the
// cfg2 shape gave other last bits, and a resumed run was bit-identical with the uninterrupted one only most of the time.)
// Pass A: block (segment s of EMB_SEG sorted positions, column group) sums every piece of equal ids inside its segment in row order; a piece that
// STARTS its run is added to dW by this block (one writer), the one piece that continues a run from the previous segment goes to partial[s].
// Pass B: the block of the FIRST continuation segment of a run adds the partials of the run's segments in segment order.  (One block per run with a
// serial walk was 6.2 ms at the cfg2 shape -- a 4 256-row run of <image> tokens, two dependent loads per row; the atomic form 1.3 ms.)
#define EMB_SEG 64
__global__ __launch_bounds__(256) void embedding_bwd_sorted_a(const int64_t* __restrict__ sid, const int64_t* __restrict__ perm, const bf16* __restrict__ dout,
                                                              long lddo, float* __restrict__ dW, long lddw, float* __restrict__ partial, int rows, int D, int vocab) {
  __shared__ long ps[EMB_SEG], is[EMB_SEG];
  const int s = blockIdx.x, t0 = s * EMB_SEG, n = min(EMB_SEG, rows - t0);
  if ((int)threadIdx.x < n) { ps[threadIdx.x] = perm[t0 + threadIdx.x]; is[threadIdx.x] = sid[t0 + threadIdx.x]; }
  __syncthreads();
  const int c = blockIdx.y * blockDim.x + threadIdx.x;
  if (c >= (D >> 2)) return;
  const bool cont0 = t0 > 0 && sid[t0 - 1] == is[0];             // the segment's first piece continues a run
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int piece0 = 0;
  for (int t = 0; t < n; ++t) {
    const long id = is[t];
    const bf16x4 v = *(const bf16x4*)(dout + ps[t] * lddo + c * 4);
    a0 += bf2f(v[0]); a1 += bf2f(v[1]); a2 += bf2f(v[2]); a3 += bf2f(v[3]);
    if (t == n - 1 || is[t + 1] != id) {                          // the piece ends here
      if (id >= 0 && id < vocab) {
        float* d = (piece0 == 0 && cont0) ? partial + (long)s * D + c * 4 : dW + id * lddw + c * 4;
        if (piece0 == 0 && cont0) { d[0] = a0; d[1] = a1; d[2] = a2; d[3] = a3; }
        else { d[0] += a0; d[1] += a1; d[2] += a2; d[3] += a3; }
      }
      a0 = a1 = a2 = a3 = 0.f;
      piece0 = t + 1;
    }
  }
}
__global__ __launch_bounds__(256) void embedding_bwd_sorted_b(const int64_t* __restrict__ sid, float* __restrict__ dW, long lddw, const float* __restrict__ partial,
                                                              int rows, int D, int vocab) {
  const int s = blockIdx.x, nseg = gridDim.x;
  const long id = sid[(long)s * EMB_SEG];
  if (s == 0 || id < 0 || id >= vocab || sid[(long)s * EMB_SEG - 1] != id) return;                        // not a continuation
  if (s > 1 && sid[(long)(s - 1) * EMB_SEG] == id && sid[(long)(s - 1) * EMB_SEG - 1] == id) return;      // not the FIRST continuation of its run
  const int c = blockIdx.y * blockDim.x + threadIdx.x;
  if (c >= (D >> 2)) return;
  float* d = dW + id * lddw + c * 4;
  float a0 = d[0], a1 = d[1], a2 = d[2], a3 = d[3];
  for (int q = s; q < nseg && sid[(long)q * EMB_SEG] == id; ++q) {       // sorted: a segment that begins with id after one that ended with it continues the run
    const float* pp = partial + (long)q * D + c * 4;
    a0 += pp[0]; a1 += pp[1]; a2 += pp[2]; a3 += pp[3];
  }
  d[0] = a0; d[1] = a1; d[2] = a2; d[3] = a3;
}

extern "C" int unimp_embedding_fwd(const int64_t* ids, const void* W, int64_t ldw, const int64_t* pos, const void* P, int64_t ldp,
                                   void* out, int64_t ldo, int rows, int D, int vocab, void* stream) {
  if (!ids || !W || !out || (P && !pos)) return unimp_set_error(UNIMP_ERR_ARG, "embedding_fwd: null pointer");
  if ((D & 7) || (ldw & 7) || (ldo & 7)) return unimp_set_error(UNIMP_ERR_SHAPE, "embedding_fwd: D, ld must be multiples of 8");
  if (rows <= 0) return UNIMP_OK;
  long total = (long)rows * (D >> 3);
  hipLaunchKernelGGL(embedding_fwd_kernel, GRID1D(total, 256), dim3(256), 0, (hipStream_t)stream, ids, (const bf16*)W, (long)ldw, pos,
                     (const bf16*)P, (long)ldp, (bf16*)out, (long)ldo, rows, D, vocab);
  return unimp_check_launch("embedding_fwd");
}

extern "C" int unimp_embedding_bwd(const int64_t* ids, const void* dout, int64_t lddo, float* dW32, int64_t lddw, int rows, int D,
                                   int vocab, void* stream) {
  if (!ids || !dout || !dW32) return unimp_set_error(UNIMP_ERR_ARG, "embedding_bwd: null pointer");
  if ((D & 3) || (lddo & 3)) return unimp_set_error(UNIMP_ERR_SHAPE, "embedding_bwd: D, ld must be multiples of 4");
  if (rows <= 0) return UNIMP_OK;
  long total = (long)rows * (D >> 2);
  hipLaunchKernelGGL(embedding_bwd_kernel, GRID1D(total, 256), dim3(256), 0, (hipStream_t)stream, ids, (const bf16*)dout, (long)lddo,
                     dW32, (long)lddw, rows, D, vocab);
  return unimp_check_launch("embedding_bwd");
}

extern "C" int64_t unimp_embedding_bwd_sorted_scratch(int rows, int D) {          // floats
  return rows <= 0 ? 0 : (int64_t)((rows + EMB_SEG - 1) / EMB_SEG) * D;
}
extern "C" int unimp_embedding_bwd_sorted(const int64_t* sorted_ids, const int64_t* perm, const void* dout, int64_t lddo, float* dW32, int64_t lddw,
                                          float* scratch, int rows, int D, int vocab, void* stream) {
  if (!sorted_ids || !perm || !dout || !dW32 || !scratch) return unimp_set_error(UNIMP_ERR_ARG, "embedding_bwd_sorted: null pointer");
  if ((D & 3) || (lddo & 3) || (lddw & 3)) return unimp_set_error(UNIMP_ERR_SHAPE, "embedding_bwd_sorted: D, ld must be multiples of 4");
  if (rows <= 0) return UNIMP_OK;
  dim3 grid((rows + EMB_SEG - 1) / EMB_SEG, ((D >> 2) + 255) / 256);
  hipLaunchKernelGGL(embedding_bwd_sorted_a, grid, dim3(256), 0, (hipStream_t)stream, sorted_ids, perm, (const bf16*)dout, (long)lddo,
                     dW32, (long)lddw, scratch, rows, D, vocab);
  hipLaunchKernelGGL(embedding_bwd_sorted_b, grid, dim3(256), 0, (hipStream_t)stream, sorted_ids, dW32, (long)lddw, (const float*)scratch, rows, D, vocab);
  return unimp_check_launch("embedding_bwd");
}

// ------------------------------------------------------------------------------------------- ViT input path
template <typename T>
__global__ void patchify_kernel(const T* __restrict__ px, bf16* __restrict__ cols, long ldc, int N, int Hi, int Wi, int P) {
  int gx = Wi / P, gy = Hi / P, K = 3 * P * P;
  int cpr = ldc >> 3;
  long total = (long)N * gx * gy * cpr;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int c = i % cpr; long row = i / cpr;
    int pxi = row % gx; long t = row / gx; int pyi = t % gy; long n = t / gy;
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int k = c * 8 + j;
      float v = 0.f;
      if (k < K) {
        int ch = k / (P * P), rem = k - ch * P * P, yy = rem / P, xx = rem - yy * P;
        v = (float)px[((n * 3 + ch) * Hi + pyi * P + yy) * (long)Wi + pxi * P + xx];
      }
      o[j] = f2bf(v);
    }
    *(bf16x8*)(cols + row * ldc + c * 8) = o;
  }
}

__global__ void vit_assemble_kernel(const bf16* __restrict__ patch, long ldp, const bf16* __restrict__ cls,
                                    const bf16* __restrict__ pos, bf16* __restrict__ x, int N, int np, int D) {
  int cpr = D >> 3;
  long total = (long)N * (np + 1) * cpr;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int c = i % cpr; long t = i / cpr; int tok = t % (np + 1); long n = t / (np + 1);
    bf16x8 a = tok == 0 ? *(const bf16x8*)(cls + c * 8) : *(const bf16x8*)(patch + (n * np + tok - 1) * ldp + c * 8);
    bf16x8 b = *(const bf16x8*)(pos + (long)tok * D + c * 8), o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = f2bf(bf2f(a[j]) + bf2f(b[j]));
    *(bf16x8*)(x + t * D + c * 8) = o;
  }
}

extern "C" int unimp_vit_patchify(const void* pixels, int pixels_f32, void* cols, int64_t ldc, int N, int Hi, int Wi, int P, void* stream) {
  if (!pixels || !cols) return unimp_set_error(UNIMP_ERR_ARG, "patchify: null pointer");
  if (P <= 0 || Hi % P || Wi % P || (ldc & 7) || ldc < 3 * P * P) return unimp_set_error(UNIMP_ERR_SHAPE, "patchify: bad shape");
  if (N <= 0) return UNIMP_OK;
  long total = (long)N * (Hi / P) * (Wi / P) * (ldc >> 3);
  hipStream_t s = (hipStream_t)stream;
  if (pixels_f32) hipLaunchKernelGGL((patchify_kernel<float>), GRID1D(total, 256), dim3(256), 0, s, (const float*)pixels, (bf16*)cols, (long)ldc, N, Hi, Wi, P);
  else hipLaunchKernelGGL((patchify_kernel<bf16>), GRID1D(total, 256), dim3(256), 0, s, (const bf16*)pixels, (bf16*)cols, (long)ldc, N, Hi, Wi, P);
  return unimp_check_launch("patchify");
}

extern "C" int unimp_vit_assemble(const void* patch, int64_t ldp, const void* cls, const void* pos, void* x, int N, int n_patch, int D, void* stream) {
  if (!patch || !cls || !pos || !x) return unimp_set_error(UNIMP_ERR_ARG, "vit_assemble: null pointer");
  if ((D & 7) || (ldp & 7)) return unimp_set_error(UNIMP_ERR_SHAPE, "vit_assemble: D, ld must be multiples of 8");
  if (N <= 0) return UNIMP_OK;
  long total = (long)N * (n_patch + 1) * (D >> 3);
  hipLaunchKernelGGL(vit_assemble_kernel, GRID1D(total, 256), dim3(256), 0, (hipStream_t)stream, (const bf16*)patch, (long)ldp,
                     (const bf16*)cls, (const bf16*)pos, (bf16*)x, N, n_patch, D);
  return unimp_check_launch("vit_assemble");
}

// ------------------------------------------------------------------------------------------- add / cast / swiglu / dot
__global__ void add_kernel(const bf16* __restrict__ a, const bf16* __restrict__ b, bf16* __restrict__ o, long n) {
  long nv = n >> 3;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
    bf16x8 x = *(const bf16x8*)(a + i * 8), y = *(const bf16x8*)(b + i * 8), z;
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = f2bf(bf2f(x[j]) + bf2f(y[j]));
    *(bf16x8*)(o + i * 8) = z;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) { long i = (n & ~7L) + threadIdx.x; o[i] = f2bf(bf2f(a[i]) + bf2f(b[i])); }
}

__global__ void cast_kernel(const float* __restrict__ s, bf16* __restrict__ d, long n, float scale) {
  long nv = n >> 3;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
    f32x4 x = *(const f32x4*)(s + i * 8), y = *(const f32x4*)(s + i * 8 + 4);
    bf16x8 z = {f2bf(x[0] * scale), f2bf(x[1] * scale), f2bf(x[2] * scale), f2bf(x[3] * scale),
                f2bf(y[0] * scale), f2bf(y[1] * scale), f2bf(y[2] * scale), f2bf(y[3] * scale)};
    *(bf16x8*)(d + i * 8) = z;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) { long i = (n & ~7L) + threadIdx.x; d[i] = f2bf(s[i] * scale); }
}

// Row gather (packed token order, round 3): dst[r] = idx[r] >= 0 ? src[idx[r]] : 0 for rows of D bf16 (D % 8 == 0), one wave per row,
// 16-byte chunks.  Packs the valid tokens of right-padded sequences ([B*L] -> [M] rows, idx = flat positions) and unpacks them again
// ([M] -> [B*L], idx = the inverse map with -1 at <PAD> positions, which come out as zeros: no separate memset).
__global__ __launch_bounds__(256) void gather_rows_kernel(const bf16* __restrict__ src, long lds_, const int* __restrict__ idx,
                                                          bf16* __restrict__ dst, long ldd, int rows, int D) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= rows) return;
  const int s = idx[r];
  const u32x4* sp = (const u32x4*)(src + (long)max(s, 0) * lds_);
  u32x4* dp = (u32x4*)(dst + (long)r * ldd);
  const int nc = D >> 3;
  for (int c = lane; c < nc; c += 64) dp[c] = s >= 0 ? sp[c] : u32x4{0u, 0u, 0u, 0u};
}
extern "C" int unimp_gather_rows(const void* src, int64_t lds_, const int32_t* idx, void* dst, int64_t ldd, int rows, int D, void* stream) {
  if (rows <= 0) return UNIMP_OK;
  if (!src || !idx || !dst) return unimp_set_error(UNIMP_ERR_ARG, "gather_rows: null pointer");
  if ((D & 7) || (lds_ & 7) || (ldd & 7) || (((uintptr_t)src | (uintptr_t)dst) & 15)) return unimp_set_error(UNIMP_ERR_ALIGN, "gather_rows: D and leading dimensions % 8, 16-byte aligned bases");
  hipLaunchKernelGGL(gather_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16*)src, (long)lds_, idx, (bf16*)dst, (long)ldd, rows, D);
  return unimp_check_launch("gather_rows");
}

// Trace marker: an empty kernel whose GRID SIZE carries an id (id workgroups of 64 threads), so that a rocprofv3 --kernel-trace of
// a long process can be cut to a region of interest afterwards (tools/trace_window.py: the timed steps of bench.py without model
// construction, autotuning and warm-up).  No memory traffic, ~1.5 us.
__global__ void unimp_marker_kernel(int id) { (void)id; }
extern "C" int unimp_marker(int id, void* stream) {
  if (id < 1 || id > 65535) return unimp_set_error(UNIMP_ERR_ARG, "marker: id must be in 1..65535");
  hipLaunchKernelGGL(unimp_marker_kernel, dim3(id), dim3(64), 0, (hipStream_t)stream, id);
  return unimp_check_launch("marker");
}

extern "C" int unimp_add_bf16(const void* a, const void* b, void* out, int64_t n, void* stream) {
  if (!a || !b || !out) return unimp_set_error(UNIMP_ERR_ARG, "add: null pointer");
  if (n <= 0) return UNIMP_OK;
  hipLaunchKernelGGL(add_kernel, GRID1D((n >> 3) + 1, 256), dim3(256), 0, (hipStream_t)stream, (const bf16*)a, (const bf16*)b, (bf16*)out, (long)n);
  return unimp_check_launch("add");
}
extern "C" int unimp_cast_f32_to_bf16(const float* src, void* dst, int64_t n, float scale, void* stream) {
  if (!src || !dst) return unimp_set_error(UNIMP_ERR_ARG, "cast: null pointer");
  if (n <= 0) return UNIMP_OK;
  hipLaunchKernelGGL(cast_kernel, GRID1D((n >> 3) + 1, 256), dim3(256), 0, (hipStream_t)stream, src, (bf16*)dst, (long)n, scale);
  return unimp_check_launch("cast");
}

// gate_up rows are [gate(F) | up(F)]  (llama.py:185-199 with gate_proj / up_proj outputs concatenated)
__global__ void swiglu_fwd_kernel(const bf16* __restrict__ gu, long ld, bf16* __restrict__ out, long ldo, int rows, int F) {
  int cpr = F >> 3; long total = (long)rows * cpr;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int c = i % cpr; long r = i / cpr;
    bf16x8 g = *(const bf16x8*)(gu + r * ld + c * 8), u = *(const bf16x8*)(gu + r * ld + F + c * 8), o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = f2bf(act_fwd(ACT_SILU, bf2f(g[j])) * bf2f(u[j]));
    *(bf16x8*)(out + r * ldo + c * 8) = o;
  }
}
__global__ void swiglu_bwd_kernel(const bf16* __restrict__ gu, long ld, const bf16* __restrict__ dout, long lddo, bf16* __restrict__ dgu,
                                  long ldd, int rows, int F) {
  int cpr = F >> 3; long total = (long)rows * cpr;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int c = i % cpr; long r = i / cpr;
    bf16x8 g = *(const bf16x8*)(gu + r * ld + c * 8), u = *(const bf16x8*)(gu + r * ld + F + c * 8);
    bf16x8 d = *(const bf16x8*)(dout + r * lddo + c * 8), dg, du;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float gf = bf2f(g[j]), uf = bf2f(u[j]), df = bf2f(d[j]);
      dg[j] = f2bf(df * uf * act_bwd(ACT_SILU, gf));
      du[j] = f2bf(df * act_fwd(ACT_SILU, gf));
    }
    *(bf16x8*)(dgu + r * ldd + c * 8) = dg; *(bf16x8*)(dgu + r * ldd + F + c * 8) = du;
  }
}
extern "C" int unimp_swiglu_fwd(const void* gate_up, int64_t ld, void* out, int64_t ldo, int rows, int F, void* stream) {
  if (!gate_up || !out) return unimp_set_error(UNIMP_ERR_ARG, "swiglu: null pointer");
  if ((F & 7) || (ld & 7) || (ldo & 7)) return unimp_set_error(UNIMP_ERR_SHAPE, "swiglu: F, ld must be multiples of 8");
  if (rows <= 0) return UNIMP_OK;
  hipLaunchKernelGGL(swiglu_fwd_kernel, GRID1D((long)rows * (F >> 3), 256), dim3(256), 0, (hipStream_t)stream, (const bf16*)gate_up, (long)ld, (bf16*)out, (long)ldo, rows, F);
  return unimp_check_launch("swiglu_fwd");
}
extern "C" int unimp_swiglu_bwd(const void* gate_up, int64_t ld, const void* dout, int64_t lddo, void* dgate_up, int64_t ldd, int rows, int F, void* stream) {
  if (!gate_up || !dout || !dgate_up) return unimp_set_error(UNIMP_ERR_ARG, "swiglu_bwd: null pointer");
  if ((F & 7) || (ld & 7) || (lddo & 7) || (ldd & 7)) return unimp_set_error(UNIMP_ERR_SHAPE, "swiglu_bwd: F, ld must be multiples of 8");
  if (rows <= 0) return UNIMP_OK;
  hipLaunchKernelGGL(swiglu_bwd_kernel, GRID1D((long)rows * (F >> 3), 256), dim3(256), 0, (hipStream_t)stream, (const bf16*)gate_up, (long)ld, (const bf16*)dout, (long)lddo, (bf16*)dgate_up, (long)ldd, rows, F);
  return unimp_check_launch("swiglu_bwd");
}

// ordered two-stage reduction (like sumsq): per-block partials into out[1 + block], then one block adds them in a fixed order --
// the tanh-gate gradients are bit-reproducible from run to run (round 1 used one fp32 atomicAdd per block)
__global__ void dot_final_kernel(const float* __restrict__ part, int nb, float* __restrict__ out) {
  __shared__ float sh[256];
  float a = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) a += part[i];
  sh[threadIdx.x] = a;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) out[0] += sh[0];
}
__global__ void dot_kernel(const bf16* __restrict__ a, const bf16* __restrict__ b, long n, float* __restrict__ out) {
  __shared__ float sh[4];
  float acc = 0.f;
  long nv = n >> 3;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
    bf16x8 x = *(const bf16x8*)(a + i * 8), y = *(const bf16x8*)(b + i * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += bf2f(x[j]) * bf2f(y[j]);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) { long i = (n & ~7L) + threadIdx.x; acc += bf2f(a[i]) * bf2f(b[i]); }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[1 + blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
extern "C" int unimp_dot_bf16(const void* a, const void* b, int64_t n, float* out, void* stream) {
  if (!a || !b || !out) return unimp_set_error(UNIMP_ERR_ARG, "dot: null pointer");
  if (n <= 0) return UNIMP_OK;
  long blocks = std::min<long>(((n >> 3) + 255) / 256 + 1, 1024);
  hipLaunchKernelGGL(dot_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16*)a, (const bf16*)b, (long)n, out);
  hipLaunchKernelGGL(dot_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, out + 1, (int)blocks, out);
  return unimp_check_launch("dot");
}

// ------------------------------------------------------------------------------------------- periodic rows
__global__ void bcast_rows_kernel(const bf16* __restrict__ src, bf16* __restrict__ out, long ldo, int rows, int period, int D) {
  int cpr = D >> 3; long total = (long)rows * cpr;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int c = i % cpr; long r = i / cpr;
    *(bf16x8*)(out + r * ldo + c * 8) = *(const bf16x8*)(src + (long)(r % period) * D + c * 8);
  }
}
__global__ void reduce_rows_periodic_kernel(const bf16* __restrict__ src, long lds_, bf16* __restrict__ out, int rows, int period, int D) {
  int cpr = D >> 3; long total = (long)period * cpr;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int c = i % cpr; int j = i / cpr;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (long r = j; r < rows; r += period) {
    bf16x8 v = *(const bf16x8*)(src + r * lds_ + c * 8);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] += bf2f(v[k]);
  }
  bf16x8 o;
#pragma unroll
  for (int k = 0; k < 8; ++k) o[k] = f2bf(acc[k]);
  *(bf16x8*)(out + (long)j * D + c * 8) = o;
}
extern "C" int unimp_bcast_rows(const void* src, void* out, int64_t ldo, int rows, int period, int D, void* stream) {
  if (!src || !out) return unimp_set_error(UNIMP_ERR_ARG, "bcast_rows: null pointer");
  if ((D & 7) || (ldo & 7) || period <= 0) return unimp_set_error(UNIMP_ERR_SHAPE, "bcast_rows: bad shape");
  if (rows <= 0) return UNIMP_OK;
  hipLaunchKernelGGL(bcast_rows_kernel, GRID1D((long)rows * (D >> 3), 256), dim3(256), 0, (hipStream_t)stream, (const bf16*)src, (bf16*)out, (long)ldo, rows, period, D);
  return unimp_check_launch("bcast_rows");
}
extern "C" int unimp_reduce_rows_periodic(const void* src, int64_t lds_, void* out, int rows, int period, int D, void* stream) {
  if (!src || !out) return unimp_set_error(UNIMP_ERR_ARG, "reduce_rows_periodic: null pointer");
  if ((D & 7) || (lds_ & 7) || period <= 0) return unimp_set_error(UNIMP_ERR_SHAPE, "reduce_rows_periodic: bad shape");
  long total = (long)period * (D >> 3);
  hipLaunchKernelGGL(reduce_rows_periodic_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const bf16*)src, (long)lds_, (bf16*)out, rows, period, D);
  return unimp_check_launch("reduce_rows_periodic");
}
