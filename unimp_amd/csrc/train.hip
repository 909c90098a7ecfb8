// Training-step kernels: label mask + media_time scan, fused weighted focal cross-entropy (fwd / bwd),
// global grad-norm, fused flat AdamW.  All HBM-bound; 16-byte accesses; reductions are two-stage and
// ordered (no float atomics) so the loss and the clip coefficient are bit-reproducible from run to run for given
// gradients.  (The one float-atomic left on the training path is the embedding-table scatter-add,
// elementwise.hip::embedding_bwd_kernel: its fp32 sums may differ in the last bit between runs before they are
// rounded to bf16; replicas still agree because gradients are all-reduced.)
#include <algorithm>
#include "common.h"
#include "unimp_hip.h"

// ------------------------------------------------------------------------------------------- label mask
// One wave per row, 64 tokens per step.  Closed form of the reference's state machine (mmrec.py:143-168):
// keep[j] <=> (last <answer> strictly before j) > (last <|endofchunk|> strictly before j) and the token is none
// of {eoc, answer, pad, image} and j > 0.  media_time[j] = #(<image> tokens at positions <= j).
__global__ void label_mask_kernel(const int64_t* __restrict__ ids, int64_t* __restrict__ labels, int* __restrict__ media_time,
                                  int B, int L, long ans, long eoc, long pad, long img) {
  int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (row >= B) return;
  int lane = lane_id();
  int carry_la = -1, carry_le = -1, carry_img = 0;
  unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));     // lanes strictly below
  for (int base = 0; base < L; base += 64) {
    int j = base + lane;
    long t = j < L ? ids[(long)row * L + j] : -1;
    bool ok = j < L;
    unsigned long long ma = __ballot(ok && t == ans), me = __ballot(ok && t == eoc), mi = __ballot(ok && t == img);
    unsigned long long la_m = ma & below, le_m = me & below;
    int la = la_m ? base + 63 - __clzll(la_m) : carry_la;
    int le = le_m ? base + 63 - __clzll(le_m) : carry_le;
    bool keep = ok && la > le && t != eoc && t != ans && t != pad && t != img && j > 0;
    if (ok) {
      if (labels) labels[(long)row * L + j] = keep ? t : -100;
      if (media_time) media_time[(long)row * L + j] = carry_img + __popcll(mi & (below | (1ull << lane)));
    }
    if (ma) carry_la = base + 63 - __clzll(ma);
    if (me) carry_le = base + 63 - __clzll(me);
    carry_img += __popcll(mi);
  }
}

extern "C" int unimp_label_mask(const int64_t* ids, int64_t* labels, int32_t* media_time, int B, int L, int64_t answer_id,
                                int64_t eoc_id, int64_t pad_id, int64_t media_id, void* stream) {
  if (!ids) return unimp_set_error(UNIMP_ERR_ARG, "label_mask: null ids");
  if (B <= 0 || L <= 0) return UNIMP_OK;
  hipLaunchKernelGGL(label_mask_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, ids, labels, media_time, B, L,
                     (long)answer_id, (long)eoc_id, (long)pad_id, (long)media_id);
  return unimp_check_launch("label_mask");
}

// ------------------------------------------------------------------------------------------- focal CE
__device__ __forceinline__ void online_add(float& m, float& s, float z) {
  if (z > m) { s = s * __expf(m - z) + 1.f; m = z; } else { s += __expf(z - m); }
}
__device__ __forceinline__ void online_merge(float& m, float& s, float m2, float s2) {
  float mn = fmaxf(m, m2);
  if (mn == -INFINITY) { s = 0.f; return; }
  s = s * __expf(m - mn) + s2 * __expf(m2 - mn);
  m = mn;
}

// one 256-thread block per scored position (b, j), j < L-1; label = labels[b][j+1]
__global__ __launch_bounds__(256) void focal_fwd_kernel(const bf16* __restrict__ logits, long ldv, const int64_t* __restrict__ labels,
                                                        float* __restrict__ row_lse, float* __restrict__ row_zy, int B, int L, int V) {
  __shared__ float shm[4], shs[4];
  long r = blockIdx.x;            // r = b*L + j
  int j = r % L;
  long y = (j < L - 1) ? labels[r + 1] : -100;
  if (y < 0 || y >= V) { if (threadIdx.x == 0) { row_lse[r] = 0.f; row_zy[r] = 0.f; } return; }
  const bf16* z = logits + r * ldv;
  float m = -INFINITY, s = 0.f;
  int nch = (V + 7) >> 3;
  for (int c = threadIdx.x; c < nch; c += 256) {
    bf16x8 v = *(const bf16x8*)(z + c * 8);
#pragma unroll
    for (int k = 0; k < 8; ++k) if (c * 8 + k < V) online_add(m, s, bf2f(v[k]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64); online_merge(m, s, m2, s2); }
  if ((threadIdx.x & 63) == 0) { shm[threadIdx.x >> 6] = m; shs[threadIdx.x >> 6] = s; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) online_merge(m, s, shm[w], shs[w]);
    row_lse[r] = m + __logf(s);
    row_zy[r] = bf2f(z[y]);
  }
}

// ordered reduction of the per-row terms: out3 = {sum w*ce*(1-pt)^g, #valid, sum ce}
__global__ __launch_bounds__(256) void focal_reduce_kernel(const int64_t* __restrict__ labels, const float* __restrict__ weights,
                                                           const float* __restrict__ row_lse, const float* __restrict__ row_zy,
                                                           float gamma, int use_reweight, float* __restrict__ out3, int B, int L, int V) {
  __shared__ float sh[3][256];
  float a = 0.f, n = 0.f, c = 0.f;
  long total = (long)B * L;
  for (long r = threadIdx.x; r < total; r += 256) {
    int j = r % L; int b = r / L;
    long y = (j < L - 1) ? labels[r + 1] : -100;
    if (y < 0 || y >= V) continue;
    float ce = row_lse[r] - row_zy[r];
    float term = weights[b] * ce;
    if (use_reweight) { float pt = __expf(-ce); term *= powf(fmaxf(1.f - pt, 0.f), gamma); }
    a += term; n += 1.f; c += ce;
  }
  sh[0][threadIdx.x] = a; sh[1][threadIdx.x] = n; sh[2][threadIdx.x] = c;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) for (int k = 0; k < 3; ++k) sh[k][threadIdx.x] += sh[k][threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x < 3) out3[threadIdx.x] = sh[threadIdx.x][0];
}

// one block per position (all B*L rows incl. unlabeled ones, which are zero-filled)
__global__ __launch_bounds__(256) void focal_bwd_kernel(const bf16* __restrict__ logits, long ldv, const int64_t* __restrict__ labels,
                                                        const float* __restrict__ weights, float gamma, int use_reweight,
                                                        const float* __restrict__ row_lse, const float* __restrict__ row_zy,
                                                        const float* __restrict__ out3, const float* __restrict__ gscale,
                                                        bf16* __restrict__ dl, int B, int L, int V) {
  long r = blockIdx.x;
  int j = r % L; int b = r / L;
  long y = (j < L - 1) ? labels[r + 1] : -100;
  const bf16* z = logits + r * ldv;
  bf16* d = dl + r * ldv;
  int nch = ldv >> 3;
  if (y < 0 || y >= V) {
    bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int c = threadIdx.x; c < nch; c += 256) *(bf16x8*)(d + c * 8) = zero;
    return;
  }
  float lse = row_lse[r], ce = lse - row_zy[r];
  float coef = 1.f;
  if (use_reweight && gamma != 0.f) {
    float pt = __expf(-ce), om = fmaxf(1.f - pt, 0.f);
    coef = powf(om, gamma) + gamma * pt * powf(om, gamma - 1.f) * ce;      // log pt = -ce
  }
  float g = (gscale ? gscale[0] : 1.f) * weights[b] * coef / out3[1];
  for (int c = threadIdx.x; c < nch; c += 256) {
    bf16x8 v = *(const bf16x8*)(z + c * 8), o;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      int col = c * 8 + k;
      float p = col < V ? __expf(bf2f(v[k]) - lse) : 0.f;
      if (col == y) p -= 1.f;
      o[k] = f2bf(g * p);
    }
    *(bf16x8*)(d + c * 8) = o;
  }
}

// the same gradient for a LIST of scored positions, written compactly: block i handles row rows[i] (= b*L + j) and writes
// dl[i][0..ldd).  Rows without a label never appear in the list: the dense [B*L][ldv] gradient the reference materialises is
// zero there, and neither dX = dlogits W nor dW = dlogits^T h receives anything from a zero row.
__global__ __launch_bounds__(256) void focal_bwd_rows_kernel(const bf16* __restrict__ logits, long ldv, const int64_t* __restrict__ labels,
                                                             const float* __restrict__ weights, float gamma, int use_reweight,
                                                             const float* __restrict__ row_lse, const float* __restrict__ row_zy,
                                                             const float* __restrict__ out3, const float* __restrict__ gscale,
                                                             const int64_t* __restrict__ rows, bf16* __restrict__ dl, long ldd, int L, int V) {
  long r = rows[blockIdx.x];
  int j = r % L; int b = r / L;
  long y = (j < L - 1) ? labels[r + 1] : -100;
  const bf16* z = logits + r * ldv;
  bf16* d = dl + (long)blockIdx.x * ldd;
  int nch = ldd >> 3;
  if (y < 0 || y >= V) {                                   // a listed row without a label: zeros (the dense kernel's value)
    bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int c = threadIdx.x; c < nch; c += 256) *(bf16x8*)(d + c * 8) = zero;
    return;
  }
  float lse = row_lse[r], ce = lse - row_zy[r];
  float coef = 1.f;
  if (use_reweight && gamma != 0.f) {
    float pt = __expf(-ce), om = fmaxf(1.f - pt, 0.f);
    coef = powf(om, gamma) + gamma * pt * powf(om, gamma - 1.f) * ce;
  }
  float g = (gscale ? gscale[0] : 1.f) * weights[b] * coef / out3[1];
  for (int c = threadIdx.x; c < nch; c += 256) {
    bf16x8 v = *(const bf16x8*)(z + c * 8), o;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      int col = c * 8 + k;
      float p = col < V ? __expf(bf2f(v[k]) - lse) : 0.f;
      if (col == y) p -= 1.f;
      o[k] = f2bf(g * p);
    }
    *(bf16x8*)(d + c * 8) = o;
  }
}

extern "C" int unimp_focal_ce_bwd_rows(const void* logits, int64_t ldv, const int64_t* labels, const float* weights, float gamma,
                                       int use_reweight, const float* row_lse, const float* row_zy, const float* out3,
                                       const float* gscale, const int64_t* rows, int n_rows, void* dl, int64_t ldd, int L, int V,
                                       void* stream) {
  if (!logits || !labels || !weights || !row_lse || !row_zy || !out3 || !rows || !dl) return unimp_set_error(UNIMP_ERR_ARG, "focal_ce_bwd_rows: null pointer");
  if ((ldv & 7) || (ldd & 7) || ldv < V || ldd < V || ldd > ldv || ((uintptr_t)logits & 15) || ((uintptr_t)dl & 15))
    return unimp_set_error(UNIMP_ERR_ALIGN, "focal_ce_bwd_rows: ldv, ldd %% 8 == 0, V <= ldd <= ldv, 16-B aligned");
  if (n_rows <= 0 || L <= 0) return UNIMP_OK;
  hipLaunchKernelGGL(focal_bwd_rows_kernel, dim3((unsigned)n_rows), dim3(256), 0, (hipStream_t)stream, (const bf16*)logits, (long)ldv, labels,
                     weights, gamma, use_reweight, row_lse, row_zy, out3, gscale, rows, (bf16*)dl, (long)ldd, L, V);
  return unimp_check_launch("focal_ce_bwd_rows");
}

extern "C" int unimp_focal_ce_fwd(const void* logits, int64_t ldv, const int64_t* labels, const float* weights, float gamma,
                                  int use_reweight, float* row_lse, float* row_zy, float* out3, int B, int L, int V, void* stream) {
  if (!logits || !labels || !weights || !row_lse || !row_zy || !out3) return unimp_set_error(UNIMP_ERR_ARG, "focal_ce_fwd: null pointer");
  if ((ldv & 7) || ldv < V || ((uintptr_t)logits & 15)) return unimp_set_error(UNIMP_ERR_ALIGN, "focal_ce_fwd: ldv %% 8 == 0, ldv >= V, 16-B aligned logits");
  if (B <= 0 || L <= 0) return UNIMP_OK;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(focal_fwd_kernel, dim3((unsigned)((long)B * L)), dim3(256), 0, s, (const bf16*)logits, (long)ldv, labels, row_lse, row_zy, B, L, V);
  hipLaunchKernelGGL(focal_reduce_kernel, dim3(1), dim3(256), 0, s, labels, weights, row_lse, row_zy, gamma, use_reweight, out3, B, L, V);
  return unimp_check_launch("focal_ce_fwd");
}

extern "C" int unimp_focal_ce_bwd(const void* logits, int64_t ldv, const int64_t* labels, const float* weights, float gamma,
                                  int use_reweight, const float* row_lse, const float* row_zy, const float* out3, const float* gscale,
                                  void* dlogits, int B, int L, int V, void* stream) {
  if (!logits || !labels || !weights || !row_lse || !row_zy || !out3 || !dlogits) return unimp_set_error(UNIMP_ERR_ARG, "focal_ce_bwd: null pointer");
  if ((ldv & 7) || ldv < V || ((uintptr_t)logits & 15) || ((uintptr_t)dlogits & 15)) return unimp_set_error(UNIMP_ERR_ALIGN, "focal_ce_bwd: alignment");
  if (B <= 0 || L <= 0) return UNIMP_OK;
  hipLaunchKernelGGL(focal_bwd_kernel, dim3((unsigned)((long)B * L)), dim3(256), 0, (hipStream_t)stream, (const bf16*)logits, (long)ldv, labels,
                     weights, gamma, use_reweight, row_lse, row_zy, out3, gscale, (bf16*)dlogits, B, L, V);
  return unimp_check_launch("focal_ce_bwd");
}

// ------------------------------------------------------------------------------------------- grad norm
#define SUMSQ_BLOCKS 1024
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const bf16* __restrict__ g, long n, float* __restrict__ part) {
  __shared__ float sh[4];
  float acc = 0.f;
  long nv = n >> 3;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
    bf16x8 x = *(const bf16x8*)(g + i * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) { float f = bf2f(x[j]); acc += f * f; }
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) { float f = bf2f(g[(n & ~7L) + threadIdx.x]); acc += f * f; }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ part, int nb, float* __restrict__ out) {
  __shared__ float sh[256];
  float a = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) a += part[i];
  sh[threadIdx.x] = a;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) out[0] += sh[0];
}
// out[0] += sum(g^2); out[1 .. 1+SUMSQ_BLOCKS) is scratch for the ordered two-stage reduction
extern "C" int unimp_sumsq_bf16(const void* g, int64_t n, float* out, void* stream) {
  if (!g || !out) return unimp_set_error(UNIMP_ERR_ARG, "sumsq: null pointer");
  if ((uintptr_t)g & 15) return unimp_set_error(UNIMP_ERR_ALIGN, "sumsq: 16-B aligned buffer required");
  if (n <= 0) return UNIMP_OK;
  int nb = (int)std::min<long>(((n >> 3) + 255) / 256 + 1, SUMSQ_BLOCKS);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nb), dim3(256), 0, s, (const bf16*)g, (long)n, out + 1);
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, s, out + 1, nb, out);
  return unimp_check_launch("sumsq");
}

// ------------------------------------------------------------------------------------------- AdamW (flat, fused clip)
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ master, float* __restrict__ m, float* __restrict__ v,
                                                    bf16* __restrict__ p16, bf16* __restrict__ g16, long n, long n_decay, float lr,
                                                    float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt,
                                                    const float* __restrict__ sumsq, float gscale, float max_norm, int zero_grad) {
  float clip = 1.f;
  if (sumsq && max_norm > 0.f) { float nrm = sqrtf(sumsq[0]) * gscale; clip = fminf(1.f, max_norm / (nrm + 1e-6f)); }
  float gs = gscale * clip;
  long nv = (n + 7) >> 3;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
    long e0 = i * 8;
    if (e0 + 8 <= n) {
      bf16x8 g = *(const bf16x8*)(g16 + e0), po;
      f32x4 pm[2] = {*(f32x4*)(master + e0), *(f32x4*)(master + e0 + 4)};
      f32x4 mm[2] = {*(f32x4*)(m + e0), *(f32x4*)(m + e0 + 4)};
      f32x4 vv[2] = {*(f32x4*)(v + e0), *(f32x4*)(v + e0 + 4)};
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float gr = bf2f(g[k]) * gs;
        float decay = (e0 + k < n_decay) ? (1.f - lr * wd) : 1.f;
        float pv = pm[k >> 2][k & 3] * decay;
        float mn = mm[k >> 2][k & 3] * b1 + gr * (1.f - b1);
        float vn = vv[k >> 2][k & 3] * b2 + gr * gr * (1.f - b2);
        pv -= (lr / bc1) * mn / (sqrtf(vn) / bc2_sqrt + eps);
        pm[k >> 2][k & 3] = pv; mm[k >> 2][k & 3] = mn; vv[k >> 2][k & 3] = vn;
        po[k] = f2bf(pv);
      }
      *(f32x4*)(master + e0) = pm[0]; *(f32x4*)(master + e0 + 4) = pm[1];
      *(f32x4*)(m + e0) = mm[0]; *(f32x4*)(m + e0 + 4) = mm[1];
      *(f32x4*)(v + e0) = vv[0]; *(f32x4*)(v + e0 + 4) = vv[1];
      *(bf16x8*)(p16 + e0) = po;
      if (zero_grad) { bf16x8 zz = {0, 0, 0, 0, 0, 0, 0, 0}; *(bf16x8*)(g16 + e0) = zz; }
    } else {
      for (long e = e0; e < n; ++e) {
        float gr = bf2f(g16[e]) * gs;
        float decay = (e < n_decay) ? (1.f - lr * wd) : 1.f;
        float pv = master[e] * decay;
        float mn = m[e] * b1 + gr * (1.f - b1);
        float vn = v[e] * b2 + gr * gr * (1.f - b2);
        pv -= (lr / bc1) * mn / (sqrtf(vn) / bc2_sqrt + eps);
        master[e] = pv; m[e] = mn; v[e] = vn; p16[e] = f2bf(pv);
        if (zero_grad) g16[e] = f2bf(0.f);
      }
    }
  }
}

extern "C" int unimp_adamw_flat(float* master, float* m, float* v, void* param_bf16, void* grad_bf16, int64_t n, int64_t n_decay,
                                float lr, float beta1, float beta2, float eps, float wd, int step, const float* sumsq, float gscale,
                                float max_norm, int zero_grad, void* stream) {
  if (!master || !m || !v || !param_bf16 || !grad_bf16) return unimp_set_error(UNIMP_ERR_ARG, "adamw: null pointer");
  if (step < 1) return unimp_set_error(UNIMP_ERR_ARG, "adamw: step is 1-based");
  if (((uintptr_t)master | (uintptr_t)m | (uintptr_t)v | (uintptr_t)param_bf16 | (uintptr_t)grad_bf16) & 15)
    return unimp_set_error(UNIMP_ERR_ALIGN, "adamw: 16-B aligned buffers required");
  if (n <= 0) return UNIMP_OK;
  float bc1 = 1.f - powf(beta1, (float)step), bc2s = sqrtf(1.f - powf(beta2, (float)step));
  long nv = (n + 7) >> 3;
  int nb = (int)std::min<long>((nv + 255) / 256, 8192);
  hipLaunchKernelGGL(adamw_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, master, m, v, (bf16*)param_bf16, (bf16*)grad_bf16, (long)n,
                     (long)n_decay, lr, beta1, beta2, eps, wd, bc1, bc2s, sumsq, gscale, max_norm, zero_grad);
  return unimp_check_launch("adamw");
}
