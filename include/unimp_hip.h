/* unimp_hip.h -- C ABI of libunimp_hip.so: the MI355X (gfx950) kernels behind UniMP's Flamingo train step.
 *
 * The reference (weitianxin/UniMP) has no native FFI: its hot path is reached through the Python API of
 * the pip package open-flamingo==2.0.1 (UniMP/mmrec.py:20-22,476-524 construction; :177-181 forward;
 * :190-213 loss; :215-256 backward/clip/AdamW).  Each entry point below names the reference arithmetic it
 * replaces.  The Python package `unimp_amd` (the open_flamingo-surface drop-in) is the only caller.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer on the current HIP device unless marked host;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing synchronises;
 *   - bf16 tensors are row-major with an explicit leading dimension in ELEMENTS; vector-accessed bases
 *     must be 16-byte aligned and leading dimensions multiples of 8 unless stated;
 *   - no ownership transfer, no allocation inside any entry point (graph-capture safe);
 *   - return 0 on success, a UNIMP_ERR_* code otherwise (message via unimp_last_error()).
 */
#ifndef UNIMP_HIP_H
#define UNIMP_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* bumped whenever a signature, a descriptor layout or a buffer-size requirement changes incompatibly (2: round-2 additions --
 * layernorm_bwd(wgrad_accumulate), dot_bf16's fp32[1+1024] scratch, grown gemm / attention descriptors; 3: round 3) */
#define UNIMP_ABI_VERSION 8
enum { UNIMP_OK = 0, UNIMP_ERR_ARG = 1, UNIMP_ERR_SHAPE = 2, UNIMP_ERR_ALIGN = 3, UNIMP_ERR_LAUNCH = 4,
       UNIMP_ERR_UNSUPPORTED = 5 };
enum { UNIMP_ACT_NONE = 0, UNIMP_ACT_GELU = 1, UNIMP_ACT_QUICKGELU = 2, UNIMP_ACT_RELU = 3, UNIMP_ACT_SILU = 4,
       UNIMP_ACT_DERIV = 5 /* dact only: aux already holds act'(z) */ };
enum { UNIMP_MASK_NONE = 0, UNIMP_MASK_CAUSAL = 1, UNIMP_MASK_SEGMENT = 2 };

int unimp_abi_version(void);
int unimp_struct_size(int which);            /* sizeof(unimp_gemm_desc / unimp_attn_desc / unimp_image_desc / unimp_mx_gemm_desc / unimp_decode_step_desc) for which = 0 ... 4 */
const char* unimp_last_error(void);          /* thread-local, valid until the next failing call */
/* internal helpers shared by the translation units (exported for the tests' benefit only) */
int unimp_set_error(int code, const char* msg);
int unimp_check_launch(const char* what);

/* ---- dense layers: nn.Linear forward / dX / dW, ViT conv1-as-GEMM --------------------------------------
 * replaces: F.linear in every tower (clip.py:98-141,144-156; gpt_neox modelling :180-283; open_flamingo
 * helpers to_q/to_kv/to_out/ff), their autograd backward, and the lm-head GEMM (mmrec.py:177-190).
 *   C[M,N] = epi( alpha * sum_k A(m,k) B(n,k) )
 *   a_kstrided=0: A(m,k)=A[m*lda+k]   a_kstrided=1: A(m,k)=A[k*lda+m]      (same for B with n)
 *   epi: v += bias[n]; pre[m,n] = v (or act'(v) if pre_deriv); v = act(v); v *= act'(aux[m,n]) (dact); v *= tanh(*gate);
 *        v += res[m,n]; if accumulate v += C[m,n]; C = bf16(v) or f32(v)
 */
typedef struct {
  const void* A; const void* B; void* C;
  int M, N, K;
  int64_t lda, ldb, ldc;
  int a_kstrided, b_kstrided;
  const void* bias;
  const void* res; int64_t ldres;
  const void* aux; int64_t ldaux;
  void* pre; int64_t ldpre;
  const void* gate;
  float alpha;
  int act, dact;
  int out_f32, accumulate;
  int pre_deriv;   /* 1: pre receives act'(v) instead of v: the backward multiply then needs no transcendental (dact = DERIV = 5).
                    * 2: the same as uint8 [M][ldpre BYTES], q = round(202 act'(v) + 27) -- every derivative served lies in [-0.129, 1.129],
                    * 0 and 1 are exact, |error| <= 0.0025 (finer than bf16 above 0.63); read back with dact = 6 (aux uint8, ldaux in bytes).
                    * Half the bytes of the up-projection's second output and of the backward's aux operand. */
  /* optional rotary epilogue (rope_rot > 0; the fused QKV projection of a GPT-NeoX / Llama block): C is [tokens][N] with
   * position = row % rope_L; column n belongs to a rotated vector iff (n % rope_period) < rope_span, its index inside the
   * head vector is n % rope_hd, and the first rope_rot indices are rotated.  The rotation pairs are ADJACENT in memory:
   * index 8g + j pairs with 8g + 4 + j (j < 4) at frequency i = 4g + j, theta_i = base^(-2i / rope_rot) -- i.e. the caller
   * permuted the projection's rows from the half-split order (i, i + rope_rot / 2) once (attention is invariant to a common
   * permutation of q and k).  cos / sin of position * theta_i are computed in the epilogue (v_sin / v_cos on the fractional
   * number of turns; |error| <= 6e-5 against fp32 tables at position 2048).  Plain epilogue (alpha, bias) only, N % 8 == 0,
   * ldc % 8 == 0, k-contiguous A and B, variants PP256 / PP256P (UNIMP_ERR_UNSUPPORTED otherwise).  rope_log2_base = log2(base). */
  int rope_rot, rope_hd, rope_period, rope_span, rope_L;
  float rope_log2_base;
  /* ABI 4: NULL, or int32 [M]: the position of row m (instead of m % rope_L) -- packed rows, where a sequence starts at any row */
  const int32_t* rope_pos;
  /* ABI 8, decode rows only (variant SKINNY with M <= 16, K % 64 == 0, K <= 4096: unimp_gemm_skinny_ln_ok): ln_gamma != NULL -- the rows of A
   * are LAYER-NORMALISED on their way into the product, y = (a - mean) * rstd * ln_gamma + ln_beta (ln_beta may be NULL) in fp32, rounded to
   * bf16: what unimp_layernorm_fwd would have stored, without that launch.  Replaces the nn.LayerNorm in front of a projection in a
   * cached decode step (GPTNeoXLayer.input_layernorm / post_attention_layernorm, open_flamingo MaskedCrossAttention.norm, FeedForward[0],
   * final_layer_norm in front of embed_out).  Other variants refuse it. */
  const void* ln_gamma; const void* ln_beta; float ln_eps;
} unimp_gemm_desc;
/* Pre-packed B operand for FROZEN weights (b_kstrided = 2 in the descriptor; ping-pong variants only): every MFMA B fragment
 * of the 16x16x32 instruction -- (n-tile of 16, 32-k step) -- stored as one contiguous 1-KiB block in lane order, so a wave
 * fetches it with one coalesced global_load_dwordx4 and the B operand bypasses the LDS.  pack_b_bytes gives the image size
 * (N rounded up to 256 columns, K to 32); `kstrided` describes the SOURCE (0: X[n*ld + k], 1: X[k*ld + n]).  Results are
 * bit-identical to the unpacked ping-pong kernel (same k grouping inside the MFMAs). */
int64_t unimp_pack_b_bytes(int N, int K);
int unimp_pack_b_bf16(const void* X, int64_t ld, int N, int K, int kstrided, void* out, void* stream);
int unimp_gemm_bf16(const unimp_gemm_desc* d, void* stream);
/* same, with an explicit kernel variant (all compute identical results up to fp32 summation order):
 *   V1      128x128x64 tiles, 4 waves, register-staged double buffer, 2 workgroups / CU (small or ragged problems)
 *   DMA256/128  256 x {256,128} tiles, 8 waves, LDS-DMA staging, 2-stage ring
 *   PP256/128   256 x {256,128} tiles, 8 waves, LDS-DMA, 4-deep ring of 32-k half-stages, SIMD partners ping-ponged
 * AUTO picks by shape; the Python layer autotunes per (shape, layout) on first use. */
enum { UNIMP_GEMM_AUTO = 0, UNIMP_GEMM_V1 = 1, UNIMP_GEMM_DMA256 = 2, UNIMP_GEMM_DMA128 = 3, UNIMP_GEMM_PP256 = 4,
       UNIMP_GEMM_PP128 = 5, UNIMP_GEMM_SKINNY = 6 /* M <= 64 (decode): W streamed once, no LDS staging */,
       UNIMP_GEMM_W4 = 7 /* 256 x 256 tiles, 4 waves of 128 x 128 (one wave per SIMD) */,
       UNIMP_GEMM_W8 = 8 /* 256 x 256 tiles, 8 self-interleaving waves, one barrier per half-stage */,
       UNIMP_GEMM_PP256P = 9 /* PP256 as a persistent kernel: one workgroup per CU walks its tiles, the next tile's first
                                LDS-DMA half-stages are issued before the current tile's epilogue */,
       UNIMP_GEMM_PP256A = 13, UNIMP_GEMM_PP128A = 14 /* PP256X / PP128X with a k-contiguous A operand staged in whole 128-byte rows
          (64-k stages, a ring of its own); other operand forms run the X kernels */,
       UNIMP_GEMM_PP256X = 10, UNIMP_GEMM_PP128X = 11, UNIMP_GEMM_PP256PX = 12,
       UNIMP_GEMM_W4X = 15 /* round 5: W4's blocking (one wave per SIMD, 128 x 128 per wave) with a hand-ordered two-register-set main loop over 64-k stages
                              (gemm7.hip); k-contiguous A, K % 64 == 0; other forms are refused */,
       UNIMP_GEMM_W4X_S1 = 16 /* measurement build of W4X: its first schedule (one release barrier per stage, separate M0 writes) */,
       UNIMP_GEMM_PP256B = 18, UNIMP_GEMM_PP128B = 19 /* measurement builds: PP256A / PP128A with the steady-state L phase spelled in asm (M0 writes fused with the fragment reads) */,
       UNIMP_GEMM_DW = 20 /* round 6: 128 x 256 tiles, 4 waves (the ping-pong kernel's 128 x 64 wave tile), 72 KiB of LDS: TWO independent workgroups per CU -- one
                                 multiplies while the other is between tiles (gemm9.hip); every operand form, fixed epilogue kinds, no rotary epilogue, no packed B */,
       UNIMP_GEMM_W4X_PF = 17 /* measurement build of W4X: each workgroup prefetches its share of the A / B panels' lines into L2 two stages ahead of the DMA */
       /* PP256 / PP128 / PP256P with ONE fragment register set: the L phase of a half-step reads that half-step's own fragments
          (before its LDS-DMA issue) and leaves two half-stages in flight instead of one -- 48 registers fewer, same bits */ };
int unimp_gemm_bf16_variant(const unimp_gemm_desc* d, int variant, void* stream);
/* decode rows (ABI 8): M <= 16 rows with K <= 4096 run the second-generation weight-streaming kernel (gemm.hip skinny2: every load of a wave
 * issued before the first wait, optional fused LayerNorm) instead of round 3's.  set_skinny2(0) restores the latter (A/B, tests; env
 * UNIMP_SKINNY2); returns the previous setting.  skinny_ln_ok: may a decode GEMM of M rows, depth K take ln_gamma? */
int unimp_gemm_set_skinny2(int on);
int unimp_gemm_skinny_ln_ok(int M, int K);
/* how the decode-row GEMM splits N weight rows over workgroups (host arithmetic only): rows per MFMA tile (<= 16; the busiest CU's rows are minimised
 * below 512 sixteen-row tiles: 10 for N = 2560, 15 for 7680) and, through *tiles_per_workgroup, 1 or 2 tiles per workgroup (max_tiles = 2: the
 * fused-LayerNorm forms at M >= 4).  UNIMP_SKINNY2_ROWS = 1 ... 16 fixes the rows. */
int unimp_gemm_skinny_rows(int N, int max_tiles, int* tiles_per_workgroup);
/* split-K form for outputs much smaller than the chip (weight gradients of narrow projections): `splits` K slices into
 * f32 slabs [splits][M][N] (caller-provided workspace), then an ordered reduction applying alpha*tanh(gate).  Only the
 * alpha / gate epilogue is allowed. */
int unimp_gemm_bf16_splitk(const unimp_gemm_desc* d, int splits, float* slabs, void* stream);

/* ---- LayerNorm / RMSNorm ---------------------------------------------------------------------------------
 * replaces nn.LayerNorm (clip.py:164-166,423; gpt_neox input/post_attention/final LN; open_flamingo norms)
 * and LlamaRMSNorm (llama.py:101-118).  rows x D, one wave per row, statistics in fp32.
 * Output row r goes to row (r / grp) * grp_stride + (r % grp) + grp_off of y (grp = 0: identity) so the
 * Perceiver's cat(x, latents) (open_flamingo PerceiverAttention) is written in place.
 * bwd: dx = LN'(dy) (+ dres if given); dgamma/dbeta (bf16 [D]; overwritten, or added to when wgrad_accumulate != 0: the
 * parameter's slot of a gradient buffer) through `partial` (fp32
 * [partial_blocks*2*D]) when gamma grads are wanted (dgamma != NULL); dy rows are read through the same row map;
 * dy2 (optional, plain row layout) is added to dy first (the Perceiver's norm_latents output feeds both q and kv).
 */
int unimp_layernorm_fwd(const void* x, int64_t ldx, const void* gamma, const void* beta, void* y, int64_t ldy,
                        float* mean, float* rstd, int rows, int D, float eps, int rms,
                        int grp, int grp_stride, int grp_off, void* stream);
/* the same row arithmetic with the normalised row leaving as an MX-fp8 operand (e4m3 bytes [rows][D], ldyq in bytes; E8M0 [rows][D/32]):
 * the bytes unimp_mx_quantize makes of the bf16 row -- LayerNorm feeding a frozen projection on the MX GEMM (F4).  D % 32 == 0. */
int unimp_layernorm_fwd_mx(const void* x, int64_t ldx, const void* gamma, const void* beta, void* yq, int64_t ldyq, void* scales,
                           int64_t ldsc, float* mean, float* rstd, int rows, int D, float eps, int rms, void* stream);
int unimp_layernorm_bwd(const void* dy, int64_t lddy, const void* dy2, int64_t lddy2, const void* x, int64_t ldx, const void* gamma,
                        const float* mean, const float* rstd, const void* dres, int64_t lddres,
                        void* dx, int64_t lddx, void* dgamma, void* dbeta, float* partial, int partial_blocks,
                        int rows, int D, int rms, int grp, int grp_stride, int grp_off, int wgrad_accumulate, void* stream);

/* ---- rotary embedding, GPT-NeoX / Llama half-split (gpt_neox modelling :107-160; llama.py:121-182) ------
 * in place on x viewed as [rows = B*L][heads][head_stride]; rotates the first `rot` dims of the `nvec`
 * vectors (q and k) found at element offsets vec_off[0..nvec) inside each head slot; position = row % L.
 * inverse != 0 applies the transpose rotation (backward).  cos/sin: fp32 [L][rot/2].
 */
int unimp_rope_halfsplit(void* x, int64_t row_stride, int64_t head_stride, int rows, int L, int heads, int rot,
                         int nvec, int vec_off0, int vec_off1, const float* cos_t, const float* sin_t, int inverse,
                         void* stream);
/* the same with the position of every row given (int32 [rows], each < the tables' row count) instead of row % L: packed rows (ABI 4) */
int unimp_rope_halfsplit_pos(void* x, int64_t row_stride, int64_t head_stride, int rows, const int32_t* pos, int heads, int rot,
                             int nvec, int vec_off0, int vec_off1, const float* cos_t, const float* sin_t, int inverse,
                             void* stream);

/* decode step (ABI 8; F1: eval_rec.py:100-110 / eval_img_gen.py:102-111 through Flamingo.generate with a KV cache -- transformers'
 * GPTNeoXAttention with layer_past: apply_rotary_pos_emb on the new token's q / k, then torch.cat of k / v onto the cache): ONE launch
 * rotates q and k of every row in place (row r at its own position: cos_rows / sin_rows fp32 [rows][rot / 2] hold that position's table
 * row; rot = 0: no rotation) and writes the rotated k and v into slot pos_idx[r] of the caches
 * (element strides: row, slot, head).  Same per-element arithmetic as unimp_rope_halfsplit.  q / k / v at element offsets
 * q_off / k_off / v_off inside each head slot of qkv [rows][heads][head_stride]. */
int unimp_decode_rope_append(void* qkv, int64_t row_stride, int64_t head_stride, int rows, int heads, int hd, int q_off, int k_off,
                             int v_off, int rot, const float* cos_rows, const float* sin_rows, void* kcache, void* vcache,
                             int64_t c_row_stride, int64_t c_slot_stride, int64_t c_head_stride, const int64_t* pos_idx, void* stream);

/* beam search (ABI 8): transformers' GenerationMixin._reorder_cache (index_select of past_key_values per step; eval_rec.py:100-110, K = 10) on the
 * generated tail of the cache, in place, one launch: kv = [n_planes = layers x 2][rows = n_groups x K][capacity][row_elems] (element strides s_plane,
 * s_row, s_slot); row j of group g takes what row src_local[g K + j] (index INSIDE the group) held, for the slots slot0[g] .. pos_idx[g K] - 1 (the
 * tokens generated so far; max_new bounds the launch).  K <= 16. */
int unimp_kv_reorder_beams(void* kv, int64_t s_plane, int n_planes, int64_t s_row, int64_t s_slot, int row_elems, int K, int n_groups,
                           const int64_t* src_local, const int32_t* slot0, const int64_t* pos_idx, int max_new, void* stream);

/* ---- attention (flash-style, MFMA) -------------------------------------------------------------------------
 * replaces xformers.ops.memory_efficient_attention (clip.py:130-136; llama.py:287-301), the GPT-NeoX causal
 * SDPA, open_flamingo PerceiverAttention and MaskedCrossAttention softmax(QK^T)V.
 * q/k/v/o are [B][S][H][D] views with element strides (batch, seq, head); D in {64, 80, 128}.
 *   MASK_NONE   : keys j < kv_len[b] (kv_len NULL: all Sk).  K/V rows in [kv_len[b], Sk) are still loaded (whole 64-key
 *                 tiles) and weighted with p = 0: they must hold finite values (a KV cache is zero-initialised)
 *   MASK_CAUSAL : keys j <= i and j < kv_len[b]
 *   MASK_SEGMENT: query i attends keys [(t-1)*seg_len, t*seg_len) with t = seg[b*Sq+i]; t == 0: output 0
 *                 (open_flamingo only_attend_immediate_media=True, rows before the first <image> zeroed)
 * lse: fp32 [B][H][Sq] (natural log; -inf for empty rows).
 */
typedef struct {
  const void* q; const void* k; const void* v; void* o; float* lse;
  int64_t q_bs, q_ss, q_hs, k_bs, k_ss, k_hs, v_bs, v_ss, v_hs, o_bs, o_ss, o_hs;
  int B, H, Sq, Sk, D;
  float scale;
  int mask_mode;
  const int32_t* kv_len;
  const int32_t* seg; int seg_len;
  /* backward only */
  const void* d_o; void* dq; void* dk; void* dv; float* delta;   /* delta: fp32 [B][H][Sq] workspace */
  int64_t do_bs, do_ss, do_hs, dq_bs, dq_ss, dq_hs, dk_bs, dk_ss, dk_hs, dv_bs, dv_ss, dv_hs;
  const float* alibi_slopes;   /* NULL, or fp32 [H]: score(i, j) += slope[h] * j (MPT's key-position ALiBi: softmax is shift-invariant
                                  per row, so this equals slope * (j - i) and transformers' slope * (j - (Sk - 1))) */
  /* backward only, optional: q and k entered the forward rotated (unimp_rope_halfsplit); with rope_cos / rope_sin set
   * (fp32 [>= max(Sq, Sk)][rope_half], row = position = sequence index) unimp_attn_bwd applies the transpose rotation to
   * the first 2 * rope_half dims of every dq and dk row on their way out -- the same arithmetic as
   * unimp_rope_halfsplit(inverse = 1) on the stored gradients, without that pass over HBM.  rope_half % 8 == 0,
   * 2 * rope_half <= D; needs 16-byte aligned dq / dk / dv views (UNIMP_ERR_UNSUPPORTED otherwise, and under kernel
   * generations other than 2). */
  const float* rope_cos; const float* rope_sin; int rope_half;
  /* rope_log2_base != 0 (and rope_cos / rope_sin NULL, rope_half set): the ADJACENT-pair layout of the GEMM's rotary epilogue
   * (unimp_gemm_desc.rope_*): dims 8g + j and 8g + 4 + j of every dq / dk row are rotated back at frequency 4g + j, cos / sin
   * computed in the epilogue -- no tables, no partner chunk. */
  float rope_log2_base;
  /* packed rows (ABI 4; training entry points only, NULL = the padded layout above): q_row_off int32 [B] -- the query-side
   * tensors (q, o, d_o, dq) are ONE [rows][H][D] buffer in which sequence b owns rows q_row_off[b] .. q_row_off[b] + q_len[b] - 1
   * (q_len int32 [B], required with q_row_off); the *_bs strides are ignored for them, Sq is the LONGEST sequence (it sizes the grid
   * and the per-row statistics lse / delta, which stay [B][H][Sq]), and seg, if set, is packed like q (seg[q_row_off[b] + i]).
   * k_row_off int32 [B]: the same for k, v, dk, dv with kv_len[b] rows each (kv_len required; a sequence with kv_len 0 must have
   * q_len 0).  Rows outside a sequence are neither read as data nor written.  collate_rec.py:38-74 right-pads every sequence to the
   * longest of the batch; this is the layout in which the language tower never computes the <PAD> rows.  Needs kernel
   * generation >= 2 in the backward (UNIMP_ERR_UNSUPPORTED otherwise); the decode entry points refuse it. */
  const int32_t* q_row_off; const int32_t* q_len; const int32_t* k_row_off;
  /* per-call switches (ABI 8).  UNIMP_ATTN_NO_PERSISTENT: unimp_attn_bwd keeps this launch off the persistent dK/dV kernel
   * (attention3.hip: one workgroup per CU) whatever unimp_attn_set_dkv3 says -- the caller runs beside collectives that hold CUs
   * (mmrec.py:215's gradient exchange overlapped with backward); same gradients from the first-generation kernel. */
  int flags;
} unimp_attn_desc;
#define UNIMP_ATTN_NO_PERSISTENT 1
int unimp_attn_fwd(const unimp_attn_desc* d, void* stream);
int unimp_attn_bwd(const unimp_attn_desc* d, void* stream);
/* Decode step of Flamingo.generate with a KV cache (eval_rec.py:100-110; eval_exp.py:103-113 and eval_img_gen.py:102-111 at
 * 256 / 600 new tokens): ONE query row per (batch row, head) -- d->Sq == 1, mask_mode NONE, kv_len[b] keys visible (NULL: Sk),
 * optional ALiBi slopes -- against the cached K / V [B][Sk = capacity][H][D].  The keys are split over `splits` workgroups per
 * (row, head); partial (max, sum, o[D]) go to `workspace` (fp32 [B*H*splits*(D+2)], caller-provided, may be NULL when splits
 * == 1) and are merged in a fixed order (no atomics: reproducible, graph-capturable; the grid depends on Sk only, never on
 * kv_len).  `splits` must be unimp_attn_decode_splits(B, H, Sk) (a fixed number of keys per workgroup: a row's bits do not depend
 * on the batch it is decoded in or on the cache capacity).  d->lse is not written.  HBM-bound: algorithmic bytes =
 * 2 * sum_b kv_len[b] * H * D * 2. */
int unimp_attn_decode(const unimp_attn_desc* d, float* workspace, int splits, void* stream);
/* Beam search form: rows [g * group, (g + 1) * group) are the beams of prompt g and hold IDENTICAL K / V for key positions below
 * shared_len[g] (int32 [B / group], device; the prompt: generate() repeats vision_x / the prompt per beam, eval_rec.py:100-110 with
 * num_beams = 10).  Those keys are read once per prompt (from the group's first row) for all its queries; each row's own keys from
 * shared_len[g] to kv_len[b] as above.  group <= 16; workspace B*H*2*splits*(D+2) floats.  group = 1 is unimp_attn_decode. */
int unimp_attn_decode_grouped(const unimp_attn_desc* d, float* workspace, int splits, int group, const int* shared_len, void* stream);
int unimp_attn_decode_splits(int B, int H, int Sk);
/* The whole self-attention of a cached decode step in ONE launch (ABI 8; transformers' GPTNeoXAttention / MPT attention with layer_past inside
 * Flamingo.generate -- eval_rec.py:100-110, eval_img_gen.py:102-111, eval_exp.py:103-113): what unimp_decode_rope_append followed by
 * unimp_attn_decode compute, bit for bit -- rotate the new token's q and k (row r with ITS table row; rot = 0: none), write the rotated k and v
 * to cache slot pos_idx[r], attend the keys [0, pos_idx[r]] (ALiBi slopes optional) -- in unimp_attn_decode's grid (the keys of a (row, head)
 * split over unimp_attn_decode_splits(rows, heads, capacity) workgroups); the last workgroup of a (row, head) to arrive merges the partials (fixed
 * order; the hand-over uses agent-scope accesses, no fence).  A decode step is bound by its launches, not by the K / V bytes.  qkv is NOT modified
 * (the rotated q / k exist in registers and in the cache only).  workspace: rows*heads*splits*(hd+2) floats; arrived: rows*heads uint32, ZERO
 * before the first launch (every launch leaves them zero).  All strides in elements. */
typedef struct {
  const void* qkv;                 /* bf16 [rows][heads][head_stride]: q / k / v at q_off / k_off / v_off inside a head slot */
  int64_t row_stride, head_stride;
  int q_off, k_off, v_off;
  int rows, heads, hd, rot;
  const float* cos_rows; const float* sin_rows;     /* fp32 [rows][rot / 2] */
  void* kcache; void* vcache;      /* bf16 [rows][capacity][heads][hd] views: strides below */
  int64_t c_row_stride, c_slot_stride, c_head_stride;
  int capacity;                    /* slots per row: sizes the launch (never the positions: graph-capturable) */
  const int64_t* pos_idx;          /* [rows], device: slot of the new token (< capacity) */
  float scale; const float* alibi_slopes;           /* fp32 [heads] or NULL */
  void* out;                       /* bf16 [rows][heads][hd] */
  int64_t o_row_stride, o_head_stride;
  float* workspace; void* arrived;
  int group_mode;                                   /* 0: by ADDRESS -- every row reads the keys below shared_len[g] from its group's FIRST row (identical
                                                     * copies: the bits of group = 1; the repeated lines come from L2 / the memory-side cache);
                                                     * 1: by prefix workgroups (below) */
  int group; const int32_t* shared_len;             /* beam search as in unimp_attn_decode_grouped (group <= 1: none): the keys below shared_len[g] are read
                                                     * once per prompt, in chunks of 32 keys by extra workgroups of the same launch (workspace:
                                                     * unimp_attn_decode_step_slots) */
} unimp_decode_step_desc;
int unimp_attn_decode_step(const unimp_decode_step_desc* d, void* stream);
int unimp_attn_decode_step_slots(int rows, int heads, int capacity, int group);      /* partial slots per (row, head): workspace = slots * rows*heads*(hd+2) floats */
/* tuning / test knob: which generation of attention kernels serves the calls above.  2 (default; env UNIMP_ATTN_GEN):
 * the 32x32x16-MFMA / LDS-DMA kernels of attention2.hip for the forward and dQ; dK/dV by attention3.hip (64 keys per wave, one
 * wave per SIMD) where it serves the form -- head dim 80, causal / no mask, Sq and Sk multiples of 32, padded rows, a (batch,
 * head) pair per CU -- else by
 * the first generation; 1: the first-generation kernels (kept for A/B measurements and run by the tests as a second
 * implementation of the same contract); 3: attention2.hip throughout; 4 (ABI 7): generation 2 without attention3.hip.
 * Returns the previous value. */
int unimp_attn_set_generation(int generation);
int unimp_attn_get_generation(void);
/* measurement knob (ABI 6; env UNIMP_ATTN_VIT): 1 (default) = the ViT forward (S = 257, no mask) seeds its online softmax with the 257th
 * key and walks four full key tiles; 0 = the general five-tile path.  Same contract, different rounding of one key's contribution
 * (bench.py's parity leg reports both).  Returns the previous value. */
int unimp_attn_set_vit_tail(int on);
/* measurement / test knob (ABI 7; env UNIMP_DKV3): 0 = generation 2 keeps the first-generation dK/dV kernel everywhere (what
 * generation 4 selects); 1 (default) = attention3.hip where it serves the form AND the launch has a (batch, head) pair per CU
 * (one persistent workgroup per CU: below that the first generation's many small workgroups are faster); 2 = wherever it serves
 * the form (the tests' small cases).  Returns the previous value. */
int unimp_attn_set_dkv3(int mode);
/* which dK/dV kernel the last unimp_attn_bwd of this process launched: 1 = first generation (attention.hip), 2 = attention2.hip,
 * 3 = attention3.hip; 0 = none yet (tests: the size threshold, and that a data-parallel group keeps the persistent kernel out) */
int unimp_attn_last_dkv(void);

/* ---- token embedding (gpt_neox.embed_in / OPT embed_tokens+embed_positions) -------------------------------
 * fwd: out[r] = W[ids[r]] (+ P[pos[r]]);  bwd: dW[ids[r]] += dout[r]  (fp32 atomics into dW32, then cast)
 */
int unimp_embedding_fwd(const int64_t* ids, const void* W, int64_t ldw, const int64_t* pos, const void* P, int64_t ldp,
                        void* out, int64_t ldo, int rows, int D, int vocab, void* stream);
int unimp_embedding_bwd(const int64_t* ids, const void* dout, int64_t lddo, float* dW32, int64_t lddw,
                        int rows, int D, int vocab, void* stream);
/* the same, bit-reproducible (ABI 8): the caller passes the ids SORTED (stable) and perm[t] = the row at sorted position t; the rows of an id are
 * summed in a fixed order (row order inside segments of 64 sorted positions, the segments of a run in order) by one writer per element, no
 * atomics.  (The atomic form's sums into a hot row -- pad, <image> -- depend on the order the adds land in: other last bits in 3 of 20 launches at
 * the cfg2 shape.)  scratch: unimp_embedding_bwd_sorted_scratch(rows, D) floats.  The package uses this one. */
int64_t unimp_embedding_bwd_sorted_scratch(int rows, int D);
int unimp_embedding_bwd_sorted(const int64_t* sorted_ids, const int64_t* perm, const void* dout, int64_t lddo, float* dW32, int64_t lddw,
                               float* scratch, int rows, int D, int vocab, void* stream);

/* ---- ViT input path: conv1(k=s=P, no bias) as im2col + GEMM, class token, position embedding ------------
 * (clip.py:60-84).  patchify: pixels [N,3,Hi,Wi] (fp32 or bf16) -> cols [N*g*g][ldc] bf16, k = c*P*P+py*P+px,
 * zero padded to ldc.  assemble: x[n][0] = cls + pos[0]; x[n][1+i] = patch[n*g*g+i] + pos[1+i].
 */
int unimp_vit_patchify(const void* pixels, int pixels_f32, void* cols, int64_t ldc, int N, int Hi, int Wi, int P,
                       void* stream);
int unimp_vit_assemble(const void* patch, int64_t ldp, const void* cls, const void* pos, void* x, int N, int n_patch,
                       int D, void* stream);

/* ---- elementwise helpers ----------------------------------------------------------------------------------*/
/* dst[r][0..D) = idx[r] >= 0 ? src[idx[r]][0..D) : 0 (bf16 rows, D % 8 == 0): packs the valid tokens of right-padded sequences into a
 * dense row range and unpacks them again (idx = the inverse map, -1 at the <PAD> positions of collate_rec.py:38-74's batches), so that
 * the row-wise kernels of the step skip the padding rows (Trainer(packed=True)). */
int unimp_gather_rows(const void* src, int64_t ld_src, const int32_t* idx, void* dst, int64_t ld_dst, int rows, int D, void* stream);
/* trace marker (measurement aid for mmrec.py:259-296's step timing): an empty kernel launched with `id` workgroups of 64 threads,
 * so a kernel trace can be cut to the region between two markers (tools/trace_window.py). */
int unimp_marker(int id, void* stream);
int unimp_add_bf16(const void* a, const void* b, void* out, int64_t n, void* stream);            /* out = a + b */
int unimp_cast_f32_to_bf16(const float* src, void* dst, int64_t n, float scale, void* stream);
int unimp_swiglu_fwd(const void* gate_up, int64_t ld, void* out, int64_t ldo, int rows, int F, void* stream);
int unimp_swiglu_bwd(const void* gate_up, int64_t ld, const void* dout, int64_t lddo, void* dgate_up, int64_t ldd,
                     int rows, int F, void* stream);
/* out[0] += sum(a*b) over n elements (fp32; out is fp32[1 + 1024]: out[0] zeroed by the caller, out[1..1025) scratch for the ordered
 * two-stage reduction -- no float atomics): d tanh-gate = dot(dy, y_pre_gate) */
int unimp_dot_bf16(const void* a, const void* b, int64_t n, float* out, void* stream);
/* beam search step (ABI 8; transformers GenerationMixin.beam_search inside Flamingo.generate: eval_rec.py:100-110 K = 10, eval_exp.py:103-113 K = 5):
 * out = top-C (sorted, descending; ties to the smaller index) of log_softmax(logits[r]) + beam_scores[r] over the K rows of every prompt viewed as one
 * [K * V] vector -- scores fp32 and flat indices r_in_group * V + token (int64), [rows / K][C] each.  logits bf16 (logits_f32 = 0) or fp32 [rows][ld];
 * beam_scores fp32 [rows]; scratch unimp_beam_topk_scratch(rows) bytes.  Two launches instead of a float copy, log_softmax, an add and a sort-based
 * top-k over K * V elements (233 -> about 40 us per token-step at K = 10, V = 74 053).  K <= 16, C <= 32. */
int64_t unimp_beam_topk_scratch(int rows);
int unimp_beam_topk(const void* logits, int logits_f32, int64_t ld, int rows, int V, int K, int C, const float* beam_scores, void* scratch,
                    float* out_scores, int64_t* out_idx, void* stream);
/* decode (ABI 8): read `bytes` at p and discard them -- launched on a second stream beside a weight-streaming GEMM it pulls the NEXT GEMM's weights
 * into the memory-side cache (256 MB Infinity Cache) while the launch gaps and tails of the chain leave HBM idle.  `blocks` workgroups of 256 threads
 * (<= 0: 256); sink: any 4 writable bytes (never written in practice) or NULL. */
int unimp_prefetch(const void* p, int64_t bytes, int blocks, void* sink, void* stream);
/* out[r] = src[r % period] (Perceiver latents repeat "n d -> b T n d") and its adjoint out[j] = sum_{r%period==j} src[r] (bf16) */
int unimp_bcast_rows(const void* src, void* out, int64_t ldo, int rows, int period, int D, void* stream);
int unimp_reduce_rows_periodic(const void* src, int64_t lds_, void* out, int rows, int period, int D, void* stream);

/* ---- MX-fp8 path for the frozen towers (reference: the "9b" model of mmrec.py:515-524, BASELINE config 5) ----
 * mx_quantize: bf16 x[rows][K] (row stride ldx elements) -> OCP e4m3 q[rows][K] (row stride ldq bytes) + one E8M0 scale byte per 32
 *   consecutive k: scales[rows][K/32] (row stride lds bytes).  MX rule: shared exponent floor(log2 amax) - 8, saturating cast.
 * gemm_mxfp8:  C[M][N] (bf16) = epi( sum_k A[m][k] B[n][k] ) with both operands quantised along k (k-contiguous rows), fp32
 *   accumulate on v_mfma_scale_f32_16x16x128_f8f6f4.  Epilogue: + bias[N], act (UNIMP_ACT_*; with `pre` != NULL the derivative
 *   act'(z) is written there), x aux[M][N], + res[M][N].  K % 128 == 0. */
typedef struct unimp_mx_gemm_desc {
  const void* A; const void* B;            /* e4m3 bytes, [M][K] and [N][K] */
  const void* scale_a; const void* scale_b; /* E8M0 bytes, [M][K/32] and [N][K/32] */
  void* C;
  const void* bias; const void* res; const void* aux; void* pre;
  int64_t lda, ldb, ldsa, ldsb, ldc, ldres, ldaux, ldpre;
  int32_t M, N, K, act;
  int32_t deriv_u8;                         /* != 0: the derivative act'(z) written to `pre` / read from `aux` is the uint8 form the bf16
                                             * GEMM descriptor selects with pre_deriv = 2 / dact = UNIMP_ACT_DERIV_U8: one byte per element,
                                             * ldpre / ldaux in bytes */
  int32_t reserved0;
  void* scale_c; int64_t ldsc;              /* scale_c != NULL: C leaves as an MX operand for the NEXT product -- e4m3 bytes [M][N] (ldc in bytes)
                                             * + E8M0 [M][N/32], quantised along N exactly as unimp_mx_quantize would quantise the bf16 result
                                             * (same bytes); N % 32 == 0.  The bf16 round trip and the quantiser pass disappear. */
} unimp_mx_gemm_desc;
int unimp_mx_quantize(const void* x, int64_t ldx, void* q, int64_t ldq, void* scales, int64_t lds, int rows, int K, void* stream);
int unimp_gemm_mxfp8(const unimp_mx_gemm_desc* d, void* stream);

/* ---- training-step host logic moved to the device ---------------------------------------------------------
 * label mask state machine (mmrec.py:143-168): labels[b][j] = keep ? ids : -100, one thread block per row.
 * media_time (open_flamingo MaskedCrossAttention text_time = cumsum(ids == media_id)) int32 [B][L].
 */
int unimp_label_mask(const int64_t* ids, int64_t* labels, int32_t* media_time, int B, int L, int64_t answer_id,
                     int64_t eoc_id, int64_t pad_id, int64_t media_id, void* stream);

/* ---- weighted focal cross-entropy (mmrec.py:190-213) -------------------------------------------------------
 * logits bf16 [B][L][ldv] (V valid columns); labels int64 [B][L] (un-shifted; row (b,j) is scored against
 * labels[b][j+1], j < L-1); weights fp32 [B].
 * fwd: per-row stats (lse, z_y) + out[0] = sum_rows w*ce*(1-pt)^gamma, out[1] = #valid, out[2] = sum ce (HF mean CE numerator).
 * bwd: dlogits[b][j][k] = gscale/out[1] * w_b * (p_k - [k==y]) * coef  (focal term NOT detached); rows without a
 *      label and pad columns [V,ldv) are written as zeros.  dlogits may alias logits.
 */
int unimp_focal_ce_fwd(const void* logits, int64_t ldv, const int64_t* labels, const float* weights, float gamma,
                       int use_reweight, float* row_lse, float* row_zy, float* out3, int B, int L, int V, void* stream);
int unimp_focal_ce_bwd(const void* logits, int64_t ldv, const int64_t* labels, const float* weights, float gamma,
                       int use_reweight, const float* row_lse, const float* row_zy, const float* out3,
                       const float* gscale, void* dlogits, int B, int L, int V, void* stream);
/* bwd_rows: the same gradient for a list of scored positions rows[i] = b*L + j (int64, n_rows entries), written compactly:
 *      dl[i][0..ldd) = dlogits[rows[i]][0..ldd), V <= ldd <= ldv.  The reference (mmrec.py:190-215) materialises the dense
 *      [B*L][V] gradient, which is zero on every row without a label (~98 % of them); the head's dX / dW GEMMs then run on
 *      the listed rows only -- same gradients, no zero-row products. */
int unimp_focal_ce_bwd_rows(const void* logits, int64_t ldv, const int64_t* labels, const float* weights, float gamma,
                            int use_reweight, const float* row_lse, const float* row_zy, const float* out3,
                            const float* gscale, const int64_t* rows, int n_rows, void* dl, int64_t ldd, int L, int V,
                            void* stream);

/* ---- optimizer (mmrec.py:247-256,609-631,671): global-norm clip + AdamW, flat buffers ---------------------
 * sumsq: out[0] += sum(g^2) (fp32; caller zeroes out[0]; out[1..1025) is scratch for the ordered reduction).  adamw: for i in [0,n): g = grad[i]*gscale*clip,
 * clip = min(1, max_norm/(sqrt(sumsq[0])*gscale + 1e-6)); decoupled weight decay `wd` for i < n_decay, 0 after;
 * fp32 master/m/v updated, bf16 shadow param rewritten, grad zeroed when zero_grad != 0.  step is 1-based.
 */
int unimp_sumsq_bf16(const void* g, int64_t n, float* out, void* stream);
int unimp_adamw_flat(float* master, float* m, float* v, void* param_bf16, void* grad_bf16, int64_t n, int64_t n_decay,
                     float lr, float beta1, float beta2, float eps, float wd, int step, const float* sumsq,
                     float gscale, float max_norm, int zero_grad, void* stream);

/* ---- image preprocessing ----------------------------------------------------------------
 * replaces `patch_resize_transform` (UniMP/pipeline/mm_utils/rec_dataset.py:91-107): RandomResize([(S, S)]) =
 * transforms.py:102-136 `F.resize(image, (S, S), interpolation=Image.BICUBIC)` on a PIL image (= Pillow Image.resize,
 * 8-bit two-pass fixed-point resampler), ToTensor (uint8 / 255 in fp32), Normalize(FLAMINGO_MEAN, FLAMINGO_STD)
 * (rec_dataset.py:30-31), then the cast of mmrec.py:135-141.  JPEG decoding stays on the host.
 * src: all decoded RGB images of the batch packed as [H][W][3] uint8 (device); descs: one per image (device);
 * tables: int32 tap tables (device), one row of (2 + ks) ints per output coordinate: first source index, tap count,
 * taps in 22-bit fixed point exactly as Pillow computes them (the host side builds them: unimp_amd/data.py);
 * tmp: scratch for the horizontal pass ([H][out_w][3] per image at tmp_off); mean / std: HOST arrays of 3 floats.
 * out: [n][3][out_h][out_w] bf16 (or f32 when out_f32); out_u8 (optional, may be NULL): the resized bytes
 * [n][out_h][out_w][3] -- bit-exact with Pillow.  max_h: largest H in the batch (grid sizing only).
 */
typedef struct unimp_image_desc {
  int64_t src_off;          /* byte offset of the image in src */
  int64_t H, W;
  int64_t kx_off, ky_off;   /* int32 element offsets of the per-axis tap tables in `tables` */
  int64_t ksx, ksy;         /* taps per row of the table; 0 = this axis already has the output size (no resampling) */
  int64_t tmp_off;          /* byte offset of this image's horizontal-pass result in tmp */
} unimp_image_desc;
int unimp_image_resize_normalize(const uint8_t* src, const unimp_image_desc* descs, int n_images, int max_h,
                                 const int32_t* tables, uint8_t* tmp, int out_h, int out_w, const float* mean,
                                 const float* std, void* out, int out_f32, uint8_t* out_u8, void* stream);

#ifdef __cplusplus
}
#endif
#endif
