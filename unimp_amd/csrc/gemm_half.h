// Shared by the kernels that eat K in 32-wide half-stages staged by LDS-DMA (gemm3.hip, gemm4.hip): LDS images,
// fragment reads, DMA issue.
#pragma once
#include "gemm_tile.h"

__device__ __forceinline__ int kc32_off(int row, int chunk) { return row * 64 + ((chunk ^ (((row >> 3) & 1) * 3)) << 4); }

__device__ __forceinline__ bf16x8 frag_kc32(const char* tile, int r0) {
  int l = lane_id();
  return *(const bf16x8*)(tile + kc32_off(r0 + (l & 15), l >> 4));
}
template <int ROWS>
__device__ __forceinline__ int ks32_off(int krow, int col) {
  return krow * (ROWS * 2) + ((((col >> 4) ^ ks_h(krow)) << 5) | ((col & 15) << 1));
}
// Transposed fragment of a k-strided image, issued as INLINE ASM: with the builtin, hipcc puts `s_waitcnt vmcnt(0)` in
// front of every ds_read_b64_tr_b16 while an LDS-DMA is in flight (it cannot tell the DMA's LDS destination from the
// read), which drains the prefetch ring every half-step.  The two 64-bit halves stay separate values until the C phase
// (after the explicit lgkmcnt(0) + barrier of G3_BARRIER), so the compiler never touches the destination registers
// before the data has landed (cdna guide §5.7 item 1, form iii).
typedef __attribute__((ext_vector_type(8))) short s16x8;
// per-lane LDS byte offset of fragment 0 of a wave's column range starting at c0 (multiple of 64, or of 32 for WN = 32):
// fragment i (columns c0 + 16 i ..) is then at  (lane_base + buffer_base) ^ (i << 5)  because the granule XOR only
// touches bits 5..7 -- ONE live address register per operand instead of one per fragment.
template <int ROWS>
__device__ __forceinline__ uint32_t ks32_lane_base(int c0) {
  int l = lane_id();
  int g = l >> 4, q = (l >> 2) & 3, p = l & 3;
  return (uint32_t)ks32_off<ROWS>(g * 8 + q, c0 + 4 * p);
}
template <int ROWS>
__device__ __forceinline__ void frag_ks32_asm(uint32_t a0, int i, s16x4& lo, s16x4& hi) {
  uint32_t addr = a0 ^ ((uint32_t)i << 5);
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(addr) : "memory");
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "n"(4 * ROWS * 2) : "memory");
}
__device__ __forceinline__ bf16x8 join_halves(s16x4 lo, s16x4 hi) {
  s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

// LDS-DMA in the saddr form: wave-uniform 64-bit base in SGPRs + per-lane 32-bit byte offset.  The builtin only emits
// the 64-bit-VGPR-address form, which costs two 64-bit VALU adds per instruction and half-stage (per-lane base + the
// half-stage's uniform offset) right where the matrix pipe waits for them; here the half-stage offset is scalar math and
// the per-lane offsets are loop-invariant registers.  M0 = LDS destination of lane 0 (1 wait state after writing M0).
__device__ __forceinline__ void glds16_s(const void* sbase, uint32_t voff, uint32_t lds_dst) {
#ifdef GLDS_BUILTIN
  glds16((const char*)sbase + voff, (char*)(uintptr_t)lds_dst);
#else
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
               :: "v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
#endif
}
__device__ __forceinline__ uint32_t lds_addr(const char* p) { return (uint32_t)(uintptr_t)LDS_PTR(const char, p); }

// LDS-DMA of one operand's [ROWS x 32] half-stage by NW waves: ROWS / (16 NW) wave-instructions per wave.
// dma_setup() computes, once, each lane's byte offset of its chunk at half-step 0.  Rows / columns beyond the matrix are
// CLAMPED to the last valid chunk instead of predicated: they only feed output rows / columns that are never stored, so
// the steady-state issue is  `scalar base of the half-step + per-lane constant`  with no VALU work and no branches.
// Only a ragged K tail (K % 32 != 0) takes the predicated path, where chunks with k >= K must read as zeros.
template <bool KS, int ROWS, int NW = 8>
__device__ __forceinline__ void dma_setup(long ld, int r0, int R, int wave, uint32_t (&off)[ROWS / (16 * NW)]) {
  constexpr int NI = ROWS / (16 * NW);
  int l = lane_id();
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    int slot0 = (wave * NI + i) * 64;
    if (!KS) {
      int row = (slot0 >> 2) + (l >> 2), s = l & 3;
      int c = s ^ (((row >> 3) & 1) * 3);
      int gr = min(r0 + row, R - 1);
      off[i] = (uint32_t)(((long)gr * ld + c * 8) * 2);
    } else {
      constexpr int SPR = ROWS / 8;
      int krow = (slot0 + l) / SPR, s = (slot0 + l) % SPR;
      int col = (((s >> 1) ^ ks_h(krow)) << 4) | ((s & 1) << 3);
      int gr = min(r0 + col, ((R + 7) & ~7) - 8);
      off[i] = (uint32_t)(((long)krow * ld + gr) * 2);
    }
  }
}
template <bool KS, int ROWS, int NW = 8>
__device__ __forceinline__ void dma_issue(const bf16* __restrict__ X, long ld, int h, int K, char* img, int wave,
                                          const uint32_t (&off)[ROWS / (16 * NW)]) {
  constexpr int NI = ROWS / (16 * NW);
  const char* ub = (const char*)X + (KS ? (long)h * 32 * ld * 2 : (long)h * 64);       // wave-uniform
  if (h * 32 + 32 <= K) {
    uint32_t dst = __builtin_amdgcn_readfirstlane(lds_addr(img) + wave * NI * 1024);
#pragma unroll
    for (int i = 0; i < NI; ++i) glds16_s(ub, off[i], dst + i * 1024);
  } else {                                                                                // ragged K tail
    asm volatile("" : "+s"(ub));      // keep this path's 64-bit address math inside the branch (LICM would hoist it into every half-stage)
    int l = lane_id();
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      int slot0 = (wave * NI + i) * 64;
      int gk;
      if (!KS) { int row = (slot0 >> 2) + (l >> 2); gk = h * 32 + (((l & 3) ^ (((row >> 3) & 1) * 3)) << 3); }
      else gk = h * 32 + (slot0 + l) / (ROWS / 8);
      const char* src = gk < K ? ub + off[i] : (const char*)g_zero16;
      glds16(src, img + slot0 * 16);
    }
  }
}


// dma_issue without its ragged-k branch: the caller guarantees a full 32-k half-stage (K % 32 == 0 up to and including half-step h)
template <bool KS, int ROWS, int NW = 8>
__device__ __forceinline__ void dma_full(const bf16* __restrict__ X, long ld, int h, char* img, int wave, const uint32_t (&off)[ROWS / (16 * NW)]) {
  constexpr int NI = ROWS / (16 * NW);
  const char* ub = (const char*)X + (KS ? (long)h * 32 * ld * 2 : (long)h * 64);       // wave-uniform
  uint32_t dst = __builtin_amdgcn_readfirstlane(lds_addr(img) + wave * NI * 1024);
#pragma unroll
  for (int i = 0; i < NI; ++i) glds16_s(ub, off[i], dst + i * 1024);
}

// one wave-instruction of dma_issue's steady-state path (full 32-k half-stage), so a kernel can spread a half-stage's
// DMA between its MFMAs
template <bool KS, int ROWS, int NW>
__device__ __forceinline__ void dma_one(const bf16* __restrict__ X, long ld, int h, char* img, int wave,
                                        const uint32_t (&off)[ROWS / (16 * NW)], int i) {
  constexpr int NI = ROWS / (16 * NW);
  const char* ub = (const char*)X + (KS ? (long)h * 32 * ld * 2 : (long)h * 64);
  glds16_s(ub, off[i], __builtin_amdgcn_readfirstlane(lds_addr(img) + (wave * NI + i) * 1024));
}
