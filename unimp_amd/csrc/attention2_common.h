// Helpers shared by the second-generation attention kernels (attention2.hip, attention3.hip): LDS-DMA statements, the XCD-aware
// block decode, the image pitches and the transposed-accumulator epilogue.
#pragma once
#include <stdlib.h>
#include "common.h"
#include "unimp_hip.h"
#include "attention_params.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#define EXP2(x) __builtin_amdgcn_exp2f(x)
__device__ __forceinline__ float a2_max3(float a, float b, float c) {
  float d;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f

// one LDS-DMA wave-instruction: 64 lanes x 16 B, lane i lands at lds_dst + 16 i; saddr form (scalar 64-bit base + per-lane
// 32-bit byte offset).  M0 carries the LDS destination.
__device__ __forceinline__ void a2_glds(const void* sbase, uint32_t voff, uint32_t lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
               :: "v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
}

// the same with a full 64-bit address per lane (sources that do not share a base)
__device__ __forceinline__ void a2_glds_v(const void* vaddr, uint32_t lds_dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
               :: "v"(vaddr), "s"(lds_dst) : "memory", "m0");
}

// The compiler counts only ITS OWN loads in vmcnt: a fragment loaded from global before the tile loop would get its wait at
// the first use INSIDE the loop, executed every iteration, where it also waits for the hand-issued LDS-DMA of the next tile
// (measured: the Q K^T MFMAs sat behind `s_waitcnt vmcnt(4..0)` every tile).  Passing the value through an empty asm
// statement before the loop makes the compiler retire its load there; nothing it knows of is outstanding afterwards.
template <typename T> __device__ __forceinline__ void a2_pin(T& v) { asm volatile("" : "+v"(v)); }

// 1-D launch, XCD-aware: hardware deals consecutive workgroup ids round-robin over the 8 XCDs (each with its own L2), so the
// blocks of one (batch, head) -- which read the same K / V (or Q / dO) tiles -- would meet in 4 different L2s.  xcd_remap
// gives the workgroups of one XCD a contiguous range of logical ids; x (the block inside the (b, h) pair) varies fastest in
// the logical id, so a pair's blocks run on ONE XCD, back to back, and all but the first read of a tile is an L2 hit.
// Heaviest block first inside a pair (causal: the last query block sees the most key tiles).
__device__ __forceinline__ void a2_decode(int nx, int H, int B, int& x, int& h, int& b) {
  int id = xcd_remap(blockIdx.x, nx * H * B);
  x = nx - 1 - id % nx;
  int t = id / nx;
  h = t % H; b = t / H;
}

template <int D> struct A2Cfg {
  static constexpr int CPR = D / 8;                          // 16-byte chunks per row
  static constexpr int PK = CPR + 1 + (CPR & 1);             // odd pitch (slots) of the K image: 9 / 11 / 17
  static constexpr int PV = D == 128 ? 20 : 12;              // V image pitch: (4 PV) % 64 in {16, 48}
  static constexpr int KS = D / 16;                          // k-steps of the Q K^T contraction
  static constexpr int ND = (D + 31) / 32;                   // 32-row blocks of O^T
  static constexpr int NI = PK + PV;                         // DMA wave-instructions per 64-key tile
  static constexpr int STAGE = 64 * (PK + PV) * 16;
};

// Epilogue: a wave's transposed accumulator tile X^T[d][row] (lane: row = l & 31, d = 32 nd + 8 g + 4 hi5 + e) -> 32 global
// rows of D bf16 each.  Storing from the accumulator layout is 12-16 8-byte stores per lane, each touching 32 different rows
// (measured: the store tail of the dK/dV kernel cost 80 us of its 190 us fixed cost); through a wave-private LDS region the
// tile goes out as whole 16-byte chunks of consecutive row segments (D = 80: five instructions per lane).  The caller must
// have passed a barrier after the last LDS read of the tile loop; the region is [32][D * 2 + 16] bytes per wave.
template <int D, int ND>
__device__ __forceinline__ void a2_store_rows(char* lds_wave, const f32x16 (&acc)[ND], float mul, bf16* __restrict__ gbase, long row_stride,
                                              int row0, int nrows, const float* rope_cos = nullptr, const float* rope_sin = nullptr,
                                              int rope_half = 0, float rope_step = 0.f, int lane = -1) {
  constexpr int PITCH = D * 2 + 16, CPR = D / 8;
  // lane: the caller's (possibly opaque) copy of the lane id -- a persistent kernel passes one it re-made opaque inside its item loop, so
  // that hipcc does not hoist this function's per-lane constants out of that loop and spill them (attention3.hip)
  const int l = lane >= 0 ? lane : lane_id(), hi5 = l >> 5, rl = l & 31;
#pragma unroll
  for (int nd = 0; nd < ND; ++nd)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      int d0 = 32 * nd + 8 * g + 4 * hi5;
      if (d0 < D) {
        bf16x4 w = {f2bf(acc[nd][4 * g] * mul), f2bf(acc[nd][4 * g + 1] * mul), f2bf(acc[nd][4 * g + 2] * mul), f2bf(acc[nd][4 * g + 3] * mul)};
        *(bf16x4*)(lds_wave + rl * PITCH + d0 * 2) = w;
      }
    }
  __builtin_amdgcn_s_waitcnt(0xc07f);                         // lgkmcnt(0): the wave's own LDS writes have landed (wave-private region)
  __builtin_amdgcn_wave_barrier();
  constexpr int NCH = (32 * CPR + 63) / 64;
  if (rope_step != 0.f) {                                      // adjacent-pair layout: no tables, no partner chunk
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      int id = l + 64 * i;
      int r = id / CPR, c = id - r * CPR;
      if (id < 32 * CPR && row0 + r < nrows)
        *(u32x4*)(gbase + (long)(row0 + r) * row_stride + c * 8) = attn_rope_inv_adjacent(lds_wave + r * PITCH, c, rope_half, (float)(row0 + r), rope_step);
    }
    return;
  }
  if (rope_cos) {
    AttnRopeChunk ch[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {                            // branch-free: a lane without a chunk / row re-reads a valid one
      int id = min(l + 64 * i, 32 * CPR - 1);
      int r = id / CPR, c = id - r * CPR;
      long pos = min(row0 + r, nrows - 1);
      attn_rope_inv_load(ch[i], lds_wave + r * PITCH, c, rope_half, rope_cos + pos * rope_half, rope_sin + pos * rope_half);
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      int id = l + 64 * i;
      int r = id / CPR, c = id - r * CPR;
      u32x4 v = attn_rope_inv_apply(ch[i]);
      if (id < 32 * CPR && row0 + r < nrows) *(u32x4*)(gbase + (long)(row0 + r) * row_stride + c * 8) = v;
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    int id = l + 64 * i;
    int r = id / CPR, c = id - r * CPR;
    if (id < 32 * CPR && row0 + r < nrows) {
      u32x4 v = *(const u32x4*)(lds_wave + r * PITCH + c * 16);
      *(u32x4*)(gbase + (long)(row0 + r) * row_stride + c * 8) = v;
    }
  }
}

// key range [lo, hi) attended by query row `qr` of batch b
__device__ __forceinline__ void a2_key_range(const AttnP& p, int b, int qr, int& lo, int& hi) {
  lo = 0; hi = 0;
  if (qr >= p.Sq) return;
  int kvl = p.kv_len ? p.kv_len[b] : p.Sk;
  if (p.mask_mode == UNIMP_MASK_NONE) { hi = kvl; }
  else if (p.mask_mode == UNIMP_MASK_CAUSAL) { hi = min(qr + 1, kvl); }
  else { int t = p.seg[(long)b * p.SqS + qr]; if (t > 0) { lo = (t - 1) * p.seg_len; hi = min(t * p.seg_len, p.Sk); } }
}

// the same with the batch row's key count already in a register (no load inside a tile loop)
__device__ __forceinline__ void a2_key_range_kvl(const AttnP& p, int b, int qr, int kvl, int& lo, int& hi) {
  lo = 0; hi = 0;
  if (qr >= p.Sq) return;
  if (p.mask_mode == UNIMP_MASK_NONE) { hi = kvl; }
  else if (p.mask_mode == UNIMP_MASK_CAUSAL) { hi = min(qr + 1, kvl); }
  else { int t = p.seg[(long)b * p.SqS + qr]; if (t > 0) { lo = (t - 1) * p.seg_len; hi = min(t * p.seg_len, p.Sk); } }
}

