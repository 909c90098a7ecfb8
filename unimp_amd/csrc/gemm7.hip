// One-wave-per-SIMD bf16 MFMA GEMM with a hand-ordered, software-pipelined main loop (round 5): 256 x 256 tiles, 256 threads = 4 waves,
// each wave owns 128 x 128 of the tile (64 accumulator tiles in 256 AGPRs), K eaten in 64-k stages of whole 128-byte rows.
//
// What is different from gemm4.hip (same blocking, compiler-scheduled, 32-k half-stages, one barrier + one vmcnt per 64 MFMAs):
//   * EVERY instruction of the loop body is an `asm volatile` statement, so the source order IS the issue order: no ds_read, LDS-DMA or
//     wait ever sits between two MFMAs unless it was put there -- at most one rider per MFMA gap, the next MFMA is already queued when
//     the matrix pipe frees up (MI355X_MICROARCH "vector-instruction ISSUE cost": a 16x16x32 MFMA holds the issue port 8 of its 16 cycles).
//   * two fragment register sets (F0 = k 0..31, F1 = k 32..63 of the stage): while the 64 MFMAs of one k-half run, the other set is
//     refilled -- a wave never waits on its own ds_reads, and there is no SIMD partner to cover for it.
//   * two 64-k stages in LDS (2 x 64 KiB) with the DMA two tiles ahead: stage s is refilled for tile t+2 as soon as the last fragment
//     read of tile t has returned (one barrier), tile t+1 is waited for with a counted vmcnt in the middle of the second k-half (second
//     barrier).  Two barriers and two counted waits per 128 MFMAs; the data a wait covers was requested ~120 MFMAs (~2000 cycles) earlier.
//   * A and a k-contiguous B are staged in whole 128-byte rows (gemm3.hip G3_AFULL's image), a k-strided B in whole 512-byte k-rows.
//
// Per stage and wave: 128 MFMAs, 16 LDS-DMA, 16 ds_read_b128 + 32 ds_read_b64_tr_b16 (k-strided B) or 32 ds_read_b128, 2 barriers.
// Schedule of one iteration (tile t in stage s; F0 holds its k-half 0, read during the previous iteration):
//   block 0 (64 MFMAs on F0): reads of k-half 1 -> F1 in the even gaps 0..46; lgkmcnt(0) + barrier at gap 54 (stage s is now free);
//                             DMA of tile t+2 -> stage s from gap 55 on, one per two gaps
//   block 1 (64 MFMAs on F1): the rest of the DMA; vmcnt(N) + barrier at gap 12 (tile t+1 has landed everywhere); reads of its k-half 0
//                             -> F0 in the gaps 13..59; lgkmcnt(0) after the last MFMA
// Same k grouping inside every MFMA and the same k order per accumulator as every other variant: bit-identical results.
// Serves: k-contiguous A, B k-contiguous or k-strided (not packed), K % 64 == 0, K >= 128; every epilogue kind of gemm_tile.h.
#include <stdlib.h>
#include <type_traits>
#include "gemm_half.h"

// measurement builds of this file under their own symbols (Makefile: gemm7n.o = -DG7_POLICY=1 -DG7_TAG=n, gemm7s.o = sc1)
#ifdef G7_TAG
#define G7_CAT_(a, b, c) a##b##c
#define G7_CAT(a, b, c) G7_CAT_(a, b, c)
#define gemm7_bf16_kernel G7_CAT(gemm7, G7_TAG, _bf16_kernel)
#define unimp_gemm7_launch G7_CAT(unimp_gemm7, G7_TAG, _launch)
#define launch7 G7_CAT(launch7, G7_TAG, _)
#define g7_stamps G7_CAT(g7, G7_TAG, _stamps)
#define unimp_debug_g7_stamps G7_CAT(unimp_debug_g7, G7_TAG, _stamps)
#endif
#define G7_BM 256
#define G7_BN 256
#define G7_STG 65536          // one stage: A [256][64] + B [256][64] bf16
#define G7_ASUB 32768

#ifdef G7_STAMP       // debug build: s_memtime around the main loop of every block (tools/stamp_gemm7.py)
__device__ unsigned long long g7_stamps[8192 * 8];      // per block: s_memtime (shader cycles) at 4 points, s_memrealtime (100 MHz) at the same 4
extern "C" int unimp_debug_g7_stamps(void* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g7_stamps), sizeof(g7_stamps)); }
#define G7_T(K_) do { if (threadIdx.x == 0 && blockIdx.x < 8192) { unsigned long long t_, r_;                                      \
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_), "=s"(r_) :: "memory");                      \
    g7_stamps[blockIdx.x * 8 + (K_)] = t_; g7_stamps[blockIdx.x * 8 + 4 + (K_)] = r_; } } while (0)
#else
#define G7_T(K_) do {} while (0)
#endif

__device__ __forceinline__ void g7_mfma(f32x4& c, bf16x8 b, bf16x8 a) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(b), "v"(a));
}
template <int OFF>
__device__ __forceinline__ void g7_read128(bf16x8& out, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(out) : "v"(addr), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void g7_read_tr(s16x4& out, uint32_t addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(out) : "v"(addr), "n"(OFF) : "memory");
}
// cache policy of the staging loads (measurement knob, compile time): 0 default, 1 nt, 2 sc1, 3 sc0
#ifndef G7_POLICY
#define G7_POLICY 0
#endif
__device__ __forceinline__ void g7_dma(const void* sbase, uint32_t voff, uint32_t lds_dst) {
#if G7_POLICY == 1
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" :: "v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
#elif G7_POLICY == 2
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 sc1" :: "v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
#elif G7_POLICY == 3
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 sc0" :: "v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
#else
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
#endif
}
// schedule 2: an MFMA and the LDS-DMA that rides in its gap as ONE statement -- M0 (the DMA's LDS destination = wave base + immediate) is
// written before the MFMA, which then is the wait state the DMA needs after an M0 write: one SALU + one VMEM per DMA, like a ds_read gap
template <int IMM>
__device__ __forceinline__ void g7_mfma_dma(f32x4& c, bf16x8 b, bf16x8 a, uint32_t m0base, const void* sbase, uint32_t voff) {
  asm volatile("s_add_u32 m0, %3, %4\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\tglobal_load_lds_dwordx4 %5, %6"
               : "+a"(c) : "v"(b), "v"(a), "s"(m0base), "n"(IMM), "v"(voff), "s"(sbase) : "memory", "m0", "scc");
}
__device__ __forceinline__ void g7_xor_stage(uint32_t& x) { asm volatile("v_xor_b32 %0, 0x10000, %0" : "+v"(x)); }
#ifndef G7_SCHED
#define G7_SCHED 2
#endif
#ifndef G7_PF
#define G7_PF 0
#endif
#define G7_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define G7_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory")
#define G7_BAR() asm volatile("s_barrier" ::: "memory")

// EPI: as gemm3.hip (-1 = kind chosen per tile at run time; EK_* = that kind only)
// AKS: the A operand k-strided too (X[k * lda + m]: the weight-gradient form dy^T . x, both operands [tokens][features]) -- staged and
// read exactly like a k-strided B (two 32-k half-stage images per stage, ds_read_b64_tr_b16 fragments); served together with BKS only
template <bool AKS, bool BKS, int EPI>
__global__ __launch_bounds__(256, 1) void gemm7_bf16_kernel(Gemm2Params p) {
  static_assert(!AKS || BKS, "a k-strided A comes with a k-strided B (the dW form)");
  constexpr bool ROPE = EPI == EK_ROPE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  G7_T(0);
  const int nwg = p.nbm * p.nbn;
  const int id = xcd_remap(blockIdx.x, nwg);
  const int GM = p.gm > 0 ? p.gm : 4;
  const int per_group = GM * p.nbn;
  const int grp_ = id / per_group;
  const int first_m = grp_ * GM;
  const int gsz = min(p.nbm - first_m, GM);
  const int in_g = id - grp_ * per_group;
  const int tm = first_m + in_g % gsz, tn = in_g / gsz;
  const int m0 = tm * G7_BM, n0 = tn * G7_BN;

  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = lane_id();
  const int wm = wave >> 1, wn = wave & 1;

  f32x4 acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 fa0[8], fa1[8], fb0[8], fb1[8];            // fragments of k-half 0 / 1; fb*: k-contiguous B
  s16x4 bl0[8], bh0[8], bl1[8], bh1[8];             // k-strided B: the two transposed halves of a fragment
  s16x4 al0[8], ah0[8], al1[8], ah1[8];             // k-strided A likewise

  // ---- LDS-DMA source offsets (bytes, per lane; loop-invariant) and the wave's destination bases
  uint32_t aoff[8], boff[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int P = (wave * 8 + i) * 64 + lane, row = P >> 3, c = (P & 7) ^ ((row >> 1) & 7);
    if (!AKS) aoff[i] = (uint32_t)(((long)min(m0 + row, p.M - 1) * p.lda + c * 8) * 2);
    if (!BKS) boff[i] = (uint32_t)(((long)min(n0 + row, p.N - 1) * p.ldb + c * 8) * 2);
  }
  uint32_t boffs[4], aoffs[4];
  if (BKS) dma_setup<true, G7_BN, 4>(p.ldb, n0, p.N, wave, boffs);
  if (AKS) dma_setup<true, G7_BM, 4>(p.lda, m0, p.M, wave, aoffs);
  const uint32_t smem_lds = lds_addr(smem);
  const uint32_t dA = __builtin_amdgcn_readfirstlane(smem_lds + (AKS ? wave * 4096 : wave * 8192));  // + stage * G7_STG + i * 1024 (k-strided: + kh * 16384)
  const uint32_t dB = __builtin_amdgcn_readfirstlane(smem_lds + G7_ASUB + (BKS ? wave * 4096 : wave * 8192));   // k-strided: + kh * 16384 + i * 1024
  // one LDS-DMA instruction of tile T (stage S_): I = 0..7 A, 8..15 B
#define G7_DMA(T, S_, I) do { constexpr int I_ = (I) & 15, J_ = (I_ - 8) & 7;      /* masked: dead branches may name any gap */           \
    if (I_ < 8 && AKS) g7_dma((const char*)p.A + ((long)(2 * (T) + ((I_ & 7) >> 2)) * 32 * p.lda) * 2, aoffs[I_ & 3],             \
                              dA + (S_) * G7_STG + ((I_ & 7) >> 2) * 16384 + (I_ & 3) * 1024);                                  \
    else if (I_ < 8) g7_dma((const char*)p.A + (long)(T) * 128, aoff[I_ & 7], dA + (S_) * G7_STG + I_ * 1024);                 \
    else if (!BKS) g7_dma((const char*)p.B + (long)(T) * 128, boff[J_], dB + (S_) * G7_STG + J_ * 1024);                        \
    else g7_dma((const char*)p.B + ((long)(2 * (T) + (J_ >> 2)) * 32 * p.ldb) * 2, boffs[J_ & 3],                               \
                dB + (S_) * G7_STG + (J_ >> 2) * 16384 + (J_ & 3) * 1024); } while (0)

  // ---- fragment read addresses (stage 0; the stage bit is toggled per iteration)
  uint32_t va0 = smem_lds + kc_off(wm * 128 + (lane & 15), lane >> 4), va1 = va0 ^ 64;
  uint32_t vb0 = 0, vb1 = 0, vbs[8], vas[8];
  if (AKS) {
    const uint32_t la = ks32_lane_base<G7_BM>(wm * 128);
#pragma unroll
    for (int i = 0; i < 8; ++i) vas[i] = smem_lds + (la ^ ((uint32_t)i << 5));
  }
  if (!BKS) { vb0 = smem_lds + G7_ASUB + kc_off(wn * 128 + (lane & 15), lane >> 4); vb1 = vb0 ^ 64; }
  else {
    const uint32_t lb = ks32_lane_base<G7_BN>(wn * 128);
#pragma unroll
    for (int j = 0; j < 8; ++j) vbs[j] = smem_lds + G7_ASUB + (lb ^ ((uint32_t)j << 5));
  }
#if G7_PF > 0
  // ---- L2 prefetch (measurement knob G7_PF = lead in stages beyond the DMA's two): the 8 workgroups that share an A panel (same tile row,
  // consecutive in the XCD's raster) and the 4 that share a B panel run the same k at about the same time, so a line's first request is a
  // fabric miss that all of them wait on.  Here each workgroup touches ITS SHARE of the two panels' lines G7_PF stages before anybody's
  // DMA asks for them (one byte per 128-byte line, ONE instruction per wave and stage, result never read): 1/8 of the A stage (32 rows)
  // and 1/4 of the B stage (64 lines), split over the 4 waves -- lanes 0..7 A rows, lanes 8..23 B lines, the rest repeat lane 0.
  uint64_t pfaddr; uint32_t pfstride; uint32_t pfdummy = 0;
  {
    const int ga = tn & 7, gb = tm & 3;
    // opaque scalar copies of everything uniform that goes into the per-lane address: the selects below must not become THE values the DMA
    // statements' "s" operands are built from (hipcc then hands those a VGPR pair: "invalid operand for instruction")
    uint64_t bA = (uint64_t)(uintptr_t)p.A, bB = (uint64_t)(uintptr_t)p.B;
    uint32_t ldb2 = (uint32_t)(p.ldb * 2), lda2 = (uint32_t)(p.lda * 2);
    asm volatile("" : "+s"(bA), "+s"(bB), "+s"(ldb2), "+s"(lda2));
    const bool isB = lane >= 8 && lane < 24;
    const int li = 64 * gb + 16 * wave + ((lane - 8) & 15);                      // B line of this lane (0 .. 255)
    const int rowA = min(m0 + 32 * ga + 8 * wave + (lane & 7), p.M - 1);
    const uint64_t offA = (uint64_t)rowA * lda2 + 256;                             // tile 2 is the first one nobody has requested yet
    const uint64_t offB = BKS ? (uint64_t)(li >> 2) * ldb2 + (uint64_t)min(n0 + (li & 3) * 64, ((p.N + 7) & ~7) - 8) * 2 + (uint64_t)128 * ldb2
                              : (uint64_t)min(n0 + li, p.N - 1) * ldb2 + 256;
    pfaddr = isB ? bB + offB : bA + offA;
    pfstride = isB ? (BKS ? 64u * ldb2 : 128u) : 128u;
  }
// the destination register stays RESERVED from the first prefetch to the vmcnt(0) of the second-to-last iteration ("+v" here, a use after
// the loop): the byte lands a microsecond after the statement, in a register the compiler would otherwise hand to a fragment
#define G7_PF_ISSUE() do { asm volatile("global_load_ubyte %0, %1, off" : "+v"(pfdummy) : "v"(pfaddr) : "memory"); } while (0)
#endif
  // rider R of the 24 (k-strided B) / 16 (k-contiguous B) fragment reads of k-half KH into set F: B first (block KH needs all of B at once)
#define G7_RD_A(F, KH, I) g7_read128<(I) * 2048>(fa##F[I], (KH) ? va1 : va0)
#define G7_RD_ASL(F, KH, I) g7_read_tr<(KH) * 16384>(al##F[I], vas[I])
#define G7_RD_ASH(F, KH, I) g7_read_tr<(KH) * 16384 + 2048>(ah##F[I], vas[I])
#define G7_FA(F, I) (AKS ? join_halves(al##F[I], ah##F[I]) : fa##F[I])
#define G7_RD_BC(F, KH, J) g7_read128<(J) * 2048>(fb##F[J], (KH) ? vb1 : vb0)
#define G7_RD_BSL(F, KH, J) g7_read_tr<(KH) * 16384>(bl##F[J], vbs[J])
#define G7_RD_BSH(F, KH, J) g7_read_tr<(KH) * 16384 + 2048>(bh##F[J], vbs[J])
#define G7_FB(F, J) (BKS ? join_halves(bl##F[J], bh##F[J]) : fb##F[J])
#define G7_MF(F, I, J) g7_mfma(acc[I][J], G7_FB(F, J), G7_FA(F, I))

  const int nt = p.K >> 6;                         // host-checked: K % 64 == 0, nt >= 2
  // ---- prologue: tiles 0 and 1 on their way, k-half 0 of tile 0 in F0
#define G7_DMA_ALL(T, S_) do { G7_DMA(T, S_, 0); G7_DMA(T, S_, 1); G7_DMA(T, S_, 2); G7_DMA(T, S_, 3); G7_DMA(T, S_, 4); G7_DMA(T, S_, 5);    \
    G7_DMA(T, S_, 6); G7_DMA(T, S_, 7); G7_DMA(T, S_, 8); G7_DMA(T, S_, 9); G7_DMA(T, S_, 10); G7_DMA(T, S_, 11); G7_DMA(T, S_, 12);    \
    G7_DMA(T, S_, 13); G7_DMA(T, S_, 14); G7_DMA(T, S_, 15); } while (0)
  G7_DMA_ALL(0, 0);
  G7_DMA_ALL(1, 1);
#if G7_PF > 0
  int pf_tile = 2;                                   // next tile to prefetch; never beyond the last one (the address stops advancing)
#pragma unroll
  for (int i = 0; i < G7_PF; ++i) { G7_PF_ISSUE(); if (pf_tile + 1 < nt) { pfaddr += pfstride; ++pf_tile; } }
#endif
  G7_WAIT_VM(16 + G7_PF);
  G7_BAR();
#define G7_READ_ALL(F, KH) do {                                                                                             \
    if (BKS) { G7_RD_BSL(F, KH, 0); G7_RD_BSH(F, KH, 0); G7_RD_BSL(F, KH, 1); G7_RD_BSH(F, KH, 1); G7_RD_BSL(F, KH, 2); G7_RD_BSH(F, KH, 2);   \
               G7_RD_BSL(F, KH, 3); G7_RD_BSH(F, KH, 3); G7_RD_BSL(F, KH, 4); G7_RD_BSH(F, KH, 4); G7_RD_BSL(F, KH, 5); G7_RD_BSH(F, KH, 5);   \
               G7_RD_BSL(F, KH, 6); G7_RD_BSH(F, KH, 6); G7_RD_BSL(F, KH, 7); G7_RD_BSH(F, KH, 7); }                                          \
    else { G7_RD_BC(F, KH, 0); G7_RD_BC(F, KH, 1); G7_RD_BC(F, KH, 2); G7_RD_BC(F, KH, 3); G7_RD_BC(F, KH, 4); G7_RD_BC(F, KH, 5);             \
           G7_RD_BC(F, KH, 6); G7_RD_BC(F, KH, 7); }                                                                                          \
    if (AKS) { G7_RD_ASL(F, KH, 0); G7_RD_ASH(F, KH, 0); G7_RD_ASL(F, KH, 1); G7_RD_ASH(F, KH, 1); G7_RD_ASL(F, KH, 2); G7_RD_ASH(F, KH, 2);   \
               G7_RD_ASL(F, KH, 3); G7_RD_ASH(F, KH, 3); G7_RD_ASL(F, KH, 4); G7_RD_ASH(F, KH, 4); G7_RD_ASL(F, KH, 5); G7_RD_ASH(F, KH, 5);   \
               G7_RD_ASL(F, KH, 6); G7_RD_ASH(F, KH, 6); G7_RD_ASL(F, KH, 7); G7_RD_ASH(F, KH, 7); }                                          \
    else { G7_RD_A(F, KH, 0); G7_RD_A(F, KH, 1); G7_RD_A(F, KH, 2); G7_RD_A(F, KH, 3); G7_RD_A(F, KH, 4); G7_RD_A(F, KH, 5); G7_RD_A(F, KH, 6);  \
           G7_RD_A(F, KH, 7); } } while (0)
  G7_READ_ALL(0, 0);
  G7_WAIT_LGKM0();
  G7_T(1);

  // ---- one 64-k iteration.  MODE 0: steady state (DMA of tile t+2, reads of tile t+1); 1: second to last (no DMA); 2: last (no DMA, no
  // next reads).  A block's 64 MFMAs go row by row (A fragment i x the 8 B fragments); rider(g) is what follows MFMA number g.
  // The riders are spelled per gap through the macros below so that every template instantiation gets straight-line code.
  // k-strided B: 24 reads per k-half (B lo/hi x 8, then A x 8) in the even gaps 0 .. 46; k-contiguous B: 16 reads in the gaps 0, 3, 6 .. 45.
#define G7_RIDER_RD(F, KH, R) do {                                                                                          \
    if (BKS) { if ((R) < 16) { if ((R) & 1) G7_RD_BSH(F, KH, ((R) >> 1) & 7); else G7_RD_BSL(F, KH, ((R) >> 1) & 7); }             \
               else if (AKS) { if ((R) & 1) G7_RD_ASH(F, KH, (((R) - 16) >> 1) & 7); else G7_RD_ASL(F, KH, (((R) - 16) >> 1) & 7); } \
               else G7_RD_A(F, KH, ((R) - 16) & 7); }                                                                       \
    else { if ((R) < 8) G7_RD_BC(F, KH, (R) & 7); else G7_RD_A(F, KH, ((R) - 8) & 7); } } while (0)
  constexpr int NRD = BKS ? 24 : 16;               // fragment reads per k-half
  // gap of read R: k-strided 2 R (0 .. 46); k-contiguous 3 R (0 .. 45)
#define G7_RD_GAP(R) (BKS ? 2 * (R) : 3 * (R))

#if G7_SCHED == 1
  auto iter = [&](auto mode_c, int t, int s) __attribute__((always_inline)) {
    constexpr int MODE = decltype(mode_c)::value;
    // ---------------- block 0: MFMAs on F0; reads of k-half 1 -> F1; release of stage s; first DMAs of tile t+2
#define G7_B0_GAP(G) do {                                                                                                    \
      if (BKS ? (((G) & 1) == 0 && (G) / 2 < NRD) : ((G) % 3 == 0 && (G) / 3 < NRD)) G7_RIDER_RD(1, 1, BKS ? (G) / 2 : (G) / 3);  \
      if ((G) == 54) { G7_WAIT_LGKM0(); if (MODE == 0) G7_BAR(); }                                                              \
      if (MODE == 0 && (G) >= 55 && (((G) - 55) & 1) == 0) G7_DMA(t + 2, s, ((G) - 55) >> 1); } while (0)
#define G7_B0_ROW(I) do {                                                                                                    \
      G7_MF(0, I, 0); G7_B0_GAP((I) * 8 + 0); G7_MF(0, I, 1); G7_B0_GAP((I) * 8 + 1); G7_MF(0, I, 2); G7_B0_GAP((I) * 8 + 2);    \
      G7_MF(0, I, 3); G7_B0_GAP((I) * 8 + 3); G7_MF(0, I, 4); G7_B0_GAP((I) * 8 + 4); G7_MF(0, I, 5); G7_B0_GAP((I) * 8 + 5);    \
      G7_MF(0, I, 6); G7_B0_GAP((I) * 8 + 6); G7_MF(0, I, 7); G7_B0_GAP((I) * 8 + 7); } while (0)
    G7_B0_ROW(0); G7_B0_ROW(1); G7_B0_ROW(2); G7_B0_ROW(3); G7_B0_ROW(4); G7_B0_ROW(5); G7_B0_ROW(6); G7_B0_ROW(7);
    // gaps 55, 57, 59, 61, 63 carried DMA 0 .. 4 of tile t+2
    // ---------------- block 1: MFMAs on F1; DMA 5 .. 15 in the gaps 0, 2, .. 10 and 13, 15, .. 21; tile t+1 waited for at gap 12
    //                  (vmcnt counts this wave's younger DMAs of tile t+2: 5 + 6 = 11 issued by then); its k-half 0 -> F0 from gap 14 on
#define G7_B1_GAP(G) do {                                                                                                    \
      if (MODE == 0 && (G) <= 10 && ((G) & 1) == 0) G7_DMA(t + 2, s, 5 + ((G) >> 1));                                          \
      if ((G) == 12 && MODE != 2) { if (MODE == 0) G7_WAIT_VM(11); else G7_WAIT_VM(0); G7_BAR(); }                              \
      if (MODE == 0 && (G) >= 13 && (G) <= 21 && ((G) & 1) == 1) G7_DMA(t + 2, s, 11 + (((G) - 13) >> 1));                       \
      if (MODE != 2 && (G) >= 14 && (BKS ? ((((G) - 14) & 1) == 0 && ((G) - 14) / 2 < NRD) : (((G) - 14) % 3 == 0 && ((G) - 14) / 3 < NRD)))   \
        G7_RIDER_RD(0, 0, BKS ? ((G) - 14) / 2 : ((G) - 14) / 3); } while (0)
#define G7_B1_ROW(I) do {                                                                                                    \
      G7_MF(1, I, 0); G7_B1_GAP((I) * 8 + 0); G7_MF(1, I, 1); G7_B1_GAP((I) * 8 + 1); G7_MF(1, I, 2); G7_B1_GAP((I) * 8 + 2);    \
      G7_MF(1, I, 3); G7_B1_GAP((I) * 8 + 3); G7_MF(1, I, 4); G7_B1_GAP((I) * 8 + 4); G7_MF(1, I, 5); G7_B1_GAP((I) * 8 + 5);    \
      G7_MF(1, I, 6); G7_B1_GAP((I) * 8 + 6); G7_MF(1, I, 7); G7_B1_GAP((I) * 8 + 7); } while (0)
    // the reads of tile t+1 go to the OTHER stage: addresses toggled for them, and they stay toggled for the next iteration
    if (MODE != 2) {
      va0 ^= G7_STG; va1 ^= G7_STG;
      if (!BKS) { vb0 ^= G7_STG; vb1 ^= G7_STG; }
      else {
#pragma unroll
        for (int j = 0; j < 8; ++j) vbs[j] ^= G7_STG;
      }
    }
    G7_B1_ROW(0); G7_B1_ROW(1); G7_B1_ROW(2); G7_B1_ROW(3); G7_B1_ROW(4); G7_B1_ROW(5); G7_B1_ROW(6); G7_B1_ROW(7);
    if (MODE != 2) G7_WAIT_LGKM0();
  };
#else
  // Schedule 2 (the default).  Differences from schedule 1: (a) the A region of stage s is released as soon as the A fragments of k-half 1
  // are in (barrier at gap 17), the B region at gap 53 -- the A half of tile t+2 is requested ~36 MFMAs earlier; (b) every DMA is fused
  // with the MFMA it follows (g7_mfma_dma); (c) the stage toggles of the read addresses are riders too (no VALU cluster in one gap);
  // (d) all 16 DMAs of tile t+2 are out before tile t+1 is waited for: vmcnt(16).
  //   block 0: A reads (k-half 1) gaps 0,2..14 | gap 17 lgkmcnt(0)+barrier | B reads gaps 18.. | A DMA fused at MFMA 19,21..33 |
  //            gap 53 lgkmcnt(0)+barrier | B DMA fused at MFMA 54,56..62 | address toggles in the free gaps 49..63
  //   block 1: B DMA fused at MFMA 0,2,4 | gap 11 vmcnt(16)+barrier | reads of tile t+1 (B then A) from gap 12 | lgkmcnt(0) at the end
  uint32_t dAs = dA, dBs = dB;                       // DMA destination bases of the stage being refilled (SGPR; toggled per iteration)
  auto iter = [&](auto mode_c, int t, int s) __attribute__((always_inline)) {
    constexpr int MODE = decltype(mode_c)::value;
    const char* srcA = AKS ? (const char*)p.A + ((long)(2 * (t + 2)) * 32 * p.lda) * 2 : (const char*)p.A + (long)(t + 2) * 128;
    const char* srcA1 = AKS ? srcA + (long)32 * p.lda * 2 : srcA;
    const char* srcB0 = BKS ? (const char*)p.B + ((long)(2 * (t + 2)) * 32 * p.ldb) * 2 : (const char*)p.B + (long)(t + 2) * 128;
    const char* srcB1 = BKS ? srcB0 + (long)32 * p.ldb * 2 : srcB0;
    // MFMA number M of block F (row M / 8, column M % 8), alone or fused with DMA number D of tile t+2 (0..7 A, 8..15 B)
#define G7_MFD(F, M, D) do { constexpr int D_ = (D) & 15, J_ = (D_ - 8) & 7;                                                         \
      if (D_ < 8 && AKS) g7_mfma_dma<((D_ & 7) >> 2) * 16384 + (D_ & 3) * 1024>(acc[(M) / 8][(M) % 8], G7_FB(F, (M) % 8), G7_FA(F, (M) / 8), dAs,   \
                                                                               ((D_ & 7) >> 2) ? srcA1 : srcA, aoffs[D_ & 3]);               \
      else if (D_ < 8) g7_mfma_dma<D_ * 1024>(acc[(M) / 8][(M) % 8], G7_FB(F, (M) % 8), G7_FA(F, (M) / 8), dAs, srcA, aoff[D_ & 7]);       \
      else if (!BKS) g7_mfma_dma<J_ * 1024>(acc[(M) / 8][(M) % 8], G7_FB(F, (M) % 8), G7_FA(F, (M) / 8), dBs, srcB0, boff[J_]);            \
      else g7_mfma_dma<(J_ >> 2) * 16384 + (J_ & 3) * 1024>(acc[(M) / 8][(M) % 8], G7_FB(F, (M) % 8), G7_FA(F, (M) / 8), dBs,              \
                                                            (J_ >> 2) ? srcB1 : srcB0, boffs[J_ & 3]); } while (0)
#define G7_XOR_RIDER(X) do { constexpr int X_ = (X);                                                                                 \
      if (AKS) { if (X_ < 8) g7_xor_stage(vas[X_ & 7]); else if (X_ < 16) g7_xor_stage(vbs[(X_ - 8) & 7]); }                          \
      else if (X_ == 0) g7_xor_stage(va0); else if (X_ == 1) g7_xor_stage(va1);                                                      \
      else if (!BKS) { if (X_ == 2) g7_xor_stage(vb0); else if (X_ == 3) g7_xor_stage(vb1); }                                        \
      else if (X_ < 10) g7_xor_stage(vbs[(X_ - 2) & 7]); } while (0)
    // ---- block 0
#define G7_S2_B0(M) do {                                                                                                             \
      if (MODE == 0 && (M) >= 19 && (M) <= 33 && ((M) & 1)) G7_MFD(0, M, ((M) - 19) >> 1);                                           \
      else if (MODE == 0 && (M) >= 54 && ((M) & 1) == 0) G7_MFD(0, M, 8 + (((M) - 54) >> 1));                                        \
      else G7_MF(0, (M) / 8, (M) % 8);                                                                                               \
      if (!AKS && (M) <= 14 && ((M) & 1) == 0) G7_RD_A(1, 1, ((M) >> 1) & 7);                                                       \
      if (AKS && (M) <= 15) { if ((M) & 1) G7_RD_ASH(1, 1, ((M) >> 1) & 7); else G7_RD_ASL(1, 1, ((M) >> 1) & 7); }                   \
      if ((M) == 17) { G7_WAIT_LGKM0(); if (MODE == 0) G7_BAR(); }                                                                    \
      if (BKS && (M) >= 18 && (M) <= 48 && ((M) & 1) == 0) { if ((((M) - 18) >> 1) & 1) G7_RD_BSH(1, 1, (((M) - 18) >> 2) & 7); else G7_RD_BSL(1, 1, (((M) - 18) >> 2) & 7); }   \
      if (!BKS && (M) >= 18 && (M) <= 46 && (((M) - 18) & 3) == 0) G7_RD_BC(1, 1, (((M) - 18) >> 2) & 7);                            \
      if ((M) == 53) { G7_WAIT_LGKM0(); if (MODE == 0) G7_BAR(); }                                                                    \
      if (MODE != 2 && (M) >= 49 && (M) <= 52) G7_XOR_RIDER((M) - 49);                                                                \
      if (MODE != 2 && (M) >= 55 && ((M) & 1)) G7_XOR_RIDER(4 + (((M) - 55) >> 1)); } while (0)
#define G7_S2_ROW0(I) do { G7_S2_B0((I) * 8 + 0); G7_S2_B0((I) * 8 + 1); G7_S2_B0((I) * 8 + 2); G7_S2_B0((I) * 8 + 3);               \
                           G7_S2_B0((I) * 8 + 4); G7_S2_B0((I) * 8 + 5); G7_S2_B0((I) * 8 + 6); G7_S2_B0((I) * 8 + 7); } while (0)
    G7_S2_ROW0(0); G7_S2_ROW0(1); G7_S2_ROW0(2); G7_S2_ROW0(3); G7_S2_ROW0(4); G7_S2_ROW0(5); G7_S2_ROW0(6); G7_S2_ROW0(7);
    // ---- block 1
#if G7_PF > 0
#define G7_PF_RIDER(M) do { if (MODE == 0 && (M) == 13) G7_PF_ISSUE(); } while (0)
#else
#define G7_PF_RIDER(M) do {} while (0)
#endif
#define G7_S2_B1(M) do {                                                                                                             \
      if (MODE == 0 && (M) <= 4 && ((M) & 1) == 0) G7_MFD(1, M, 13 + ((M) >> 1));                                                    \
      else G7_MF(1, (M) / 8, (M) % 8);                                                                                               \
      if (MODE != 2 && (M) == 1) G7_XOR_RIDER(9);                                                                                     \
      if (MODE != 2 && AKS && ((M) == 3 || ((M) >= 5 && (M) <= 9))) G7_XOR_RIDER((M) == 3 ? 10 : 6 + (M));                            \
      if ((M) == 11 && MODE != 2) { if (MODE == 0) G7_WAIT_VM(16 + (G7_PF > 0)); else G7_WAIT_VM(0); G7_BAR(); }                      \
      G7_PF_RIDER(M);                                                                                                                \
      if (MODE != 2 && BKS && !AKS && (M) >= 12 && (M) <= 58 && ((M) & 1) == 0) G7_RIDER_RD(0, 0, ((M) - 12) >> 1);                   \
      if (MODE != 2 && AKS && (M) >= 12 && (M) <= 42 && ((M) & 1) == 0) G7_RIDER_RD(0, 0, ((M) - 12) >> 1);                           \
      if (MODE != 2 && AKS && (M) >= 43 && (M) <= 58) G7_RIDER_RD(0, 0, 16 + (M) - 43);                                               \
      if (MODE != 2 && !BKS && (M) >= 12 && (M) <= 57 && ((M) - 12) % 3 == 0) G7_RIDER_RD(0, 0, ((M) - 12) / 3); } while (0)
#define G7_S2_ROW1(I) do { G7_S2_B1((I) * 8 + 0); G7_S2_B1((I) * 8 + 1); G7_S2_B1((I) * 8 + 2); G7_S2_B1((I) * 8 + 3);               \
                           G7_S2_B1((I) * 8 + 4); G7_S2_B1((I) * 8 + 5); G7_S2_B1((I) * 8 + 6); G7_S2_B1((I) * 8 + 7); } while (0)
    G7_S2_ROW1(0); G7_S2_ROW1(1); G7_S2_ROW1(2); G7_S2_ROW1(3); G7_S2_ROW1(4); G7_S2_ROW1(5); G7_S2_ROW1(6); G7_S2_ROW1(7);
    if (MODE != 2) G7_WAIT_LGKM0();
    dAs ^= G7_STG; dBs ^= G7_STG;                    // the next iteration refills the other stage
#if G7_PF > 0
    if (MODE == 0 && pf_tile + 1 < nt) { pfaddr += pfstride; ++pf_tile; }
#endif
  };
#endif
  // NOTE on the address toggle above: block 0's reads (k-half 1 of tile t) use the addresses of stage s, block 1's (k-half 0 of tile
  // t+1) those of stage s ^ 1 -- the XOR sits between the two blocks, after the last read of block 0 has been issued (gap <= 46).
  {
    int t = 0;
#pragma unroll 1
    for (; t + 2 < nt; ++t) iter(std::integral_constant<int, 0>{}, t, t & 1);
    iter(std::integral_constant<int, 1>{}, t, t & 1);
    iter(std::integral_constant<int, 2>{}, t + 1, (t + 1) & 1);
  }
#if G7_PF > 0
  asm volatile("" :: "v"(pfdummy));                  // end of the prefetch register's reservation (every prefetch has landed: vmcnt(0) above)
#endif
  // asm MFMAs are opaque to the hazard recogniser: wait out the last results before anything reads an accumulator
#define G7_SETTLE(I) asm volatile("s_nop 7" : "+a"(acc[I][0]), "+a"(acc[I][1]), "+a"(acc[I][2]), "+a"(acc[I][3]),    \
                                              "+a"(acc[I][4]), "+a"(acc[I][5]), "+a"(acc[I][6]), "+a"(acc[I][7]))
  asm volatile("s_nop 15" ::: "memory");
  G7_SETTLE(0); G7_SETTLE(1); G7_SETTLE(2); G7_SETTLE(3); G7_SETTLE(4); G7_SETTLE(5); G7_SETTLE(6); G7_SETTLE(7);
  G7_BAR();                                          // every wave is done with the ring: the epilogue reuses it
  G7_T(2);

  // ---- epilogue through LDS (gemm_tile.h): wave-private [64][128] f32 region (32 KiB), 16-B units XOR-swizzled by row, two passes
  float gate = 1.f;
  if (p.gate) gate = tanhf(bf2f(*p.gate));
  constexpr int WN = 128, ESTR = WN * 4, UNITS = WN / 4;
  char* er = smem + wave * (64 * ESTR);
  const bool fast = ((p.ldc | p.ldres | p.ldaux | p.ldpre) & 7) == 0;
#define G7_EPI_STAGE(PASS) do {                                                                                    \
    _Pragma("unroll") for (int i2 = 0; i2 < 4; ++i2)                                                               \
      _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                              \
        int row = i2 * 16 + (lane & 15), u = j * 4 + (lane >> 4);                                                  \
        *(f32x4*)(er + row * ESTR + ((u ^ (row & (UNITS - 1))) << 4)) = acc[(PASS) * 4 + i2][j];                   \
      }                                                                                                            \
    __builtin_amdgcn_s_waitcnt(0xc07f); } while (0)
  const int em = m0 + wm * 128, en = n0 + wn * WN;
  const int kind = EPI >= 0 ? EPI : epi_kind(p, fast);
  EpiPre<WN> pre0, pre1;
  bf16x8 biasv = epi_bias<WN>(p, lane, en, kind);
  epi_fetch<WN>(p, lane, em, en, kind, pre0);
  G7_EPI_STAGE(0);
  epi_fetch<WN>(p, lane, em + 64, en, kind, pre1);
  if (kind != EK_GENERIC) epi_inputs_ready();
  // a wave is alone on its SIMD here: row groups in flight are its only cover for LDS and store latency (G7_EPI_UNR per kind)
  constexpr int EUNR = (EPI == EK_PLAIN || EPI == EK_GELU2) ? 8 : 0, RUNR = 4;      // EK_ACT keeps its run-time activation switch: unrolled 8 deep it is 65 KiB of code
  if (EPI >= 0 && !ROPE) epi_pass_fixed<WN, EPI < 0 ? 0 : EPI, 64, EUNR>(p, er, lane, em, en, gate, pre0, biasv);
  else epi_pass_kind<WN, 64, ROPE, RUNR>(p, er, lane, em, en, gate, fast, kind, pre0, biasv);
  __builtin_amdgcn_s_waitcnt(0xc07f);
  G7_EPI_STAGE(1);
  if (EPI >= 0 && !ROPE) epi_pass_fixed<WN, EPI < 0 ? 0 : EPI, 64, EUNR>(p, er, lane, em + 64, en, gate, pre1, biasv);
  else epi_pass_kind<WN, 64, ROPE, RUNR>(p, er, lane, em + 64, en, gate, fast, kind, pre1, biasv);
  G7_T(3);
}

template <bool AKS, bool BKS, int EPI>
static void launch7(const Gemm2Params& p, hipStream_t s) {
  static bool attr_set = false;
  constexpr size_t lds = 2 * G7_STG;               // the ring; the epilogue's 4 x 32 KiB staging regions reuse it
  auto kern = gemm7_bf16_kernel<AKS, BKS, EPI>;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; }
  hipLaunchKernelGGL(kern, dim3(p.nbm * p.nbn), dim3(256), lds, s, p);
}

// returns 1 if launched, 0 if this form is not served (the caller falls back to another variant)
extern "C" int unimp_gemm7_launch(const unimp_gemm_desc* d, void* stream) {
  if ((d->a_kstrided && d->b_kstrided != 1) || d->b_kstrided == 2 || (d->K & 63) || d->K < 128) return 0;
#if G7_SCHED == 1 || G7_PF > 0
  if (d->a_kstrided) return 0;                     // the measurement builds serve the k-contiguous A only
#endif
  // the per-lane source offsets are 32-bit byte offsets from the operand's base
  if ((long)(d->a_kstrided ? 32 : d->M) * d->lda * 2 >= (1L << 32) || (long)(d->b_kstrided ? 32 : d->N) * d->ldb * 2 >= (1L << 32)) return 0;
  Gemm2Params p;
  p.A = (const bf16*)d->A; p.B = (const bf16*)d->B; p.C = d->C;
  p.M = d->M; p.N = d->N; p.K = d->K; p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc;
  p.bias = (const bf16*)d->bias; p.res = (const bf16*)d->res; p.ldres = d->ldres;
  p.aux = (const bf16*)d->aux; p.ldaux = d->ldaux; p.pre = (bf16*)d->pre; p.ldpre = d->ldpre;
  p.gate = (const bf16*)d->gate; p.alpha = d->alpha; p.act = d->act; p.dact = d->dact;
  p.out_f32 = d->out_f32; p.accumulate = d->accumulate; p.pre_deriv = d->pre_deriv; GEMM2_FILL_ROPE(p, d); p.ksplit = 0;
  { static int gm = -1; if (gm < 0) { const char* e = getenv("UNIMP_GEMM_GM"); gm = e ? atoi(e) : 0; } p.gm = gm; }
  p.nbm = (d->M + G7_BM - 1) / G7_BM;
  p.nbn = (d->N + G7_BN - 1) / G7_BN;
  hipStream_t s = (hipStream_t)stream;
  const int b = d->b_kstrided;
#if G7_SCHED != 1 && G7_PF == 0
  if (d->a_kstrided) {                             // weight gradients: plain (alpha / gate / accumulate) epilogues, chosen per tile at run time
    if (epi_kind_host(p) == EK_PLAIN) launch7<true, true, EK_PLAIN>(p, s); else launch7<true, true, -1>(p, s);
    return 1;
  }
#endif
#define L7(K_) do { if (b) launch7<false, true, K_>(p, s); else launch7<false, false, K_>(p, s); return 1; } while (0)
  if (p.rope_rot) L7(EK_ROPE);
  switch (epi_kind_host(p)) {
    case EK_PLAIN: L7(EK_PLAIN);
    case EK_ACT:   L7(EK_ACT);
    case EK_GELU2: L7(EK_GELU2);
    case EK_RES:   L7(EK_RES);
    case EK_AUX:   L7(EK_AUX);
    default: break;
  }
  L7(-1);
#undef L7
}
