// Ping-pong bf16 MFMA GEMM for gfx950: 256 x BN tiles (BN = 256 / 128), 512 threads = 8 waves, one workgroup per CU.
//
// The two waves that share a SIMD (wave w and w+4) run the same program ONE PHASE APART, so the matrix pipe always has
// a wave issuing MFMAs while its partner does the memory-side work (MI355X_MICROARCH "Two waves per SIMD", item 9 and
// the 8-phase idea of the cdna guide §5): K is consumed in half-steps of 32, each half-step being
//     L phase: issue the LDS-DMA of half-step h+3, ds_read the fragments of half-step h+1, wait for them, s_barrier
//     C phase: 8 x NJ MFMAs on the fragments of half-step h (register set R[h&1]),                       s_barrier
// Group A (waves 0-3) is in its C phase while group B (waves 4-7) is in its L phase and vice versa (B passes one
// extra barrier up front, A one at the end).  Raw s_barrier + counted s_waitcnt vmcnt keep 1-2 DMA half-stages in
// flight across barriers (a __syncthreads() would drain them).
//
// LDS: ring of 4 half-stages (A [256][32] + B [BN][32] bf16 = 32 / 24 KiB each).  Hazards, with A's phases at intervals
// 2h (L) / 2h+1 (C) and B's at 2h+1 / 2h+2:
//   * DMA(h+3) overwrites the buffer of half-step h-1: its last readers ran L(h-2) (intervals 2h-4 / 2h-3) and every
//     L phase ends with lgkmcnt(0) + barrier, so all reads are complete before interval 2h.
//   * half-step h+1 is read in L(h): its DMA was issued in L(h-2) and each wave waits for it (vmcnt(newest only)) at the
//     end of L(h-1), i.e. before the barriers that precede both groups' L(h).
// Images: k-contiguous operands as [rows][32] with 64-B rows, 16-B chunk XOR ((row>>3)&1)*3 (conflict-free
// ds_read_b128); k-strided operands as [32][rows] read by ds_read_b64_tr_b16 (granule swizzle of common.h).  The DMA
// destination is lane-linear, so both swizzles are applied to the per-lane SOURCE address; out-of-range chunks come
// from a zero constant (no predication).  Epilogue: gemm_tile.h (through LDS, 16-byte row-major accesses).
// BPK form (b_kstrided == 2): the B operand is a PRE-PACKED image of a frozen weight -- [n-tile of 16][32-k step][lane][8 bf16],
// i.e. every MFMA B fragment is one contiguous, coalesced 1-KiB block (unimp_pack_b_bf16) -- and bypasses the LDS: each wave
// loads its NJ fragments of half-step h+1 with NJ global_load_dwordx4 during the L phase of half-step h (one phase pair of
// latency cover, served by L2: a B tile is shared by all row tiles).  Per half-step the LDS then carries 64 KiB of A-fragment
// reads + 16 KiB of DMA writes instead of 96 + 32 KiB: with B through the LDS the LDS pipe (128 B/clk) is as busy as the matrix
// pipe (128 KiB per 1024 MFMA cycles), which is what held every schedule of this family at 52-57 % MFMA utilisation.
// Same k grouping inside the MFMAs as the unpacked kernel: bit-identical results.
#include <stdlib.h>
#include "gemm_half.h"

#if defined(G3X) && defined(G3_AFULL) && defined(G3_FUSE)   // A/B build (Makefile: gemm3b.o): gemm3a with the steady-state L phase spelled in asm, M0 writes fused
#define gemm3_bf16_kernel gemm3b_bf16_kernel
#define unimp_gemm3_launch unimp_gemm3b_launch
#define unimp_gemm3_launch_splitk unimp_gemm3b_launch_splitk
#define launch3 launch3b
#define g3_stamps g3b_stamps
#define unimp_debug_g3_stamps unimp_debug_g3b_stamps
#define getenv_no_fixed getenv_no_fixed_b
#elif defined(G3X) && defined(G3_AFULL)   // third build (Makefile: gemm3a.o): the one-set schedule with the A operand staged in whole 128-byte rows
#define gemm3_bf16_kernel gemm3a_bf16_kernel
#define unimp_gemm3_launch unimp_gemm3a_launch
#define unimp_gemm3_launch_splitk unimp_gemm3a_launch_splitk
#define launch3 launch3a
#define g3_stamps g3a_stamps
#define unimp_debug_g3_stamps unimp_debug_g3a_stamps
#define getenv_no_fixed getenv_no_fixed_a
#elif defined(G3X)    // second build of this file with other schedule switches, under its own symbols (Makefile: gemm3x.o)
#define gemm3_bf16_kernel gemm3x_bf16_kernel
#define unimp_gemm3_launch unimp_gemm3x_launch
#define unimp_gemm3_launch_splitk unimp_gemm3x_launch_splitk
#define launch3 launch3x
#define g3_stamps g3x_stamps
#define unimp_debug_g3_stamps unimp_debug_g3x_stamps
#define getenv_no_fixed getenv_no_fixed_x
#endif
#if defined(G3_AFULL) && !defined(G3_ONESET)
#error "G3_AFULL builds on the one-set schedule"
#endif
// G3_AFULL: a k-contiguous A operand ([M][K] activations -- every forward and dX GEMM) is staged in 64-k stages of WHOLE 128-byte rows
// (8 rows x 128 B per LDS-DMA instruction) through a ring of its own, instead of 32-k half-stages whose instructions fetch 64-byte
// halves of 16 different rows: every 128-byte line is requested once, by one instruction, where the half-stage form asks for it twice,
// a half-step apart (the B operand read k-strided gained 2-5 % from the same change of access shape: tools/bench_gemm_ab.py W^T rows).
//   LDS (160 KiB): A ring 3 stages x [256 rows][64 k] = 96 KiB, B ring 4 half-stages x [BN][32 k] = 64 KiB (BN = 256).
//   A stage s (k = 64 s ..) lives in slot s % 3 as the v1 kernel's KC image (common.h kc_off: 16-byte chunk XOR (row >> 1) & 7,
//   conflict-free ds_read_b128); half-step h reads chunks 4 (h & 1) .. + 3 of stage h >> 1.  DMA of stage s + 2 is issued in L(2 s):
//   its slot's last readers ran L(2 s - 1) (both groups, two and one intervals earlier); it is needed in L(2 s + 4), four half-steps on.
//   vmcnt: per wave an even half-step issues 4 (A) + NB (B) instructions, an odd one NB; after L(h)'s issues everything younger than
//   B(h + 1) may stay in flight = 4 + 2 NB in steady state (either parity), 2 NB / NB / 0 over the last four half-steps.
// Same k grouping inside every MFMA as the other builds: bit-identical results.

#define G3_BM 256
#ifndef G3_NST
#define G3_NST 4          // LDS ring depth in 32-k half-stages (4 x 32 KiB for 256 x 256 tiles; 5 = the whole 160 KiB LDS measured no faster)
#endif

#ifdef G3_NO_PRIO
#define G3_PRIO(x) do {} while (0)
#else
#define G3_PRIO(x) __builtin_amdgcn_s_setprio(x)
#endif
#ifdef G3_STAMP      // debug build: per-block wall-clock stamps (100 MHz s_memrealtime) + the CU it ran on (tools/stamp_gemm3.py)
__device__ unsigned long long g3_stamps[8192 * 12];
extern "C" int unimp_debug_g3_stamps(void* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g3_stamps), sizeof(g3_stamps)); }
#define G3_T(K_) do { if (threadIdx.x == 0 && blockIdx.x < 8192) g3_stamps[blockIdx.x * 12 + (K_)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define G3_T(K_) do {} while (0)
#endif
#define G3_FENCE() do { __builtin_amdgcn_sched_barrier(0); asm volatile("" ::: "memory"); } while (0)
#define G3_BARRIER() do { G3_FENCE(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); G3_FENCE(); } while (0)

// one fragment of the packed B image: 64 lanes x 16 B contiguous.  Inline asm: the result registers are only touched after
// the counted vmcnt + barrier of the L phase (same discipline as frag_ks32_asm).
__device__ __forceinline__ void frag_packed_asm(const void* sbase, uint32_t voff, bf16x8& out) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(out) : "v"(voff), "s"(sbase) : "memory");
}

// EPI: -1 = epilogue kind chosen per tile at run time (every form); EK_ROPE / EK_PLAIN / EK_ACT / EK_GELU2 / EK_AUX / EK_RES = that kind only
template <bool AKS, bool BKS, int BN, bool BPK = false, int EPI = -1>
__global__ __launch_bounds__(512, 2) void gemm3_bf16_kernel(Gemm2Params p) {
  constexpr bool ROPE = EPI == EK_ROPE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  G3_T(0);
#ifdef G3_STAMP
  if (threadIdx.x == 0 && blockIdx.x < 8192) {
    g3_stamps[blockIdx.x * 12 + 10] = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_ID
    g3_stamps[blockIdx.x * 12 + 11] = __builtin_amdgcn_s_getreg((31 << 11) | 20);     // XCC_ID
  }
#endif
  constexpr int NJ = BN / 64, WN = BN / 4;
  constexpr int A_SUB = G3_BM * 64, B_SUB = BPK ? 0 : BN * 64, SUB = A_SUB + B_SUB;
  constexpr int NEW = G3_BM / 128 + (BPK ? NJ : BN / 128);   // vector-memory instructions a wave issues per half-step (DMA + packed-B loads)

  if (!BPK && p.ksplit > 0) {       // split-K: this block reduces one K slice into its own f32 slab (summed by splitk_reduce)
    int k_off = blockIdx.y * p.ksplit;
    p.A += AKS ? (long)k_off * p.lda : (long)k_off;
    p.B += BKS ? (long)k_off * p.ldb : (long)k_off;
    p.K = min(p.K - k_off, p.ksplit);
    p.C = (float*)p.C + (long)blockIdx.y * p.M * p.ldc;
  }
  int nwg = p.nbm * p.nbn;
  int id = xcd_remap(blockIdx.x, nwg);
  const int GM = p.gm > 0 ? p.gm : 4;
  int per_group = GM * p.nbn;
  int grp_ = id / per_group;
  int first_m = grp_ * GM;
  int gsz = min(p.nbm - first_m, GM);
  int in_g = id - grp_ * per_group;
  int tm = first_m + in_g % gsz, tn = in_g / gsz;
  int m0 = tm * G3_BM, n0 = tn * BN;

  int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = lane_id();
  int wm = wave >> 2, wn = wave & 3;                    // wm doubles as the ping-pong group

  f32x4 acc[8][NJ];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#ifndef G3_ONESET
  bf16x8 ra0[8], rb0[NJ], ra1[8], rb1[NJ];                  // k-contiguous operands: whole fragments
  s16x4 la0[8], ha0[8], lb0[NJ], hb0[NJ], la1[8], ha1[8], lb1[NJ], hb1[NJ];   // k-strided operands: two tr halves
#else
  bf16x8 ra0[8], rb0[NJ];
  s16x4 la0[8], ha0[8], lb0[NJ], hb0[NJ];
#endif

  int nh = (p.K + 31) >> 5;

  uint32_t aoff[G3_BM / 128], boff[BN / 128];
  dma_setup<AKS, G3_BM>(p.lda, m0, p.M, wave, aoff);
  if (!BPK) dma_setup<BKS, BN>(p.ldb, n0, p.N, wave, boff);
  // packed B: fragment (n-tile nt, half-step h) sits at ((nt * nh + h) * 64 + lane) * 16 bytes; n-tiles beyond N exist (zero padded)
  const long pk_nh = (p.K + 31) >> 5;
  const char* pk_base = BPK ? (const char*)p.B + ((long)((n0 + wn * WN) >> 4) * pk_nh) * 1024 : nullptr;
  const uint32_t pk_lane = (uint32_t)lane * 16u;
#define DMA(H) do { char* b_ = smem + ((H) % G3_NST) * SUB;                                                             \
    dma_issue<AKS, G3_BM>(p.A, p.lda, (H), p.K, b_, wave, aoff);                                                   \
    if (!BPK) dma_issue<BKS, BN>(p.B, p.ldb, (H), p.K, b_ + A_SUB, wave, boff); } while (0)
  const uint32_t smem_lds = (uint32_t)(uintptr_t)LDS_PTR(char, smem);
  const uint32_t lbA = AKS ? ks32_lane_base<G3_BM>(wm * 128) : 0u, lbB = (BKS && !BPK) ? ks32_lane_base<BN>(wn * WN) : 0u;
#define LOADF(S, H) do { const char* b_ = smem + ((H) % G3_NST) * SUB;                                                  \
    uint32_t ub_ = smem_lds + ((H) % G3_NST) * SUB;                                                                     \
    _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                                               \
      if (BPK) frag_packed_asm(pk_base + ((long)j * pk_nh + (H)) * 1024, pk_lane, rb##S[j]);                       \
      else if (BKS) frag_ks32_asm<BN>(lbB + ub_ + A_SUB, j, lb##S[j], hb##S[j]);                                   \
      else rb##S[j] = frag_kc32(b_ + A_SUB, wn * WN + j * 16); }                                                   \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                                \
      if (AKS) frag_ks32_asm<G3_BM>(lbA + ub_, i, la##S[i], ha##S[i]);                                             \
      else ra##S[i] = frag_kc32(b_, wm * 128 + i * 16); } } while (0)
#define MFMAS(S) do { G3_PRIO(1);                                                                                  \
    bf16x8 fb_[NJ];                                                                                                \
    _Pragma("unroll") for (int j = 0; j < NJ; ++j) fb_[j] = (BKS && !BPK) ? join_halves(lb##S[j], hb##S[j]) : rb##S[j];      \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                                \
      bf16x8 fa_ = AKS ? join_halves(la##S[i], ha##S[i]) : ra##S[i];                                               \
      _Pragma("unroll") for (int j = 0; j < NJ; ++j) acc[i][j] = MFMA16(fb_[j], fa_, acc[i][j]); }                 \
    G3_PRIO(0); } while (0)
#ifndef G3_ONESET
// one half-step: L phase (prefetch h+3, fragments of h+1 -> RN), barrier, C phase (MFMA on RC), barrier
#define HALF_STEP(H, SC, SN) do {                                                                                  \
    if ((H) + PD < nh) DMA((H) + PD);                                                                              \
    if ((H) + 1 < nh) LOADF(SN, (H) + 1);                                                                          \
    /* half-step H+2 must have landed; the (PD-2) younger ones may stay in flight (fewer near the end of K) */     \
    if ((H) + PD < nh) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((PD - 2) * NEW) : "memory");                      \
    else if ((H) + PD - 1 < nh && PD >= 4) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((PD - 3) * NEW) : "memory");  \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                          \
    G3_BARRIER();                                                                                                  \
    MFMAS(SC);                                                                                                     \
    G3_BARRIER(); } while (0)
#else
// ONE fragment register set: the L phase of half-step h reads the fragments of h ITSELF (they are complete at the lgkmcnt(0) of the
// barrier that ends the phase, which is all the C phase needs), fragment reads first so that they land while the wave is held by its
// LDS-DMA issue, and the DMA wait is one half-step more relaxed: half-step h+1 must have landed (the other group reads it in the next
// interval), h+2 and h+3 stay in flight.  48 registers fewer than the two-set form.
static_assert(G3_NST == 4, "the one-set schedule is written for a prefetch distance of 3");
#define HALF_STEP(H, SC, SN) do {                                                                                  \
    LOADF(0, (H));                                                                                                 \
    if ((H) + PD < nh) { DMA((H) + PD); asm volatile("s_waitcnt vmcnt(%0)" :: "n"((PD - 1) * NEW) : "memory"); }   \
    else if ((H) + PD - 1 < nh) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((PD - 2) * NEW) : "memory");             \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                          \
    G3_BARRIER();                                                                                                  \
    MFMAS(0);                                                                                                      \
    G3_BARRIER(); } while (0)
#endif

  constexpr int PD = G3_NST - 1;                        // prefetch distance in half-steps
#ifdef G3_AFULL
  if constexpr (!AKS && !BPK) {
    constexpr int A_STG = G3_BM * 128, A_RING = 3 * A_STG, NB = BN / 128, NFULL = 4 + 2 * NB;
    uint32_t aoffF[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int P = (wave * 4 + i) * 64 + lane, row = P >> 3, cpos = P & 7, c = cpos ^ ((row >> 1) & 7);
      aoffF[i] = (uint32_t)(((long)min(m0 + row, p.M - 1) * p.lda + c * 8) * 2);
    }
    const uint32_t a_dst0 = smem_lds + wave * 4096;
    const int a_lane = kc_off(wm * 128 + (lane & 15), lane >> 4);
#define DMA_A(S) do { const char* ub_ = (const char*)p.A + (long)(S) * 128;                                         \
      uint32_t d_ = __builtin_amdgcn_readfirstlane(a_dst0 + ((S) % 3) * A_STG);                                        \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) glds16_s(ub_, aoffF[i], d_ + i * 1024); } while (0)
#define DMA_B(H) dma_issue<BKS, BN>(p.B, p.ldb, (H), p.K, smem + A_RING + ((H) % 4) * B_SUB, wave, boff)
#define LOADF_AF(H) do { const char* bb_ = smem + A_RING + ((H) % 4) * B_SUB;                                       \
      uint32_t ubb_ = smem_lds + A_RING + ((H) % 4) * B_SUB;                                                           \
      _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                                             \
        if (BKS) frag_ks32_asm<BN>(lbB + ubb_, j, lb0[j], hb0[j]);                                                 \
        else rb0[j] = frag_kc32(bb_, wn * WN + j * 16); }                                                          \
      const char* ab_ = smem + (((H) >> 1) % 3) * A_STG + (a_lane ^ (((H) & 1) << 6));                             \
      _Pragma("unroll") for (int i = 0; i < 8; ++i) ra0[i] = *(const bf16x8*)(ab_ + i * 2048); } while (0)
#define HALF_STEP_AF(H, EVEN) do {                                                                                 \
      LOADF_AF(H);                                                                                                 \
      if ((EVEN) && (H) + 4 < nh) DMA_A(((H) >> 1) + 2);                                                           \
      if ((H) + 3 < nh) DMA_B((H) + 3);                                                                            \
      if ((H) + 4 < nh) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NFULL) : "memory");                              \
      else if ((H) + 4 == nh) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NB) : "memory");                       \
      else if ((H) + 3 == nh) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NB) : "memory");                           \
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                        \
      G3_BARRIER();                                                                                                \
      MFMAS(0);                                                                                                    \
      G3_BARRIER(); } while (0)
    // host-checked: K % 64 == 0, K >= 256 (nh >= 8, even)
    DMA_A(0); DMA_B(0); DMA_A(1); DMA_B(1); DMA_B(2);
#if defined(G3_FUSE) && defined(G3_PF)
    // L2 prefetch (gemm7.hip's G7_PF, measured there: +2.5 ... 5 % on the long-K and ViT up-projection shapes): the 8 workgroups that share
    // this tile's A panel and the 4 that share its B panel each touch THEIR share of the two panels' lines ahead of everybody's DMA (stage
    // 3 here, then one stage per even half-step: two stages ahead of the A stage's DMA) -- one byte per 128-byte line, one instruction per
    // wave and 64-k stage, issued right BEHIND the counted wait of the even half-step so that it is the oldest operation of the next window
    // and has three half-steps to land; every steady-state vmcnt allows one more operation in flight (G3_PFN).  The destination register
    // is reserved for the whole loop ("+v", a use after it); lanes 0..3 carry A rows, the next BSH / 8 lanes B lines, the rest repeat lane 0.
    uint64_t pfaddr; uint32_t pfstride, pfdummy = 0; int pf_stage = 3;
    {
      uint64_t bA = (uint64_t)(uintptr_t)p.A, bB = (uint64_t)(uintptr_t)p.B;
      uint32_t ldb2 = (uint32_t)(p.ldb * 2), lda2 = (uint32_t)(p.lda * 2);
      asm volatile("" : "+s"(bA), "+s"(bB), "+s"(ldb2), "+s"(lda2));
      const int ga = tn & 7, gb = tm & 3;
      constexpr int LPS = BN / 64;                         // 128-byte lines per k-row of a k-strided B stage (BN columns)
      constexpr int BSH = BKS ? 64 * LPS / 4 : BN / 4;      // this workgroup's share of the B stage's lines (1/4); BSH / 8 lanes per wave
      const bool isB = lane >= 4 && lane < 4 + BSH / 8;
      const int li = BSH * gb + (BSH / 8) * wave + ((lane - 4) & (BSH / 8 - 1));
      const int rowA = min(m0 + 32 * ga + 4 * wave + (lane & 3), p.M - 1);
      const uint64_t offA = (uint64_t)rowA * lda2 + 3 * 128;
      const uint64_t offB = BKS ? (uint64_t)(li / LPS) * ldb2 + (uint64_t)min(n0 + (li % LPS) * 64, ((p.N + 7) & ~7) - 8) * 2 + (uint64_t)(3 * 64) * ldb2
                                : (uint64_t)min(n0 + li, p.N - 1) * ldb2 + 3 * 128;
      pfaddr = isB ? bB + offB : bA + offA;
      pfstride = isB ? (BKS ? 64u * ldb2 : 128u) : 128u;
    }
#define G3_PFN 1
#define G3_PF_ISSUE() do { asm volatile("global_load_ubyte %0, %1, off" : "+v"(pfdummy) : "v"(pfaddr) : "memory");       \
      if (pf_stage + 1 < (nh >> 1)) { pfaddr += pfstride; ++pf_stage; } } while (0)
    G3_PF_ISSUE();
#else
#define G3_PFN 0
#define G3_PF_ISSUE() do {} while (0)
#endif
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NFULL + G3_PFN) : "memory");       // stage 0 / half-stage 0 have landed; A(1), B(1), B(2) (and the prefetch) stay in flight
    G3_BARRIER();
    G3_T(1);
    if (wm == 1) G3_BARRIER();
#ifndef G3_NO_PEEL
    // Round 5: the same half-steps with the K tail PEELED (-DG3_NO_PEEL restores round 4's loop for A/B; profiles/r05_gemm_ab_peel_w4x.txt:
    // +5 ... 10 % with a k-strided B, +1 ... 3.5 % with a k-contiguous one, same bits): HALF_STEP_AF decides per half-step whether to issue the A / B DMA and which of
    // four vmcnt counts applies -- eight scalar branches in every L phase (ISA of round 4), the phase that sets the pace of the ping-pong.
    // Steady state (h <= nh - 6) has no decision left; the last four half-steps are written out with their constants.  dma_full: K % 64 == 0
    // here, so the ragged-k path of dma_issue (and its branch) is not compiled in.  Same instructions otherwise: same bits.
#undef DMA_B
#define DMA_B(H) dma_full<BKS, BN>(p.B, p.ldb, (H), smem + A_RING + ((H) % 4) * B_SUB, wave, boff)
#define HS_AF(H, DOA, DOB, VM) do {                                                                                \
      LOADF_AF(H);                                                                                                 \
      if (DOA) DMA_A(((H) >> 1) + 2);                                                                              \
      if (DOB) DMA_B((H) + 3);                                                                                     \
      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(VM) : "memory");                                                   \
      G3_BARRIER();                                                                                                \
      MFMAS(0);                                                                                                    \
      G3_BARRIER(); } while (0)
#ifdef G3_FUSE
    // Steady-state L phase with every instruction placed: the fragment reads are asm statements too, and each LDS-DMA's destination goes
    // into M0 with ONE scalar add whose wait state is the fragment read that follows it (glds16_s spends s_add + s_mov + s_nop per DMA):
    //   even half-step: B reads | [M0, A read, DMA] x 4 (A stage) | [M0, A read, DMA] x NB (B half-stage) | remaining A reads
    //   odd  half-step: B reads | [M0, A read, DMA] x NB | remaining A reads
#define M0_SET(BASE, IMM) asm volatile("s_add_u32 m0, %0, %1" :: "s"(BASE), "n"(IMM) : "m0", "scc")
#define DMA_M0(SB, VOFF) asm volatile("global_load_lds_dwordx4 %0, %1" :: "v"(VOFF), "s"(SB) : "memory")
#define RD_A(I) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ra0[I]) : "v"(va_), "n"((I) * 2048) : "memory")
#define RD_B_ALL(H) do { uint32_t ubb_ = smem_lds + A_RING + ((H) % 4) * B_SUB;                                     \
      if (BKS) { _Pragma("unroll") for (int j = 0; j < NJ; ++j) frag_ks32_asm<BN>(lbB + ubb_, j, lb0[j], hb0[j]); }  \
      else { uint32_t vb_ = ubb_ + kc32_off(wn * WN + (lane & 15), lane >> 4);                                       \
        _Pragma("unroll") for (int j = 0; j < NJ; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(rb0[j]) : "v"(vb_), "n"(j * 1024) : "memory"); } } while (0)
    static_assert(NJ <= 4 && NB <= 2, "rider slots below");
#define HS_FUSE(H, DOA) do {                                                                                         \
      const uint32_t va_ = smem_lds + (((H) >> 1) % 3) * A_STG + (a_lane ^ (((H) & 1) << 6));                        \
      RD_B_ALL(H);                                                                                                   \
      if (DOA) {                                                                                                     \
        const char* ua_ = (const char*)p.A + (long)(((H) >> 1) + 2) * 128;                                           \
        const uint32_t da_ = __builtin_amdgcn_readfirstlane(a_dst0 + ((((H) >> 1) + 2) % 3) * A_STG);                 \
        M0_SET(da_, 0); RD_A(0); DMA_M0(ua_, aoffF[0]); M0_SET(da_, 1024); RD_A(1); DMA_M0(ua_, aoffF[1]);            \
        M0_SET(da_, 2048); RD_A(2); DMA_M0(ua_, aoffF[2]); M0_SET(da_, 3072); RD_A(3); DMA_M0(ua_, aoffF[3]);         \
      } else { RD_A(0); RD_A(1); RD_A(2); RD_A(3); }                                                                 \
      { const char* ub_ = (const char*)p.B + (BKS ? (long)((H) + 3) * 32 * p.ldb * 2 : (long)((H) + 3) * 64);        \
        const uint32_t db_ = __builtin_amdgcn_readfirstlane(smem_lds + A_RING + (((H) + 3) % 4) * B_SUB + wave * NB * 1024);   \
        M0_SET(db_, 0); RD_A(4); DMA_M0(ub_, boff[0]);                                                               \
        if (NB > 1) { M0_SET(db_, 1024); RD_A(5); DMA_M0(ub_, boff[NB > 1 ? 1 : 0]); } else RD_A(5); }                \
      RD_A(6); RD_A(7);                                                                                              \
      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NFULL + G3_PFN) : "memory");                                         \
      if (DOA) G3_PF_ISSUE();                                                                                        \
      G3_BARRIER();                                                                                                  \
      MFMAS(0);                                                                                                      \
      G3_BARRIER(); } while (0)
#endif
    {
      int h = 0;
#pragma unroll 1
#ifdef G3_FUSE
      for (; h + 6 <= nh; h += 2) { HS_FUSE(h, true); HS_FUSE(h + 1, false); }
#else
      for (; h + 6 <= nh; h += 2) { HS_AF(h, true, true, NFULL); HS_AF(h + 1, false, true, NFULL); }
#endif
      HS_AF(h, false, true, 2 * NB); HS_AF(h + 1, false, false, NB); HS_AF(h + 2, false, false, 0); HS_AF(h + 3, false, false, 0);
    }
#if defined(G3_FUSE) && defined(G3_PF)
    asm volatile("" :: "v"(pfdummy));                  // every prefetch has landed (vmcnt(0) of the last half-steps): the register is free again
#endif
#undef HS_AF
#else
    for (int h = 0; h < nh; h += 2) {
      HALF_STEP_AF(h, true);
      HALF_STEP_AF(h + 1, false);
    }
#endif
    if (wm == 0) G3_BARRIER();
    G3_T(2);
#undef DMA_A
#undef DMA_B
#undef LOADF_AF
#undef HALF_STEP_AF
  } else {
#endif
  for (int h0 = 0; h0 < PD && h0 < nh; ++h0) DMA(h0);
#ifndef G3_ONESET
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  G3_BARRIER();
  LOADF(0, 0);
#else
  if (nh >= PD) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((PD - 1) * NEW) : "memory");      // half-step 0 has landed; 1 and 2 stay in flight
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  G3_BARRIER();
#endif
  G3_T(1);
  if (wm == 1) G3_BARRIER();                            // group B runs one phase behind group A
  {
    int h = 0;
#if defined(G3_ONESET) && !defined(G3_NO_PEEL)
    // Round 5, as in the whole-row-A path: the steady state without its per-half-step decisions.  While h + PD + 2 < nh both half-steps
    // of a trip prefetch a half-step that is neither beyond K nor the (possibly ragged) last one: dma_full, one vmcnt constant, no branch.
    // The weight-gradient GEMMs (both operands k-strided, K = all tokens) spend their whole K loop here.
    if constexpr (!BPK) {
#define DMA_F(H) do { char* b_ = smem + ((H) % G3_NST) * SUB;                                                        \
      dma_full<AKS, G3_BM>(p.A, p.lda, (H), b_, wave, aoff); dma_full<BKS, BN>(p.B, p.ldb, (H), b_ + A_SUB, wave, boff); } while (0)
#pragma unroll 1
      for (; h + PD + 2 < nh; h += 2) {
        LOADF(0, h); DMA_F(h + PD); asm volatile("s_waitcnt vmcnt(%0)" :: "n"((PD - 1) * NEW) : "memory");
        G3_BARRIER(); MFMAS(0); G3_BARRIER();
        LOADF(0, h + 1); DMA_F(h + 1 + PD); asm volatile("s_waitcnt vmcnt(%0)" :: "n"((PD - 1) * NEW) : "memory");
        G3_BARRIER(); MFMAS(0); G3_BARRIER();
      }
#undef DMA_F
    }
#endif
    for (; h < nh; h += 2) {
      HALF_STEP(h, 0, 1);
      if (h + 1 < nh) HALF_STEP(h + 1, 1, 0);
    }
  }
  if (wm == 0) G3_BARRIER();                            // equalise the barrier count; all LDS reads are complete
  G3_T(2);
#ifdef G3_AFULL
  }
#endif
#undef DMA
#undef LOADF
#undef MFMAS
#undef HALF_STEP

  float gate = 1.f;
  if (p.gate) gate = tanhf(bf2f(*p.gate));
  // ---- epilogue through LDS (see gemm2.hip): wave-private [64][WN] f32 region, 16-B units XOR-swizzled by row
  constexpr int ESTR = WN * 4, UNITS = WN / 4;
  char* er = smem + wave * (64 * ESTR);
  bool fast = ((p.ldc | p.ldres | p.ldaux | p.ldpre) & 7) == 0;      // N may be ragged: only its last 8-column group is element-wise
  // the two 64-row passes are written out (a loop the compiler declines to unroll would index acc at run time and
  // demote the whole accumulator array to scratch)
#define EPI_STAGE(PASS) do {                                                                                      \
    _Pragma("unroll") for (int i2 = 0; i2 < 4; ++i2)                                                               \
      _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                                           \
        int row = i2 * 16 + (lane & 15), u = j * 4 + (lane >> 4);                                                  \
        *(f32x4*)(er + row * ESTR + ((u ^ (row & (UNITS - 1))) << 4)) = acc[(PASS) * 4 + i2][j];                   \
      }                                                                                                            \
    __builtin_amdgcn_s_waitcnt(0xc07f); } while (0)
  // the epilogue kind is chosen once per tile; every global load goes out before the first store (see gemm_tile.h)
  const int em = m0 + wm * 128, en = n0 + wn * WN;
  const int kind = EPI >= 0 ? EPI : epi_kind(p, fast);
  EpiPre<WN> pre0, pre1;
  bf16x8 biasv = epi_bias<WN>(p, lane, en, kind);
  epi_fetch<WN>(p, lane, em, en, kind, pre0);
  G3_T(4);
  EPI_STAGE(0);
  G3_T(5);
  epi_fetch<WN>(p, lane, em + 64, en, kind, pre1);
  if (kind != EK_GENERIC) epi_inputs_ready();
  if (EPI >= 0 && !ROPE) epi_pass_fixed<WN, EPI < 0 ? 0 : EPI>(p, er, lane, em, en, gate, pre0, biasv);
  else epi_pass_kind<WN, 64, ROPE>(p, er, lane, em, en, gate, fast, kind, pre0, biasv);
  __builtin_amdgcn_s_waitcnt(0xc07f);
  G3_T(6);
  EPI_STAGE(1);
  G3_T(7);
  if (EPI >= 0 && !ROPE) epi_pass_fixed<WN, EPI < 0 ? 0 : EPI>(p, er, lane, em + 64, en, gate, pre1, biasv);
  else epi_pass_kind<WN, 64, ROPE>(p, er, lane, em + 64, en, gate, fast, kind, pre1, biasv);
#undef EPI_STAGE
  G3_T(3);
}

// UNIMP_GEMM_FIXED_EPI=0: every launch takes the run-time-dispatch kernel (A/B and debugging; same bits either way)
static bool getenv_no_fixed() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("UNIMP_GEMM_FIXED_EPI"); v = (e && e[0] == '0') ? 1 : 0; }
  return v == 1;
}

template <bool AKS, bool BKS, int BN, bool BPK = false, int EPI = -1>
static void launch3(const Gemm2Params& p, hipStream_t s, int slices = 1) {
  static bool attr_set = false;
  // the epilogue stages the tile through wave-private LDS regions: 8 waves x 64 rows x (BN / 4) floats
#ifdef G3_AFULL
  constexpr size_t lds_ring = (!AKS && !BPK) ? 3 * (G3_BM * 128) + 4 * (BN * 64) : G3_NST * (G3_BM * 64 + (BPK ? 0 : BN * 64));
  constexpr size_t lds_epi = 8 * 64 * (BN / 4) * 4;
#else
  constexpr size_t lds_ring = G3_NST * (G3_BM * 64 + (BPK ? 0 : BN * 64)), lds_epi = 8 * 64 * (BN / 4) * 4;
#endif
  constexpr size_t lds = lds_ring > lds_epi ? lds_ring : lds_epi;
  auto kern = gemm3_bf16_kernel<AKS, BKS, BN, BPK, EPI>;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; }
  hipLaunchKernelGGL(kern, dim3(p.nbm * p.nbn, slices), dim3(512), lds, s, p);
}

// variant: 256 or 128 = tile width.  Returns 1 if launched.
// splits > 1: K slices into the f32 slabs [slices][M][N] (plain stores; gemm.hip's splitk_reduce applies alpha / gate)
extern "C" int unimp_gemm3_launch_splitk(const unimp_gemm_desc* d, int bn, int splits, float* slabs, void* stream);
extern "C" int unimp_gemm3_launch(const unimp_gemm_desc* d, int bn, void* stream) {
  return unimp_gemm3_launch_splitk(d, bn, 1, nullptr, stream);
}
#ifdef G3_AFULL
extern "C" int unimp_gemm3x_launch_splitk(const unimp_gemm_desc* d, int bn, int splits, float* slabs, void* stream);
#endif
extern "C" int unimp_gemm3_launch_splitk(const unimp_gemm_desc* d, int bn, int splits, float* slabs, void* stream) {
#ifdef G3_AFULL
  // the whole-row A staging serves a k-contiguous A with K a multiple of the 64-k stage; everything else runs the plain one-set build (same bits)
  if (d->a_kstrided || d->b_kstrided == 2 || (d->K & 63) || d->K < 256 || splits > 1) return unimp_gemm3x_launch_splitk(d, bn, splits, slabs, stream);
#endif
  Gemm2Params p;
  p.A = (const bf16*)d->A; p.B = (const bf16*)d->B; p.C = d->C;
  p.M = d->M; p.N = d->N; p.K = d->K; p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc;
  p.bias = (const bf16*)d->bias; p.res = (const bf16*)d->res; p.ldres = d->ldres;
  p.aux = (const bf16*)d->aux; p.ldaux = d->ldaux; p.pre = (bf16*)d->pre; p.ldpre = d->ldpre;
  p.gate = (const bf16*)d->gate; p.alpha = d->alpha; p.act = d->act; p.dact = d->dact;
  p.out_f32 = d->out_f32; p.accumulate = d->accumulate; p.pre_deriv = d->pre_deriv; GEMM2_FILL_ROPE(p, d);
  p.nbm = (d->M + G3_BM - 1) / G3_BM;
  p.nbn = (d->N + bn - 1) / bn;
  p.ksplit = 0;
  { static int gm = -1; if (gm < 0) { const char* e = getenv("UNIMP_GEMM_GM"); gm = e ? atoi(e) : 0; } p.gm = gm; }
  int slices = 1;
  if (splits > 1) {
    p.ksplit = ((d->K + splits - 1) / splits + 63) & ~63;
    slices = (d->K + p.ksplit - 1) / p.ksplit;
    p.C = slabs; p.ldc = d->N; p.out_f32 = 1; p.accumulate = 0; p.alpha = 1.f;
    p.bias = nullptr; p.res = nullptr; p.aux = nullptr; p.pre = nullptr; p.gate = nullptr; p.act = 0; p.dact = 0; p.pre_deriv = 0;
  }
  hipStream_t s = (hipStream_t)stream;
  int a = d->a_kstrided, b = d->b_kstrided;
#define L3(AK, BK_) do { if (bn == 256) launch3<AK, BK_, 256>(p, s, slices); else launch3<AK, BK_, 128>(p, s, slices); } while (0)
#define L3P(AK) do { if (bn == 256) launch3<AK, false, 256, true>(p, s, 1); else launch3<AK, false, 128, true>(p, s, 1); } while (0)
  if (p.rope_rot) {       // host-validated: k-contiguous A, 256-wide tiles; B k-contiguous or k-strided (the transposed copy of a frozen projection)
    if (b == 1) launch3<false, true, 256, false, EK_ROPE>(p, s, 1); else launch3<false, false, 256, false, EK_ROPE>(p, s, 1);
    return 1;
  }
  // fixed-kind instantiations for the forms the training step spends its time in (256-wide tiles, k-contiguous A, unpacked B, no split-K)
  if (bn == 256 && !a && b != 2 && slices == 1 && !getenv_no_fixed()) {
    const int ek = epi_kind_host(p);
#define L3F(K_) do { if (b) launch3<false, true, 256, false, K_>(p, s, 1); else launch3<false, false, 256, false, K_>(p, s, 1); return 1; } while (0)
    switch (ek) {
      case EK_PLAIN: L3F(EK_PLAIN);
      case EK_ACT:   L3F(EK_ACT);
      case EK_GELU2: L3F(EK_GELU2);
      case EK_RES:   L3F(EK_RES);
      case EK_AUX:   if (b) { launch3<false, true, 256, false, EK_AUX>(p, s, 1); return 1; } break;      // dX through the activation: W is read k-strided
      default: break;
    }
#undef L3F
  }
#ifndef G3_ONESET
  if (b == 2) { if (a) L3P(true); else L3P(false); }
  else
#else
  if (b == 2) return 0;      // the packed-B loads need their own look-ahead register set: served by the two-set build only
#endif
  if (!a && !b) L3(false, false); else if (!a && b) L3(false, true); else if (a && b) L3(true, true); else L3(true, false);
#undef L3
#undef L3P
  return 1;
}
