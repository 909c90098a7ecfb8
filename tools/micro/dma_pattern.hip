// Does the per-instruction address pattern of LDS-DMA matter on gfx950?  Every wave streams rows of a [R][K] bf16 matrix
// (row stride 5120 B like the step's activations) into LDS, 4 KB per wave per "half-step" (= 32 KB per CU, the 256 x 256
// GEMM tile's appetite), with at most 3 half-steps in flight, while issuing 32 MFMAs per half-step:
//   pattern 64:  one instruction = 16 rows x 64 B   (a [rows][32 k] half-stage of a k-contiguous operand)
//   pattern 128: one instruction =  8 rows x 128 B  (a [rows][64 k] stage: whole cache lines)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)

template <int PAT, bool DO_MFMA, int SPREAD = 16384, bool OFFSET = false>   // OFFSET: waves 4-7 (the SIMD partners of 0-3) run their MFMA block BEFORE their DMA block      // SPREAD: rows the blocks are spread over (16384 = 84 MB, 2048 = 10 MB ~ L2 + MALL, 512 = 2.6 MB in L2)
__global__ __launch_bounds__(512, 2) void stream(const char* __restrict__ src, long ldb, int K2 /* bytes per row */, float* out, int nh,
                                                 unsigned long long* clk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  f32x4 acc[8][4];
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
  bf16x8 fa[8], fb[4];
  for (int i = 0; i < 8; ++i) for (int e = 0; e < 8; ++e) fa[i][e] = (__bf16)(1.0f + i);
  for (int j = 0; j < 4; ++j) for (int e = 0; e < 8; ++e) fb[j][e] = (__bf16)(0.5f + j);
  // block owns 512 rows (A 256 + B 256 of a tile); wave owns 64 of them
  long row0 = ((long)blockIdx.x * 512 + wave * 64) % SPREAD;
  for (int h = 0; h < nh; ++h) {
    if (OFFSET && DO_MFMA && wave >= 4) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = MFMA(fb[j], fa[i], acc[i][j]);
    }
    // 4 instructions of 1 KB per wave per half-step
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const char* g;
      if (PAT == 64) { int r = q * 16 + (lane >> 2); g = src + (row0 + r) * ldb + ((long)h * 64) % K2 + (lane & 3) * 16; }
      else { int hh = h >> 1, half = h & 1; int r = half * 32 + q * 8 + (lane >> 3); g = src + (row0 + r) * ldb + ((long)hh * 128) % K2 + (lane & 7) * 16; }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                       (__attribute__((address_space(3))) void*)(smem + ((h & 3) * 8 + wave) * 4096 + q * 1024), 16, 0, 0);
    }
    if (DO_MFMA && !(OFFSET && wave >= 4)) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = MFMA(fb[j], fa[i], acc[i][j]);
    }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");        // at most 2 older half-steps + this one in flight
  }
  f32x4 t = acc[0][0];
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) t += acc[i][j];
  out[blockIdx.x * 512 + threadIdx.x] = t[0] + t[1] + t[2] + t[3] + smem[threadIdx.x];
  if (blockIdx.x == 17 && threadIdx.x == 0) { clk[0] = __builtin_readcyclecounter() - c0; clk[1] = wall_clock64() - r0; }
}

template <int PAT, bool M, int SPREAD = 16384, bool OFF = false> void run(const char* name, const char* src, float* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int nh = 4096;
  (void)hipFuncSetAttribute((const void*)stream<PAT, M, SPREAD, OFF>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    static unsigned long long* clk = nullptr; if (!clk) hipMalloc(&clk, 16);
    hipLaunchKernelGGL((stream<PAT, M, SPREAD, OFF>), dim3(256), dim3(512), 131072, 0, src, 5120L, 5120, out, nh, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double bytes = 256.0 * 8 * 4096.0 * nh, fl = 256.0 * 8 * nh * 32 * 16384.0;
    unsigned long long hc[2]; hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
    if (rep) printf("%-40s %7.2f ms  shader clock %4.0f MHz  fill %6.2f TB/s (%5.1f GB/s per CU)%s\n", name, ms, (double)hc[0] / hc[1] * 100.0, bytes / ms / 1e9, bytes / ms / 1e6 / 256,
                    M ? (std::string("  MFMA ") + std::to_string((int)(fl / ms / 1e9)) + " TFLOP/s").c_str() : "");
  }
}
#include <string>
int main() {
  char* src; float* out; size_t n = 16384UL * 5120 + 65536; hipMalloc(&src, n); hipMemset(src, 0, n); hipMalloc(&out, 256 * 512 * 4);
  run<64, false>("16 rows x 64 B per instr, no MFMA", src, out);
  run<128, false>("8 rows x 128 B per instr, no MFMA", src, out);
  run<64, true>("16 rows x 64 B per instr + 32 MFMA", src, out);
  run<128, true>("8 rows x 128 B per instr + 32 MFMA", src, out);
  run<64, true, 4096>("64 B + MFMA, 21 MB footprint (8 blocks/region)", src, out);
  run<64, true, 512>("64 B + MFMA, 2.6 MB footprint (L2 resident)", src, out);
  run<128, true, 512>("128 B + MFMA, 2.6 MB footprint (L2 resident)", src, out);
  run<128, false, 512>("128 B no MFMA, 2.6 MB footprint", src, out);
  run<64, true, 16384, true>("64 B + MFMA, SIMD partners out of phase", src, out);
  run<128, true, 16384, true>("128 B + MFMA, SIMD partners out of phase", src, out);
  run<64, true, 512, true>("64 B + MFMA, out of phase, L2 resident", src, out);
  run<128, true, 512, true>("128 B + MFMA, out of phase, L2 resident", src, out);
  return 0;
}
