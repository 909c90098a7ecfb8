// What do the GEMM's side streams cost the matrix pipe on gfx950?  Per "half-step" a wave issues 32 MFMAs (8 x 4 tile,
// operands rotating through 12 fragment registers like the real kernel) and optionally 12 ds_read_b128 into the other
// fragment set and 4 global_load_lds (1 KB each, L2-resident source).  No barriers.  512 threads = 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)

template <bool LDSR, bool DMA>
__global__ __launch_bounds__(512, 2) void mix(const char* __restrict__ src, float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 32768 / 16; i += 512) ((uint4*)smem)[i] = uint4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  __syncthreads();
  f32x4 acc[8][4];
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
  bf16x8 fa[2][8], fb[2][4];
  for (int s = 0; s < 2; ++s) { for (int i = 0; i < 8; ++i) fa[s][i] = *(const bf16x8*)(smem + ((lane * 16 + i * 1024 + s * 8192) & 32767));
                                for (int j = 0; j < 4; ++j) fb[s][j] = *(const bf16x8*)(smem + ((lane * 16 + j * 1024 + 16384 + s * 4096) & 32767)); }
  const char* g = src + (size_t)(blockIdx.x * 8 + wave) * 4096 + lane * 16;
  char* dma_dst = smem + 32768 + wave * 4096;
  for (int it = 0; it < iters; it += 2) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = MFMA(fb[s][j], fa[s][i], acc[i][j]);
        if (LDSR && i < 4) {
          fa[s ^ 1][2 * i] = *(const bf16x8*)(smem + ((lane * 16 + (2 * i) * 1024 + it * 64) & 32767));
          fa[s ^ 1][2 * i + 1] = *(const bf16x8*)(smem + ((lane * 16 + (2 * i + 1) * 1024 + it * 64) & 32767));
          fb[s ^ 1][i] = *(const bf16x8*)(smem + ((lane * 16 + i * 1024 + 16384 + it * 64) & 32767));
        }
        if (DMA && !(i & 1))
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + (i >> 1) * 1024),
                                           (__attribute__((address_space(3))) void*)(dma_dst + (i >> 1) * 1024), 16, 0, 0);
      }
    }
  }
  f32x4 t = acc[0][0];
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) t += acc[i][j];
  out[blockIdx.x * 512 + threadIdx.x] = t[0] + t[1] + t[2] + t[3];
}

template <bool L, bool D> void run(const char* name, const char* src, float* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int iters = 20000;
  (void)hipFuncSetAttribute((const void*)mix<L, D>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((mix<L, D>), dim3(256), dim3(512), 65536, 0, src, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = 256.0 * 8 * iters * 32 * 16384.0;
    if (rep) printf("%-44s %7.2f ms  %7.1f TFLOP/s  (%.0f cycles per 32-MFMA half-step per wave at 2.4 GHz; ideal 2 waves/SIMD = 1024)\n",
                    name, ms, fl / ms / 1e9, ms * 1e-3 * 2.4e9 / iters);
  }
}
int main() {
  char* src; float* out; hipMalloc(&src, 256 * 8 * 4096); hipMemset(src, 0, 256 * 8 * 4096); hipMalloc(&out, 256 * 512 * 4);
  run<false, false>("MFMA only (rotating operands)", src, out);
  run<true, false>("+ 12 ds_read_b128 per half-step", src, out);
  run<false, true>("+ 4 global_load_lds per half-step", src, out);
  run<true, true>("+ both", src, out);
  return 0;
}
