// Sustained MFMA ceiling on this part: register-only v_mfma_f32_16x16x32_bf16 streams, 1 or 2 waves per SIMD, plus the
// shader clock seen by s_memtime vs the 100 MHz s_memrealtime.   hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int NACC>
__global__ __launch_bounds__(512) void mfma_loop(float* out, int iters, unsigned long long* clk) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.0f + i * 0.01f); }
  unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
  f32x4 s = acc[0];
  for (int i = 1; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

int main() {
  float* out; unsigned long long* clk;
  hipMalloc(&out, 1024 * 512 * 4); hipMalloc(&clk, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wps = 1; wps <= 2; ++wps) {
    int threads = 256 * wps, blocks = 256, iters = 40000;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(mfma_loop<16>, dim3(blocks), dim3(threads), 0, 0, out, iters, clk);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
      double flops = (double)blocks * (threads / 64) * iters * 16 * 16384.0;
      printf("waves/SIMD %d: %.2f ms  %.1f TFLOP/s   s_memtime ticks %llu, realtime ticks %llu (100 MHz) -> memtime %.1f MHz; MFMA cycles/issue at 2.4 GHz: %.2f\n",
             wps, ms, flops / ms / 1e9, h[0], h[1], (double)h[0] / h[1] * 100.0, ms * 1e-3 * 2.4e9 / ((double)iters * 16 * wps));
    }
  }
  return 0;
}
