// s_barrier round-trip on gfx950: N back-to-back barriers, 4 / 8 / 16 waves per workgroup, one workgroup per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void bar_loop(int iters, unsigned long long* out) {
  unsigned long long r0 = wall_clock64();
  for (int i = 0; i < iters; ++i) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
  unsigned long long r1 = wall_clock64();
  if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = r1 - r0;
}
// ping-pong skeleton: two groups alternate through barriers, each "phase" busy-waits `work` cycles of s_sleep-free VALU
__global__ void pingpong(int iters, int nmf, unsigned long long* out, float* sink) {
  typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
  typedef __attribute__((ext_vector_type(4))) float f32x4;
  f32x4 acc[8]; for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
  bf16x8 a, b; for (int i = 0; i < 8; ++i) { a[i] = (__bf16)1.0f; b[i] = (__bf16)0.5f; }
  int grp = (threadIdx.x >> 6) >= (blockDim.x >> 7);
  unsigned long long r0 = wall_clock64();
  if (grp) __builtin_amdgcn_s_barrier();
  for (int i = 0; i < iters; ++i) {
    __builtin_amdgcn_s_barrier();                       // "L phase" (empty)
    for (int m = 0; m < nmf; m += 8)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
    __builtin_amdgcn_s_barrier();                       // end of "C phase"
  }
  if (!grp) __builtin_amdgcn_s_barrier();
  unsigned long long r1 = wall_clock64();
  f32x4 s = acc[0]; for (int j = 1; j < 8; ++j) s += acc[j];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s[0];
  if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = r1 - r0;
}
int main() {
  unsigned long long* d; hipMalloc(&d, 8); float* sink; hipMalloc(&sink, 256 * 1024 * 4);
  for (int threads : {256, 512, 1024}) {
    int iters = 20000; unsigned long long h;
    hipLaunchKernelGGL(bar_loop, dim3(256), dim3(threads), 0, 0, iters, d); hipDeviceSynchronize();
    hipLaunchKernelGGL(bar_loop, dim3(256), dim3(threads), 0, 0, iters, d); hipDeviceSynchronize();
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("%2d waves/WG: %.1f ns per s_barrier (%.0f cycles at 2.4 GHz)\n", threads / 64, h * 10.0 / iters, h * 10.0 / iters * 2.4);
  }
  for (int nmf : {32, 64, 128}) {
    int iters = 5000; unsigned long long h;
    hipLaunchKernelGGL(pingpong, dim3(256), dim3(512), 0, 0, iters, nmf, d, sink); hipDeviceSynchronize();
    hipLaunchKernelGGL(pingpong, dim3(256), dim3(512), 0, 0, iters, nmf, d, sink); hipDeviceSynchronize();
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    double ns = h * 10.0 / iters, ideal = 2.0 * nmf * 16 / 2.4;
    printf("ping-pong 8 waves, %3d MFMAs per C phase: %.0f ns per A+B phase pair, MFMA-only ideal %.0f ns -> %.0f %% busy\n", nmf, ns, ideal, 100 * ideal / ns);
  }
  return 0;
}
