// What does ONE wave per SIMD pay for each LDS-DMA instruction issued between its MFMAs?  4 waves per CU (one per SIMD),
// each loops over 64 MFMAs with NDMA 1-KiB global->LDS loads spread between them, source L2-resident.
//   mode 0: MFMAs only            mode 1: global_load_lds_dwordx4 (saddr form)      mode 2: buffer_load_dwordx4 ... lds
//   mode 3: global_load_dwordx4 into VGPRs (no LDS)      mode 4: ds_read_b128 only (NDMA of them)
// hipcc --offload-arch=gfx950 -O3 dma_issue.hip -o dma_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;

template <int MODE, int NDMA, int ROWB>
__global__ __launch_bounds__(256, 1) void k(const char* src, float* out, int iters, unsigned long long* clk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.0f + i * 0.01f); }
  int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const char* base = src + (size_t)blockIdx.x * 65536 + wave * 16384;
  // ROWB = bytes of one contiguous global run per row: 64 (k32 half-stage rows) or 128 (whole lines); rows 4 KiB apart
  uint32_t voff = (uint32_t)((lane / (ROWB / 16)) * 4096 + (lane % (ROWB / 16)) * 16);
  uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wave * 16384;
  i32x4 rsrc = {(int)(uintptr_t)base, (int)((uintptr_t)base >> 32), (int)0x80000000, 0x00020000};
  rsrc[0] = __builtin_amdgcn_readfirstlane(rsrc[0]); rsrc[1] = __builtin_amdgcn_readfirstlane(rsrc[1]);
  f32x4 sink = {0.f, 0.f, 0.f, 0.f};
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 64; ++i) {
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i & 15]) : "v"(a), "v"(b));
      if (NDMA && (i % (64 / NDMA)) == (64 / NDMA) - 1) {
        int d = i / (64 / NDMA);
        __builtin_amdgcn_sched_barrier(0);
        if (MODE == 1) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(base + d * 256), "s"(lds0 + d * 1024) : "memory", "m0");
        if (MODE == 2) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" :: "v"(voff), "s"(rsrc), "s"(lds0 + d * 1024), "s"(d * 256) : "memory", "m0");
        if (MODE == 3) { f32x4 t; asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(t) : "v"(voff), "s"(base + d * 256) : "memory"); (void)t; }
        if (MODE == 5) { if (d & 1) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(base + (d >> 1) * 256), "s"(lds0 + 8192 + (d >> 1) * 1024) : "memory", "m0");
                         else { f32x4 t, u; asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024" : "=v"(t), "=v"(u) : "v"(lds0 + lane * 16 + (d >> 1) * 2048) : "memory"); (void)t; (void)u; } }
        if (MODE == 4) { f32x4 t; asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"(lds0 + lane * 16 + d * 1024) : "memory"); (void)t; }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  f32x4 s = acc[0];
  for (int i = 1; i < 16; ++i) s += acc[i];
  s += sink;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3] + ((float*)smem)[threadIdx.x];
  if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;
}

template <int MODE, int NDMA, int ROWB>
void run(const char* name, const char* src, float* out, unsigned long long* clk) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int iters = 4000;
  auto kern = k<MODE, NDMA, ROWB>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  float best = 1e9; unsigned long long h = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), 131072, 0, src, out, iters, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) { best = ms; hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost); }
  }
  printf("%-44s %7.1f cycles per 64 MFMAs (ideal 1024)   %.1f TFLOP/s   fill %.1f GB/s/CU\n", name, (double)h / iters,
         256.0 * 4 * iters * 64 * 16384.0 / best / 1e9, 4.0 * (MODE == 5 ? NDMA / 2 : NDMA) * 1024 * iters / (best * 1e-3) / 1e9 * ((MODE >= 1 && MODE <= 3) || MODE == 5));
}

int main() {
  char* src; float* out; unsigned long long* clk;
  hipMalloc(&src, 256 * 65536 + (1 << 20)); hipMemset(src, 0, 256 * 65536 + (1 << 20));
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&clk, 16);
  run<0, 0, 64>("MFMA only", src, out, clk);
  run<1, 8, 64>("global_load_lds x8, 64-B rows", src, out, clk);
  run<1, 8, 128>("global_load_lds x8, 128-B rows", src, out, clk);
  run<2, 8, 64>("buffer_load lds x8, 64-B rows", src, out, clk);
  run<2, 8, 128>("buffer_load lds x8, 128-B rows", src, out, clk);
  run<3, 8, 64>("global_load (VGPR) x8, 64-B rows", src, out, clk);
  run<3, 8, 128>("global_load (VGPR) x8, 128-B rows", src, out, clk);
  run<4, 16, 64>("ds_read_b128 x16", src, out, clk);
  run<5, 16, 64>("gemm4 mix: 8 global_load_lds + 16 ds_read", src, out, clk);
  run<1, 16, 128>("global_load_lds x16, 128-B rows", src, out, clk);
  run<2, 16, 128>("buffer_load lds x16, 128-B rows", src, out, clk);
  return 0;
}
