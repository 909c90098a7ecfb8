// RCCL-footprint stand-in for the one-GPU interference measurement (tools/bench_interference.py): G persistent workgroups of
// 512 threads (RCCL's ring kernels run one 256..512-thread workgroup per channel, no LDS tiles to speak of) streaming a buffer:
// read src + read dst + write dst (a ring step's reduce: local chunk + received chunk -> send / store), 16 bytes per access.
#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ __launch_bounds__(512) void stream_reduce(uint4* __restrict__ dst, const uint4* __restrict__ src, long n16, int reps) {
  for (int r = 0; r < reps; ++r)
    for (long i = (long)blockIdx.x * 512 + threadIdx.x; i < n16; i += (long)gridDim.x * 512) {
      uint4 a = src[i], b = dst[i];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
      dst[i] = a;
    }
}
extern "C" int interfere_launch(void* dst, const void* src, long bytes, int groups, int reps, void* stream) {
  hipLaunchKernelGGL(stream_reduce, dim3(groups), dim3(512), 0, (hipStream_t)stream, (uint4*)dst, (const uint4*)src, bytes / 16, reps);
  return (int)hipGetLastError();
}
