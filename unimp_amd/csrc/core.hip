// Error state + launch check shared by every translation unit of libunimp_hip.so.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include "unimp_hip.h"

static thread_local char g_err[512] = "";

extern "C" int unimp_abi_version(void) { return UNIMP_ABI_VERSION; }
extern "C" const char* unimp_last_error(void) { return g_err; }
extern "C" int unimp_set_error(int code, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg ? msg : "");
  return code;
}
extern "C" int unimp_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "%s: launch failed: %s", what, hipGetErrorString(e));
    return UNIMP_ERR_LAUNCH;
  }
  return UNIMP_OK;
}

// sizeof of the descriptor structs, so a binding can verify its mirror of them (0: gemm, 1: attention, 2: image, 3: MX gemm, 4: decode step)
extern "C" int unimp_struct_size(int which) {
  switch (which) {
    case 0: return (int)sizeof(unimp_gemm_desc);
    case 1: return (int)sizeof(unimp_attn_desc);
    case 2: return (int)sizeof(unimp_image_desc);
    case 3: return (int)sizeof(unimp_mx_gemm_desc);
    case 4: return (int)sizeof(unimp_decode_step_desc);
    default: return -1;
  }
}
