// api.hip -- library identification and the last-error slot of the C ABI.
#include <stdarg.h>
#include <string.h>

#include "common.h"

static thread_local char g_err[512] = "";

void ag_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int ag_abi_version(void) { return 1; }
extern "C" const char* ag_arch(void) { return "gfx950"; }
extern "C" const char* ag_last_error(void) { return g_err; }

// ---- workspace binding for the two-stage reductions (common.h) -------------------------------------------------
static thread_local AgWs g_ws = {nullptr, 0};

extern "C" int ag_bind_workspace(float* ws, int64_t numel) {
  AG_REQUIRE(numel >= 0 && (((uintptr_t)ws) & 15) == 0, "ag_bind_workspace: workspace must be 16-byte aligned");
  g_ws.p = numel > 0 ? ws : nullptr;
  g_ws.numel = ws ? numel : 0;
  return AG_OK;
}

AgWs ag_ws_take() {
  AgWs w = g_ws;
  g_ws.p = nullptr;
  g_ws.numel = 0;
  return w;
}

__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ ws, int Z, int64_t n,
                                                          float* __restrict__ dst, int accumulate) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float s = ws[i];
    for (int z = 1; z < Z; ++z) s += ws[(int64_t)z * n + i];
    dst[i] = accumulate ? dst[i] + s : s;
  }
}

__global__ __launch_bounds__(256) void slab_reduce4_kernel(const f32x4* __restrict__ ws, int Z, int64_t n4,
                                                           f32x4* __restrict__ dst, int accumulate) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    f32x4 s = ws[i];
    for (int z = 1; z < Z; ++z) s += ws[(int64_t)z * n4 + i];
    dst[i] = accumulate ? dst[i] + s : s;
  }
}

int ag_slab_reduce(const float* ws, int Z, int64_t n, float* dst, int accumulate, hipStream_t st) {
  if (n <= 0 || Z <= 0) return AG_OK;
  if ((n & 3) == 0 && (((uintptr_t)ws | (uintptr_t)dst) & 15) == 0) {
    const int64_t n4 = n >> 2;
    int g = (int)ag_cdiv64(n4, 256);
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(slab_reduce4_kernel, dim3(g), dim3(256), 0, st, (const f32x4*)ws, Z, n4, (f32x4*)dst, accumulate);
  } else {
    int g = (int)ag_cdiv64(n, 256);
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(g), dim3(256), 0, st, ws, Z, n, dst, accumulate);
  }
  AG_CHECK_LAUNCH("ag_slab_reduce");
  return AG_OK;
}
