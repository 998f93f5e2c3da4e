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

// bumped whenever an entry point, a struct layout or the meaning of an argument changes (round 3: sticky persistent status,
// AG_ACT_LEAKY_GATE, AG_PREC_F32X3, GRU front, Conv2DLSTMCell pieces; round 4: v9 - pitched x / split external gradients of
// the front's persistent launches, process-wide deferral with pause / resume, ag_build_zc, ag_critic_batch; v10 - bf16 storage: ag_gemm_h, bf16 in / out flags of the transposes, rowdot, col_sum and the
// persistent LSTM launches): audiogan_amd/_lib.py refuses a library of another
// version, so Python that relies on a new mode can never drive an older build
extern "C" int ag_abi_version(void) { return 10; }
extern "C" const char* ag_arch(void) { return "gfx950"; }
extern "C" const char* ag_last_error(void) { return g_err; }

// ---- precision mode of the contractions (common.h) ---------------------------------------------------------------
// process-wide (NOT thread-local): torch runs a module's backward on its autograd engine thread, and both halves of a
// step must contract in the same precision
#include <atomic>
static std::atomic<int> g_prec{AG_PREC_F32};

extern "C" int ag_set_precision(int mode) {
  AG_REQUIRE(mode == AG_PREC_F32 || mode == AG_PREC_BF16 || mode == AG_PREC_F32X3,
             "ag_set_precision: mode must be 0 (f32), 1 (bf16) or 2 (f32x3)");
  g_prec.store(mode);
  return AG_OK;
}
extern "C" int ag_get_precision(void) { return g_prec.load(); }
int ag_precision() { return g_prec.load(); }

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

// One output (a float4, or a float in the scalar form) is summed by 8 threads: thread zq takes slabs z = zq, zq + 8, ...
// (ascending, 4 independent loads in flight), then the 8 partial sums are added in the order zq = 0..7.  The order is
// fixed by (Z, thread layout) alone, so the result is bitwise reproducible; the loads of a slab row are coalesced.
template <typename V>
__device__ __forceinline__ V slab_sum(const V* __restrict__ ws, int Z, int64_t n, int64_t i, int zq, bool ok, V* sh) {
  V s = V{};
  if (ok) {
    int z = zq;
    for (; z + 24 < Z; z += 32) {
      const V a = ws[(int64_t)z * n + i], b = ws[(int64_t)(z + 8) * n + i], c = ws[(int64_t)(z + 16) * n + i],
              d = ws[(int64_t)(z + 24) * n + i];
      s += a; s += b; s += c; s += d;
    }
    for (; z < Z; z += 8) s += ws[(int64_t)z * n + i];
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  V t = sh[threadIdx.x & 31];
#pragma unroll
  for (int q = 1; q < 8; ++q) t += sh[q * 32 + (threadIdx.x & 31)];
  return t;
}

__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ ws, int Z, int64_t n,
                                                          float* __restrict__ dst, int accumulate) {
  __shared__ float sh[256];
  const int64_t i = (int64_t)blockIdx.x * 32 + (threadIdx.x & 31);
  const float t = slab_sum<float>(ws, Z, n, i, threadIdx.x >> 5, i < n, sh);
  if (threadIdx.x < 32 && i < n) dst[i] = accumulate ? dst[i] + t : t;
}

__global__ __launch_bounds__(256) void slab_reduce4_kernel(const f32x4* __restrict__ ws, int Z, int64_t n4,
                                                           f32x4* __restrict__ dst, int accumulate) {
  __shared__ f32x4 sh[256];
  const int64_t i = (int64_t)blockIdx.x * 32 + (threadIdx.x & 31);
  const f32x4 t = slab_sum<f32x4>(ws, Z, n4, i, threadIdx.x >> 5, i < n4, sh);
  if (threadIdx.x < 32 && i < n4) dst[i] = accumulate ? dst[i] + t : t;
}

// ---- deferred second stages --------------------------------------------------------------------------------------
// A network's backward issues one second stage per weight-gradient / bias-sum launch (50+ per step, ~5 us each, nothing
// reads their results before the weight-norm backward / the optimiser at the end).  Between ag_defer_reduces(1) and
// ag_flush_reduces() ag_slab_reduce only RECORDS its arguments; the flush sums every recorded output in ONE launch, each
// output in exactly the order the single launches use (bitwise the same result).  The caller keeps the partial-sum
// workspaces alive until the flush.  The state is process-wide (round 4; it was per thread): a scope is opened by the
// thread that calls loss.backward() and the recording calls come from torch's autograd thread - never concurrently, the
// opener blocks inside backward() - so one scope can span every block of a network's backward.
// ag_defer_reduces modes: 1 = start recording (drops anything recorded), 0 = stop (an error while stages are pending),
// 2 = pause (stages issued now run at once, recorded ones stay), 3 = resume.
struct SlabDesc {
  const float* ws;
  float* dst;
  int64_t n;     // outputs (floats)
  int Z, acc, vec, blk0;
  int ncol, ld;  // dst is a [n / ncol, ncol] block of rows of pitch ld (ld == ncol: contiguous)
};
#define SLAB_MAXD 40
struct SlabMulti {
  SlabDesc d[SLAB_MAXD];
  int nd;
};
static bool g_defer = false;          // recording
static bool g_defer_open = false;     // a scope exists (recording or paused)
static SlabMulti g_multi = {};
static int g_multi_blocks = 0;

__global__ __launch_bounds__(256) void slab_reduce_multi_kernel(const SlabMulti m) {
  __shared__ f32x4 sh[256];
  // (constant indices only: a run-time index into the by-value argument would put all of it into scratch)
  SlabDesc D = m.d[0];
#pragma unroll
  for (int k = 1; k < SLAB_MAXD; ++k)
    if (k < m.nd && (int)blockIdx.x >= m.d[k].blk0) D = m.d[k];
  const int64_t i = (int64_t)((int)blockIdx.x - D.blk0) * 32 + (threadIdx.x & 31);
  if (D.vec) {
    const int64_t n4 = D.n >> 2;
    const f32x4 t = slab_sum<f32x4>((const f32x4*)D.ws, D.Z, n4, i, threadIdx.x >> 5, i < n4, sh);
    if (threadIdx.x < 32 && i < n4) {
      const int64_t e = 4 * i;
      f32x4* dst = (f32x4*)(D.dst + (D.ld == D.ncol ? e : (e / D.ncol) * D.ld + e % D.ncol));
      *dst = D.acc ? *dst + t : t;
    }
  } else {
    const float t = slab_sum<float>(D.ws, D.Z, D.n, i, threadIdx.x >> 5, i < D.n, (float*)sh);
    if (threadIdx.x < 32 && i < D.n) {
      float* dst = D.dst + (D.ld == D.ncol ? i : (i / D.ncol) * D.ld + i % D.ncol);
      *dst = D.acc ? *dst + t : t;
    }
  }
}

static int slab_flush(hipStream_t st) {
  if (g_multi.nd > 0) {
    hipLaunchKernelGGL(slab_reduce_multi_kernel, dim3((unsigned)g_multi_blocks), dim3(256), 0, st, g_multi);
    g_multi.nd = 0;
    g_multi_blocks = 0;
    AG_CHECK_LAUNCH("ag_flush_reduces");
  }
  return AG_OK;
}

extern "C" int ag_defer_reduces(int mode) {
  AG_REQUIRE(mode >= 0 && mode <= 3, "ag_defer_reduces: mode must be 0 (off), 1 (on), 2 (pause) or 3 (resume)");
  if (mode == 0) {
    AG_REQUIRE(g_multi.nd == 0, "ag_defer_reduces: %d recorded second stages were never flushed", g_multi.nd);
    g_defer = g_defer_open = false;
  } else if (mode == 1) {
    g_defer = g_defer_open = true;
    g_multi.nd = 0;
    g_multi_blocks = 0;
  } else if (mode == 2) {
    g_defer = false;
  } else {
    AG_REQUIRE(g_defer_open, "ag_defer_reduces: resume without a scope");
    g_defer = true;
  }
  return AG_OK;
}

// is ag_slab_reduce recording right now?  (ag_gemm asks before it hands a split-K second stage over)
bool ag_reduces_deferred() { return g_defer; }

extern "C" int ag_flush_reduces(void* stream) { return slab_flush((hipStream_t)stream); }

// record one second stage: dst = a [rows, ncol] block of pitch ld (the floats it spans: (rows - 1) * ld + ncol)
static int slab_record(const float* ws, int Z, int64_t n, float* dst, int ncol, int ld, int accumulate, hipStream_t st) {
  const int64_t span = (n / ncol - 1) * (int64_t)ld + ncol;
  // two recorded stages must not write the same floats in one launch, and a launch holds SLAB_MAXD of them (spans of
  // pitched blocks are compared whole: conservative)
  bool clash = g_multi.nd == SLAB_MAXD;
  for (int k = 0; k < g_multi.nd && !clash; ++k) {
    const SlabDesc& E = g_multi.d[k];
    const int64_t espan = (E.n / E.ncol - 1) * (int64_t)E.ld + E.ncol;
    clash = dst < E.dst + espan && E.dst < dst + span;
  }
  if (clash) {
    const int rc = slab_flush(st);
    if (rc != AG_OK) return rc;
  }
  SlabDesc& D = g_multi.d[g_multi.nd++];
  D.ws = ws; D.dst = dst; D.n = n; D.Z = Z; D.acc = accumulate; D.ncol = ncol; D.ld = ld;
  D.vec = ((n & 3) == 0 && (ncol & 3) == 0 && (ld & 3) == 0 && (((uintptr_t)ws | (uintptr_t)dst) & 15) == 0) ? 1 : 0;
  D.blk0 = g_multi_blocks;
  g_multi_blocks += (int)ag_cdiv64(D.vec ? n >> 2 : n, 32);
  return AG_OK;
}

// the second stage of a split-K product into a [M, N] block of row pitch ldc, inside a recording scope only
int ag_slab_defer_2d(const float* ws, int Z, int M, int N, float* dst, int ldc, int accumulate, hipStream_t st) {
  return slab_record(ws, Z, (int64_t)M * N, dst, N, ldc, accumulate, st);
}

int ag_slab_reduce(const float* ws, int Z, int64_t n, float* dst, int accumulate, hipStream_t st) {
  if (n <= 0 || Z <= 0) return AG_OK;
  if (g_defer && n < ((int64_t)1 << 30)) return slab_record(ws, Z, n, dst, (int)n, (int)n, accumulate, st);
  if ((n & 3) == 0 && (((uintptr_t)ws | (uintptr_t)dst) & 15) == 0) {
    const int64_t n4 = n >> 2;
    hipLaunchKernelGGL(slab_reduce4_kernel, dim3((unsigned)ag_cdiv64(n4, 32)), dim3(256), 0, st, (const f32x4*)ws, Z, n4,
                       (f32x4*)dst, accumulate);
  } else {
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)ag_cdiv64(n, 32)), dim3(256), 0, st, ws, Z, n, dst, accumulate);
  }
  AG_CHECK_LAUNCH("ag_slab_reduce");
  return AG_OK;
}
