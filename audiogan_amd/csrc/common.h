// common.h -- shared device/host helpers for libaudiogan_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/audiogan_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define AG_WAVE 64

// last-error text (host side, one per process; the library is driven by one thread per GPU)
void ag_set_error(const char* fmt, ...);

// Two-stage reductions (deterministic: no float atomics).  A caller binds a workspace with ag_bind_workspace() right
// before a reducing entry point; that entry point takes it (the binding is consumed) and sums partial results in a
// fixed order with ag_slab_reduce().  Without a bound workspace the legacy float-atomic path runs.
struct AgWs {
  float* p;
  int64_t numel;
};
AgWs ag_ws_take();
// dst[i] (+)= sum_{z=0}^{Z-1} ws[z*n + i], z ascending
int ag_slab_reduce(const float* ws, int Z, int64_t n, float* dst, int accumulate, hipStream_t st);
bool ag_reduces_deferred();   // inside a recording ag_defer_reduces scope (api.hip)
// record (scope recording only) the second stage of a split-K product into a [M,N] block of row pitch ldc
int ag_slab_defer_2d(const float* ws, int Z, int M, int N, float* dst, int ldc, int accumulate, hipStream_t st);

// C[row*ldc+col] = beta*C + sum_{z<Z} part[z*pitch + row*N+col] + bias[col] + res[row*ldres+col]   (gemm.hip)
int ag_splitk_reduce(const float* part, int Z, int64_t pitch, int M, int N, float* C, int ldc, float beta,
                     const float* bias, const float* res, int ldres, hipStream_t st);

// Precision mode of the contractions (api.hip, process-wide, set by ag_set_precision):
//   AG_PREC_F32   operands as stored (exact fp32 MFMA / FMA chains)
//   AG_PREC_BF16  EVERY contraction (conv, transposed conv, linear, recurrent products; forward, backward-data and
//                 backward-weight forms) rounds BOTH operands to bf16 (round-to-nearest-even) and accumulates in fp32.
//                 The GEMM and the persistent recurrent kernels then run v_mfma_f32_32x32x16_bf16; kernels that still
//                 issue fp32 MFMA / FMA round their operands in registers (ag_rbf) - products of bf16 values are exact
//                 in fp32, so both forms compute the same sums up to the order of the fp32 additions.
int ag_precision();
#define AG_PREC_F32 0
#define AG_PREC_BF16 1

// fp32 -> bf16 by the hardware convert (v_cvt_pk_bf16_f32: round-to-nearest-even, NaN stays NaN), two values per
// instruction.  ag_pack_bf16: packed pair (lo = a, hi = b); ag_rbf*: rounded values kept in fp32 registers.
typedef __bf16 ag_bf2 __attribute__((ext_vector_type(2)));
typedef float ag_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned ag_pack_bf16(float a, float b) {
  const ag_f2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, ag_bf2));
}
// Every prepared conv weight layout [Cp2][taps][Mpad] (fp32, conv_engine.hip) is followed, in the same buffer, by its bf16
// image in the order the bf16 conv kernel keeps it in LDS: [ceil(Cp2/16)][taps][2][Mpad][8 x bf16] (8 consecutive channels
// of one row per 16-byte slot; channels beyond Cp2 stay zero).  Offsets / sizes in floats:
__host__ __device__ inline int64_t ag_wq_offset(int cp2, int taps, int mpad) { return (int64_t)cp2 * taps * mpad; }
__host__ __device__ inline int64_t ag_wq_floats(int cp2, int taps, int mpad) {
  return (int64_t)((cp2 + 15) / 16) * 16 * taps * mpad / 2;
}
__host__ __device__ inline int64_t ag_wq_index(int c, int tap, int row, int taps, int mpad) {   // in bf16 elements
  return ((((int64_t)(c >> 4) * taps + tap) * 2 + ((c >> 3) & 1)) * mpad + row) * 8 + (c & 7);
}
__device__ __forceinline__ float ag_rbf(float x) { return __uint_as_float(ag_pack_bf16(x, x) << 16); }
__device__ __forceinline__ float ag_rbf_if(float x, int rb) { return rb ? ag_rbf(x) : x; }
__device__ __forceinline__ f32x4 ag_rbf4_if(f32x4 v, int rb) {
  if (rb) {
    const unsigned lo = ag_pack_bf16(v[0], v[1]), hi = ag_pack_bf16(v[2], v[3]);
    v[0] = __uint_as_float(lo << 16); v[1] = __uint_as_float(lo & 0xFFFF0000u);
    v[2] = __uint_as_float(hi << 16); v[3] = __uint_as_float(hi & 0xFFFF0000u);
  }
  return v;
}

#define AG_REQUIRE(cond, ...)     \
  do {                            \
    if (!(cond)) {                \
      ag_set_error(__VA_ARGS__);  \
      return AG_ERR_ARG;          \
    }                             \
  } while (0)

#define AG_CHECK_LAUNCH(name)                                              \
  do {                                                                     \
    hipError_t e__ = hipGetLastError();                                    \
    if (e__ != hipSuccess) {                                               \
      ag_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return AG_ERR_LAUNCH;                                                \
    }                                                                      \
  } while (0)

static inline int ag_cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t ag_cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int ag_roundup(int a, int b) { return ag_cdiv(a, b) * b; }

__device__ __forceinline__ float ag_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block-wide sum for blocks of up to 1024 threads; result valid in every thread
__device__ __forceinline__ float ag_block_sum(float v, float* sh /* >= 17 floats */) {
  v = ag_wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) sh[wid] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < nw; ++i) r += sh[i];
  return r;
}

// "Aligned" scatter layout of a strided conv's weight (mode 1 of the conv engine, conv_engine.hip).
// Row (o, r) of the polyphase GEMM writes output position u = s*n + r - pad.  When pad % s != 0 the phases start at
// different n (q_r = ceil((pad - r) / s)), the column range is L/s + 1 wide and the ragged last column used to cost
// a second launch.  Storing phase r's taps one slot later by shift_r = ceil(pad/s) - q_r (0 or 1) lines all phases
// up on the same columns.  It is free whenever no phase needs more than the ceil(K/s) slots that exist
// (k7 s2 p3 - the critic's convs - does; k8 s4 p2 would need a third slot and keeps the plain layout).
__host__ __device__ inline int ag_scatter_shift(int s, int pad, int r) {
  const int qmax = (pad + s - 1) / s;
  const int qr = pad > r ? (pad - r + s - 1) / s : 0;
  return qmax - qr;
}
__host__ __device__ inline int ag_scatter_aligned(int K, int s, int pad) {
  if (s <= 1 || pad % s == 0) return 0;
  const int mt = (K + s - 1) / s;
  for (int r = 0; r < s; ++r) {
    const int taps_r = r < K ? (K - r + s - 1) / s : 0;
    if (taps_r + ag_scatter_shift(s, pad, r) > mt) return 0;
  }
  return 1;
}

__device__ __forceinline__ float ag_apply_act(float v, int act, float slope) {
  if (act == AG_ACT_LEAKY) return v > 0.f ? v : v * slope;
  if (act == AG_ACT_TANH) return tanhf(v);
  return v;
}

// the tail of a GEMM epilogue.  AG_ACT_LEAKY_GATE: `res` is not added - it is the SAVED OUTPUT of a LeakyReLU whose
// derivative scales this result (the activation backward rides in the epilogue of the product that feeds it)
__device__ __forceinline__ float ag_res_act(float v, bool has_res, float r, int act, float slope) {
  if (act == AG_ACT_LEAKY_GATE) return r > 0.f ? v : v * slope;
  if (has_res) v += r;
  return ag_apply_act(v, act, slope);
}

__device__ __forceinline__ float ag_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

// single-input-channel convolutions (conv_c1.hip): taken by ag_conv1d_engine / ag_conv1d_wgrad for the layer shapes they
// are instantiated for.  try_*: 1 = launched (status in *rc), 0 = not this kernel's case.
int ag_conv_c1_try_fwd(const ag_conv_args& a, int rb, hipStream_t st, int* rc);
int ag_conv_c1_try_bwdx(const ag_conv_args& a, int rb, hipStream_t st, int* rc);
int ag_conv_c1_wgrad_slabs(int B, int A, int Lt, int s, int K, int* bper_out);
int ag_conv_c1_wgrad(const float* dy, int64_t dy_bs, int64_t dy_cs, const float* x, int64_t x_bs, float* part, int B, int A,
                     int Lt, int Lx, int s, int K, int pad, int rb, hipStream_t st);
