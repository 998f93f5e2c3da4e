// conv_c1.hip -- the critic's first layer, a convolution with ONE input channel (D1: Conv1d 1 -> 16, k7 s2 p3,
// audiogan.py:490).
//
// With one input channel the implicit GEMM has a reduction depth of K = 7: an MFMA tile is mostly padding, and SURVEY 8(d)
// puts the layer on the HBM roofline (3 FLOP per byte).  Through the general engine it was an 18 us (forward) / 27 us
// (backward-weight) launch at 0.7-1.0 TB/s.  Here it is a streaming kernel in the style of the 113 -> 1 final conv
// (conv_grad.hip, conv_o1_*): every byte of the big tensor (the output of the forward, dy of the two backward forms) is
// touched once, in 16-byte pieces, with the waveform window of a thread kept in registers: 12 us / 23 us at batch 64.
// The same templates instantiated for the generator's G1.conv (1 -> 128, k17 s8) measured SLOWER than the engine
// (forward 24 vs 22 us, backward-data 93 vs 27, backward-weight 105 vs 26: a 41-sample window per thread at a 32-byte
// lane stride is 41 uncoalesced dword loads, re-read by every channel group), so that layer stays on the engine.
//
//   forward   y[b,o,t]  = act(bias[o] + sum_k w[o,k] x[b, s t + k - p]) * [t < len_b]
//   bwd-data  dx[b,u] (+)= sum_o sum_{k : s t + k - p = u} w[o,k] dy[b,o,t]
//   bwd-weight dw[o,k]  = sum_{b,t} dy[b,o,t] x[b, s t + k - p]           (two-stage, fixed order)
//
// Weights arrive in the engine's prepared layouts (gather layout for the forward, scatter layout for backward-data), so
// the callers do not change.  Taken for k7 s2 only; everything else keeps the engine.
#include "common.h"

template <int S, int K>
struct C1Win {
  static constexpr int N = 3 * S + K;        // waveform samples behind 4 consecutive outputs
};

// ---- forward: a thread owns 4 consecutive output steps of OC output channels -------------------------------------
template <int S, int K, int OC>
__global__ __launch_bounds__(256) void conv_c1_fwd_kernel(const float* __restrict__ x, int64_t x_bs,
                                                          const float* __restrict__ wpa, int opad,
                                                          const float* __restrict__ bias, float* __restrict__ y,
                                                          int64_t y_bs, int64_t y_cs, const int64_t* __restrict__ lens,
                                                          int O, int Lin, int Lout, int pad, int act, float slope, int rb) {
  __shared__ float ws[OC * K];
  __shared__ float bs[OC];
  const int b = blockIdx.y, o0 = blockIdx.z * OC;
  const int t0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  for (int i = threadIdx.x; i < OC * K; i += 256) {
    const int o = i / K, k = i - o * K;
    ws[i] = (o0 + o < O) ? ag_rbf_if(wpa[(int64_t)k * opad + o0 + o], rb) : 0.f;      // gather layout [k][O up 32], C = 1
  }
  if (threadIdx.x < OC) bs[threadIdx.x] = (bias && o0 + threadIdx.x < O) ? bias[o0 + threadIdx.x] : 0.f;
  constexpr int WN = C1Win<S, K>::N;
  float win[WN];
  {
    const float* xb = x + (int64_t)b * x_bs;
    const int g0 = S * t0 - pad;
#pragma unroll
    for (int i = 0; i < WN; ++i) {
      const int g = g0 + i;
      const float v = xb[min(max(g, 0), Lin - 1)];                 // unconditional load, then select
      win[i] = (g >= 0 && g < Lin) ? ag_rbf_if(v, rb) : 0.f;
    }
  }
  __syncthreads();
  if (t0 >= Lout) return;
  int64_t lenb = (int64_t)1 << 60;
  if (lens) lenb = lens[b];
  float* yb = y + (int64_t)b * y_bs + t0;
  const bool full = t0 + 3 < Lout;
#pragma unroll 4
  for (int o = 0; o < OC; ++o) {
    if (o0 + o >= O) break;
    float acc[4] = {bs[o], bs[o], bs[o], bs[o]};
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const float wk = ws[o * K + k];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] += wk * win[S * j + k];
    }
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (t0 + j < lenb) ? ag_apply_act(acc[j], act, slope) : 0.f;
    float* dst = yb + (int64_t)(o0 + o) * y_cs;
    if (full) {
      *reinterpret_cast<f32x4*>(dst) = v;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (t0 + j < Lout) dst[j] = v[j];
    }
  }
}

// ---- backward-data: a thread owns TT consecutive steps t of dy, i.e. the S*TT outputs u in [S tb - p, S (tb + TT) - p)
// (each dy element feeds K outputs; an output window of S*TT samples collects from TT + H steps, H = ceil((K-1)/S))
template <int S, int K, int TT>
__global__ __launch_bounds__(256) void conv_c1_bwdx_kernel(const float* __restrict__ dy, int64_t dy_bs, int64_t dy_cs,
                                                           const float* __restrict__ wpb, int mt, int aligned,
                                                           float* __restrict__ dx, int64_t dx_bs, int accumulate, int C,
                                                           int Lt, int Lu, int pad, int rb) {
  extern __shared__ float ws[];                 // [C][K]
  constexpr int H = (K - 1 + S - 1) / S, NW = TT + H, NO = S * TT;
  const int b = blockIdx.y;
  const int tb = (blockIdx.x * 256 + threadIdx.x) * TT;
  for (int i = threadIdx.x; i < C * K; i += 256) {
    const int c = i / K, k = i - c * K;
    int m = k / S;
    const int rr = k - m * S;
    if (aligned) m += ag_scatter_shift(S, pad, rr);
    ws[i] = ag_rbf_if(wpb[((int64_t)c * mt + m) * 32 + rr], rb);     // scatter layout [c][m][(1 * S) up 32]
  }
  __syncthreads();
  if (tb - H >= Lt) return;
  float acc[NO];
#pragma unroll
  for (int i = 0; i < NO; ++i) acc[i] = 0.f;
  const float* db = dy + (int64_t)b * dy_bs;
  constexpr int UC = 4;                         // channels in flight
  for (int c0 = 0; c0 < C; c0 += UC) {
    float v[UC][NW];
#pragma unroll
    for (int u = 0; u < UC; ++u) {
      const float* dc = db + (int64_t)min(c0 + u, C - 1) * dy_cs;
#pragma unroll
      for (int j = 0; j < NW; ++j) {
        const int t = tb - H + j;
        const float r = dc[min(max(t, 0), Lt - 1)];
        v[u][j] = (t >= 0 && t < Lt && c0 + u < C) ? ag_rbf_if(r, rb) : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < UC; ++u) {
      const float* wc = ws + (size_t)min(c0 + u, C - 1) * K;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const float wk = wc[k];
#pragma unroll
        for (int j = 0; j < NW; ++j) {
          constexpr int dummy = 0; (void)dummy;
          const int idx = S * (j - H) + k;      // output index inside the thread's window (compile time after unrolling)
          if (idx >= 0 && idx < NO) acc[idx] += wk * v[u][j];
        }
      }
    }
  }
  float* xb = dx + (int64_t)b * dx_bs;
  const int u0 = S * tb - pad;
#pragma unroll
  for (int i = 0; i < NO; ++i) {
    const int u = u0 + i;
    if (u >= 0 && u < Lu) xb[u] = accumulate ? xb[u] + acc[i] : acc[i];
  }
}

// ---- backward-weight: a thread owns 4 consecutive steps (one 16-byte piece of dy per channel) of AG channels and walks
// over its share of the clips; AG*K sums per thread, reduced over the block once at the end, one partial per block
template <int S, int K, int AG>
__global__ __launch_bounds__(256) void conv_c1_wgrad4_kernel(const float* __restrict__ dy, int64_t dy_bs, int64_t dy_cs,
                                                             const float* __restrict__ x, int64_t x_bs,
                                                             float* __restrict__ part, int B, int A, int Lt, int Lx,
                                                             int pad, int bper, int rb) {
  __shared__ float red[4][AG * K];
  const int a0 = blockIdx.y * AG;
  const int t0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  const int b0 = blockIdx.z * bper, b1 = min(B, b0 + bper);
  constexpr int WN = C1Win<S, K>::N;
  float acc[AG][K];
#pragma unroll
  for (int i = 0; i < AG; ++i)
#pragma unroll
    for (int k = 0; k < K; ++k) acc[i][k] = 0.f;
  if (t0 < Lt) {
    const int g0 = S * t0 - pad;
    const bool full = t0 + 3 < Lt;
    for (int b = b0; b < b1; ++b) {
      const float* xb = x + (int64_t)b * x_bs;
      const float* db = dy + (int64_t)b * dy_bs + t0;
      f32x4 g[AG];
#pragma unroll
      for (int i = 0; i < AG; ++i) {
        const float* dc = db + (int64_t)min(a0 + i, A - 1) * dy_cs;
        if (full) {
          g[i] = *reinterpret_cast<const f32x4*>(dc);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) g[i][j] = (t0 + j < Lt) ? dc[j] : 0.f;
        }
      }
      float win[WN];
#pragma unroll
      for (int i = 0; i < WN; ++i) {
        const int q = g0 + i;
        const float v = xb[min(max(q, 0), Lx - 1)];
        win[i] = (q >= 0 && q < Lx) ? ag_rbf_if(v, rb) : 0.f;
      }
#pragma unroll
      for (int i = 0; i < AG; ++i) {
        if (rb) g[i] = ag_rbf4_if(g[i], 1);
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][k] += g[i][j] * win[S * j + k];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < AG; ++i)
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const float v = ag_wave_sum(acc[i][k]);
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][i * K + k] = v;
    }
  __syncthreads();
  if (threadIdx.x < AG * K) {
    const int i = threadIdx.x / K, k = threadIdx.x - i * K;
    if (a0 + i < A)
      part[((int64_t)(blockIdx.z * gridDim.x + blockIdx.x) * A + a0 + i) * K + k] =
          red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
  }
}

static bool c1_shape(int s, int K) { return s == 2 && K == 7; }

// Forward of a Conv1d with one input channel straight from the engine's arguments; returns 1 when it took the launch.
int ag_conv_c1_try_fwd(const ag_conv_args& a, int rb, hipStream_t st, int* rc) {
  if (!(a.mode == 0 && a.C == 1 && !a.res && !a.accumulate && c1_shape(a.stride, a.K) && a.act != AG_ACT_LEAKY_GATE)) return 0;
  if ((((uintptr_t)a.y & 15) != 0) || a.y_bs % 4 != 0 || a.y_cs % 4 != 0) return 0;
  const int opad = ag_roundup(a.O, 32);
  constexpr int OC = 4;
  hipLaunchKernelGGL((conv_c1_fwd_kernel<2, 7, OC>), dim3(ag_cdiv(a.Lout, 1024), a.B, ag_cdiv(a.O, OC)), dim3(256), 0, st, a.x,
                     a.x_bs, a.wp, opad, a.bias, a.y, a.y_bs, a.y_cs, a.lens_i64, a.O, a.Lin, a.Lout, a.pad, a.act, a.slope, rb);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { ag_set_error("ag_conv1d_engine(c1 fwd): launch failed: %s", hipGetErrorString(e)); *rc = AG_ERR_LAUNCH; }
  else *rc = AG_OK;
  return 1;
}

// Backward-data of such a conv: the engine's mode 1 call with ONE output channel (x = dy [B,C,Lt], y = dx [B,1,Lu]).
int ag_conv_c1_try_bwdx(const ag_conv_args& a, int rb, hipStream_t st, int* rc) {
  if (!(a.mode == 1 && a.O == 1 && !a.res && !a.bias && !a.lens_i64 && a.act == AG_ACT_NONE && c1_shape(a.stride, a.K))) return 0;
  if (a.C > 512) return 0;
  const int mt = ag_cdiv(a.K, a.stride);
  const int aligned = (a.wp_pad == a.pad && ag_scatter_aligned(a.K, a.stride, a.pad)) ? 1 : 0;
  const size_t lds = (size_t)a.C * a.K * sizeof(float);
  hipLaunchKernelGGL((conv_c1_bwdx_kernel<2, 7, 4>), dim3(ag_cdiv(a.Lin + 3, 1024), a.B), dim3(256), lds, st, a.x, a.x_bs, a.x_cs,
                     a.wp, mt, aligned, a.y, a.y_bs, a.accumulate, a.C, a.Lin, a.Lout, a.pad, rb);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { ag_set_error("ag_conv1d_engine(c1 bwd-data): launch failed: %s", hipGetErrorString(e)); *rc = AG_ERR_LAUNCH; }
  else *rc = AG_OK;
  return 1;
}

// Backward-weight: blocks / partials for (B, A, Lt); returns the number of partial slabs (0: shape not taken)
int ag_conv_c1_wgrad_slabs(int B, int A, int Lt, int s, int K, int* bper_out) {
  if (!c1_shape(s, K)) return 0;
  const int ag = 8;
  const int gx = ag_cdiv(Lt, 1024), gy = ag_cdiv(A, ag);
  int gz = ag_cdiv(512, gx * gy);
  if (gz > B) gz = B;
  if (gz < 1) gz = 1;
  const int bper = ag_cdiv(B, gz);
  gz = ag_cdiv(B, bper);
  if (bper_out) *bper_out = bper;
  return gx * gz;
}

int ag_conv_c1_wgrad(const float* dy, int64_t dy_bs, int64_t dy_cs, const float* x, int64_t x_bs, float* part, int B, int A,
                     int Lt, int Lx, int s, int K, int pad, int rb, hipStream_t st) {
  int bper = 1;
  const int slabs = ag_conv_c1_wgrad_slabs(B, A, Lt, s, K, &bper);
  const int gx = ag_cdiv(Lt, 1024), gz = slabs / gx;
  hipLaunchKernelGGL((conv_c1_wgrad4_kernel<2, 7, 8>), dim3(gx, ag_cdiv(A, 8), gz), dim3(256), 0, st, dy, dy_bs, dy_cs, x, x_bs,
                     part, B, A, Lt, Lx, pad, bper, rb);
  AG_CHECK_LAUNCH("ag_conv1d_wgrad(c1)");
  return AG_OK;
}
