// pointwise.hip -- weight norm, LSTM cell pointwise, masked BCE, activations and the
// fused optimiser.  Each kernel names the reference lines it replaces.
#include "common.h"

// ------------------------------------------------------------------------------------------
// weight norm (audiogan.py:77-80): one WAVE per row, rows reduced with wavefront shuffles;
// a table of tensors is handled by one launch (grid.y = tensor).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void weight_norm_fwd_kernel(const ag_wn_desc* __restrict__ descs) {
  const ag_wn_desc d = descs[blockIdx.y];
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= d.rows) return;
  const float* v = d.v + (int64_t)r * d.cols;
  float ss = 0.f;
  for (int i = lane; i < d.cols; i += 64) {
    const float x = v[i];
    ss += x * x;
  }
  ss = ag_wave_sum(ss);
  const float nrm = sqrtf(ss);
  const float sc = d.g[r] / nrm;
  if (lane == 0 && d.inv_norm) d.inv_norm[r] = 1.f / nrm;
  const int d0p32 = (d.rows + 31) / 32 * 32;
  const int s = d.stride > 0 ? d.stride : 1;
  const int mt = (d.K + s - 1) / s;
  const int mp = (d.d1 * s + 31) / 32 * 32;
  const bool aligned = d.wpb && ag_scatter_aligned(d.K, s, d.pad);
  // bf16 images behind the fp32 layouts (common.h ag_wq_*): what the bf16 conv kernel stages
  unsigned short* qa = d.wpa ? reinterpret_cast<unsigned short*>(d.wpa + ag_wq_offset((d.d1 + 1) & ~1, d.K, d0p32)) : nullptr;
  unsigned short* qb = d.wpb ? reinterpret_cast<unsigned short*>(d.wpb + ag_wq_offset((d.rows + 1) & ~1, mt, mp)) : nullptr;
  for (int i = lane; i < d.cols; i += 64) {
    const float w = v[i] * sc;
    const unsigned short wq = (unsigned short)ag_pack_bf16(w, w);
    if (d.w) d.w[(int64_t)r * d.cols + i] = w;
    const int kk = d.K > 0 ? d.K : 1;
    const int o = i / kk, k = i - o * kk;          // i = c*K + k
    if (d.wpa) {
      d.wpa[(int64_t)i * d0p32 + r] = w;
      qa[ag_wq_index(o, k, r, d.K, d0p32)] = wq;
    }
    if (d.wpb) {
      int m = k / s;
      const int rr = k - m * s;
      if (aligned) m += ag_scatter_shift(s, d.pad, rr);
      d.wpb[((int64_t)r * mt + m) * mp + o * s + rr] = w;
      qb[ag_wq_index(r, m, o * s + rr, mt, mp)] = wq;
    }
  }
}

extern "C" int ag_weight_norm_fwd(const ag_wn_desc* descs_dev, int n, int max_rows, void* stream) {
  AG_REQUIRE(descs_dev && n > 0 && max_rows > 0, "ag_weight_norm_fwd: bad args");
  AG_REQUIRE(n <= 65535, "ag_weight_norm_fwd: too many tensors");
  hipLaunchKernelGGL(weight_norm_fwd_kernel, dim3(ag_cdiv(max_rows, 4), n), dim3(256), 0,
                     (hipStream_t)stream, descs_dev);
  AG_CHECK_LAUNCH("ag_weight_norm_fwd");
  return AG_OK;
}

__global__ __launch_bounds__(256) void weight_norm_bwd_kernel(const ag_wn_bwd_desc* __restrict__ descs) {
  const ag_wn_bwd_desc d = descs[blockIdx.y];
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= d.rows) return;
  const float* v = d.v + (int64_t)r * d.cols;
  const float* dw = d.dw + (int64_t)r * d.cols;
  float ss = 0.f, dot = 0.f;
  for (int i = lane; i < d.cols; i += 64) {
    const float x = v[i];
    ss += x * x;
    dot += x * dw[i];
  }
  ss = ag_wave_sum(ss);
  dot = ag_wave_sum(dot);
  const float inv = 1.f / sqrtf(ss);
  const float g = d.g[r];
  const float a = g * inv, bcoef = dot * inv * inv;
  float* dv = d.dv + (int64_t)r * d.cols;
  if (d.accumulate) {
    if (lane == 0) d.dg[r] += dot * inv;
    for (int i = lane; i < d.cols; i += 64) dv[i] += a * (dw[i] - v[i] * bcoef);
    return;
  }
  if (lane == 0) d.dg[r] = dot * inv;
  for (int i = lane; i < d.cols; i += 64) dv[i] = a * (dw[i] - v[i] * bcoef);
}

extern "C" int ag_weight_norm_bwd(const ag_wn_bwd_desc* descs_dev, int n, int max_rows, void* stream) {
  AG_REQUIRE(descs_dev && n > 0 && max_rows > 0, "ag_weight_norm_bwd: bad args");
  AG_REQUIRE(n <= 65535, "ag_weight_norm_bwd: too many tensors");
  hipLaunchKernelGGL(weight_norm_bwd_kernel, dim3(ag_cdiv(max_rows, 4), n), dim3(256), 0,
                     (hipStream_t)stream, descs_dev);
  AG_CHECK_LAUNCH("ag_weight_norm_bwd");
  return AG_OK;
}

// ------------------------------------------------------------------------------------------
// LSTM cell pointwise (audiogan.py:380,440-442 NN.LSTMCell; :498 NN.LSTM), gate order i|f|g|o
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lstm_cell_fwd_kernel(
    float* __restrict__ gates, int ldg, const float* __restrict__ c_prev, int ldcp,
    float* __restrict__ h_out, int ldh, float* __restrict__ c_out, int ldc, float* __restrict__ y_out,
    int ldy, const float* __restrict__ h_prev, int ldhp, const int64_t* __restrict__ valid, int t, int B,
    int H) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (j >= H) return;
  float* gr = gates + (int64_t)b * ldg;
  const float cp = c_prev[(int64_t)b * ldcp + j];
  if (valid && t >= valid[b]) {
    // padded step of a packed sequence: state is carried, output is zero
    if (h_out) h_out[(int64_t)b * ldh + j] = h_prev ? h_prev[(int64_t)b * ldhp + j] : 0.f;
    c_out[(int64_t)b * ldc + j] = cp;
    if (y_out) y_out[(int64_t)b * ldy + j] = 0.f;
    return;
  }
  const float ig = ag_sigmoid(gr[j]);
  const float fg = ag_sigmoid(gr[H + j]);
  const float gg = tanhf(gr[2 * H + j]);
  const float og = ag_sigmoid(gr[3 * H + j]);
  const float cn = fg * cp + ig * gg;
  const float hn = og * tanhf(cn);
  gr[j] = ig;
  gr[H + j] = fg;
  gr[2 * H + j] = gg;
  gr[3 * H + j] = og;
  c_out[(int64_t)b * ldc + j] = cn;
  if (h_out) h_out[(int64_t)b * ldh + j] = hn;
  if (y_out) y_out[(int64_t)b * ldy + j] = hn;
}

extern "C" int ag_lstm_cell_fwd(float* gates, int ldg, const float* c_prev, int ldcp, float* h_out,
                                int ldh, float* c_out, int ldc, float* y_out, int ldy,
                                const float* h_prev, int ldhp, const int64_t* valid_i64, int t, int B,
                                int H, void* stream) {
  AG_REQUIRE(gates && c_prev && c_out && (h_out || y_out), "ag_lstm_cell_fwd: null tensor");
  AG_REQUIRE(B > 0 && H > 0 && B <= 65535 && ldg >= 4 * H, "ag_lstm_cell_fwd: bad shape");
  hipLaunchKernelGGL(lstm_cell_fwd_kernel, dim3(ag_cdiv(H, 256), B), dim3(256), 0, (hipStream_t)stream,
                     gates, ldg, c_prev, ldcp, h_out, ldh, c_out, ldc, y_out, ldy, h_prev, ldhp,
                     valid_i64, t, B, H);
  AG_CHECK_LAUNCH("ag_lstm_cell_fwd");
  return AG_OK;
}

__global__ __launch_bounds__(256) void lstm_cell_bwd_kernel(
    const float* __restrict__ ga, int ldg, const float* __restrict__ c_prev, int ldcp,
    const float* __restrict__ c_new, int ldc, const float* __restrict__ dh, int lddh,
    const float* __restrict__ dy, int lddy, const float* __restrict__ dc_next, int lddcn,
    float* __restrict__ dgates, int lddg,
    float* __restrict__ dc_prev, int lddcp, float* __restrict__ dh_pass, int lddhp,
    const int64_t* __restrict__ valid, int t, int B, int H) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (j >= H) return;
  float* dg = dgates + (int64_t)b * lddg;
  const float dhf = dh ? dh[(int64_t)b * lddh + j] : 0.f;
  const float dcn = dc_next ? dc_next[(int64_t)b * lddcn + j] : 0.f;
  if (valid && t >= valid[b]) {
    dg[j] = 0.f;
    dg[H + j] = 0.f;
    dg[2 * H + j] = 0.f;
    dg[3 * H + j] = 0.f;
    dc_prev[(int64_t)b * lddcp + j] = dcn;
    if (dh_pass) dh_pass[(int64_t)b * lddhp + j] = dhf;
    return;
  }
  const float dhv = dhf + (dy ? dy[(int64_t)b * lddy + j] : 0.f);
  const float* gr = ga + (int64_t)b * ldg;
  const float ig = gr[j], fg = gr[H + j], gg = gr[2 * H + j], og = gr[3 * H + j];
  const float cp = c_prev[(int64_t)b * ldcp + j];
  const float tc = tanhf(c_new[(int64_t)b * ldc + j]);
  const float dc = dcn + dhv * og * (1.f - tc * tc);
  dg[j] = dc * gg * ig * (1.f - ig);
  dg[H + j] = dc * cp * fg * (1.f - fg);
  dg[2 * H + j] = dc * ig * (1.f - gg * gg);
  dg[3 * H + j] = dhv * tc * og * (1.f - og);
  dc_prev[(int64_t)b * lddcp + j] = dc * fg;
  if (dh_pass) dh_pass[(int64_t)b * lddhp + j] = 0.f;
}

extern "C" int ag_lstm_cell_bwd(const float* gates_act, int ldg, const float* c_prev, int ldcp,
                                const float* c_new, int ldc, const float* dh, int lddh,
                                const float* dy, int lddy, const float* dc_next, int lddcn,
                                float* dgates, int lddg,
                                float* dc_prev, int lddcp, float* dh_pass, int lddhp,
                                const int64_t* valid_i64, int t, int B, int H, void* stream) {
  AG_REQUIRE(gates_act && c_prev && c_new && (dh || dy) && dgates && dc_prev,
             "ag_lstm_cell_bwd: null tensor");
  AG_REQUIRE(B > 0 && H > 0 && B <= 65535, "ag_lstm_cell_bwd: bad shape");
  hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3(ag_cdiv(H, 256), B), dim3(256), 0, (hipStream_t)stream,
                     gates_act, ldg, c_prev, ldcp, c_new, ldc, dh, lddh, dy, lddy, dc_next, lddcn, dgates,
                     lddg,
                     dc_prev, lddcp, dh_pass, lddhp, valid_i64, t, B, H);
  AG_CHECK_LAUNCH("ag_lstm_cell_bwd");
  return AG_OK;
}

// ------------------------------------------------------------------------------------------
// masked BCE-with-logits (audiogan.py:187-197, :204-211): one wave per sample
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bce_fwd_kernel(const float* __restrict__ x, int ldx, float target,
                                                      const int64_t* __restrict__ nfr,
                                                      float* __restrict__ per, float* __restrict__ loss,
                                                      float scale, int B, int T) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  int64_t n = nfr ? nfr[b] : T;
  const int lim = n < T ? (int)n : T;
  float s = 0.f;
  for (int t = lane; t < lim; t += 64) {
    const float v = x[(int64_t)b * ldx + t];
    const float m = fmaxf(-v, 0.f);
    s += v - v * target + m + logf(expf(-m) + expf(-v - m));
  }
  s = ag_wave_sum(s);
  if (lane == 0) per[b] = s;
}

// loss[0] += scale * sum_b per[b] / n[b], in a fixed order (one wave)
__global__ __launch_bounds__(64) void bce_finish_kernel(const float* __restrict__ per, const int64_t* __restrict__ nfr,
                                                        float* __restrict__ loss, float scale, int B, int T) {
  float s = 0.f;
  for (int b = threadIdx.x; b < B; b += 64) s += per[b] / (float)(nfr ? nfr[b] : (int64_t)T);
  s = ag_wave_sum(s);
  if (threadIdx.x == 0) loss[0] += scale * s;
}

extern "C" int ag_bce_logits_fwd(const float* x, int ldx, float target, const int64_t* nframes_i64,
                                 float* per_sample, float* loss, float scale, int B, int T,
                                 void* stream) {
  AG_REQUIRE(x && per_sample && B > 0 && T > 0 && ldx >= T, "ag_bce_logits_fwd: bad args (per_sample is required)");
  hipLaunchKernelGGL(bce_fwd_kernel, dim3(ag_cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream, x, ldx,
                     target, nframes_i64, per_sample, loss, scale, B, T);
  AG_CHECK_LAUNCH("ag_bce_logits_fwd");
  if (loss) {
    hipLaunchKernelGGL(bce_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, per_sample, nframes_i64, loss,
                       scale, B, T);
    AG_CHECK_LAUNCH("ag_bce_logits_fwd(finish)");
  }
  return AG_OK;
}

__global__ __launch_bounds__(256) void bce_bwd_kernel(const float* __restrict__ x, int ldx, float target,
                                                      const int64_t* __restrict__ nfr,
                                                      const float* __restrict__ gscale, float scale,
                                                      float* __restrict__ dx, int lddx, int B, int T) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (t >= T) return;
  const int64_t n = nfr ? nfr[b] : T;
  float g = 0.f;
  if (t < n) {
    const float gs = gscale ? gscale[0] : 1.f;
    g = gs * scale / (float)n * (ag_sigmoid(x[(int64_t)b * ldx + t]) - target);
  }
  dx[(int64_t)b * lddx + t] = g;
}

extern "C" int ag_bce_logits_bwd(const float* x, int ldx, float target, const int64_t* nframes_i64,
                                 const float* gscale_dev, float scale, float* dx, int lddx, int B, int T,
                                 void* stream) {
  AG_REQUIRE(x && dx && B > 0 && T > 0 && B <= 65535, "ag_bce_logits_bwd: bad args");
  hipLaunchKernelGGL(bce_bwd_kernel, dim3(ag_cdiv(T, 256), B), dim3(256), 0, (hipStream_t)stream, x, ldx,
                     target, nframes_i64, gscale_dev, scale, dx, lddx, B, T);
  AG_CHECK_LAUNCH("ag_bce_logits_bwd");
  return AG_OK;
}

// The same loss in ONE launch on logits of any (row, column) pitch, with an optional target per row: the critic iteration
// scores real and fake clips in one pass (targets 0.9 / 0 per row), and the logits arrive as a transposed view
// [B,T'] of the heads' [T',B] rows - no contiguous copy, no zero fill, no finishing launch, no slicing in autograd.
//   per[b] = sum_{t < n_b} bce(x[b,t], tgt_b);   loss[0] = scale * sum_b per[b] / n_b      (written, not accumulated)
// One workgroup: wave w takes rows w, w + 16, ...; the 16 partial sums are added in wave order (fixed order).
__global__ __launch_bounds__(1024) void bce_fwd_one_kernel(const float* __restrict__ x, int64_t sxb, int64_t sxt, float target,
                                                           const float* __restrict__ tgt_rows,
                                                           const int64_t* __restrict__ nfr, float* __restrict__ per,
                                                           float* __restrict__ loss, float scale, int B, int T) {
  __shared__ float wsum[16];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  float acc = 0.f;
  for (int b = wid; b < B; b += 16) {
    const int64_t n = nfr ? nfr[b] : T;
    const int lim = n < T ? (int)n : T;
    const float tg = tgt_rows ? tgt_rows[b] : target;
    float s = 0.f;
    for (int t = lane; t < lim; t += 64) {
      const float v = x[(int64_t)b * sxb + (int64_t)t * sxt];
      const float m = fmaxf(-v, 0.f);
      s += v - v * tg + m + logf(expf(-m) + expf(-v - m));
    }
    s = ag_wave_sum(s);
    if (lane == 0 && per) per[b] = s;
    acc += s / (float)n;
  }
  if (lane == 0) wsum[wid] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < 16; ++w) t += wsum[w];
    loss[0] = scale * t;
  }
}

extern "C" int ag_bce_logits_fwd_strided(const float* x, int64_t sxb, int64_t sxt, float target, const float* target_rows,
                                         const int64_t* nframes_i64, float* per_sample, float* loss, float scale, int B,
                                         int T, void* stream) {
  AG_REQUIRE(x && loss && B > 0 && T > 0, "ag_bce_logits_fwd_strided: bad args");
  hipLaunchKernelGGL(bce_fwd_one_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, sxb, sxt, target, target_rows,
                     nframes_i64, per_sample, loss, scale, B, T);
  AG_CHECK_LAUNCH("ag_bce_logits_fwd_strided");
  return AG_OK;
}

__global__ __launch_bounds__(256) void bce_bwd_strided_kernel(const float* __restrict__ x, int64_t sxb, int64_t sxt,
                                                              float target, const float* __restrict__ tgt_rows,
                                                              const int64_t* __restrict__ nfr,
                                                              const float* __restrict__ gscale, float scale,
                                                              float* __restrict__ dx, int64_t sdb, int64_t sdt, int B, int T,
                                                              int tfast) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)B * T) return;
  // consecutive threads walk whichever index is contiguous in dx
  const int b = tfast ? (int)(i / T) : (int)(i % B), t = tfast ? (int)(i % T) : (int)(i / B);
  const int64_t n = nfr ? nfr[b] : T;
  float g = 0.f;
  if (t < n) {
    const float gs = gscale ? gscale[0] : 1.f;
    g = gs * scale / (float)n * (ag_sigmoid(x[(int64_t)b * sxb + (int64_t)t * sxt]) - (tgt_rows ? tgt_rows[b] : target));
  }
  dx[(int64_t)b * sdb + (int64_t)t * sdt] = g;
}

extern "C" int ag_bce_logits_bwd_strided(const float* x, int64_t sxb, int64_t sxt, float target, const float* target_rows,
                                         const int64_t* nframes_i64, const float* gscale_dev, float scale, float* dx,
                                         int64_t sdb, int64_t sdt, int B, int T, void* stream) {
  AG_REQUIRE(x && dx && B > 0 && T > 0, "ag_bce_logits_bwd_strided: bad args");
  const int64_t n = (int64_t)B * T;
  AG_REQUIRE(ag_cdiv64(n, 256) < ((int64_t)1 << 31), "ag_bce_logits_bwd_strided: too large");
  hipLaunchKernelGGL(bce_bwd_strided_kernel, dim3((unsigned)ag_cdiv64(n, 256)), dim3(256), 0, (hipStream_t)stream, x, sxb,
                     sxt, target, target_rows, nframes_i64, gscale_dev, scale, dx, sdb, sdt, B, T, sdt == 1 ? 1 : 0);
  AG_CHECK_LAUNCH("ag_bce_logits_bwd_strided");
  return AG_OK;
}

// ------------------------------------------------------------------------------------------
// activations / axpby on contiguous buffers
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                      int64_t n, int act, float slope) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    y[i] = ag_apply_act(x[i], act, slope);
}
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dy,
                                                      const float* __restrict__ y, float* __restrict__ dx,
                                                      int64_t n, int act, float slope) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float o = y[i];
    float g = dy[i];
    if (act == AG_ACT_LEAKY) g = o > 0.f ? g : g * slope;
    else if (act == AG_ACT_TANH) g = g * (1.f - o * o);
    dx[i] = g;
  }
}
__global__ __launch_bounds__(256) void axpby_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                    int64_t n, float a, float b) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    y[i] = a * x[i] + (b == 0.f ? 0.f : b * y[i]);
}

static unsigned ew_grid(int64_t n) {
  int64_t g = ag_cdiv64(n, 256);
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (unsigned)g;
}

extern "C" int ag_act_fwd(const float* x, float* y, int64_t n, int act, float slope, void* stream) {
  AG_REQUIRE(x && y && n > 0, "ag_act_fwd: bad args");
  hipLaunchKernelGGL(act_fwd_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, y, n, act,
                     slope);
  AG_CHECK_LAUNCH("ag_act_fwd");
  return AG_OK;
}
extern "C" int ag_act_bwd(const float* dy, const float* y, float* dx, int64_t n, int act, float slope,
                          void* stream) {
  AG_REQUIRE(dy && y && dx && n > 0, "ag_act_bwd: bad args");
  hipLaunchKernelGGL(act_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, dy, y, dx, n,
                     act, slope);
  AG_CHECK_LAUNCH("ag_act_bwd");
  return AG_OK;
}
extern "C" int ag_axpby(const float* x, float* y, int64_t n, float a, float b, void* stream) {
  AG_REQUIRE(x && y && n > 0, "ag_axpby: bad args");
  hipLaunchKernelGGL(axpby_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, y, n, a, b);
  AG_CHECK_LAUNCH("ag_axpby");
  return AG_OK;
}

// ------------------------------------------------------------------------------------------
// fused optimiser (audiogan.py:232-253 check_grad/clip_grad, :693-694 RMSprop;
// computation_graph.py:58-59 Adam).  grid.y = tensor, grid.x = chunk.
// ------------------------------------------------------------------------------------------
#define OPT_CHUNKS 256

__global__ __launch_bounds__(256) void grad_norms_kernel(const ag_opt_desc* __restrict__ descs, float gscale,
                                                         float* __restrict__ part, int32_t* __restrict__ step_dev) {
  if (step_dev && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) step_dev[0] += 1;   // (read by ag_opt_step only)
  // part: [n][OPT_CHUNKS] partial sums of squares, then [n][OPT_CHUNKS] flag words (as floats' bit patterns) - every slot
  // is written by its workgroup, so nothing needs zeroing in front of the launch
  __shared__ float red[17];
  __shared__ int fsh;
  const ag_opt_desc d = descs[blockIdx.y];
  float* fpart = part + (int64_t)gridDim.y * OPT_CHUNKS;
  const int64_t slot = (int64_t)blockIdx.y * OPT_CHUNKS + blockIdx.x;
  if ((int64_t)blockIdx.x * 256 >= d.n) {  // uniform per block (a longer tensor's 16-byte path needs no more chunks either)
    if (threadIdx.x == 0) { part[slot] = 0.f; fpart[slot] = __int_as_float(0); }
    return;
  }
  if (threadIdx.x == 0) fsh = 0;
  float s = 0.f;
  int f = 0;
  // (a workgroup's elements and their order do not depend on the path: chunk c owns elements [256 c, 256 c + 256) of every
  // stripe of 256 * gridDim.x; the 16-byte path reads the same elements four at a time when the tensor is aligned)
  if ((((uintptr_t)d.grad) & 15) == 0 && (d.n & 3) == 0) {
    const int64_t n4 = d.n >> 2;
    const f32x4* g4 = reinterpret_cast<const f32x4*>(d.grad);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
      const f32x4 v = g4[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float g = v[e] * gscale;
        s += g * g;
        if (g != g) f |= AG_FLAG_NAN;
        if (fabsf(g) > 1e5f) f |= AG_FLAG_BIG;
      }
    }
  } else
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < d.n; i += (int64_t)gridDim.x * 256) {
    const float g = d.grad[i] * gscale;
    s += g * g;
    if (g != g) f |= AG_FLAG_NAN;
    if (fabsf(g) > 1e5f) f |= AG_FLAG_BIG;
  }
  s = ag_block_sum(s, red);             // (its first barrier also orders the fsh = 0 above)
  if (f) atomicOr(&fsh, f);             // LDS, integer: order-independent
  __syncthreads();
  if (threadIdx.x == 0) { part[slot] = s; fpart[slot] = __int_as_float(fsh); }
}

// norms of all tensors from the per-chunk partials; one workgroup.  Shared by the stand-alone finishing launch of
// ag_grad_norms and by workgroup (0, 0) of ag_opt_step's fused form.
__device__ __forceinline__ void grad_norms_finish(float* __restrict__ sq_to_norm, float* __restrict__ norm_sum,
                                                  int32_t* __restrict__ flags, int n, const float* __restrict__ part);

__global__ void grad_norms_finish_kernel(float* __restrict__ sq_to_norm, float* __restrict__ norm_sum, int32_t* __restrict__ flags,
                                         int n, const float* __restrict__ part) {
  grad_norms_finish(sq_to_norm, norm_sum, flags, n, part);
}

__device__ __forceinline__ void grad_norms_finish(float* __restrict__ sq_to_norm, float* __restrict__ norm_sum,
                                                  int32_t* __restrict__ flags, int n, const float* __restrict__ part) {
  // single block: norms[i] = sqrt(sum of tensor i's per-chunk partials in chunk order); norm_sum = sum_i norms[i]
  // (deterministic order); flags = OR of every chunk's flag word (WRITTEN, not accumulated)
  __shared__ float red[17];
  __shared__ int fsh;
  if (threadIdx.x == 0) fsh = 0;
  __syncthreads();
  const float* fpart = part + (int64_t)n * OPT_CHUNKS;
  float s = 0.f;
  int f = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    float q = 0.f;
    for (int c = 0; c < OPT_CHUNKS; ++c) {
      q += part[(int64_t)i * OPT_CHUNKS + c];
      f |= __float_as_int(fpart[(int64_t)i * OPT_CHUNKS + c]);
    }
    const float nr = sqrtf(q);
    sq_to_norm[i] = nr;
    s += nr;
  }
  if (f) atomicOr(&fsh, f);
  s = ag_block_sum(s, red);
  if (threadIdx.x == 0 && norm_sum) norm_sum[0] = s;
  if (threadIdx.x == 0 && flags) flags[0] = fsh;
}

extern "C" int ag_grad_norms(const ag_opt_desc* descs_dev, int n, float* norms, float* norm_sum,
                             int32_t* flags, float grad_scale, int32_t* step_dev, int finish, void* stream) {
  const AgWs ws = ag_ws_take();     // FIRST: an argument error below must not leave a stale binding behind
  AG_REQUIRE(descs_dev && norms && n > 0 && n <= 65535, "ag_grad_norms: bad args");
  AG_REQUIRE(ws.p && ws.numel >= (int64_t)2 * n * OPT_CHUNKS,
             "ag_grad_norms: bind a workspace of >= %lld floats (ag_bind_workspace): per-chunk sums of squares and flag words, "
             "summed in a fixed order (there is no float-atomic accumulation)", (long long)2 * n * OPT_CHUNKS);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(grad_norms_kernel, dim3(OPT_CHUNKS, n), dim3(256), 0, st, descs_dev, grad_scale, ws.p, step_dev);
  AG_CHECK_LAUNCH("ag_grad_norms");
  if (finish) {
    hipLaunchKernelGGL(grad_norms_finish_kernel, dim3(1), dim3(256), 0, st, norms, norm_sum, flags, n, ws.p);
    AG_CHECK_LAUNCH("ag_grad_norms(finish)");
  }
  return AG_OK;
}

__global__ __launch_bounds__(256) void opt_step_kernel(const ag_opt_desc* __restrict__ descs,
                                                       const float* __restrict__ norms, int kind, float lr,
                                                       float clip, float gscale, float a1, float b2,
                                                       float eps, float bc1, float bc2sqrt,
                                                       const int32_t* __restrict__ step_dev, const float* __restrict__ part,
                                                       float* __restrict__ norms_out, float* __restrict__ norm_sum,
                                                       int32_t* __restrict__ flags) {
  const ag_opt_desc d = descs[blockIdx.y];
  float nr;
  if (part) {
    // fused form: ag_grad_norms(finish = 0) left the per-chunk partials; every workgroup sums its tensor's 256 partials
    // itself (chunk order: the same value in every workgroup, and the same as the stand-alone finish computes), and
    // workgroup (0, 0) also writes the outputs the finishing launch would have written
    __shared__ float nsh;
    if (blockIdx.x == 0 && blockIdx.y == 0) grad_norms_finish(norms_out, norm_sum, flags, gridDim.y, part);
    if (threadIdx.x == 0) {
      float q = 0.f;
      for (int c = 0; c < OPT_CHUNKS; ++c) q += part[(int64_t)blockIdx.y * OPT_CHUNKS + c];
      nsh = sqrtf(q);
    }
    __syncthreads();
    nr = nsh;
  } else {
    nr = norms[blockIdx.y];
  }
  if (step_dev && kind == AG_OPT_ADAM) {
    const float st = (float)step_dev[0];
    bc1 = 1.f - powf(a1, st);
    bc2sqrt = sqrtf(1.f - powf(b2, st));
  }
  if ((int64_t)blockIdx.x * 256 >= d.n) return;
  const bool do_clip = clip > 0.f && nr > clip;
  const float div = do_clip ? nr / clip : 1.f;  // reference: grad /= (norm / clip), audiogan.py:252
  auto upd = [&](float gr, float& pv, float& m1, float& m2) {
    float g = gr * gscale;
    if (do_clip) g = g / div;
    if (kind == AG_OPT_RMSPROP) {
      m1 = a1 * m1 + (1.f - a1) * g * g;
      pv = pv - lr * (g / (sqrtf(m1) + eps));
    } else {
      m1 = a1 * m1 + (1.f - a1) * g;
      m2 = b2 * m2 + (1.f - b2) * g * g;
      pv = pv - (lr / bc1) * (m1 / (sqrtf(m2) / bc2sqrt + eps));
    }
  };
  const bool adam = kind == AG_OPT_ADAM;
  // 16-byte path: the same arithmetic per element, four elements per load / store (this pass moves 7 floats per element;
  // scalar accesses ran it at 3 TB/s)
  if (((((uintptr_t)d.p | (uintptr_t)d.grad | (uintptr_t)d.s1 | (adam ? (uintptr_t)d.s2 : 0)) & 15) == 0) && (d.n & 3) == 0) {
    const int64_t n4 = d.n >> 2;
    f32x4* p4 = reinterpret_cast<f32x4*>(d.p);
    const f32x4* g4 = reinterpret_cast<const f32x4*>(d.grad);
    f32x4* s14 = reinterpret_cast<f32x4*>(d.s1);
    f32x4* s24 = reinterpret_cast<f32x4*>(d.s2);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
      const f32x4 gv = g4[i];
      f32x4 pv = p4[i], m1 = s14[i], m2 = adam ? s24[i] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a = pv[e], b = m1[e], c = m2[e];
        upd(gv[e], a, b, c);
        pv[e] = a; m1[e] = b; m2[e] = c;
      }
      p4[i] = pv; s14[i] = m1;
      if (adam) s24[i] = m2;
    }
    return;
  }
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < d.n; i += (int64_t)gridDim.x * 256) {
    float pv = d.p[i], m1 = d.s1[i], m2 = adam ? d.s2[i] : 0.f;
    upd(d.grad[i], pv, m1, m2);
    d.p[i] = pv; d.s1[i] = m1;
    if (adam) d.s2[i] = m2;
  }
}

extern "C" int ag_opt_step(const ag_opt_desc* descs_dev, int n, const float* norms, int kind, float lr,
                           float clip, float grad_scale, float alpha_or_beta1, float beta2, float eps,
                           int step, const int32_t* step_dev, const float* part, float* norms_out, float* norm_sum,
                           int32_t* flags, void* stream) {
  AG_REQUIRE(descs_dev && (norms || part) && n > 0 && n <= 65535, "ag_opt_step: bad args");
  AG_REQUIRE(!part || norms_out, "ag_opt_step: the fused form (part != NULL) writes the norms: norms_out must be given");
  AG_REQUIRE(kind == AG_OPT_RMSPROP || kind == AG_OPT_ADAM, "ag_opt_step: bad optimiser kind");
  float bc1 = 1.f, bc2s = 1.f;
  if (kind == AG_OPT_ADAM && !step_dev) {
    AG_REQUIRE(step >= 1, "ag_opt_step: Adam step must be >= 1");
    bc1 = (float)(1.0 - pow((double)alpha_or_beta1, (double)step));
    bc2s = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  }
  hipLaunchKernelGGL(opt_step_kernel, dim3(OPT_CHUNKS, n), dim3(256), 0, (hipStream_t)stream, descs_dev,
                     norms, kind, lr, clip, grad_scale, alpha_or_beta1, beta2, eps, bc1, bc2s, step_dev, part, norms_out,
                     norm_sum, flags);
  AG_CHECK_LAUNCH("ag_opt_step");
  return AG_OK;
}

// ------------------------------------------------------------------------------------------
// GRU cell pointwise (BASELINE config C4; torch.nn.GRUCell semantics, gate order r|z|n):
//   r = s(gi_r + gh_r), z = s(gi_z + gh_z), n = tanh(gi_n + r * gh_n), h' = (1-z) n + z h
// gi = x W_ih^T + b_ih, gh = h W_hh^T + b_hh are complete on entry.  fwd overwrites gi with the
// activated (r, z, n); gh keeps gh_n for backward.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gru_cell_fwd_kernel(float* __restrict__ gi, const float* __restrict__ gh,
                                                           const float* __restrict__ h_prev, int ldhp,
                                                           float* __restrict__ h_out, int ldh, int B, int H) {
  const int j = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (j >= H) return;
  float* a = gi + (int64_t)b * 3 * H;
  const float* c = gh + (int64_t)b * 3 * H;
  const float r = ag_sigmoid(a[j] + c[j]);
  const float z = ag_sigmoid(a[H + j] + c[H + j]);
  const float n = tanhf(a[2 * H + j] + r * c[2 * H + j]);
  const float hp = h_prev[(int64_t)b * ldhp + j];
  a[j] = r;
  a[H + j] = z;
  a[2 * H + j] = n;
  h_out[(int64_t)b * ldh + j] = (1.f - z) * n + z * hp;
}

extern "C" int ag_gru_cell_fwd(float* gi, const float* gh, const float* h_prev, int ldhp, float* h_out,
                               int ldh, int B, int H, void* stream) {
  AG_REQUIRE(gi && gh && h_prev && h_out && B > 0 && H > 0 && B <= 65535, "ag_gru_cell_fwd: bad args");
  hipLaunchKernelGGL(gru_cell_fwd_kernel, dim3(ag_cdiv(H, 256), B), dim3(256), 0, (hipStream_t)stream, gi, gh,
                     h_prev, ldhp, h_out, ldh, B, H);
  AG_CHECK_LAUNCH("ag_gru_cell_fwd");
  return AG_OK;
}

// dgi = (dr_pre, dz_pre, dn_pre), dgh = (dr_pre, dz_pre, dn_pre * r), dh_prev = dh * z (direct path)
__global__ __launch_bounds__(256) void gru_cell_bwd_kernel(const float* __restrict__ ga, const float* __restrict__ gh,
                                                           const float* __restrict__ h_prev, int ldhp,
                                                           const float* __restrict__ dh, int lddh,
                                                           float* __restrict__ dgi, float* __restrict__ dgh,
                                                           float* __restrict__ dh_prev, int lddhp, int B, int H) {
  const int j = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (j >= H) return;
  const float* a = ga + (int64_t)b * 3 * H;
  const float r = a[j], z = a[H + j], n = a[2 * H + j];
  const float hn = gh[(int64_t)b * 3 * H + 2 * H + j];
  const float hp = h_prev[(int64_t)b * ldhp + j];
  const float g = dh[(int64_t)b * lddh + j];
  const float dn_pre = g * (1.f - z) * (1.f - n * n);
  const float dz_pre = g * (hp - n) * z * (1.f - z);
  const float dr_pre = dn_pre * hn * r * (1.f - r);
  float* di = dgi + (int64_t)b * 3 * H;
  float* dhh = dgh + (int64_t)b * 3 * H;
  di[j] = dr_pre; di[H + j] = dz_pre; di[2 * H + j] = dn_pre;
  dhh[j] = dr_pre; dhh[H + j] = dz_pre; dhh[2 * H + j] = dn_pre * r;
  dh_prev[(int64_t)b * lddhp + j] = g * z;
}

extern "C" int ag_gru_cell_bwd(const float* gates_act, const float* gh, const float* h_prev, int ldhp,
                               const float* dh, int lddh, float* dgi, float* dgh, float* dh_prev, int lddhp,
                               int B, int H, void* stream) {
  AG_REQUIRE(gates_act && gh && h_prev && dh && dgi && dgh && dh_prev && B > 0 && H > 0 && B <= 65535,
             "ag_gru_cell_bwd: bad args");
  hipLaunchKernelGGL(gru_cell_bwd_kernel, dim3(ag_cdiv(H, 256), B), dim3(256), 0, (hipStream_t)stream, gates_act,
                     gh, h_prev, ldhp, dh, lddh, dgi, dgh, dh_prev, lddhp, B, H);
  AG_CHECK_LAUNCH("ag_gru_cell_bwd");
  return AG_OK;
}

// dx[r, c] = dy[r, c] * act'(.) from the saved output y, all three row-strided 2-D views
__global__ __launch_bounds__(256) void act_bwd2d_kernel(const float* __restrict__ dy, int lddy,
                                                        const float* __restrict__ y, int ldy,
                                                        float* __restrict__ dx, int lddx, int rows, int cols,
                                                        int act, float slope) {
  const int c = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
  if (c >= cols) return;
  const float o = y[(int64_t)r * ldy + c];
  float g = dy[(int64_t)r * lddy + c];
  if (act == AG_ACT_LEAKY) g = o > 0.f ? g : g * slope;
  else if (act == AG_ACT_TANH) g = g * (1.f - o * o);
  dx[(int64_t)r * lddx + c] = g;
}

extern "C" int ag_act_bwd2d(const float* dy, int lddy, const float* y, int ldy, float* dx, int lddx, int rows,
                            int cols, int act, float slope, void* stream) {
  AG_REQUIRE(dy && y && dx && rows > 0 && cols > 0 && rows <= 65535, "ag_act_bwd2d: bad args");
  hipLaunchKernelGGL(act_bwd2d_kernel, dim3(ag_cdiv(cols, 256), rows), dim3(256), 0, (hipStream_t)stream, dy, lddy,
                     y, ldy, dx, lddx, rows, cols, act, slope);
  AG_CHECK_LAUNCH("ag_act_bwd2d");
  return AG_OK;
}


// ------------------------------------------------------------------------------------------
// Feature-matching statistics over time (audiogan.py:341-350, calc_dists): for every (clip, channel) row of an
// activation h [B,C,L] with valid length len[b]
//   m = sum_t h_t / len                       (the sum runs over ALL t: D has zeroed the padded steps)
//   cen_t = h_t - m * [t < len]
//   s = sqrt(sum_t cen_t^2) / len ,   f = (sum_t cen_t^4)^(1/4) / len
// one wave per row, two passes over the row (mean, then central moments).  Backward:
//   dh_t = gm/len + gs * (cen_t - A1/len) / (len * sqrt(S2)) + gf * (cen_t^3 - A3/len) / (len * S4^(3/4)),
//   A1 = sum_t cen_t [t<len],  A3 = sum_t cen_t^3 [t<len]
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void time_moments_fwd_kernel(const float* __restrict__ h, int64_t bs, int64_t cs,
                                                               const int64_t* __restrict__ lens, float* __restrict__ m,
                                                               float* __restrict__ s, float* __restrict__ f, int B,
                                                               int C, int L) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= B * C) return;
  const int b = row / C, c = row - b * C;
  const float* hr = h + (int64_t)b * bs + (int64_t)c * cs;
  const int len = (int)lens[b];
  const float lf = (float)len;
  float sum = 0.f;
  for (int t = lane; t < L; t += 64) sum += hr[t];
  sum = ag_wave_sum(sum);
  const float mean = sum / lf;
  float s2 = 0.f, s4 = 0.f;
  for (int t = lane; t < L; t += 64) {
    const float cen = hr[t] - (t < len ? mean : 0.f);
    const float c2 = cen * cen;
    s2 += c2;
    s4 += c2 * c2;
  }
  s2 = ag_wave_sum(s2);
  s4 = ag_wave_sum(s4);
  if (lane == 0) {
    m[row] = mean;
    s[row] = sqrtf(s2) / lf;
    f[row] = sqrtf(sqrtf(s4)) / lf;
  }
}

__global__ __launch_bounds__(256) void time_moments_bwd_kernel(const float* __restrict__ h, int64_t bs, int64_t cs,
                                                               const int64_t* __restrict__ lens,
                                                               const float* __restrict__ gm, const float* __restrict__ gs,
                                                               const float* __restrict__ gf, float* __restrict__ dh,
                                                               int64_t dbs, int64_t dcs, int B, int C, int L) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= B * C) return;
  const int b = row / C, c = row - b * C;
  const float* hr = h + (int64_t)b * bs + (int64_t)c * cs;
  float* dr = dh + (int64_t)b * dbs + (int64_t)c * dcs;
  const int len = (int)lens[b];
  const float lf = (float)len;
  float sum = 0.f;
  for (int t = lane; t < L; t += 64) sum += hr[t];
  const float mean = ag_wave_sum(sum) / lf;
  float s2 = 0.f, s4 = 0.f, a1 = 0.f, a3 = 0.f;
  for (int t = lane; t < L; t += 64) {
    const bool in = t < len;
    const float cen = hr[t] - (in ? mean : 0.f);
    const float c2 = cen * cen;
    s2 += c2;
    s4 += c2 * c2;
    if (in) { a1 += cen; a3 += c2 * cen; }
  }
  s2 = ag_wave_sum(s2); s4 = ag_wave_sum(s4); a1 = ag_wave_sum(a1); a3 = ag_wave_sum(a3);
  const float km = (gm ? gm[row] : 0.f) / lf;
  // zero rows: sqrt'(0) is infinite; torch gives nan/inf there too - keep finite by dropping the term
  const float ks = (gs && s2 > 0.f) ? gs[row] / (lf * sqrtf(s2)) : 0.f;
  const float kf = (gf && s4 > 0.f) ? gf[row] / (lf * sqrtf(sqrtf(s4)) * sqrtf(s4)) : 0.f;
  const float m1 = a1 / lf, m3 = a3 / lf;
  for (int t = lane; t < L; t += 64) {
    const float cen = hr[t] - (t < len ? mean : 0.f);
    dr[t] = km + ks * (cen - m1) + kf * (cen * cen * cen - m3);
  }
}

extern "C" int ag_time_moments_fwd(const float* h, int64_t bs, int64_t cs, const int64_t* lens_i64, float* m, float* s,
                                   float* f, int B, int C, int L, void* stream) {
  AG_REQUIRE(h && lens_i64 && m && s && f && B > 0 && C > 0 && L > 0, "ag_time_moments_fwd: bad args");
  hipLaunchKernelGGL(time_moments_fwd_kernel, dim3(ag_cdiv(B * C, 4)), dim3(256), 0, (hipStream_t)stream, h, bs, cs,
                     lens_i64, m, s, f, B, C, L);
  AG_CHECK_LAUNCH("ag_time_moments_fwd");
  return AG_OK;
}

extern "C" int ag_time_moments_bwd(const float* h, int64_t bs, int64_t cs, const int64_t* lens_i64, const float* gm,
                                   const float* gs, const float* gf, float* dh, int64_t dbs, int64_t dcs, int B, int C,
                                   int L, void* stream) {
  AG_REQUIRE(h && lens_i64 && dh && B > 0 && C > 0 && L > 0, "ag_time_moments_bwd: bad args");
  hipLaunchKernelGGL(time_moments_bwd_kernel, dim3(ag_cdiv(B * C, 4)), dim3(256), 0, (hipStream_t)stream, h, bs, cs,
                     lens_i64, gm, gs, gf, dh, dbs, dcs, B, C, L);
  AG_CHECK_LAUNCH("ag_time_moments_bwd");
  return AG_OK;
}

// ------------------------------------------------------------------------------------------
// Batched 2-D transpose through LDS: out[b][j][i] = in[b][i][j], inner index contiguous on both sides.  The critic hands
// its conv features [B,C,T'] to the biLSTM as [T',B,C] (audiogan.py:542) and back (:544 under .backward()): as a strided
// elementwise copy one side of that is uncoalesced (45 us for 33 MB); 32 x 32 tiles staged in LDS read and write rows.
//   in : element (b, i, j) at in  + b*ibs + i*irs + j        (i < R, j < Cc)
//   out: element (b, j, i) at out + b*obs + j*ors + i
// ------------------------------------------------------------------------------------------
// (IN16 / OUT16: that side is stored as bfloat16 - strides in elements of its own type; fp32 -> bf16 rounds to nearest even)
template <bool IN16, bool OUT16>
__global__ __launch_bounds__(256) void transpose_batched_kernel(const void* __restrict__ in, int64_t ibs, int64_t irs,
                                                                void* __restrict__ out, int64_t obs, int64_t ors,
                                                                int R, int Cc) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = i0 + ty + 8 * k, j = j0 + tx;
    float v = 0.f;
    if (i < R && j < Cc) {
      const int64_t o = (int64_t)b * ibs + (int64_t)i * irs + j;
      v = IN16 ? __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(in)[o] << 16) : reinterpret_cast<const float*>(in)[o];
    }
    tile[ty + 8 * k][tx] = v;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int j = j0 + ty + 8 * k, i = i0 + tx;
    if (j < Cc && i < R) {
      const int64_t o = (int64_t)b * obs + (int64_t)j * ors + i;
      const float v = tile[tx][ty + 8 * k];
      if (OUT16) reinterpret_cast<unsigned short*>(out)[o] = (unsigned short)(ag_pack_bf16(v, v) & 0xFFFFu);
      else reinterpret_cast<float*>(out)[o] = v;
    }
  }
}

extern "C" int ag_transpose_batched(const void* in, int in_bf16, int64_t ibs, int64_t irs, void* out, int out_bf16, int64_t obs,
                                    int64_t ors, int B, int R, int Cc, void* stream) {
  AG_REQUIRE(in && out && B > 0 && R > 0 && Cc > 0 && B <= 65535, "ag_transpose_batched: bad args");
  AG_REQUIRE(irs >= Cc && ors >= R, "ag_transpose_batched: a row pitch is smaller than its row");
  AG_REQUIRE(ag_cdiv(R, 32) <= 65535, "ag_transpose_batched: too many rows");
  const dim3 grid(ag_cdiv(Cc, 32), ag_cdiv(R, 32), B);
  hipStream_t st = (hipStream_t)stream;
  if (!in_bf16 && !out_bf16) hipLaunchKernelGGL((transpose_batched_kernel<false, false>), grid, dim3(256), 0, st, in, ibs, irs, out, obs, ors, R, Cc);
  if (!in_bf16 && out_bf16) hipLaunchKernelGGL((transpose_batched_kernel<false, true>), grid, dim3(256), 0, st, in, ibs, irs, out, obs, ors, R, Cc);
  if (in_bf16 && !out_bf16) hipLaunchKernelGGL((transpose_batched_kernel<true, false>), grid, dim3(256), 0, st, in, ibs, irs, out, obs, ors, R, Cc);
  if (in_bf16 && out_bf16) hipLaunchKernelGGL((transpose_batched_kernel<true, true>), grid, dim3(256), 0, st, in, ibs, irs, out, obs, ors, R, Cc);
  AG_CHECK_LAUNCH("ag_transpose_batched");
  return AG_OK;
}

// ------------------------------------------------------------------------------------------
// Input assembly of the two networks - what the reference does with T.cat / expand / transpose / + on the host side
// of every iteration (audiogan.py:433-436, :724-728, :749-751, :844) as ONE launch each instead of 3-8 torch launches.
// ------------------------------------------------------------------------------------------
// zc[t, b, :] = [ z[b, t, :ns] | c[b, :es] ]      (audiogan.py:439: the frame's LSTM input besides the fed-back frame)
__global__ __launch_bounds__(256) void build_zc_kernel(const float* __restrict__ z, const float* __restrict__ c,
                                                       float* __restrict__ zc, int B, int T, int ns, int es) {
  const int F = ns + es;
  const int64_t n = (int64_t)T * B * F;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int j = (int)(i % F);
    const int64_t tb = i / F;
    const int b = (int)(tb % B), t = (int)(tb / B);
    zc[i] = j < ns ? z[((int64_t)b * T + t) * ns + j] : c[(int64_t)b * es + (j - ns)];
  }
}

extern "C" int ag_build_zc(const float* z, const float* c, float* zc, int B, int T, int ns, int es, void* stream) {
  AG_REQUIRE(z && c && zc && B > 0 && T > 0 && ns >= 0 && es >= 0 && ns + es > 0, "ag_build_zc: bad args");
  const int64_t n = (int64_t)T * B * (ns + es);
  int gx = (int)ag_cdiv64(n, 256 * 4);
  if (gx > 4096) gx = 4096;
  hipLaunchKernelGGL(build_zc_kernel, dim3(gx), dim3(256), 0, (hipStream_t)stream, z, c, zc, B, T, ns, es);
  AG_CHECK_LAUNCH("ag_build_zc");
  return AG_OK;
}

// The critic's minibatch: rows [0, nA) = xa + na, rows [nA, nA + nB) = xb + nb (instance noise optional), the rows'
// lengths after every conv layer (ceil(len / (s_1 ... s_i)), audiogan.py:533) and the conditioning rows [cA ; cB].
struct CriticBatchP {
  const float *xa, *na, *xb, *nb;
  int64_t xa_ld, na_ld, xb_ld, nb_ld;
  float* x;                    // [nA + nB, L] contiguous
  const int64_t *lenA, *lenB;  // [nA], [nB] (NULL: L)
  int64_t* lens;               // [nl, nA + nB] (or NULL)
  const float *cA, *cB;        // [nA, E], [nB, E] (or NULL)
  float* c;                    // [nA + nB, E] (or NULL)
  int nA, nB, L, nl, E;
  int prods[8];
};

__global__ __launch_bounds__(256) void critic_batch_kernel(const CriticBatchP p) {
  const int n = p.nA + p.nB;
  const int row = blockIdx.y;
  const bool a = row < p.nA;
  const int r = a ? row : row - p.nA;
  const float* x = a ? p.xa + (int64_t)r * p.xa_ld : p.xb + (int64_t)r * p.xb_ld;
  const float* nz = a ? (p.na ? p.na + (int64_t)r * p.na_ld : nullptr) : (p.nb ? p.nb + (int64_t)r * p.nb_ld : nullptr);
  float* y = p.x + (int64_t)row * p.L;
  const bool vec = ((((uintptr_t)x | (uintptr_t)nz | (uintptr_t)y) & 15) == 0) && (p.L & 3) == 0;
  if (vec) {
    const int L4 = p.L >> 2;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < L4; i += gridDim.x * 256) {
      f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
      if (nz) v += reinterpret_cast<const f32x4*>(nz)[i];
      reinterpret_cast<f32x4*>(y)[i] = v;
    }
  } else {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < p.L; i += gridDim.x * 256) y[i] = x[i] + (nz ? nz[i] : 0.f);
  }
  if (blockIdx.x == 0) {
    if (p.lens && threadIdx.x < p.nl) {
      const int64_t* lp = a ? p.lenA : p.lenB;
      const int64_t len = lp ? lp[r] : (int64_t)p.L;
      const int64_t d = p.prods[threadIdx.x];
      p.lens[(int64_t)threadIdx.x * n + row] = (len + d - 1) / d;
    }
    if (p.c) {
      const float* cs = a ? p.cA + (int64_t)r * p.E : p.cB + (int64_t)r * p.E;
      for (int j = threadIdx.x; j < p.E; j += 256) p.c[(int64_t)row * p.E + j] = cs[j];
    }
  }
}

extern "C" int ag_critic_batch(const float* xa, int64_t xa_ld, const float* na, int64_t na_ld, int nA, const float* xb,
                               int64_t xb_ld, const float* nb, int64_t nb_ld, int nB, int L, float* x_out,
                               const int64_t* lenA_i64, const int64_t* lenB_i64, const int32_t* prods_host, int nl,
                               int64_t* lens_out_i64, const float* cA, const float* cB, int E, float* c_out, void* stream) {
  AG_REQUIRE(x_out && L > 0 && nA >= 0 && nB >= 0 && nA + nB > 0 && nA + nB <= 65535, "ag_critic_batch: bad shape");
  AG_REQUIRE((nA == 0 || xa) && (nB == 0 || xb), "ag_critic_batch: null clips");
  AG_REQUIRE(nl >= 0 && nl <= 8 && (nl == 0 || (prods_host && lens_out_i64)), "ag_critic_batch: at most 8 conv layers");
  AG_REQUIRE(!c_out || (E > 0 && (nA == 0 || cA) && (nB == 0 || cB)), "ag_critic_batch: null conditioning rows");
  CriticBatchP p;
  p.xa = xa; p.na = na; p.xb = xb; p.nb = nb; p.xa_ld = xa_ld; p.na_ld = na_ld; p.xb_ld = xb_ld; p.nb_ld = nb_ld;
  p.x = x_out; p.lenA = lenA_i64; p.lenB = lenB_i64; p.lens = nl > 0 ? lens_out_i64 : nullptr;
  p.cA = cA; p.cB = cB; p.c = c_out; p.nA = nA; p.nB = nB; p.L = L; p.nl = nl; p.E = E;
  for (int i = 0; i < 8; ++i) {
    p.prods[i] = i < nl ? prods_host[i] : 1;
    AG_REQUIRE(p.prods[i] > 0, "ag_critic_batch: stride products must be positive");
  }
  int gx = ag_cdiv(L, 256 * 4 * 2);
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(critic_batch_kernel, dim3(gx, nA + nB), dim3(256), 0, (hipStream_t)stream, p);
  AG_CHECK_LAUNCH("ag_critic_batch");
  return AG_OK;
}


// ------------------------------------------------------------------------------------------
// A Linear layer with ONE output (the critic's last layer 512 -> 1, audiogan.py:511, :549; the Generator's stop head
// 1024 -> 1, :410, :445): on the MFMA GEMM a product with one column is a 64-wide tile that is 98 % padding plus a split-K
// second stage, its backward an outer product and a "matrix" with one row - three launches of 22 - 29 us for 33 MB.
//   rowdot_fwd:  y[m] = x[m,:] . w + b                                    one wave per row, 16-byte loads
//   rowdot_bwd:  dx[m,k] = dy[m] * w[k] (* LeakyReLU'(x[m,k]) when gate)   the layer's input gradient, gated by the saved
//                dw[k] (+)= sum_m dy[m] * x[m,k];  db (+)= sum_m dy[m]     activation below, and both parameter gradients
//                from ONE pass over x (partials per workgroup, fixed-order second stage through the bound workspace)
// AG_PREC_BF16: x and w (forward) / dy, w and x (backward products) rounded to bf16, fp32 accumulation, like ag_gemm.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rowdot_fwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ y, int64_t ldy,
                                                         int M, int K, int rb, int x16) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const float* xr = x + (int64_t)row * ldx;
  float s = 0.f;
  if (x16) {          // x stored as bfloat16 (ldx in 2-byte elements); w rounded like every operand of a bf16 contraction
    const unsigned short* xh = reinterpret_cast<const unsigned short*>(x) + (int64_t)row * ldx;
    for (int k = lane; k < K; k += 64) s += __uint_as_float((unsigned)xh[k] << 16) * ag_rbf(w[k]);
  } else if ((K & 3) == 0 && (ldx & 3) == 0 && ((((uintptr_t)x | (uintptr_t)w) & 15) == 0)) {
    for (int k = lane * 4; k < K; k += 256) {
      const f32x4 a = ag_rbf4_if(*reinterpret_cast<const f32x4*>(xr + k), rb), b = ag_rbf4_if(*reinterpret_cast<const f32x4*>(w + k), rb);
      s += a[0] * b[0]; s += a[1] * b[1]; s += a[2] * b[2]; s += a[3] * b[3];
    }
  } else {
    for (int k = lane; k < K; k += 64) s += ag_rbf_if(xr[k], rb) * ag_rbf_if(w[k], rb);
  }
  s = ag_wave_sum(s);
  if (lane == 0) y[(int64_t)row * ldy] = s + (bias ? bias[0] : 0.f);
}

#define RD_ROWS 64      // rows per workgroup of the backward kernel
__global__ __launch_bounds__(256) void rowdot_bwd_kernel(const float* __restrict__ dy, int64_t lddy, const float* __restrict__ x,
                                                         int ldx, const float* __restrict__ w, float* __restrict__ dx, int lddx,
                                                         float* __restrict__ part, int M, int K, int gate, float slope, int rb,
                                                         int h16) {
  // thread <-> column k (grid.y tiles K by 256), loop over the workgroup's RD_ROWS rows: coalesced rows of x / dx
  const int k = blockIdx.y * 256 + threadIdx.x;
  const int m0 = blockIdx.x * RD_ROWS, m1 = min(M, m0 + RD_ROWS);
  const bool ok = k < K;
  const float wk = ok ? ag_rbf_if(w[k], rb) : 0.f;
  float sw = 0.f, sb = 0.f;
  for (int m = m0; m < m1; ++m) {
    const float g = ag_rbf_if(dy[(int64_t)m * lddy], rb);
    if (ok) {
      // h16: x and dx are stored as bfloat16 (pitches in 2-byte elements)
      const float xv = h16 ? __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(x)[(int64_t)m * ldx + k] << 16)
                           : x[(int64_t)m * ldx + k];
      const float dv = (gate && !(xv > 0.f)) ? g * wk * slope : g * wk;
      if (dx) {
        if (h16) reinterpret_cast<unsigned short*>(dx)[(int64_t)m * lddx + k] = (unsigned short)(ag_pack_bf16(dv, dv) & 0xFFFFu);
        else dx[(int64_t)m * lddx + k] = dv;
      }
      sw += g * ag_rbf_if(xv, rb);
    }
    sb += dy[(int64_t)m * lddy];
  }
  if (part) {
    float* pr = part + (int64_t)blockIdx.x * (K + 1);
    if (ok) pr[k] = sw;
    if (k == 0) pr[K] = sb;
  }
}

extern "C" int ag_rowdot_fwd(const void* x, int x_bf16, int ldx, const float* w, const float* bias, float* y, int64_t ldy, int M,
                             int K, void* stream) {
  AG_REQUIRE(x && w && y && M > 0 && K > 0 && ldx >= K && ldy >= 1, "ag_rowdot_fwd: bad args");
  hipLaunchKernelGGL(rowdot_fwd_kernel, dim3(ag_cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, (const float*)x, ldx, w, bias, y,
                     ldy, M, K, (int)(ag_precision() == AG_PREC_BF16), x_bf16 ? 1 : 0);
  AG_CHECK_LAUNCH("ag_rowdot_fwd");
  return AG_OK;
}

extern "C" int64_t ag_rowdot_bwd_ws_numel(int M, int K) { return (int64_t)ag_cdiv(M, RD_ROWS) * (K + 1); }

extern "C" int ag_rowdot_bwd(const float* dy, int64_t lddy, const void* x, int ldx, const float* w, void* dx, int lddx, int h16,
                             float* dw, float* db, int accumulate, int M, int K, int gate, float slope, void* stream) {
  const AgWs ws = ag_ws_take();     // FIRST: an argument error below must not leave a stale binding behind
  AG_REQUIRE(dy && x && w && M > 0 && K > 0 && ldx >= K && (!dx || lddx >= K), "ag_rowdot_bwd: bad args");
  AG_REQUIRE((dw != nullptr) == (db != nullptr), "ag_rowdot_bwd: dw and db come together");
  float* part = nullptr;
  const int gx = ag_cdiv(M, RD_ROWS);
  if (dw) {
    AG_REQUIRE(ws.p && ws.numel >= (int64_t)gx * (K + 1),
               "ag_rowdot_bwd: bind a workspace of >= %lld floats (ag_bind_workspace; ag_rowdot_bwd_ws_numel)", (long long)gx * (K + 1));
    AG_REQUIRE(db == dw + K, "ag_rowdot_bwd: db must follow dw in memory (one [K+1] gradient row: the second stage sums both)");
    part = ws.p;
  }
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(rowdot_bwd_kernel, dim3(gx, ag_cdiv(K, 256)), dim3(256), 0, st, dy, lddy, (const float*)x, ldx, w, (float*)dx,
                     lddx, part, M, K, gate, slope, (int)(ag_precision() == AG_PREC_BF16) || h16, h16 ? 1 : 0);
  AG_CHECK_LAUNCH("ag_rowdot_bwd");
  if (part) return ag_slab_reduce(part, gx, K + 1, dw, accumulate ? 1 : 0, st);
  return AG_OK;
}
