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

__global__ __launch_bounds__(256) void grad_norms_kernel(const ag_opt_desc* __restrict__ descs,
                                                         float* __restrict__ sq, int32_t* __restrict__ flags,
                                                         float gscale, float* __restrict__ part) {
  __shared__ float red[17];
  const ag_opt_desc d = descs[blockIdx.y];
  float s = 0.f;
  int f = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < d.n; i += (int64_t)gridDim.x * 256) {
    const float g = d.grad[i] * gscale;
    s += g * g;
    if (g != g) f |= AG_FLAG_NAN;
    if (fabsf(g) > 1e5f) f |= AG_FLAG_BIG;
  }
  if ((int64_t)blockIdx.x * 256 >= d.n) {  // uniform per block
    if (part && threadIdx.x == 0) part[(int64_t)blockIdx.y * OPT_CHUNKS + blockIdx.x] = 0.f;
    return;
  }
  s = ag_block_sum(s, red);
  if (threadIdx.x == 0) {
    if (part) part[(int64_t)blockIdx.y * OPT_CHUNKS + blockIdx.x] = s;
    else atomicAdd(sq + blockIdx.y, s);
  }
  if (f && flags) atomicOr(flags, f);
}

__global__ void grad_norms_finish_kernel(float* __restrict__ sq_to_norm, float* __restrict__ norm_sum,
                                         int n, int32_t* __restrict__ step_dev, const float* __restrict__ part) {
  if (threadIdx.x == 0 && step_dev) step_dev[0] += 1;
  // single block: norms[i] = sqrt(sq[i]); norm_sum = sum_i norms[i] (deterministic order)
  __shared__ float red[17];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    if (part) {          // sum of squares of tensor i = its per-chunk partials in chunk order
      float q = 0.f;
      for (int c = 0; c < OPT_CHUNKS; ++c) q += part[(int64_t)i * OPT_CHUNKS + c];
      sq_to_norm[i] = q;
    }
    const float nr = sqrtf(sq_to_norm[i]);
    sq_to_norm[i] = nr;
    s += nr;
  }
  s = ag_block_sum(s, red);
  if (threadIdx.x == 0 && norm_sum) norm_sum[0] = s;
}

extern "C" int ag_grad_norms(const ag_opt_desc* descs_dev, int n, float* norms, float* norm_sum,
                             int32_t* flags, float grad_scale, int32_t* step_dev, void* stream) {
  const AgWs ws = ag_ws_take();     // FIRST: an argument error below must not leave a stale binding behind
  AG_REQUIRE(descs_dev && norms && n > 0 && n <= 65535, "ag_grad_norms: bad args");
  hipStream_t st = (hipStream_t)stream;
  float* part = (ws.p && ws.numel >= (int64_t)n * OPT_CHUNKS) ? ws.p : nullptr;    // two-stage sum of squares
  if (!part && hipMemsetAsync(norms, 0, sizeof(float) * n, st) != hipSuccess) {
    ag_set_error("ag_grad_norms: memset failed");
    return AG_ERR_LAUNCH;
  }
  if (flags) (void)hipMemsetAsync(flags, 0, sizeof(int32_t), st);
  hipLaunchKernelGGL(grad_norms_kernel, dim3(OPT_CHUNKS, n), dim3(256), 0, st, descs_dev, norms, flags,
                     grad_scale, part);
  AG_CHECK_LAUNCH("ag_grad_norms");
  hipLaunchKernelGGL(grad_norms_finish_kernel, dim3(1), dim3(256), 0, st, norms, norm_sum, n, step_dev, part);
  AG_CHECK_LAUNCH("ag_grad_norms(finish)");
  return AG_OK;
}

__global__ __launch_bounds__(256) void opt_step_kernel(const ag_opt_desc* __restrict__ descs,
                                                       const float* __restrict__ norms, int kind, float lr,
                                                       float clip, float gscale, float a1, float b2,
                                                       float eps, float bc1, float bc2sqrt,
                                                       const int32_t* __restrict__ step_dev) {
  const ag_opt_desc d = descs[blockIdx.y];
  if (step_dev && kind == AG_OPT_ADAM) {
    const float st = (float)step_dev[0];
    bc1 = 1.f - powf(a1, st);
    bc2sqrt = sqrtf(1.f - powf(b2, st));
  }
  const float nr = norms[blockIdx.y];
  if ((int64_t)blockIdx.x * 256 >= d.n) return;
  const bool do_clip = clip > 0.f && nr > clip;
  const float div = do_clip ? nr / clip : 1.f;  // reference: grad /= (norm / clip), audiogan.py:252
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < d.n; i += (int64_t)gridDim.x * 256) {
    float g = d.grad[i] * gscale;
    if (do_clip) g = g / div;
    if (kind == AG_OPT_RMSPROP) {
      const float sq = a1 * d.s1[i] + (1.f - a1) * g * g;
      d.s1[i] = sq;
      d.p[i] = d.p[i] - lr * (g / (sqrtf(sq) + eps));
    } else {
      const float m = a1 * d.s1[i] + (1.f - a1) * g;
      const float v = b2 * d.s2[i] + (1.f - b2) * g * g;
      d.s1[i] = m;
      d.s2[i] = v;
      const float denom = sqrtf(v) / bc2sqrt + eps;
      d.p[i] = d.p[i] - (lr / bc1) * (m / denom);
    }
  }
}

extern "C" int ag_opt_step(const ag_opt_desc* descs_dev, int n, const float* norms, int kind, float lr,
                           float clip, float grad_scale, float alpha_or_beta1, float beta2, float eps,
                           int step, const int32_t* step_dev, void* stream) {
  AG_REQUIRE(descs_dev && norms && n > 0 && n <= 65535, "ag_opt_step: bad args");
  AG_REQUIRE(kind == AG_OPT_RMSPROP || kind == AG_OPT_ADAM, "ag_opt_step: bad optimiser kind");
  float bc1 = 1.f, bc2s = 1.f;
  if (kind == AG_OPT_ADAM && !step_dev) {
    AG_REQUIRE(step >= 1, "ag_opt_step: Adam step must be >= 1");
    bc1 = (float)(1.0 - pow((double)alpha_or_beta1, (double)step));
    bc2s = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  }
  hipLaunchKernelGGL(opt_step_kernel, dim3(OPT_CHUNKS, n), dim3(256), 0, (hipStream_t)stream, descs_dev,
                     norms, kind, lr, clip, grad_scale, alpha_or_beta1, beta2, eps, bc1, bc2s, step_dev);
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
__global__ __launch_bounds__(256) void transpose_batched_kernel(const float* __restrict__ in, int64_t ibs, int64_t irs,
                                                                float* __restrict__ out, int64_t obs, int64_t ors,
                                                                int R, int Cc) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
  const float* src = in + (int64_t)b * ibs;
  float* dst = out + (int64_t)b * obs;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = i0 + ty + 8 * k, j = j0 + tx;
    tile[ty + 8 * k][tx] = (i < R && j < Cc) ? src[(int64_t)i * irs + j] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int j = j0 + ty + 8 * k, i = i0 + tx;
    if (j < Cc && i < R) dst[(int64_t)j * ors + i] = tile[tx][ty + 8 * k];
  }
}

extern "C" int ag_transpose_batched(const float* in, int64_t ibs, int64_t irs, float* out, int64_t obs, int64_t ors,
                                    int B, int R, int Cc, void* stream) {
  AG_REQUIRE(in && out && B > 0 && R > 0 && Cc > 0 && B <= 65535, "ag_transpose_batched: bad args");
  AG_REQUIRE(irs >= Cc && ors >= R, "ag_transpose_batched: a row pitch is smaller than its row");
  AG_REQUIRE(ag_cdiv(R, 32) <= 65535, "ag_transpose_batched: too many rows");
  hipLaunchKernelGGL(transpose_batched_kernel, dim3(ag_cdiv(Cc, 32), ag_cdiv(R, 32), B), dim3(256), 0, (hipStream_t)stream,
                     in, ibs, irs, out, obs, ors, R, Cc);
  AG_CHECK_LAUNCH("ag_transpose_batched");
  return AG_OK;
}
