// convlstm.hip -- pointwise / normalisation kernels of the reference's Conv2DLSTMCell (cells.py:4-103: a convolutional LSTM
// with peepholes and layer normalisation, TF 1.x).  The convolution itself runs on the 1-D conv engine (one launch per kernel
// row over row-shifted views, audiogan_amd/cells.py); everything here is elementwise over the state maps or a per-sample
// reduction, i.e. HBM-bound.
//
// Layout of every map: [H, B, C, W] contiguous ("rows x batch" is the 1-D engine's batch axis, W its time axis).
// Peephole weights: [H, F, W] (the cell keeps TF's [H, W, F] parameter and hands over a permuted copy).
// Gate blocks of the convolution output along C, as TF.split(y, 4): j | i | f | o  (cells.py:66).
#include "common.h"

__device__ __forceinline__ float cl_sig(float x) { return 1.f / (1.f + expf(-x)); }

// ---- (1) split + peepholes on the previous cell state (cells.py:66-70):  i += W_ci * c,  f += W_cf * c
__global__ __launch_bounds__(256) void convlstm_peep_fwd_kernel(const float* __restrict__ y, const float* __restrict__ c,
                                                                const float* __restrict__ wci, const float* __restrict__ wcf,
                                                                float* __restrict__ j, float* __restrict__ ip,
                                                                float* __restrict__ fp, float* __restrict__ o, int H, int B,
                                                                int F, int W) {
  const int64_t n = (int64_t)H * B * F * W;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * 256) {
    const int w = (int)(idx % W);
    const int f = (int)((idx / W) % F);
    const int hb = (int)(idx / ((int64_t)W * F));
    const int h = hb / B;
    const int64_t yb = ((int64_t)hb * 4 * F + f) * W + w;
    const float cv = c[idx];
    const int64_t pw = ((int64_t)h * F + f) * W + w;
    j[idx] = y[yb];
    ip[idx] = y[yb + (int64_t)F * W] + (wci ? wci[pw] * cv : 0.f);
    fp[idx] = y[yb + (int64_t)2 * F * W] + (wcf ? wcf[pw] * cv : 0.f);
    o[idx] = y[yb + (int64_t)3 * F * W];
  }
}

// backward: dy blocks = (dj, di, df, do);  dc (+)= di * W_ci + df * W_cf
__global__ __launch_bounds__(256) void convlstm_peep_bwd_kernel(const float* __restrict__ dj, const float* __restrict__ di,
                                                                const float* __restrict__ df, const float* __restrict__ dob,
                                                                const float* __restrict__ wci, const float* __restrict__ wcf,
                                                                float* __restrict__ dy, float* __restrict__ dc, int H, int B,
                                                                int F, int W) {
  const int64_t n = (int64_t)H * B * F * W;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * 256) {
    const int w = (int)(idx % W);
    const int f = (int)((idx / W) % F);
    const int hb = (int)(idx / ((int64_t)W * F));
    const int h = hb / B;
    const int64_t yb = ((int64_t)hb * 4 * F + f) * W + w;
    const int64_t pw = ((int64_t)h * F + f) * W + w;
    const float a = di[idx], b = df[idx];
    dy[yb] = dj[idx];
    dy[yb + (int64_t)F * W] = a;
    dy[yb + (int64_t)2 * F * W] = b;
    dy[yb + (int64_t)3 * F * W] = dob[idx];
    dc[idx] += (wci ? a * wci[pw] : 0.f) + (wcf ? b * wcf[pw] : 0.f);
  }
}

// peephole weight gradient: dW[h,f,w] = sum_b g[h,b,f,w] * c[h,b,f,w]  (one thread per weight, b ascending: deterministic)
__global__ __launch_bounds__(256) void convlstm_peep_wgrad_kernel(const float* __restrict__ g, const float* __restrict__ c,
                                                                  float* __restrict__ dw, int H, int B, int F, int W) {
  const int64_t n = (int64_t)H * F * W;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
  const int64_t fw = idx % ((int64_t)F * W);
  const int h = (int)(idx / ((int64_t)F * W));
  float s = 0.f;
  for (int b = 0; b < B; ++b) {
    const int64_t e = ((int64_t)h * B + b) * F * W + fw;
    s += g[e] * c[e];
  }
  dw[idx] = s;
}

// ---- (2) cell update (cells.py:77-82):  c' = c * sigmoid(f + forget_bias) + sigmoid(i) * act(j);  o += W_co * c'
__global__ __launch_bounds__(256) void convlstm_cell_fwd_kernel(const float* __restrict__ j, const float* __restrict__ i_,
                                                                const float* __restrict__ f_, const float* __restrict__ c,
                                                                const float* __restrict__ o, const float* __restrict__ wco,
                                                                float fb, float* __restrict__ cn, float* __restrict__ op,
                                                                int H, int B, int F, int W) {
  const int64_t n = (int64_t)H * B * F * W;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * 256) {
    const int64_t fw = idx % ((int64_t)F * W);
    const int h = (int)(idx / ((int64_t)B * F * W));
    const float v = c[idx] * cl_sig(f_[idx] + fb) + cl_sig(i_[idx]) * tanhf(j[idx]);
    cn[idx] = v;
    op[idx] = o[idx] + (wco ? wco[(int64_t)h * F * W + fw] * v : 0.f);
  }
}

// backward from (dcn = dL/dc', dop = dL/d(o + W_co c')): dj, di, df, dc (written), do = dop; dcn_tot left in dcn
__global__ __launch_bounds__(256) void convlstm_cell_bwd_kernel(const float* __restrict__ j, const float* __restrict__ i_,
                                                                const float* __restrict__ f_, const float* __restrict__ c,
                                                                const float* __restrict__ wco, float fb,
                                                                float* __restrict__ dcn, const float* __restrict__ dop,
                                                                float* __restrict__ dj, float* __restrict__ di,
                                                                float* __restrict__ df, float* __restrict__ dc, int H, int B,
                                                                int F, int W) {
  const int64_t n = (int64_t)H * B * F * W;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * 256) {
    const int64_t fw = idx % ((int64_t)F * W);
    const int h = (int)(idx / ((int64_t)B * F * W));
    const float g = dcn[idx] + (wco ? dop[idx] * wco[(int64_t)h * F * W + fw] : 0.f);
    const float sf = cl_sig(f_[idx] + fb), si = cl_sig(i_[idx]), tj = tanhf(j[idx]);
    dcn[idx] = g;
    dc[idx] = g * sf;
    df[idx] = g * c[idx] * sf * (1.f - sf);
    di[idx] = g * tj * si * (1.f - si);
    dj[idx] = g * si * (1.f - tj * tj);
  }
}

// ---- (3) output (cells.py:88-89):  h = sigmoid(o) * act(c)
__global__ __launch_bounds__(256) void convlstm_out_fwd_kernel(const float* __restrict__ o, const float* __restrict__ c,
                                                               float* __restrict__ hout, int64_t n) {
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * 256)
    hout[idx] = cl_sig(o[idx]) * tanhf(c[idx]);
}

__global__ __launch_bounds__(256) void convlstm_out_bwd_kernel(const float* __restrict__ o, const float* __restrict__ c,
                                                               const float* __restrict__ dh, float* __restrict__ dob,
                                                               float* __restrict__ dc, int64_t n) {
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * 256) {
    const float so = cl_sig(o[idx]), tc = tanhf(c[idx]), g = dh[idx];
    dob[idx] = g * tc * so * (1.f - so);
    dc[idx] = g * so * (1.f - tc * tc);
  }
}

// ---- (4) TF contrib layer_norm (begin_norm_axis = 1, begin_params_axis = -1, variance epsilon 1e-12): per sample over
//      (H, W, F); gamma / beta per feature f.  One workgroup per sample; two passes (mean, then centred second moment).
__global__ __launch_bounds__(256) void layer_norm_hbfw_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, float eps,
                                                                  float* __restrict__ y, float* __restrict__ mean,
                                                                  float* __restrict__ rstd, int H, int B, int F, int W) {
  __shared__ float red[17];
  const int b = blockIdx.x;
  const int64_t FW = (int64_t)F * W, n = (int64_t)H * FW;
  float s = 0.f;
  for (int64_t e = threadIdx.x; e < n; e += 256) s += x[((e / FW) * B + b) * FW + e % FW];
  const float m = ag_block_sum(s, red) / (float)n;
  float q = 0.f;
  for (int64_t e = threadIdx.x; e < n; e += 256) {
    const float d = x[((e / FW) * B + b) * FW + e % FW] - m;
    q += d * d;
  }
  const float r = rsqrtf(ag_block_sum(q, red) / (float)n + eps);
  if (threadIdx.x == 0) { mean[b] = m; rstd[b] = r; }
  for (int64_t e = threadIdx.x; e < n; e += 256) {
    const int64_t a = ((e / FW) * B + b) * FW + e % FW;
    const int f = (int)((e % FW) / W);
    y[a] = (x[a] - m) * r * gamma[f] + beta[f];
  }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma;  per-sample partials of dgamma / dbeta in [B, F]
__global__ __launch_bounds__(256) void layer_norm_hbfw_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                  const float* __restrict__ gamma, const float* __restrict__ mean,
                                                                  const float* __restrict__ rstd, float* __restrict__ dx,
                                                                  float* __restrict__ dgp, float* __restrict__ dbp, int H, int B,
                                                                  int F, int W) {
  __shared__ float red[17];
  const int b = blockIdx.x;
  const int64_t FW = (int64_t)F * W, n = (int64_t)H * FW;
  const float m = mean[b], r = rstd[b];
  float s1 = 0.f, s2 = 0.f;
  for (int64_t e = threadIdx.x; e < n; e += 256) {
    const int64_t a = ((e / FW) * B + b) * FW + e % FW;
    const float g = dy[a] * gamma[(int)((e % FW) / W)];
    s1 += g;
    s2 += g * (x[a] - m) * r;
  }
  const float m1 = ag_block_sum(s1, red) / (float)n;
  const float m2 = ag_block_sum(s2, red) / (float)n;
  for (int64_t e = threadIdx.x; e < n; e += 256) {
    const int64_t a = ((e / FW) * B + b) * FW + e % FW;
    const float xh = (x[a] - m) * r;
    dx[a] = r * (dy[a] * gamma[(int)((e % FW) / W)] - m1 - xh * m2);
  }
  // dgamma / dbeta partials of this sample: thread f sums its feature's (h, w) entries in a fixed order
  for (int f = threadIdx.x; f < F; f += 256) {
    float ga = 0.f, be = 0.f;
    for (int h = 0; h < H; ++h)
      for (int w = 0; w < W; ++w) {
        const int64_t a = (((int64_t)h * B + b) * F + f) * W + w;
        const float d = dy[a];
        ga += d * (x[a] - m) * r;
        be += d;
      }
    dgp[(int64_t)b * F + f] = ga;
    dbp[(int64_t)b * F + f] = be;
  }
}

static inline unsigned cl_grid(int64_t n) {
  int64_t g = ag_cdiv64(n, 256);
  return (unsigned)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

#define CL_DIMS_OK(H, B, F, W) ((H) > 0 && (B) > 0 && (F) > 0 && (W) > 0 && (int64_t)(H) * (B) * (F) * (W) * 4 < ((int64_t)1 << 40))

extern "C" int ag_convlstm_peephole_fwd(const float* y, const float* c, const float* w_ci, const float* w_cf, float* j,
                                        float* i_pre, float* f_pre, float* o_raw, int H, int B, int F, int W, void* stream) {
  AG_REQUIRE(y && c && j && i_pre && f_pre && o_raw && CL_DIMS_OK(H, B, F, W), "ag_convlstm_peephole_fwd: bad args");
  hipLaunchKernelGGL(convlstm_peep_fwd_kernel, dim3(cl_grid((int64_t)H * B * F * W)), dim3(256), 0, (hipStream_t)stream, y, c,
                     w_ci, w_cf, j, i_pre, f_pre, o_raw, H, B, F, W);
  AG_CHECK_LAUNCH("ag_convlstm_peephole_fwd");
  return AG_OK;
}

extern "C" int ag_convlstm_peephole_bwd(const float* dj, const float* di, const float* df, const float* d_o, const float* c,
                                        const float* w_ci, const float* w_cf, float* dy, float* dc, float* dw_ci,
                                        float* dw_cf, int H, int B, int F, int W, void* stream) {
  AG_REQUIRE(dj && di && df && d_o && c && dy && dc && CL_DIMS_OK(H, B, F, W), "ag_convlstm_peephole_bwd: bad args");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(convlstm_peep_bwd_kernel, dim3(cl_grid((int64_t)H * B * F * W)), dim3(256), 0, st, dj, di, df, d_o, w_ci,
                     w_cf, dy, dc, H, B, F, W);
  const unsigned gw = (unsigned)ag_cdiv64((int64_t)H * F * W, 256);
  if (dw_ci) hipLaunchKernelGGL(convlstm_peep_wgrad_kernel, dim3(gw), dim3(256), 0, st, di, c, dw_ci, H, B, F, W);
  if (dw_cf) hipLaunchKernelGGL(convlstm_peep_wgrad_kernel, dim3(gw), dim3(256), 0, st, df, c, dw_cf, H, B, F, W);
  AG_CHECK_LAUNCH("ag_convlstm_peephole_bwd");
  return AG_OK;
}

extern "C" int ag_convlstm_cell_fwd(const float* j, const float* i_, const float* f_, const float* c, const float* o_raw,
                                    const float* w_co, float forget_bias, float* c_new, float* o_pre, int H, int B, int F,
                                    int W, void* stream) {
  AG_REQUIRE(j && i_ && f_ && c && o_raw && c_new && o_pre && CL_DIMS_OK(H, B, F, W), "ag_convlstm_cell_fwd: bad args");
  hipLaunchKernelGGL(convlstm_cell_fwd_kernel, dim3(cl_grid((int64_t)H * B * F * W)), dim3(256), 0, (hipStream_t)stream, j, i_,
                     f_, c, o_raw, w_co, forget_bias, c_new, o_pre, H, B, F, W);
  AG_CHECK_LAUNCH("ag_convlstm_cell_fwd");
  return AG_OK;
}

extern "C" int ag_convlstm_cell_bwd(const float* j, const float* i_, const float* f_, const float* c, const float* c_new,
                                    const float* w_co, float forget_bias, float* dc_new, const float* do_pre, float* dj,
                                    float* di, float* df, float* dc, float* dw_co, int H, int B, int F, int W, void* stream) {
  AG_REQUIRE(j && i_ && f_ && c && c_new && dc_new && do_pre && dj && di && df && dc && CL_DIMS_OK(H, B, F, W),
             "ag_convlstm_cell_bwd: bad args");
  hipStream_t st = (hipStream_t)stream;
  if (dw_co)    // dW_co = sum_b do_pre * c'
    hipLaunchKernelGGL(convlstm_peep_wgrad_kernel, dim3((unsigned)ag_cdiv64((int64_t)H * F * W, 256)), dim3(256), 0, st,
                       do_pre, c_new, dw_co, H, B, F, W);
  hipLaunchKernelGGL(convlstm_cell_bwd_kernel, dim3(cl_grid((int64_t)H * B * F * W)), dim3(256), 0, st, j, i_, f_, c, w_co,
                     forget_bias, dc_new, do_pre, dj, di, df, dc, H, B, F, W);
  AG_CHECK_LAUNCH("ag_convlstm_cell_bwd");
  return AG_OK;
}

extern "C" int ag_convlstm_out_fwd(const float* o, const float* c, float* h, int64_t n, void* stream) {
  AG_REQUIRE(o && c && h && n > 0, "ag_convlstm_out_fwd: bad args");
  hipLaunchKernelGGL(convlstm_out_fwd_kernel, dim3(cl_grid(n)), dim3(256), 0, (hipStream_t)stream, o, c, h, n);
  AG_CHECK_LAUNCH("ag_convlstm_out_fwd");
  return AG_OK;
}

extern "C" int ag_convlstm_out_bwd(const float* o, const float* c, const float* dh, float* d_o, float* dc, int64_t n,
                                   void* stream) {
  AG_REQUIRE(o && c && dh && d_o && dc && n > 0, "ag_convlstm_out_bwd: bad args");
  hipLaunchKernelGGL(convlstm_out_bwd_kernel, dim3(cl_grid(n)), dim3(256), 0, (hipStream_t)stream, o, c, dh, d_o, dc, n);
  AG_CHECK_LAUNCH("ag_convlstm_out_bwd");
  return AG_OK;
}

extern "C" int ag_layer_norm_hbfw_fwd(const float* x, const float* gamma, const float* beta, float eps, float* y, float* mean,
                                      float* rstd, int H, int B, int F, int W, void* stream) {
  AG_REQUIRE(x && gamma && beta && y && mean && rstd && CL_DIMS_OK(H, B, F, W), "ag_layer_norm_hbfw_fwd: bad args");
  hipLaunchKernelGGL(layer_norm_hbfw_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, eps, y, mean, rstd,
                     H, B, F, W);
  AG_CHECK_LAUNCH("ag_layer_norm_hbfw_fwd");
  return AG_OK;
}

extern "C" int ag_layer_norm_hbfw_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                                      float* dx, float* dgamma_part, float* dbeta_part, int H, int B, int F, int W,
                                      void* stream) {
  AG_REQUIRE(dy && x && gamma && mean && rstd && dx && dgamma_part && dbeta_part && CL_DIMS_OK(H, B, F, W),
             "ag_layer_norm_hbfw_bwd: bad args");
  hipLaunchKernelGGL(layer_norm_hbfw_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, dy, x, gamma, mean, rstd, dx,
                     dgamma_part, dbeta_part, H, B, F, W);
  AG_CHECK_LAUNCH("ag_layer_norm_hbfw_bwd");
  return AG_OK;
}
