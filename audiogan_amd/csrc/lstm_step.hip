// lstm_step.hip -- the sequential part of the recurrent layers (NN.LSTMCell loop of the
// Generator, audiogan.py:437-443; NN.LSTM of the Discriminator, :498-503,:543).
//
// Per time step the work is a "skinny" product  [B<=64, K] x [K, N]  (B = clips per GPU):
// far too small for a tiled GEMM grid, and strictly sequential over time.  Two kernels:
//
//  * skinny_gemm_kernel    C[M<=64, N] (+)= A[M,K] * op(B): every wave owns a 64x32 output
//    tile and a K slice; MFMA operands are fetched STRAIGHT from global/L2 as 16-byte
//    row pieces (no LDS staging: nothing is reused inside a workgroup), K slices are summed
//    through LDS inside a workgroup and with fp32 atomics across workgroups.
//  * lstm_step_fwd_kernel  one LSTM time step fused: gates = pre + [x,h] * [Wx|Whh]^T for the
//    4 gates of 8 hidden units per workgroup, then the cell non-linearity in the epilogue.
//    Both directions of a bidirectional layer run in the same launch (grid.y).
//
// MFMA k-slot trick: v_mfma_f32_32x32x2_f32 sums two k values per instruction, lane half h
// supplying k-slot h.  Each lane loads a float4 A[row][8q+4h .. +3]; MFMA e (0..3) then
// contracts k = {8q+e, 8q+4+e}.  Any pairing is valid as long as A and B agree.
#include "common.h"

#define SK_MAXW 16

struct SkinnyP {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  int lda, ldb, ldc;
  int M, N, K;
  int tb;        // 1: B stored [N][K]; 0: B stored [K][N]
  int act;
  float slope, beta;
};

// accumulate A[0..64, k0..k1) x Brows over the wave's K range; brow = this lane's B row (tb=1)
template <int MT>
__device__ __forceinline__ void skinny_core_nt(f32x16 (&acc)[MT], const float* __restrict__ A, int lda,
                                               int M, const float* __restrict__ brow, bool bok, int k0,
                                               int k1, int l31, int h) {
  const float* ar[MT];
  bool aok[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    aok[t] = (32 * t + l31) < M;
    ar[t] = A + (int64_t)(aok[t] ? 32 * t + l31 : 0) * lda + 4 * h;
  }
  const float* br = brow + 4 * h;
#pragma unroll 4
  for (int k = k0; k < k1; k += 8) {
    f32x4 b = bok ? *reinterpret_cast<const f32x4*>(br + k) : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 a[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
      a[t] = aok[t] ? *reinterpret_cast<const f32x4*>(ar[t] + k) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int t = 0; t < MT; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][e], b[e], acc[t], 0, 0, 0);
  }
}

// B stored [K][N]: lane j reads B[k][n0+j] (coalesced along n)
template <int MT>
__device__ __forceinline__ void skinny_core_nn(f32x16 (&acc)[MT], const float* __restrict__ A, int lda,
                                               int M, const float* __restrict__ bcol, int ldb, bool bok,
                                               int k0, int k1, int l31, int h) {
  const float* ar[MT];
  bool aok[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    aok[t] = (32 * t + l31) < M;
    ar[t] = A + (int64_t)(aok[t] ? 32 * t + l31 : 0) * lda + 4 * h;
  }
#pragma unroll 2
  for (int k = k0; k < k1; k += 8) {
    float b[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) b[e] = bok ? bcol[(int64_t)(k + 4 * h + e) * ldb] : 0.f;
    f32x4 a[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
      a[t] = aok[t] ? *reinterpret_cast<const f32x4*>(ar[t] + k) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int t = 0; t < MT; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][e], b[e], acc[t], 0, 0, 0);
  }
}

// sum the per-wave accumulators through LDS; afterwards red[0][..] holds the block total
template <int MT>
__device__ __forceinline__ void block_reduce_acc(const f32x16 (&acc)[MT], float* red, int nw, int wid,
                                                 int lane) {
  float* mine = red + (size_t)wid * (MT * 16 * 64);
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) mine[(t * 16 + e) * 64 + lane] = acc[t][e];
  __syncthreads();
  const int nout = MT * 16 * 64;
  for (int o = threadIdx.x; o < nout; o += blockDim.x) {
    float s = red[o];
    for (int w = 1; w < nw; ++w) s += red[(size_t)w * nout + o];
    red[o] = s;
  }
  __syncthreads();
}

template <int MT>
__global__ __launch_bounds__(1024) void skinny_gemm_kernel(const SkinnyP p) {
  extern __shared__ float red[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int n0 = blockIdx.x * 32;
  const int W = gridDim.y * nw, wg = blockIdx.y * nw + wid;
  const int KU = p.K >> 3;
  const int k0 = (int)((int64_t)KU * wg / W) * 8, k1 = (int)((int64_t)KU * (wg + 1) / W) * 8;
  f32x16 acc[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
  const bool bok = (n0 + l31) < p.N;
  if (p.tb)
    skinny_core_nt<MT>(acc, p.A, p.lda, p.M, p.B + (int64_t)(bok ? n0 + l31 : 0) * p.ldb, bok, k0, k1, l31, h);
  else
    skinny_core_nn<MT>(acc, p.A, p.lda, p.M, p.B + (bok ? n0 + l31 : 0), p.ldb, bok, k0, k1, l31, h);
  block_reduce_acc<MT>(acc, red, nw, wid, lane);
  const int nout = MT * 16 * 64;
  const bool direct = gridDim.y == 1;
  for (int o = threadIdx.x; o < nout; o += blockDim.x) {
    const int ln = o & 63, e = (o >> 6) & 15, t = o >> 10;
    const int m = 32 * t + (e & 3) + 8 * (e >> 2) + 4 * (ln >> 5);
    const int n = n0 + (ln & 31);
    if (m >= p.M || n >= p.N) continue;
    float* dst = p.C + (int64_t)m * p.ldc + n;
    float v = red[o];
    if (direct) {
      if (p.beta != 0.f) v += p.beta * *dst;
      if (p.bias) v += p.bias[n];
      *dst = ag_apply_act(v, p.act, p.slope);
    } else {
      if (p.bias && blockIdx.y == 0) v += p.bias[n];
      atomicAdd(dst, v);
    }
  }
}

extern "C" int ag_skinny_gemm(const float* A, int lda, const float* B, int ldb, int tb, float* C, int ldc,
                              int M, int N, int K, float beta, const float* bias, int act, float slope,
                              int accumulate_atomic, void* stream) {
  AG_REQUIRE(A && B && C, "ag_skinny_gemm: null tensor");
  AG_REQUIRE(M > 0 && M <= 64 && N > 0 && K > 0, "ag_skinny_gemm: needs 0 < M <= 64");
  AG_REQUIRE(K % 8 == 0 && lda % 4 == 0 && ((uintptr_t)A & 15) == 0, "ag_skinny_gemm: A must be 16-B aligned, K%%8==0");
  if (tb) AG_REQUIRE(ldb % 4 == 0 && ((uintptr_t)B & 15) == 0, "ag_skinny_gemm: B must be 16-B aligned");
  AG_REQUIRE(!(accumulate_atomic && act != AG_ACT_NONE), "ag_skinny_gemm: atomic mode has a linear epilogue");
  SkinnyP p;
  p.A = A; p.B = B; p.C = C; p.bias = bias;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.M = M; p.N = N; p.K = K; p.tb = tb; p.act = act;
  p.slope = slope; p.beta = beta;
  const int gx = ag_cdiv(N, 32);
  const int KU = K / 8;
  int nw, gy;
  if (accumulate_atomic) {
    // C already holds the value to add to (beta == 1 semantics); spread K over ~512 waves
    nw = 4;
    gy = ag_cdiv(512, gx * nw);
    if (gy * nw > KU) gy = ag_cdiv(KU, nw);
    if (gy < 1) gy = 1;
    if (gy == 1) gy = 2;  // keep the atomic epilogue (C holds the addend)
  } else {
    gy = 1;
    nw = ag_cdiv(1024, gx * 2);  // aim at ~4 waves per CU overall, at least 4 per block
    if (nw < 4) nw = 4;
    if (nw > SK_MAXW) nw = SK_MAXW;
    if (nw > KU) nw = KU;
  }
  const int MT = M > 32 ? 2 : 1;
  const size_t lds = (size_t)nw * MT * 16 * 64 * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (MT == 2) {
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)skinny_gemm_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(skinny_gemm_kernel<2>, dim3(gx, gy), dim3(64 * nw), lds, st, p);
  } else {
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)skinny_gemm_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(skinny_gemm_kernel<1>, dim3(gx, gy), dim3(64 * nw), lds, st, p);
  }
  AG_CHECK_LAUNCH("ag_skinny_gemm");
  return AG_OK;
}

// ------------------------------------------------------------------------------------------
// fused LSTM time step
// ------------------------------------------------------------------------------------------
struct LstmDir {
  float* pre;          // [B, 4H] gate pre-activations for this step (x-part + biases); overwritten
                       //           with the ACTIVATED gates (saved for backward)
  const float* x;      // optional fed-back input [B, Kx] (Generator: x_{t-1}); NULL if none
  const float* wx;     // [4H, ldwx] weight rows for x
  const float* h_prev; // [B, H]
  const float* whh;    // [4H, H]
  const float* c_prev; // [B, H]
  float* c_out;        // [B, H]
  float* h_out;        // [B, H]  (state; carried for padded rows)
  float* y_out;        // [B, ldy] slice of the layer output (0 for padded rows); may be NULL
  int ldx, ldwx, Kx, ldy;
  int t;               // time index (for the valid mask)
};

struct LstmStepP {
  LstmDir d[2];
  const int64_t* valid;
  int B, H;
  int skip_h;          // 1: h_prev is known to be zero (first step) -> skip that product
};

template <int MT>
__global__ __launch_bounds__(512) void lstm_step_fwd_kernel(const LstmStepP p) {
  extern __shared__ float red[];
  const LstmDir& D = p.d[blockIdx.y];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int H = p.H, B = p.B;
  const int u0 = blockIdx.x * 8;
  // column j of this workgroup's 32-wide tile = gate (j>>3), hidden unit u0 + (j&7)
  const int unit = u0 + (l31 & 7);
  const bool bok = unit < H;
  const int row = (l31 >> 3) * H + (bok ? unit : 0);
  f32x16 acc[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
  // split the concatenated K axis [x | h] over the waves in units of 8
  const int KxU = D.x ? (D.Kx >> 3) : 0;
  const int KhU = p.skip_h ? 0 : (H >> 3);
  const int KU = KxU + KhU;
  const int s0 = (int)((int64_t)KU * wid / nw), s1 = (int)((int64_t)KU * (wid + 1) / nw);
  if (s0 < KxU) {
    const int e1 = s1 < KxU ? s1 : KxU;
    skinny_core_nt<MT>(acc, D.x, D.ldx, B, D.wx + (int64_t)row * D.ldwx, bok, s0 * 8, e1 * 8, l31, h);
  }
  if (s1 > KxU) {
    const int b0 = (s0 > KxU ? s0 : KxU) - KxU;
    skinny_core_nt<MT>(acc, D.h_prev, H, B, D.whh + (int64_t)row * H, bok, b0 * 8, (s1 - KxU) * 8, l31, h);
  }
  block_reduce_acc<MT>(acc, red, nw, wid, lane);
  // epilogue: (m, unit) pairs; the 4 gates of a pair sit at columns uu, 8+uu, 16+uu, 24+uu
  const int npair = MT * 32 * 8;
  for (int q = threadIdx.x; q < npair; q += blockDim.x) {
    const int uu = q & 7, m = q >> 3;
    const int u = u0 + uu;
    if (m >= B || u >= H) continue;
    const int t = m >> 5, mm = m & 31;
    // inverse of row = (e&3) + 8*(e>>2) + 4*(lane>>5)
    const int hh = (mm >> 2) & 1, e = (mm & 3) + 4 * (mm >> 3);
    const float* rr = red + (size_t)(t * 16 + e) * 64 + 32 * hh;
    float* pre = D.pre + (int64_t)m * 4 * H;
    const float cp = D.c_prev[(int64_t)m * H + u];
    if (p.valid && D.t >= p.valid[m]) {
      D.h_out[(int64_t)m * H + u] = D.h_prev[(int64_t)m * H + u];
      D.c_out[(int64_t)m * H + u] = cp;
      if (D.y_out) D.y_out[(int64_t)m * D.ldy + u] = 0.f;
      continue;
    }
    const float ig = ag_sigmoid(pre[u] + rr[uu]);
    const float fg = ag_sigmoid(pre[H + u] + rr[8 + uu]);
    const float gg = tanhf(pre[2 * H + u] + rr[16 + uu]);
    const float og = ag_sigmoid(pre[3 * H + u] + rr[24 + uu]);
    const float cn = fg * cp + ig * gg;
    const float hn = og * tanhf(cn);
    pre[u] = ig;
    pre[H + u] = fg;
    pre[2 * H + u] = gg;
    pre[3 * H + u] = og;
    D.c_out[(int64_t)m * H + u] = cn;
    D.h_out[(int64_t)m * H + u] = hn;
    if (D.y_out) D.y_out[(int64_t)m * D.ldy + u] = hn;
  }
}

static int launch_lstm_step(const LstmStepP& p, int ndir, hipStream_t st) {
  const int nw = 8;
  const int MT = p.B > 32 ? 2 : 1;
  const size_t lds = (size_t)nw * MT * 16 * 64 * sizeof(float);
  dim3 grid(ag_cdiv(p.H, 8), ndir);
  if (MT == 2) {
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)lstm_step_fwd_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(lstm_step_fwd_kernel<2>, grid, dim3(64 * nw), lds, st, p);
  } else {
    hipLaunchKernelGGL(lstm_step_fwd_kernel<1>, grid, dim3(64 * nw), lds, st, p);
  }
  AG_CHECK_LAUNCH("ag_lstm_step_fwd");
  return AG_OK;
}

static bool step_ok(int B, int H, int Kx, int ldx, int ldwx, const void* x, const void* wx, const void* whh,
                    const void* h) {
  if (B > 64 || H % 8 != 0) return false;
  if (((uintptr_t)whh & 15) || ((uintptr_t)h & 15)) return false;
  if (x && (Kx % 8 != 0 || ldx % 4 != 0 || ldwx % 4 != 0 || ((uintptr_t)x & 15) || ((uintptr_t)wx & 15)))
    return false;
  return true;
}

// One fused step of a single LSTMCell (Generator front): gates_pre [B,4H] (in: zc-part + biases, out:
// activated gates), optional fed-back input x [B,Kx] with its weight columns wx [4H, ldwx].
extern "C" int ag_lstm_step_fwd(float* gates_pre, const float* x, int ldx, const float* wx, int ldwx, int Kx,
                                const float* h_prev, const float* whh, const float* c_prev, float* c_out,
                                float* h_out, int B, int H, int first_step, void* stream) {
  AG_REQUIRE(gates_pre && h_prev && whh && c_prev && c_out && h_out, "ag_lstm_step_fwd: null tensor");
  AG_REQUIRE(step_ok(B, H, Kx, ldx, ldwx, x, wx, whh, h_prev),
             "ag_lstm_step_fwd: needs B<=64, H%%8==0, Kx%%8==0 and 16-B aligned rows");
  LstmStepP p;
  p.valid = nullptr; p.B = B; p.H = H; p.skip_h = first_step ? 1 : 0;
  LstmDir& d = p.d[0];
  d.pre = gates_pre; d.x = (first_step ? nullptr : x); d.wx = wx; d.h_prev = h_prev; d.whh = whh;
  d.c_prev = c_prev; d.c_out = c_out; d.h_out = h_out; d.y_out = nullptr;
  d.ldx = ldx; d.ldwx = ldwx; d.Kx = Kx; d.ldy = 0; d.t = 0;
  p.d[1] = d;
  return launch_lstm_step(p, 1, (hipStream_t)stream);
}

// Whole (bi)directional NN.LSTM layer forward: T fused steps enqueued by ONE call.
//   pre   [ndir][T,B,4H]  x-projections + biases (overwritten with activated gates)
//   whh   [ndir][4H,H];  c_all [ndir][T+1,B,H] (c_all[.,0] = 0 on entry);  hbuf [ndir][2][B,H] scratch
//   y     [T,B,ndir*H] layer output.  Direction 1 runs the sequence in reverse.
extern "C" int ag_lstm_seq_fwd(float* const* pre, const float* const* whh, float* const* c_all,
                               float* const* hbuf, float* y, const int64_t* valid_i64, int T, int B, int H,
                               int ndir, void* stream) {
  AG_REQUIRE(pre && whh && c_all && hbuf && y, "ag_lstm_seq_fwd: null table");
  AG_REQUIRE(ndir == 1 || ndir == 2, "ag_lstm_seq_fwd: ndir must be 1 or 2");
  AG_REQUIRE(T > 0, "ag_lstm_seq_fwd: empty sequence");
  for (int d = 0; d < ndir; ++d)
    AG_REQUIRE(step_ok(B, H, 0, 0, 0, nullptr, nullptr, whh[d], hbuf[d]),
               "ag_lstm_seq_fwd: needs B<=64, H%%8==0 and 16-B aligned buffers");
  hipStream_t st = (hipStream_t)stream;
  const int64_t BH = (int64_t)B * H;
  for (int d = 0; d < ndir; ++d)
    if (hipMemsetAsync(hbuf[d], 0, sizeof(float) * BH, st) != hipSuccess) {
      ag_set_error("ag_lstm_seq_fwd: memset failed");
      return AG_ERR_LAUNCH;
    }
  for (int k = 0; k < T; ++k) {
    LstmStepP p;
    p.valid = valid_i64; p.B = B; p.H = H; p.skip_h = (k == 0);
    for (int d = 0; d < ndir; ++d) {
      const int t = d == 0 ? k : T - 1 - k;
      LstmDir& D = p.d[d];
      D.pre = pre[d] + (int64_t)t * B * 4 * H;
      D.x = nullptr; D.wx = nullptr; D.ldx = D.ldwx = D.Kx = 0;
      D.h_prev = hbuf[d] + (k & 1) * BH;
      D.h_out = hbuf[d] + ((k + 1) & 1) * BH;
      D.whh = whh[d];
      D.c_prev = c_all[d] + (int64_t)k * BH;
      D.c_out = c_all[d] + (int64_t)(k + 1) * BH;
      D.y_out = y + (int64_t)t * B * ndir * H + (int64_t)d * H;
      D.ldy = ndir * H;
      D.t = t;
    }
    if (ndir == 1) p.d[1] = p.d[0];
    int rc = launch_lstm_step(p, ndir, st);
    if (rc != AG_OK) return rc;
  }
  return AG_OK;
}

// Whole layer backward through time: per step one pointwise cell backward per direction and one
// K-split skinny product  dh_{k-1} += dgates_k * W_hh  (atomics into the buffer that already
// holds the pass-through term of padded rows).
//   gates [ndir][T,B,4H] activated gates (from forward);  dgates [ndir][T,B,4H] (out)
//   dy [T,B,ndir*H];  dhbuf/dcbuf [ndir][2][B,H] scratch
extern "C" int ag_lstm_seq_bwd(const float* const* gates, const float* const* whh,
                               const float* const* c_all, const float* dy, float* const* dgates,
                               float* const* dhbuf, float* const* dcbuf, const int64_t* valid_i64, int T,
                               int B, int H, int ndir, void* stream) {
  AG_REQUIRE(gates && whh && c_all && dy && dgates && dhbuf && dcbuf, "ag_lstm_seq_bwd: null table");
  AG_REQUIRE(ndir == 1 || ndir == 2, "ag_lstm_seq_bwd: ndir must be 1 or 2");
  AG_REQUIRE(T > 0 && B > 0 && B <= 64 && (4 * H) % 8 == 0, "ag_lstm_seq_bwd: bad shape");
  const int64_t BH = (int64_t)B * H, BG = (int64_t)B * 4 * H;
  for (int k = T - 1; k >= 0; --k) {
    for (int d = 0; d < ndir; ++d) {
      const int t = d == 0 ? k : T - 1 - k;
      const float* dh = (k == T - 1) ? nullptr : dhbuf[d] + (k & 1) * BH;
      const float* dcn = (k == T - 1) ? nullptr : dcbuf[d] + ((k + 1) & 1) * BH;
      float* dpass = dhbuf[d] + ((k + 1) & 1) * BH;
      float* dg = dgates[d] + (int64_t)t * BG;
      int rc = ag_lstm_cell_bwd(gates[d] + (int64_t)t * BG, 4 * H, c_all[d] + (int64_t)k * BH, H,
                                c_all[d] + (int64_t)(k + 1) * BH, H, dh, H,
                                dy + (int64_t)t * B * ndir * H + (int64_t)d * H, ndir * H, dcn, H, dg, 4 * H,
                                dcbuf[d] + (k & 1) * BH, H, dpass, H, valid_i64, t, B, H, stream);
      if (rc != AG_OK) return rc;
      if (k > 0) {
        rc = ag_skinny_gemm(dg, 4 * H, whh[d], H, 0, dpass, H, B, H, 4 * H, 1.f, nullptr, AG_ACT_NONE, 0.f,
                            1, stream);
        if (rc != AG_OK) return rc;
      }
    }
  }
  return AG_OK;
}
