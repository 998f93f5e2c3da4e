// lstm_step.hip -- the sequential part of the recurrent layers (NN.LSTMCell loop of the
// Generator, audiogan.py:437-443; NN.LSTM of the Discriminator, :498-503,:543).
//
// Per time step the work is a "skinny" product  [B<=64, K] x [K, N]  (B = clips per GPU):
// far too small for a tiled GEMM grid, and strictly sequential over time.  Kernels:
//
//  * skinny_gemm_kernel    C[M<=64, N] (+)= A[M,K] * op(B): every wave owns a 32x32 output
//    tile and a K slice; MFMA operands are fetched STRAIGHT from global/L2 as 16-byte
//    row pieces (no LDS staging: nothing is reused inside a workgroup), K slices are summed
//    through LDS inside a workgroup and with fp32 atomics across workgroups.  Up to two
//    independent problems (the two directions of a bidirectional layer) share one launch.
//  * lstm_step_fwd_kernel  one LSTM time step fused: gates = pre + [x,h] * [Wx|Whh]^T for the
//    4 gates of 8 hidden units per workgroup, then the cell non-linearity in the epilogue.
//    Both directions of a bidirectional layer run in the same launch (grid.y).
//  * lstm_cell_bwd2_kernel the pointwise cell backward for both directions in one launch.
//
// MFMA k-slot trick: v_mfma_f32_32x32x2_f32 sums two k values per instruction, lane half h
// supplying k-slot h.  Each lane loads a float4 A[row][8q+4h .. +3]; MFMA e (0..3) then
// contracts k = {8q+e, 8q+4+e}.  Any pairing is valid as long as A and B agree.
#include "common.h"

#define SK_MAXW 16

struct SkinnyOne {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
};

struct SkinnyP {
  SkinnyOne q[2];
  int lda, ldb, ldc;
  int M, N, K;
  int tb;        // 1: B stored [N][K]; 0: B stored [K][N]
  int act;
  float slope, beta;
  int rb;        // AG_PREC_BF16: operands rounded to bf16 in registers
  int mtiles;    // grid.z = nprob * mtiles
  float* part;   // K split over workgroups (gridDim.y > 1): partial [gridDim.y][nprob][M][N] (two-stage reduction)
                 // or NULL -> fp32 atomics into C
};

// A rows m0..m0+31 (k contiguous) x B rows (k contiguous), K range [k0,k1) (multiples of 8).
// Loads are UNCONDITIONAL (callers clamp out-of-range rows to a valid row; the garbage only
// reaches output rows/columns that the epilogue drops) and issued in batches of UNR k-steps
// before the first MFMA: with a predicated load hipcc branches around every load and waits
// vmcnt(0) per k-step, i.e. one dependent L2 round trip per 8 k instead of one per batch.
template <int UNR>
__device__ __forceinline__ void nt_batch(f32x16& acc, const float* __restrict__ ar,
                                         const float* __restrict__ br, int k, int rb) {
  f32x4 a[UNR], b[UNR];
#pragma unroll
  for (int u = 0; u < UNR; ++u) {
    b[u] = *reinterpret_cast<const f32x4*>(br + k + 8 * u);
    a[u] = *reinterpret_cast<const f32x4*>(ar + k + 8 * u);
  }
  if (rb) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) { a[u] = ag_rbf4_if(a[u], 1); b[u] = ag_rbf4_if(b[u], 1); }
  }
#pragma unroll
  for (int u = 0; u < UNR; ++u)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][e], b[u][e], acc, 0, 0, 0);
}

__device__ __forceinline__ void skinny_core_nt(f32x16& acc, const float* __restrict__ arow, bool aok,
                                               const float* __restrict__ brow, bool bok, int k0, int k1,
                                               int h, int rb) {
  (void)aok; (void)bok;
  const float* ar = arow + 4 * h;
  const float* br = brow + 4 * h;
  int k = k0;
  for (; k + 64 <= k1; k += 64) nt_batch<8>(acc, ar, br, k, rb);
  if (k + 32 <= k1) { nt_batch<4>(acc, ar, br, k, rb); k += 32; }
  if (k + 16 <= k1) { nt_batch<2>(acc, ar, br, k, rb); k += 16; }
  if (k + 8 <= k1) nt_batch<1>(acc, ar, br, k, rb);
}

// the same for MT 32-row tiles that share the B registers (acc[t], ar[t])
template <int UNR, int MT>
__device__ __forceinline__ void nt_batch_mt(f32x16* acc, const float* const* ar, const float* __restrict__ br, int k,
                                            int rb) {
  f32x4 a[MT][UNR], b[UNR];
#pragma unroll
  for (int u = 0; u < UNR; ++u) {
    b[u] = *reinterpret_cast<const f32x4*>(br + k + 8 * u);
#pragma unroll
    for (int t = 0; t < MT; ++t) a[t][u] = *reinterpret_cast<const f32x4*>(ar[t] + k + 8 * u);
  }
  if (rb) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      b[u] = ag_rbf4_if(b[u], 1);
#pragma unroll
      for (int t = 0; t < MT; ++t) a[t][u] = ag_rbf4_if(a[t][u], 1);
    }
  }
#pragma unroll
  for (int u = 0; u < UNR; ++u)
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][u][e], b[u][e], acc[t], 0, 0, 0);
}

template <int MT>
__device__ __forceinline__ void skinny_core_nt_mt(f32x16* acc, const float* const* arow, const float* __restrict__ brow,
                                                  int k0, int k1, int h, int rb) {
  const float* ar[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) ar[t] = arow[t] + 4 * h;
  const float* br = brow + 4 * h;
  int k = k0;
  for (; k + 64 <= k1; k += 64) nt_batch_mt<8, MT>(acc, ar, br, k, rb);
  if (k + 32 <= k1) { nt_batch_mt<4, MT>(acc, ar, br, k, rb); k += 32; }
  if (k + 16 <= k1) { nt_batch_mt<2, MT>(acc, ar, br, k, rb); k += 16; }
  if (k + 8 <= k1) nt_batch_mt<1, MT>(acc, ar, br, k, rb);
}

// B stored [K][N]: lane j reads B[k][n0+j] (coalesced along n)
template <int UNR>
__device__ __forceinline__ void nn_batch(f32x16& acc, const float* __restrict__ ar,
                                         const float* __restrict__ bc, int ldb, int k, int rb) {
  f32x4 a[UNR];
  float b[UNR][4];
#pragma unroll
  for (int u = 0; u < UNR; ++u) {
#pragma unroll
    for (int e = 0; e < 4; ++e) b[u][e] = bc[(int64_t)(k + 8 * u + e) * ldb];
    a[u] = *reinterpret_cast<const f32x4*>(ar + k + 8 * u);
  }
  if (rb) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      a[u] = ag_rbf4_if(a[u], 1);
#pragma unroll
      for (int e = 0; e < 4; ++e) b[u][e] = ag_rbf(b[u][e]);
    }
  }
#pragma unroll
  for (int u = 0; u < UNR; ++u)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][e], b[u][e], acc, 0, 0, 0);
}

__device__ __forceinline__ void skinny_core_nn(f32x16& acc, const float* __restrict__ arow, bool aok,
                                               const float* __restrict__ bcol, int ldb, bool bok, int k0,
                                               int k1, int h, int rb) {
  (void)aok; (void)bok;
  const float* ar = arow + 4 * h;
  const float* bc = bcol + (int64_t)4 * h * ldb;
  int k = k0;
  for (; k + 32 <= k1; k += 32) nn_batch<4>(acc, ar, bc, ldb, k, rb);
  if (k + 16 <= k1) { nn_batch<2>(acc, ar, bc, ldb, k, rb); k += 16; }
  if (k + 8 <= k1) nn_batch<1>(acc, ar, bc, ldb, k, rb);
}

// sum the per-wave 32x32 accumulators through LDS; afterwards red[0..1023] holds the block total,
// element (e, lane) at red[e*64 + lane]
__device__ __forceinline__ void block_reduce_acc(const f32x16& acc, float* red, int nw, int wid, int lane) {
  float* mine = red + (size_t)wid * 1024;
#pragma unroll
  for (int e = 0; e < 16; ++e) mine[e * 64 + lane] = acc[e];
  __syncthreads();
  for (int o = threadIdx.x; o < 1024; o += blockDim.x) {
    float s = red[o];
    for (int w = 1; w < nw; ++w) s += red[(size_t)w * 1024 + o];
    red[o] = s;
  }
  __syncthreads();
}

__global__ __launch_bounds__(1024) void skinny_gemm_kernel(const SkinnyP p) {
  extern __shared__ float red[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const SkinnyOne& Q = p.q[blockIdx.z / p.mtiles];
  const int m0 = (blockIdx.z % p.mtiles) * 32;
  const int n0 = blockIdx.x * 32;
  const int W = gridDim.y * nw, wg = blockIdx.y * nw + wid;
  const int KU = p.K >> 3;
  const int k0 = (int)((int64_t)KU * wg / W) * 8, k1 = (int)((int64_t)KU * (wg + 1) / W) * 8;
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const bool bok = (n0 + l31) < p.N, aok = (m0 + l31) < p.M;
  const float* arow = Q.A + (int64_t)(aok ? m0 + l31 : 0) * p.lda;
  if (p.tb)
    skinny_core_nt(acc, arow, aok, Q.B + (int64_t)(bok ? n0 + l31 : 0) * p.ldb, bok, k0, k1, h, p.rb);
  else
    skinny_core_nn(acc, arow, aok, Q.B + (bok ? n0 + l31 : 0), p.ldb, bok, k0, k1, h, p.rb);
  block_reduce_acc(acc, red, nw, wid, lane);
  const bool direct = gridDim.y == 1;
  for (int o = threadIdx.x; o < 1024; o += blockDim.x) {
    const int ln = o & 63, e = o >> 6;
    const int m = m0 + (e & 3) + 8 * (e >> 2) + 4 * (ln >> 5);
    const int n = n0 + (ln & 31);
    if (m >= p.M || n >= p.N) continue;
    float* dst = Q.C + (int64_t)m * p.ldc + n;
    float v = red[o];
    if (direct) {
      if (p.beta != 0.f) v += p.beta * *dst;
      if (Q.bias) v += Q.bias[n];
      *dst = ag_apply_act(v, p.act, p.slope);
    } else {      // K split over workgroups: partial products into the slabs (the host binds them; fixed-order second stage)
      const int prob = blockIdx.z / p.mtiles, nprob = gridDim.z / p.mtiles;
      p.part[(((int64_t)blockIdx.y * nprob + prob) * p.M + m) * p.N + n] = v;
    }
  }
}

// 16 x 16 output tiles (v_mfma_f32_16x16x4_f32), both operands k-contiguous, 16 waves splitting K, every load
// of a wave requested before its first MFMA, direct (bias + activation) epilogue.  For products whose 32-wide
// tiling leaves most CUs idle and that sit on a sequential critical path: the Generator front's projection
// x_t = tanh(h_t W_p^T + b) is 64 x 256 x 1024 - 16 workgroups as 32x32 tiles, 64 as 16x16.
__global__ __launch_bounds__(1024) void skinny16_nt_kernel(const SkinnyP p) {
  __shared__ float red[16 * 256];
  const SkinnyOne& Q = p.q[0];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 16;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int KU = p.K >> 4;
  const int u0 = KU * wid / 16, u1 = KU * (wid + 1) / 16;
  const float* ar = Q.A + (int64_t)min(m0 + li, p.M - 1) * p.lda + 4 * g;
  const float* br = Q.B + (int64_t)min(n0 + li, p.N - 1) * p.ldb + 4 * g;
  for (int ub = u0; ub < u1; ub += 4) {
    f32x4 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int u = min(ub + i, u1 - 1);
      a[i] = ag_rbf4_if(*reinterpret_cast<const f32x4*>(ar + 16 * u), p.rb);
      b[i] = ag_rbf4_if(*reinterpret_cast<const f32x4*>(br + 16 * u), p.rb);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (ub + i < u1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][e], b[i][e], acc, 0, 0, 0);
      }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[wid * 256 + (4 * g + e) * 16 + li] = acc[e];     // row 4g+e, col li
  __syncthreads();
  if (threadIdx.x >= 256) return;
  const int m = m0 + (threadIdx.x >> 4), n = n0 + (threadIdx.x & 15);
  if (m >= p.M || n >= p.N) return;
  float v = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) v += red[w * 256 + threadIdx.x];
  float* dst = Q.C + (int64_t)m * p.ldc + n;
  if (p.beta != 0.f) v += p.beta * *dst;
  if (Q.bias) v += Q.bias[n];
  *dst = ag_apply_act(v, p.act, p.slope);
}

static int launch_skinny(SkinnyP& p, int nprob, int accumulate_atomic, hipStream_t st, AgWs ws = AgWs{nullptr, 0}) {
  p.part = nullptr;
  p.rb = ag_precision() == AG_PREC_BF16;
  const int gx = ag_cdiv(p.N, 32);
  const int KU = p.K / 8;
  p.mtiles = ag_cdiv(p.M, 32);
  const int gz = nprob * p.mtiles;
  int nw, gy;
  // few 32-wide tiles, direct epilogue, both operands k-contiguous: 16x16 tiles put 4x as many CUs on it
  if (!accumulate_atomic && nprob == 1 && p.tb && p.K % 16 == 0 && p.K >= 256 && gx * gz < 64 &&
      (int64_t)ag_cdiv(p.N, 16) * ag_cdiv(p.M, 16) <= 1024) {
    hipLaunchKernelGGL(skinny16_nt_kernel, dim3(ag_cdiv(p.N, 16), ag_cdiv(p.M, 16)), dim3(1024), 0, st, p);
    AG_CHECK_LAUNCH("ag_skinny_gemm");
    return AG_OK;
  }
  if (accumulate_atomic) {
    // C already holds the value to add to; spread K over ~4096 waves on the chip: each wave's K slice is a chain
    // of dependent load batches, so shorter slices cut the latency (measured: 1024 -> 4096 waves, step -0.15 ms)
    nw = 4;
    gy = ag_cdiv(4096, gx * gz * nw);
    if (gy * nw > KU) gy = ag_cdiv(KU, nw);
    if (gy < 2) gy = 2;  // keep the split epilogue (C holds the addend)
    const int64_t slab = (int64_t)nprob * p.M * p.N;
    AG_REQUIRE(ws.p && ws.numel >= 2 * slab, "%s: this reduction spans several workgroups and needs a workspace of >= %lld floats bound with ag_bind_workspace (sums are two-stage, in a fixed order; there is no float-atomic accumulation)", "ag_skinny_gemm (accumulate mode)", (long long)2 * slab);
    if ((int64_t)gy * slab > ws.numel) gy = (int)(ws.numel / slab);     // two-stage: partial products, then a fixed-order sum into C
    p.part = ws.p;
  } else {
    gy = 1;
    nw = ag_cdiv(2048, gx * gz);
    if (nw < 4) nw = 4;
    if (nw > SK_MAXW) nw = SK_MAXW;
    if (nw > KU) nw = KU;
  }
  const size_t lds = (size_t)nw * 1024 * sizeof(float);
  hipLaunchKernelGGL(skinny_gemm_kernel, dim3(gx, gy, gz), dim3(64 * nw), lds, st, p);
  AG_CHECK_LAUNCH("ag_skinny_gemm");
  if (p.part) {
    for (int q = 0; q < nprob; ++q) {
      // slab z of problem q starts at part + (z*nprob + q)*M*N: present it as Z slabs of pitch nprob*M*N
      const int rc = ag_splitk_reduce(p.part + (int64_t)q * p.M * p.N, gy, (int64_t)nprob * p.M * p.N, p.M, p.N, p.q[q].C,
                                      p.ldc, 1.f, p.q[q].bias, nullptr, 0, st);
      if (rc != AG_OK) return rc;
    }
  }
  return AG_OK;
}

// floats of workspace ag_skinny_gemm (accumulate mode) wants bound for its two-stage reduction
extern "C" int64_t ag_skinny_ws_numel(int M, int N, int K) {
  const int gx = ag_cdiv(N, 32), gz = ag_cdiv(M, 32), KU = K / 8;
  int gy = ag_cdiv(4096, gx * gz * 4);
  if (gy * 4 > KU) gy = ag_cdiv(KU, 4);
  if (gy < 2) gy = 2;
  return (int64_t)gy * M * N;
}

extern "C" int ag_skinny_gemm(const float* A, int lda, const float* B, int ldb, int tb, float* C, int ldc,
                              int M, int N, int K, float beta, const float* bias, int act, float slope,
                              int accumulate_atomic, void* stream) {
  const AgWs ws = ag_ws_take();     // FIRST: an argument error below must not leave a stale binding behind
  AG_REQUIRE(A && B && C, "ag_skinny_gemm: null tensor");
  AG_REQUIRE(M > 0 && M <= 256 && N > 0 && K > 0, "ag_skinny_gemm: needs 0 < M <= 256");
  AG_REQUIRE(K % 8 == 0 && lda % 4 == 0 && ((uintptr_t)A & 15) == 0, "ag_skinny_gemm: A must be 16-B aligned, K%%8==0");
  if (tb) AG_REQUIRE(ldb % 4 == 0 && ((uintptr_t)B & 15) == 0, "ag_skinny_gemm: B must be 16-B aligned");
  AG_REQUIRE(!(accumulate_atomic && act != AG_ACT_NONE), "ag_skinny_gemm: atomic mode has a linear epilogue");
  SkinnyP p;
  p.q[0].A = A; p.q[0].B = B; p.q[0].C = C; p.q[0].bias = bias;
  p.q[1] = p.q[0];
  p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.M = M; p.N = N; p.K = K; p.tb = tb; p.act = act;
  p.slope = slope; p.beta = beta;
  return launch_skinny(p, 1, accumulate_atomic, (hipStream_t)stream, ws);
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
  const float* cb;     // optional [B, 4H]: time-invariant part of the pre-activations (static input + biases)
  int ldx, ldwx, Kx, ldy;
  int t;               // time index (for the valid mask)
};

struct LstmStepP {
  LstmDir d[2];
  const int64_t* valid;
  int B, H;
  int skip_h;          // 1: h_prev is known to be zero (first step) -> skip that product
  int rb;              // AG_PREC_BF16: operands rounded to bf16 in registers
};

template <int MT>      // 32-clip row tiles per workgroup (they share the weight registers)
__global__ __launch_bounds__(512) void lstm_step_fwd_kernel(const LstmStepP p) {
  extern __shared__ float red[];
  const LstmDir& D = p.d[blockIdx.y];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int H = p.H, B = p.B;
  const int u0 = blockIdx.x * 8;
  const int m0 = blockIdx.z * 32 * MT;
  // column j of this workgroup's 32-wide tile = gate (j>>3), hidden unit u0 + (j&7)
  const int unit = u0 + (l31 & 7);
  const bool bok = unit < H;
  const int row = (l31 >> 3) * H + (bok ? unit : 0);
  // epilogue operands of this thread's (clip, unit) pairs are requested FIRST so that their
  // latency overlaps the product (one dependent memory round trip less per time step)
  const int euu = threadIdx.x & 7, emm = (threadIdx.x >> 3) & 31;
  const int eu = u0 + euu;
  const bool ethread = threadIdx.x < 256 && eu < H;
  float pre4[MT][4], cp[MT], hp[MT];
  bool padded[MT], epi[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int em = m0 + 32 * t + emm;
    epi[t] = ethread && em < B;
    pre4[t][0] = pre4[t][1] = pre4[t][2] = pre4[t][3] = 0.f;
    cp[t] = hp[t] = 0.f;
    padded[t] = false;
    if (epi[t]) {
      const float* pre = D.pre + (int64_t)em * 4 * H + eu;
      pre4[t][0] = pre[0]; pre4[t][1] = pre[H]; pre4[t][2] = pre[2 * H]; pre4[t][3] = pre[3 * H];
      if (D.cb) {
        const float* cb = D.cb + (int64_t)em * 4 * H + eu;
        pre4[t][0] += cb[0]; pre4[t][1] += cb[H]; pre4[t][2] += cb[2 * H]; pre4[t][3] += cb[3 * H];
      }
      cp[t] = D.c_prev[(int64_t)em * H + eu];
      padded[t] = p.valid && D.t >= p.valid[em];
      if (padded[t]) hp[t] = D.h_prev[(int64_t)em * H + eu];
    }
  }
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
  const float* arow[MT];
  if (s0 < KxU) {
    const int e1 = s1 < KxU ? s1 : KxU;
#pragma unroll
    for (int t = 0; t < MT; ++t) arow[t] = D.x + (int64_t)min(m0 + 32 * t + l31, B - 1) * D.ldx;
    skinny_core_nt_mt<MT>(acc, arow, D.wx + (int64_t)row * D.ldwx, s0 * 8, e1 * 8, h, p.rb);
  }
  if (s1 > KxU) {
    const int b0 = (s0 > KxU ? s0 : KxU) - KxU;
#pragma unroll
    for (int t = 0; t < MT; ++t) arow[t] = D.h_prev + (int64_t)min(m0 + 32 * t + l31, B - 1) * H;
    skinny_core_nt_mt<MT>(acc, arow, D.whh + (int64_t)row * H, b0 * 8, (s1 - KxU) * 8, h, p.rb);
  }
  // per-wave accumulators -> LDS, element (t, e, lane) of wave w at red[(w*MT + t)*1024 + e*64 + lane]
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) red[(size_t)(wid * MT + t) * 1024 + e * 64 + lane] = acc[t][e];
  __syncthreads();
  // epilogue: (m, unit) pairs; the 4 gates of a pair sit at columns uu, 8+uu, 16+uu, 24+uu
  if (!ethread) return;
  // inverse of row = (e&3) + 8*(e>>2) + 4*(lane>>5)
  const int hh = (emm >> 2) & 1, e = (emm & 3) + 4 * (emm >> 3);
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    if (!epi[t]) continue;
    const int em = m0 + 32 * t + emm;
    float pr[4] = {0.f, 0.f, 0.f, 0.f};
    for (int w = 0; w < nw; ++w) {
      const float* rr = red + (size_t)(w * MT + t) * 1024 + e * 64 + 32 * hh + euu;
#pragma unroll
      for (int q = 0; q < 4; ++q) pr[q] += rr[8 * q];
    }
    float* pre = D.pre + (int64_t)em * 4 * H + eu;
    const int64_t o = (int64_t)em * H + eu;
    if (padded[t]) {
      D.h_out[o] = hp[t];
      D.c_out[o] = cp[t];
      if (D.y_out) D.y_out[(int64_t)em * D.ldy + eu] = 0.f;
      continue;
    }
    const float ig = ag_sigmoid(pre4[t][0] + pr[0]);
    const float fg = ag_sigmoid(pre4[t][1] + pr[1]);
    const float gg = tanhf(pre4[t][2] + pr[2]);
    const float og = ag_sigmoid(pre4[t][3] + pr[3]);
    const float cn = fg * cp[t] + ig * gg;
    const float hn = og * tanhf(cn);
    pre[0] = ig;
    pre[H] = fg;
    pre[2 * H] = gg;
    pre[3 * H] = og;
    D.c_out[o] = cn;
    D.h_out[o] = hn;
    if (D.y_out) D.y_out[(int64_t)em * D.ldy + eu] = hn;
  }
}

static int launch_lstm_step(LstmStepP& p, int ndir, hipStream_t st) {
  const int nw = 8;
  p.rb = ag_precision() == AG_PREC_BF16;
  // one wave of workgroups on the 256 CUs: two 32-clip tiles per workgroup once single tiles would not fit
  if ((int64_t)ag_cdiv(p.H, 8) * ndir * ag_cdiv(p.B, 32) > 256) {
    dim3 grid(ag_cdiv(p.H, 8), ndir, ag_cdiv(p.B, 64));
    hipLaunchKernelGGL(lstm_step_fwd_kernel<2>, grid, dim3(64 * nw), (size_t)nw * 2048 * sizeof(float), st, p);
  } else {
    dim3 grid(ag_cdiv(p.H, 8), ndir, ag_cdiv(p.B, 32));
    hipLaunchKernelGGL(lstm_step_fwd_kernel<1>, grid, dim3(64 * nw), (size_t)nw * 1024 * sizeof(float), st, p);
  }
  AG_CHECK_LAUNCH("ag_lstm_step_fwd");
  return AG_OK;
}

static bool step_ok(int B, int H, int Kx, int ldx, int ldwx, const void* x, const void* wx, const void* whh,
                    const void* h) {
  if (B > 256 || H % 8 != 0) return false;
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
             "ag_lstm_step_fwd: needs B<=256, H%%8==0, Kx%%8==0 and 16-B aligned rows");
  LstmStepP p;
  p.valid = nullptr; p.B = B; p.H = H; p.skip_h = first_step ? 1 : 0;
  LstmDir& d = p.d[0];
  d.pre = gates_pre; d.x = (first_step ? nullptr : x); d.wx = wx; d.h_prev = h_prev; d.whh = whh;
  d.cb = nullptr; d.c_prev = c_prev; d.c_out = c_out; d.h_out = h_out; d.y_out = nullptr;
  d.ldx = ldx; d.ldwx = ldwx; d.Kx = Kx; d.ldy = 0; d.t = 0;
  p.d[1] = d;
  return launch_lstm_step(p, 1, (hipStream_t)stream);
}

// Whole (bi)directional NN.LSTM layer forward: T fused steps enqueued by ONE call.
//   pre   [ndir][T,B,4H]  x-projections + biases (overwritten with activated gates)
//   whh   [ndir][4H,H];  c_all [ndir][T+1,B,H] (c_all[.,0] = 0 on entry);  hbuf [ndir][2][B,H] scratch
//   y     [T,B,ndir*H] layer output.  Direction 1 runs the sequence in reverse.
extern "C" int ag_lstm_seq_fwd(float* const* pre, const float* const* whh, float* const* c_all,
                               float* const* hbuf, float* y, const int64_t* valid_i64,
                               const float* const* static_pre, int T, int B, int H, int ndir, int k_begin,
                               int k_end, void* stream) {
  AG_REQUIRE(pre && whh && c_all && hbuf && y, "ag_lstm_seq_fwd: null table");
  AG_REQUIRE(ndir == 1 || ndir == 2, "ag_lstm_seq_fwd: ndir must be 1 or 2");
  AG_REQUIRE(T > 0 && 0 <= k_begin && k_begin <= k_end && k_end <= T, "ag_lstm_seq_fwd: bad step range");
  for (int d = 0; d < ndir; ++d)
    AG_REQUIRE(step_ok(B, H, 0, 0, 0, nullptr, nullptr, whh[d], hbuf[d]),
               "ag_lstm_seq_fwd: needs B<=256, H%%8==0 and 16-B aligned buffers");
  hipStream_t st = (hipStream_t)stream;
  const int64_t BH = (int64_t)B * H;
  for (int d = 0; d < ndir && k_begin == 0; ++d)
    if (hipMemsetAsync(hbuf[d], 0, sizeof(float) * BH, st) != hipSuccess) {
      ag_set_error("ag_lstm_seq_fwd: memset failed");
      return AG_ERR_LAUNCH;
    }
  for (int k = k_begin; k < k_end; ++k) {
    LstmStepP p;
    p.valid = valid_i64; p.B = B; p.H = H; p.skip_h = (k == 0);
    for (int d = 0; d < ndir; ++d) {
      const int t = d == 0 ? k : T - 1 - k;
      LstmDir& D = p.d[d];
      D.pre = pre[d] + (int64_t)t * B * 4 * H;
      D.x = nullptr; D.wx = nullptr; D.ldx = D.ldwx = D.Kx = 0;
      D.cb = static_pre ? static_pre[d] : nullptr;
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

// ------------------------------------------------------------------------------------------
// backward through time
// ------------------------------------------------------------------------------------------
struct CellBwdDir {
  const float* ga;      // activated gates [B,4H]
  const float* c_prev;  // [B,H]
  const float* c_new;   // [B,H]
  const float* dh;      // [B,H] gradient from later steps or NULL
  const float* dy;      // [B,ldy] slice or NULL
  const float* dc_next; // [B,H] or NULL
  float* dgates;        // [B,4H]
  float* dc_prev;       // [B,H]
  float* dh_pass;       // [B,H]
  int ldy, t;
};
struct CellBwd2P {
  CellBwdDir d[2];
  const int64_t* valid;
  int B, H;
};

__global__ __launch_bounds__(256) void lstm_cell_bwd2_kernel(const CellBwd2P p) {
  const CellBwdDir& D = p.d[blockIdx.z];
  const int j = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y, H = p.H;
  if (j >= H) return;
  float* dg = D.dgates + (int64_t)b * 4 * H;
  const float dhf = D.dh ? D.dh[(int64_t)b * H + j] : 0.f;
  const float dcn = D.dc_next ? D.dc_next[(int64_t)b * H + j] : 0.f;
  if (p.valid && D.t >= p.valid[b]) {
    dg[j] = 0.f; dg[H + j] = 0.f; dg[2 * H + j] = 0.f; dg[3 * H + j] = 0.f;
    D.dc_prev[(int64_t)b * H + j] = dcn;
    D.dh_pass[(int64_t)b * H + j] = dhf;
    return;
  }
  const float dhv = dhf + (D.dy ? D.dy[(int64_t)b * D.ldy + j] : 0.f);
  const float* gr = D.ga + (int64_t)b * 4 * H;
  const float ig = gr[j], fg = gr[H + j], gg = gr[2 * H + j], og = gr[3 * H + j];
  const float cp = D.c_prev[(int64_t)b * H + j];
  const float tc = tanhf(D.c_new[(int64_t)b * H + j]);
  const float dc = dcn + dhv * og * (1.f - tc * tc);
  dg[j] = dc * gg * ig * (1.f - ig);
  dg[H + j] = dc * cp * fg * (1.f - fg);
  dg[2 * H + j] = dc * ig * (1.f - gg * gg);
  dg[3 * H + j] = dhv * tc * og * (1.f - og);
  D.dc_prev[(int64_t)b * H + j] = dc * fg;
  D.dh_pass[(int64_t)b * H + j] = 0.f;
}

// ------------------------------------------------------------------------------------------
// fused backward step: product of the NEXT processed step + cell backward of THIS step.
//   dh[m,u] = dh_pass_in[m,u] + sum_kk dgates_{k+1}[m,kk] * W_hh[kk,u]        (k < T-1)
//   (dgates_k, dc_prev, dh_pass_out) = cell_bwd(dh + dy_k, dc_next, saved gates / cells of step k)
// One workgroup = 16 clips x 16 hidden units; its 16 waves split K = 4H and are summed through LDS,
// so the result is complete inside the workgroup (no atomics) and the cell backward runs in the
// epilogue: ONE launch per time step instead of a pointwise launch plus an atomic product launch.
// ------------------------------------------------------------------------------------------
struct BwdStepDir {
  const float* dg_next;  // product operand A [B,Kp] (row pitch lda): dgates of the previously processed step
                         // (k+1), or NULL when k == T-1
  const float* whh;      // product operand [Kp,H]
  const float* ax;       // optional: A is used as A * (1 - ax^2) (tanh backward applied on load), row pitch ldax
  float* aout;           // optional: that transformed A is also stored here (row pitch ldaout)
  int lda, Kp, ldax, ldaout, lddhp;   // lddhp: row pitch of dh_pass_in
  const float* ga;       // activated gates of step k [B,4H]
  const float* c_prev;   // [B,H]
  const float* c_new;    // [B,H]
  const float* dy;       // [B,ldy] slice
  const float* dc_next;  // [B,H] or NULL
  const float* dh_pass_in;  // [B,H] or NULL
  float* dgates;         // [B,4H] (out)
  float* dc_prev;        // [B,H] (out)
  float* dh_pass_out;    // [B,H] (out)
  int ldy, t;
};
struct BwdStepP {
  BwdStepDir d[2];
  const int64_t* valid;
  int B, H;
  int rb;               // AG_PREC_BF16: operands rounded to bf16 in registers
};

// 16 clips x 16 units per workgroup (v_mfma_f32_16x16x4_f32: lane (i = l&15, g = l>>4) supplies
// A[i][k-slot g] and B[k-slot g][i]).  A lane loads one float4 of its dgates row per 16-k unit and the
// four MFMAs of the unit take elements 0..3 as k = 16u + 4g + e, so the row is read 16 bytes at a time.
// The step is latency bound (W_hh comes from the fabric every step: L2 is not coherent across
// launches), so every load of a wave's K range is requested before the first MFMA.
#define BWD_UB 8      // 16-k units requested per round (8 float4 + 32 dwords in flight per lane)
// UB: 16-k units requested per round; AX: tanh backward applied to A on load (generator front)
template <int MT, int NT, int UB, bool AX>      // 16-clip row tiles x 16-unit column tiles per workgroup
__global__ __launch_bounds__(1024) void lstm_step_bwd_kernel(const BwdStepP p) {
  constexpr int TM = 16 * MT, TN = 16 * NT;
  __shared__ float red[16 * TM * TN];
  // (an XCD-aware workgroup order - one direction and a fixed quarter of the unit tiles per XCD, to cut
  // the per-step L2 fills - was measured: no change)
  const BwdStepDir& D = p.d[blockIdx.y];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int H = p.H, B = p.B;
  // 16-unit tiles split a 128-byte W_hh line between two workgroups: put both on the same XCD (same L2);
  // workgroup ids go round-robin over the 8 XCDs and blockIdx.x is the fastest index
  int bx = blockIdx.x;
  if (NT == 1 && (gridDim.x & 15) == 0) bx = (bx & ~15) | ((bx & 7) << 1) | ((bx >> 3) & 1);
  const int n0 = bx * TN, m0 = blockIdx.z * TM;
  // epilogue (clip, unit) of threads 0..TM*TN-1; operands requested up front
  const int eu = n0 + (threadIdx.x % TN), em = m0 + (threadIdx.x / TN) % TM;
  const bool epi = threadIdx.x < TM * TN && em < B && eu < H;
  float ig = 0.f, fg = 0.f, gg = 0.f, og = 0.f, cp = 0.f, cn = 0.f, dyv = 0.f, dcn = 0.f, dpi = 0.f;
  bool padded = false;
  if (epi) {
    const float* gr = D.ga + (int64_t)em * 4 * H + eu;
    ig = gr[0]; fg = gr[H]; gg = gr[2 * H]; og = gr[3 * H];
    const int64_t o = (int64_t)em * H + eu;
    cp = D.c_prev[o];
    cn = D.c_new[o];
    if (D.dy) dyv = D.dy[(int64_t)em * D.ldy + eu];
    if (D.dc_next) dcn = D.dc_next[o];
    if (D.dh_pass_in) dpi = D.dh_pass_in[(int64_t)em * D.lddhp + eu];
    padded = p.valid && D.t >= p.valid[em];
  }
  f32x4 acc[MT][NT];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int c = 0; c < NT; ++c) acc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (D.dg_next) {
    const int KU = D.Kp >> 4;                           // 16-k units
    const int u0 = KU * wid / 16, u1 = KU * (wid + 1) / 16;
    // rows / units past the end are clamped: they only feed their own (unwritten) outputs
    const float* ar[MT];
    const float* br[NT];
#pragma unroll
    for (int t = 0; t < MT; ++t) ar[t] = D.dg_next + (int64_t)min(m0 + 16 * t + li, B - 1) * D.lda + 4 * g;
#pragma unroll
    for (int c = 0; c < NT; ++c) br[c] = D.whh + (int64_t)(4 * g) * H + min(n0 + 16 * c + li, H - 1);
    for (int ub = u0; ub < u1; ub += UB) {
      f32x4 a[MT][UB];
      float b[NT][UB][4];
#pragma unroll
      for (int i = 0; i < UB; ++i) {
        const int u = min(ub + i, u1 - 1);
#pragma unroll
        for (int t = 0; t < MT; ++t) a[t][i] = *reinterpret_cast<const f32x4*>(ar[t] + 16 * u);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int c = 0; c < NT; ++c) b[c][i][e] = br[c][(int64_t)(16 * u + e) * H];
      }
      if (AX) {       // generator front, A = dL/dx_t * tanh'(pre) = dxa * (1 - x_t^2)
#pragma unroll
        for (int i = 0; i < UB; ++i) {
          const int u = min(ub + i, u1 - 1);
#pragma unroll
          for (int t = 0; t < MT; ++t) {
            const int row = min(m0 + 16 * t + li, B - 1);
            const f32x4 xv = *reinterpret_cast<const f32x4*>(D.ax + (int64_t)row * D.ldax + 4 * g + 16 * u);
            a[t][i] = a[t][i] * (1.f - xv * xv);
            if (D.aout && blockIdx.x == 0 && ub + i < u1 && m0 + 16 * t + li < B)
              *reinterpret_cast<f32x4*>(D.aout + (int64_t)row * D.ldaout + 4 * g + 16 * u) = a[t][i];
          }
        }
      }
      if (p.rb) {
#pragma unroll
        for (int i = 0; i < UB; ++i) {
#pragma unroll
          for (int t = 0; t < MT; ++t) a[t][i] = ag_rbf4_if(a[t][i], 1);
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int c = 0; c < NT; ++c) b[c][i][e] = ag_rbf(b[c][i][e]);
        }
      }
#pragma unroll
      for (int i = 0; i < UB; ++i) {
        if (ub + i < u1) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
              for (int c = 0; c < NT; ++c)
                acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][i][e], b[c][i][e], acc[t][c], 0, 0, 0);
        }
      }
    }
  }
  // C layout of a 16x16 tile: col = lane & 15, row = 4 * (lane >> 4) + e
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int c = 0; c < NT; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) red[wid * TM * TN + (16 * t + 4 * g + e) * TN + 16 * c + li] = acc[t][c][e];
  __syncthreads();
  if (!epi) return;
  const int ot = threadIdx.x;          // = (row em - m0) * TN + (col eu - n0)
  float dhf = dpi;                     // gradient reaching h_k from later steps
#pragma unroll
  for (int w = 0; w < 16; ++w) dhf += red[w * TM * TN + ot];
  float* dg = D.dgates + (int64_t)em * 4 * H + eu;
  const int64_t o = (int64_t)em * H + eu;
  if (padded) {
    dg[0] = 0.f; dg[H] = 0.f; dg[2 * H] = 0.f; dg[3 * H] = 0.f;
    D.dc_prev[o] = dcn;
    if (D.dh_pass_out) D.dh_pass_out[o] = dhf;
    return;
  }
  const float dhv = dhf + dyv;
  const float tc = tanhf(cn);
  const float dc = dcn + dhv * og * (1.f - tc * tc);
  dg[0] = dc * gg * ig * (1.f - ig);
  dg[H] = dc * cp * fg * (1.f - fg);
  dg[2 * H] = dc * ig * (1.f - gg * gg);
  dg[3 * H] = dhv * tc * og * (1.f - og);
  D.dc_prev[o] = dc * fg;
  if (D.dh_pass_out) D.dh_pass_out[o] = 0.f;
}

static void launch_bwd_step(BwdStepP& q, int ndir, hipStream_t st) {
  const int B = q.B, H = q.H;
  q.rb = ag_precision() == AG_PREC_BF16;
  // one wave of workgroups fills the 256 CUs; once 16x16 tiles would need more, widen the tile along the
  // units (full 128-byte W_hh lines per workgroup)
  if ((int64_t)(H / 16) * ndir * ag_cdiv(B, 16) > 256)
    hipLaunchKernelGGL((lstm_step_bwd_kernel<1, 2, 8, false>), dim3(ag_cdiv(H, 32), ndir, ag_cdiv(B, 16)), dim3(1024),
                       0, st, q);
  else
    hipLaunchKernelGGL((lstm_step_bwd_kernel<1, 1, 8, false>), dim3(H / 16, ndir, ag_cdiv(B, 16)), dim3(1024), 0, st,
                       q);
}

// One fused backward step of a single LSTMCell whose h feeds a tanh projection (Generator front,
// audiogan.py:428-460), for frame t:
//   gx   = dxa * (1 - x_t^2)                      (d(pre-tanh) of the projection; stored to gx_out)
//   dh   = dh_acc + gx * W_proj                   (W_proj [Kp = frame, H])
//   (dgates, dc_prev) = cell_bwd(dh, dc_next, gates, c_prev, c_new)
extern "C" int ag_lstm_front_bwd_step(const float* dxa, int lddxa, const float* x, int ldx, float* gx_out, int ldgx,
                                      int Kp, const float* w_proj, const float* dh_acc, int lddh, const float* gates,
                                      const float* c_prev, const float* c_new, const float* dc_next, float* dgates,
                                      float* dc_prev, int B, int H, void* stream) {
  AG_REQUIRE(dxa && x && w_proj && dh_acc && gates && c_prev && c_new && dgates && dc_prev,
             "ag_lstm_front_bwd_step: null tensor");
  AG_REQUIRE(B > 0 && B <= 65535 && H % 16 == 0 && Kp > 0 && Kp % 16 == 0 && lddh >= H,
             "ag_lstm_front_bwd_step: bad shape");
  AG_REQUIRE(lddxa % 4 == 0 && ldx % 4 == 0 && (!gx_out || ldgx % 4 == 0) && (((uintptr_t)dxa | (uintptr_t)x |
             (uintptr_t)gx_out) & 15) == 0, "ag_lstm_front_bwd_step: rows must be 16-byte aligned");
  BwdStepP q;
  q.valid = nullptr; q.B = B; q.H = H;
  q.rb = ag_precision() == AG_PREC_BF16;
  BwdStepDir& D = q.d[0];
  D.dg_next = dxa; D.lda = lddxa; D.Kp = Kp; D.whh = w_proj;
  D.ax = x; D.ldax = ldx; D.aout = gx_out; D.ldaout = ldgx;
  D.ga = gates; D.c_prev = c_prev; D.c_new = c_new;
  D.dy = nullptr; D.ldy = 0;
  D.dc_next = dc_next; D.dh_pass_in = dh_acc; D.lddhp = lddh;
  D.dgates = dgates; D.dc_prev = dc_prev; D.dh_pass_out = nullptr;
  D.t = 0;
  q.d[1] = q.d[0];
  if ((int64_t)(H / 16) * ag_cdiv(B, 16) > 256)
    hipLaunchKernelGGL((lstm_step_bwd_kernel<1, 2, 2, true>), dim3(ag_cdiv(H, 32), 1, ag_cdiv(B, 16)), dim3(1024), 0,
                       (hipStream_t)stream, q);
  else
    hipLaunchKernelGGL((lstm_step_bwd_kernel<1, 1, 2, true>), dim3(H / 16, 1, ag_cdiv(B, 16)), dim3(1024), 0,
                       (hipStream_t)stream, q);
  AG_CHECK_LAUNCH("ag_lstm_front_bwd_step");
  return AG_OK;
}

// Whole layer backward through time: per step ONE pointwise launch and ONE K-split skinny
// product  dh_{k-1} += dgates_k * W_hh  (atomics into the buffer that already holds the
// pass-through term of padded rows); both directions share each launch.
//   gates [ndir][T,B,4H] activated gates (from forward);  dgates [ndir][T,B,4H] (out)
//   dy [T,B,ndir*H];  dhbuf/dcbuf [ndir][2][B,H] scratch
extern "C" int ag_lstm_seq_bwd(const float* const* gates, const float* const* whh,
                               const float* const* c_all, const float* dy, float* const* dgates,
                               float* const* dhbuf, float* const* dcbuf, const int64_t* valid_i64, int T,
                               int B, int H, int ndir, int k_begin, int k_end, int phases, void* stream) {
  const AgWs ws = ag_ws_take();     // FIRST: an argument error below must not leave a stale binding behind
  AG_REQUIRE(gates && whh && c_all && dy && dgates && dhbuf && dcbuf, "ag_lstm_seq_bwd: null table");
  AG_REQUIRE(phases >= 1 && phases <= 3, "ag_lstm_seq_bwd: phases is a bit mask (1 cell, 2 product)");
  AG_REQUIRE(ndir == 1 || ndir == 2, "ag_lstm_seq_bwd: ndir must be 1 or 2");
  AG_REQUIRE(T > 0 && B > 0 && B <= 256 && (4 * H) % 8 == 0, "ag_lstm_seq_bwd: bad shape");
  AG_REQUIRE(0 <= k_begin && k_begin <= k_end && k_end <= T, "ag_lstm_seq_bwd: bad step range");
  hipStream_t st = (hipStream_t)stream;
  const int64_t BH = (int64_t)B * H, BG = (int64_t)B * 4 * H;
  if (phases == 3 && H % 16 == 0) {
    // fused path: launch k = product of step k+1 + cell backward of step k
    for (int k = k_end - 1; k >= k_begin; --k) {
      BwdStepP q;
      q.valid = valid_i64; q.B = B; q.H = H;
      for (int d = 0; d < ndir; ++d) {
        const int t = d == 0 ? k : T - 1 - k;
        const int tn = d == 0 ? k + 1 : T - 2 - k;     // time index of processing step k+1
        BwdStepDir& D = q.d[d];
        D.dg_next = (k == T - 1) ? nullptr : dgates[d] + (int64_t)tn * BG;
        D.lda = 4 * H; D.Kp = 4 * H; D.ax = nullptr; D.aout = nullptr; D.ldax = D.ldaout = 0; D.lddhp = H;
        D.whh = whh[d];
        D.ga = gates[d] + (int64_t)t * BG;
        D.c_prev = c_all[d] + (int64_t)k * BH;
        D.c_new = c_all[d] + (int64_t)(k + 1) * BH;
        D.dy = dy + (int64_t)t * B * ndir * H + (int64_t)d * H;
        D.ldy = ndir * H;
        D.dc_next = (k == T - 1) ? nullptr : dcbuf[d] + ((k + 1) & 1) * BH;
        D.dh_pass_in = (k == T - 1) ? nullptr : dhbuf[d] + ((k + 1) & 1) * BH;
        D.dgates = dgates[d] + (int64_t)t * BG;
        D.dc_prev = dcbuf[d] + (k & 1) * BH;
        D.dh_pass_out = dhbuf[d] + (k & 1) * BH;
        D.t = t;
      }
      if (ndir == 1) q.d[1] = q.d[0];
      // one wave of workgroups fills the 256 CUs: two row tiles per workgroup once a single one would not
      launch_bwd_step(q, ndir, st);
      AG_CHECK_LAUNCH("ag_lstm_seq_bwd(step)");
    }
    return AG_OK;
  }
  for (int k = k_end - 1; k >= k_begin; --k) {
    CellBwd2P c;
    c.valid = valid_i64; c.B = B; c.H = H;
    SkinnyP s;
    s.lda = 4 * H; s.ldb = H; s.ldc = H; s.M = B; s.N = H; s.K = 4 * H; s.tb = 0; s.act = AG_ACT_NONE;
    s.slope = 0.f; s.beta = 1.f;
    for (int d = 0; d < ndir; ++d) {
      const int t = d == 0 ? k : T - 1 - k;
      CellBwdDir& D = c.d[d];
      D.ga = gates[d] + (int64_t)t * BG;
      D.c_prev = c_all[d] + (int64_t)k * BH;
      D.c_new = c_all[d] + (int64_t)(k + 1) * BH;
      D.dh = (k == T - 1) ? nullptr : dhbuf[d] + (k & 1) * BH;
      D.dy = dy + (int64_t)t * B * ndir * H + (int64_t)d * H;
      D.ldy = ndir * H;
      D.dc_next = (k == T - 1) ? nullptr : dcbuf[d] + ((k + 1) & 1) * BH;
      D.dgates = dgates[d] + (int64_t)t * BG;
      D.dc_prev = dcbuf[d] + (k & 1) * BH;
      D.dh_pass = dhbuf[d] + ((k + 1) & 1) * BH;
      D.t = t;
      s.q[d].A = D.dgates; s.q[d].B = whh[d]; s.q[d].C = D.dh_pass; s.q[d].bias = nullptr;
    }
    if (ndir == 1) { c.d[1] = c.d[0]; s.q[1] = s.q[0]; }
    if (phases & 1) {
      hipLaunchKernelGGL(lstm_cell_bwd2_kernel, dim3(ag_cdiv(H, 256), B, ndir), dim3(256), 0, st, c);
      AG_CHECK_LAUNCH("ag_lstm_seq_bwd(cell)");
    }
    if (k > 0 && (phases & 2)) {
      int rc = launch_skinny(s, ndir, 1, st, ws);
      if (rc != AG_OK) return rc;
    }
  }
  return AG_OK;
}
