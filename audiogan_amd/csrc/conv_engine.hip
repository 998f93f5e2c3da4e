// conv_engine.hip -- fp32 MFMA implicit-GEMM engine for 1-D (transposed) convolutions.
//
// Replaces the cuDNN paths behind NN.Conv1d / NN.ConvTranspose1d forward and
// backward-data of the reference (audiogan.py:272,275,406,490 and loss.backward()
// at :785,:903).  See DESIGN.md "conv engine" for the derivation.
//
// One workgroup (4 waves) computes a [rows x cols] output tile of ONE clip b with
// v_mfma_f32_32x32x2_f32 (bitwise an fp32 fmaf chain):
//   MFMA row i  <-> output channel o          (mode 0)   or (o, phase r) (mode 1)
//   MFMA col j  <-> output time t             (mode 0)   or phase-time q (mode 1)
//   MFMA k pair <-> two input channels (c, c+1) at the same tap
// so the B operand is a unit-stride read of the LDS input tile (the strided input of
// mode 0 is de-interleaved into `stride` polyphase rows while staging) and the A
// operand a unit-stride read of the prepared weight layout.  C/D lanes run along
// time, so global stores are coalesced along the waveform axis.
//
// mode 1 (transposed conv) is the polyphase form: with u + p = s*q + r,
//   y[o, u] = sum_{c, m} W[c, o, r + s*m] * x[c, q - m]
// i.e. a stride-1 gather over ceil(K/s) taps whose "rows" are (o, r) pairs.
#include "common.h"

#define MAX_TAPS 32

struct ConvP {
  ag_conv_args a;
  int taps;       // taps per channel in the GEMM (K for mode 0, ceil(K/s) for mode 1)
  int sp;         // polyphase rows of the LDS input tile (stride for mode 0, 1 for mode 1)
  int sp_shift;   // log2(sp) or -1
  int ncols;      // staged columns per polyphase row
  int rowlen;     // LDS row pitch
  int chs;        // LDS channel pitch = sp * rowlen
  int CC;         // channels per chunk (even)
  int Cpad;       // channels in the prepared weight (even)
  int Mrows;      // GEMM rows: O (mode 0) or O*s (mode 1)
  int Mpad;       // row pitch of the prepared weight (multiple of 32)
  int n_lo;       // first column index (0 for mode 0, pad/s for mode 1)
  int n_cnt;      // number of columns
  int tapoff[MAX_TAPS];
};

template <int TILES_O, int TILES_T, int WAVES_O, int WAVES_T>
__global__ __launch_bounds__(256) void conv_engine_kernel(const ConvP p) {
  static_assert(WAVES_O * WAVES_T == 4, "4 waves per workgroup");
  constexpr int OT = 32 * TILES_O * WAVES_O;
  constexpr int TT = 32 * TILES_T * WAVES_T;
  extern __shared__ float smem[];
  float* xs = smem;                                // [CC][sp][rowlen]
  float* ws = smem + (size_t)p.CC * p.chs;         // [CC][taps][OT]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int wo = wid / WAVES_T, wt = wid % WAVES_T;
  const int wrow0 = wo * (32 * TILES_O), wcol0 = wt * (32 * TILES_T);

  const int b = blockIdx.z;
  const int row0 = blockIdx.y * OT;
  const int n0 = p.n_lo + blockIdx.x * TT;
  const ag_conv_args& a = p.a;
  const int base = (a.mode == 0) ? (a.stride * n0 - a.pad) : (n0 - (p.taps - 1));
  const float* xb = a.x + (int64_t)b * a.x_bs;

  f32x16 acc[TILES_O][TILES_T];
#pragma unroll
  for (int i = 0; i < TILES_O; ++i)
#pragma unroll
    for (int j = 0; j < TILES_T; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int span = p.sp * p.ncols;
  for (int c0 = 0; c0 < p.Cpad; c0 += p.CC) {
    __syncthreads();
    // ---- stage the input tile: wave w takes channels w, w+4, ...; lanes run along time
    for (int cc = wid; cc < p.CC; cc += 4) {
      const int c = c0 + cc;
      const bool cok = c < a.C;
      const float* xc = xb + (int64_t)c * a.x_cs;
      float* xr = xs + cc * p.chs;
      for (int rem = lane; rem < span; rem += 64) {
        const int g = base + rem;
        float v = 0.f;
        if (cok && g >= 0 && g < a.Lin) v = xc[g];
        int r, qq;
        if (p.sp_shift >= 0) {
          r = rem & (p.sp - 1);
          qq = rem >> p.sp_shift;
        } else {
          qq = rem / p.sp;
          r = rem - qq * p.sp;
        }
        xr[r * p.rowlen + qq] = v;
      }
    }
    // ---- stage the weight chunk: ws[cc][tau][row] <- wp[c][tau][row0 + row]
    {
      const int n4 = p.CC * p.taps * (OT / 4);
      for (int idx = tid; idx < n4; idx += 256) {
        const int r4 = idx % (OT / 4);
        const int ct = idx / (OT / 4);  // cc * taps + tau
        const int cc = ct / p.taps;
        const int c = c0 + cc;
        const int row = row0 + r4 * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (c < p.Cpad && row < p.Mpad)
          v = *reinterpret_cast<const f32x4*>(a.wp + ((int64_t)c0 * p.taps + ct) * p.Mpad + row);
        *reinterpret_cast<f32x4*>(ws + (size_t)ct * OT + r4 * 4) = v;
      }
    }
    __syncthreads();
    // ---- MFMA over (channel pair, tap)
    const int npair = p.CC >> 1;
    for (int cp = 0; cp < npair; ++cp) {
      const float* wrow = ws + (size_t)((2 * cp + h) * p.taps) * OT + wrow0 + l31;
      const float* xrow = xs + (2 * cp + h) * p.chs + wcol0 + l31;
      for (int tau = 0; tau < p.taps; ++tau) {
        float av[TILES_O], bv[TILES_T];
#pragma unroll
        for (int i = 0; i < TILES_O; ++i) av[i] = wrow[tau * OT + 32 * i];
        const int off = p.tapoff[tau];
#pragma unroll
        for (int j = 0; j < TILES_T; ++j) bv[j] = xrow[off + 32 * j];
#pragma unroll
        for (int i = 0; i < TILES_O; ++i)
#pragma unroll
          for (int j = 0; j < TILES_T; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: bias + residual + activation + length mask (+ accumulate)
  const int64_t lenb = a.lens_i64 ? a.lens_i64[b] : (int64_t)1 << 60;
  float* yb = a.y + (int64_t)b * a.y_bs;
  const float* rb = a.res ? a.res + (int64_t)b * a.res_bs : nullptr;
#pragma unroll
  for (int i = 0; i < TILES_O; ++i) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = row0 + wrow0 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
      if (row >= p.Mrows) continue;
      int o, r;
      if (a.mode == 0) {
        o = row;
        r = 0;
      } else {
        o = row / a.stride;
        r = row - o * a.stride;
      }
      const float bo = a.bias ? a.bias[o] : 0.f;
#pragma unroll
      for (int j = 0; j < TILES_T; ++j) {
        const int n = n0 + wcol0 + 32 * j + l31;
        const int t = (a.mode == 0) ? n : (a.stride * n + r - a.pad);
        if (t < 0 || t >= a.Lout) continue;
        float v = acc[i][j][e] + bo;
        if (rb) v += rb[(int64_t)o * a.res_cs + t];
        v = ag_apply_act(v, a.act, a.slope);
        if (t >= lenb) v = 0.f;
        float* dst = yb + (int64_t)o * a.y_cs + t;
        if (a.accumulate) v += *dst;
        *dst = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static int ilog2_exact(int v) {
  for (int s = 0; s < 31; ++s)
    if ((1 << s) == v) return s;
  return -1;
}

extern "C" int64_t ag_wpa_numel(int d0, int d1, int K) {
  return (int64_t)ag_roundup(d1, 2) * K * ag_roundup(d0, 32);
}
extern "C" int64_t ag_wpb_numel(int d0, int d1, int K, int stride) {
  return (int64_t)ag_roundup(d0, 2) * ag_cdiv(K, stride) * ag_roundup(d1 * stride, 32);
}

template <int TO, int TTL, int WO, int WT>
static int launch_cfg(ConvP& p, hipStream_t st) {
  constexpr int OT = 32 * TO * WO, TT = 32 * TTL * WT;
  const ag_conv_args& a = p.a;
  // input tile geometry
  const int dmax = (a.mode == 0) ? (a.K - 1) / a.stride : (p.taps - 1);
  p.ncols = TT + dmax;
  int rl = p.ncols;
  if (p.sp_shift >= 0 && p.sp > 1 && p.sp <= 32) {
    const int want = 32 / p.sp;  // rowlen == want (mod 32): polyphase rows land on disjoint banks
    rl = p.ncols + ((want - p.ncols) % 32 + 32) % 32;
  }
  p.rowlen = rl;
  p.chs = p.sp * p.rowlen;
  for (int t = 0; t < p.taps; ++t)
    p.tapoff[t] = (a.mode == 0) ? ((t % a.stride) * p.rowlen + t / a.stride) : (p.taps - 1 - t);
  // channels per chunk: as many (even, <= 16) as fit in 48 KiB of LDS
  const size_t per_c = (size_t)(p.chs + p.taps * OT) * sizeof(float);
  int cc = (int)((48 * 1024) / per_c) & ~1;
  if (cc < 2) cc = 2;
  if (cc > 16) cc = 16;
  if (cc > p.Cpad) cc = p.Cpad;
  p.CC = cc;
  const size_t lds = (size_t)cc * per_c;
  if (lds > 160 * 1024) {
    ag_set_error("conv engine: tile needs %zu B of LDS", lds);
    return AG_ERR_UNSUPPORTED;
  }
  dim3 grid(ag_cdiv(p.n_cnt, TT), ag_cdiv(p.Mrows, OT), a.B);
  auto kern = conv_engine_kernel<TO, TTL, WO, WT>;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, p);
  AG_CHECK_LAUNCH("ag_conv1d_engine");
  return AG_OK;
}

extern "C" int ag_conv1d_engine(const ag_conv_args* args, void* stream) {
  AG_REQUIRE(args != nullptr, "ag_conv1d_engine: null args");
  ConvP p;
  p.a = *args;
  const ag_conv_args& a = p.a;
  AG_REQUIRE(a.x && a.wp && a.y, "ag_conv1d_engine: null tensor");
  AG_REQUIRE(a.B > 0 && a.C > 0 && a.O > 0 && a.Lin > 0 && a.Lout > 0, "ag_conv1d_engine: bad shape");
  AG_REQUIRE(a.K > 0 && a.stride > 0 && a.pad >= 0, "ag_conv1d_engine: bad conv params");
  AG_REQUIRE(a.B <= 65535, "ag_conv1d_engine: batch > 65535");
  AG_REQUIRE(a.mode == 0 || a.mode == 1, "ag_conv1d_engine: bad mode");
  if (a.mode == 0) {
    p.taps = a.K;
    p.sp = a.stride;
    p.Mrows = a.O;
    p.n_lo = 0;
    p.n_cnt = a.Lout;
    // every output must read inside the staged window; callers pass the op's true Lout
    AG_REQUIRE((int64_t)(a.Lout - 1) * a.stride - a.pad < a.Lin, "ag_conv1d_engine: Lout too large");
  } else {
    p.taps = ag_cdiv(a.K, a.stride);
    p.sp = 1;
    p.Mrows = a.O * a.stride;
    p.n_lo = a.pad / a.stride;
    const int n_hi = (a.Lout - 1 + a.pad) / a.stride;
    p.n_cnt = n_hi - p.n_lo + 1;
  }
  AG_REQUIRE(p.taps <= MAX_TAPS, "ag_conv1d_engine: more than %d taps", MAX_TAPS);
  p.sp_shift = ilog2_exact(p.sp);
  p.Cpad = ag_roundup(a.C, 2);
  p.Mpad = ag_roundup(p.Mrows, 32);
  hipStream_t st = (hipStream_t)stream;
  // tile choice: wide in rows for fat layers, wide in time for thin ones
  if (p.Mrows <= 32) return launch_cfg<1, 2, 1, 4>(p, st);   // 32 x 256
  if (p.Mrows <= 64) return launch_cfg<2, 1, 1, 4>(p, st);   // 64 x 128
  if (p.n_cnt <= 64) return launch_cfg<2, 1, 2, 2>(p, st);   // 128 x 64
  return launch_cfg<2, 2, 2, 2>(p, st);                       // 128 x 128
}

// ------------------------------------------------------------------------------------------
// weight layouts for plain (not weight-normed) tensors
// ------------------------------------------------------------------------------------------
__global__ void prep_conv_weight_kernel(const float* __restrict__ w, float* __restrict__ wpa,
                                        float* __restrict__ wpb, int d0, int d1, int K, int s) {
  const int d0p32 = (d0 + 31) / 32 * 32;
  const int d1p2 = (d1 + 1) / 2 * 2;
  const int64_t na = (int64_t)d1p2 * K * d0p32;
  const int mt = (K + s - 1) / s;
  const int mp = (d1 * s + 31) / 32 * 32;
  const int d0p2 = (d0 + 1) / 2 * 2;
  const int64_t nb = (int64_t)d0p2 * mt * mp;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (wpa && i < na) {
    const int a0 = (int)(i % d0p32);
    const int64_t ck = i / d0p32;
    const int k = (int)(ck % K);
    const int c = (int)(ck / K);
    wpa[i] = (a0 < d0 && c < d1) ? w[((int64_t)a0 * d1 + c) * K + k] : 0.f;
  }
  if (wpb && i < nb) {
    const int row = (int)(i % mp);
    const int64_t am = i / mp;
    const int m = (int)(am % mt);
    const int a0 = (int)(am / mt);
    const int o = row / s, r = row - o * s;
    const int k = r + s * m;
    wpb[i] = (a0 < d0 && o < d1 && k < K) ? w[((int64_t)a0 * d1 + o) * K + k] : 0.f;
  }
}

extern "C" int ag_prep_conv_weight(const float* w, float* wpa, float* wpb, int d0, int d1, int K,
                                   int stride, void* stream) {
  AG_REQUIRE(w && (wpa || wpb), "ag_prep_conv_weight: null tensor");
  AG_REQUIRE(d0 > 0 && d1 > 0 && K > 0 && stride > 0, "ag_prep_conv_weight: bad shape");
  int64_t n = 0;
  if (wpa) n = ag_wpa_numel(d0, d1, K);
  if (wpb && ag_wpb_numel(d0, d1, K, stride) > n) n = ag_wpb_numel(d0, d1, K, stride);
  hipLaunchKernelGGL(prep_conv_weight_kernel, dim3((unsigned)ag_cdiv64(n, 256)), dim3(256), 0,
                     (hipStream_t)stream, w, wpa, wpb, d0, d1, K, stride);
  AG_CHECK_LAUNCH("ag_prep_conv_weight");
  return AG_OK;
}
