// conv_engine_impl.h -- device code and per-tile-configuration launchers of the conv engine.  Included by conv_engine.hip
// (host dispatch) and by conv_engine_t*.hip, ONE tile configuration each, so that the 40-odd kernel instantiations compile in
// parallel translation units instead of one 7-minute one.
#pragma once
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
#include <type_traits>

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
  int xvec;       // input rows may be read with aligned 16-byte loads
  int aligned;    // mode 1: aligned scatter layout (common.h) - phase r's outputs are shifted by -s * shift_r
  int efast;      // the buffer-addressed epilogue applies (31-bit byte offsets)
  int s_shift;    // log2(stride) or -1
  int ncols_tile; // columns of the workgroup's tile (TT)
  int pipe;       // fp32 kernel: register-pipelined staging (see the kernel)
  int solo;       // fp32 kernel launched with 4 waves and ONE LDS buffer: stage, multiply, stage, ... (see the kernel)
  int dma;        // solo form, mode 1: interior tiles stage by LDS-DMA (global_load_lds) instead of through registers
  int nbuf;       // bf16 kernel: LDS buffers (1 when the whole reduction is one chunk)
  int rb;         // AG_PREC_BF16: both operands rounded to bf16 while staging (fp32 MFMA on rounded values)
  int tapoff[MAX_TAPS];
};

// under-aligned vector types: mode-1 outputs start at u = s*n + r - pad, which is only
// guaranteed to be 4-byte aligned (gfx950 global accesses need dword alignment only)
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
typedef unsigned cu32x4 __attribute__((ext_vector_type(4)));

// Epilogue shared by the fp32 and the bf16 kernels (same 32x32 accumulator layout): bias + residual + activation +
// length mask (+ accumulate).
//
// conv_epilogue_fast is the path almost every tile takes.  It is written for instruction count (an earlier general
// version spent ~850 instructions per 16-byte store on 64-bit addresses, divisions by the stride and bounds branches:
// 12-16 us per 128x128 tile, more than the tile's MFMAs): buffer addressing with 32-bit offsets - anything out of
// range gets an offset past num_records and is dropped / reads 0, no branches -, stride by shift, rows resolved
// outside the column loop.  It covers mode 0, and mode 1 with s % 4 == 0 (one 16-byte store per lane and group) or the
// aligned s == 2 layout (8-byte stores) on tiles whose columns all map inside the signal; it returns false when the
// tile needs conv_epilogue_any - the plain element-by-element form (edge tiles of transposed convs, odd strides,
// tensors beyond 31-bit offsets).
template <int TILES_O, int TILES_T>
__device__ __forceinline__ bool conv_epilogue_fast(const ConvP& p, f32x16 (&acc)[TILES_O][TILES_T], const float* bias_s,
                                                   int b, int row0, int n0, int wrow0, int wcol0, int l31, int h) {
  const ag_conv_args& a = p.a;
  if (!p.efast) return false;
  const unsigned OOB = 0x80000000u;
  const int ycs = (int)a.y_cs, rcs = (int)a.res_cs;
  int lenb = 0x7fffffff;
  if (a.lens_i64) {
    const int64_t l = a.lens_i64[b];
    lenb = l > 0x7fffffff ? 0x7fffffff : (int)l;
  }
  __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(a.y + (int64_t)b * a.y_bs, 0, (int)((int64_t)a.O * a.y_cs * 4), 0x00020000);
  __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.res ? a.res + (int64_t)b * a.res_bs : a.y), 0, a.res ? (int)((int64_t)a.O * a.res_cs * 4) : 0, 0x00020000);
  const bool has_res = a.res != nullptr;
  // AG_ACT_LEAKY_GATE: `res` is not added - it is the SAVED OUTPUT of a LeakyReLU whose derivative scales this result
  // (the activation backward of the producing layer folded into this backward-data pass)
  const bool gate = a.act == AG_ACT_LEAKY_GATE;
  if (a.mode == 0) {
#pragma unroll
    for (int i = 0; i < TILES_O; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ro = wrow0 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
        const int o = row0 + ro;
        const bool valid = o < p.Mrows;
        const float bo = bias_s[ro];
        const int rowy = o * ycs, rowr = o * rcs;
#pragma unroll
        for (int j = 0; j < TILES_T; ++j) {
          const int t = n0 + wcol0 + 32 * j + l31;
          const bool ok = valid && t < a.Lout;
          const unsigned yo = ok ? (unsigned)((rowy + t) * 4) : OOB;
          float v = acc[i][j][e] + bo;
          if (has_res) {
            const float rv = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rr, ok ? (unsigned)((rowr + t) * 4) : OOB, 0, 0));
            v = gate ? (rv > 0.f ? v : v * a.slope) : v + rv;
          }
          v = ag_apply_act(v, a.act, a.slope);
          if (t >= lenb) v = 0.f;
          if (a.accumulate) v += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(yr, yo, 0, 0));
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), yr, yo, 0, 0);
        }
      }
    }
    return true;
  }
  // mode 1: row = o*s + r, output position u = s*n + r - pad; a lane holds 4 consecutive rows in registers 4g..4g+3.
  // Aligned layout (common.h): with rho = pad % s, phases r >= rho sit one column earlier, so the s rows of a channel
  // are the contiguous window [s*(n-a-1), +s) ROTATED by rho (pad = s*a + rho).
  const int s = a.stride;
  const int rho = a.pad % s;
  const bool rot4 = p.aligned && s == 4;                          // 16-byte window rotated by rho
  const bool al4 = p.aligned && s % 4 == 0 && rho % 4 == 0;       // 4 rows on the same side of rho: plain
  const bool sw2 = p.aligned && s == 2;                           // the critic's convs: swapped pairs, 8-byte stores
  if (!sw2 && (s % 4 != 0 || (p.aligned && !rot4 && !al4))) return false;
  {   // every column of the workgroup's tile inside the signal?  (uniform)
    const int nlo = n0, nhi = n0 + p.ncols_tile - 1;
    int lo, hi;
    if (sw2) { lo = 2 * nlo - a.pad - 1; hi = 2 * nhi - a.pad + 1; }
    else if (rot4) { lo = 4 * (nlo - 1) - (a.pad - rho); hi = 4 * nhi - (a.pad - rho); }
    else if (p.aligned) { lo = s * (nlo - 1) + rho - a.pad; hi = s * nhi + rho - a.pad; }
    else { lo = s * nlo - a.pad; hi = s * (nhi + 1) - a.pad; }
    if (lo < 0 || hi > a.Lout) return false;
  }
  const bool need_pre = has_res || a.accumulate;
  if (sw2) {
#pragma unroll
    for (int i = 0; i < TILES_O; ++i) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int ro = wrow0 + 32 * i + 8 * g + 4 * h;      // rows ro..ro+3 = (o, phase 0), (o, 1), (o+1, 0), (o+1, 1)
        const int o = (row0 + ro) >> 1;
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
          const bool valid = row0 + ro + 2 * c2 < p.Mrows;
          const float bo = bias_s[ro + 2 * c2];
          const int rowy = (o + c2) * ycs, rowr = (o + c2) * rcs;
#pragma unroll
          for (int j = 0; j < TILES_T; ++j) {
            const int u0 = 2 * (n0 + wcol0 + 32 * j + l31) - a.pad - 1;
            const unsigned yo = valid ? (unsigned)((rowy + u0) * 4) : OOB;
            float v0 = acc[i][j][4 * g + 2 * c2 + 1] + bo, v1 = acc[i][j][4 * g + 2 * c2] + bo;     // swapped pair
            if (has_res) {
              const unsigned ro_ = valid ? (unsigned)((rowr + u0) * 4) : OOB;
              const float r0 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rr, ro_, 0, 0));
              const float r1 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rr, ro_ + 4u, 0, 0));
              v0 = gate ? (r0 > 0.f ? v0 : v0 * a.slope) : v0 + r0;
              v1 = gate ? (r1 > 0.f ? v1 : v1 * a.slope) : v1 + r1;
            }
            v0 = ag_apply_act(v0, a.act, a.slope);
            v1 = ag_apply_act(v1, a.act, a.slope);
            if (u0 >= lenb) v0 = 0.f;
            if (u0 + 1 >= lenb) v1 = 0.f;
            if (a.accumulate) {
              v0 += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(yr, yo, 0, 0));
              v1 += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(yr, yo + 4u, 0, 0));
            }
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v0), yr, yo, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v1), yr, yo + 4u, 0, 0);
          }
        }
      }
    }
    return true;
  }
  const int rot = rot4 ? rho : 0;
#pragma unroll
  for (int i = 0; i < TILES_O; ++i) {
    // the residual - or for `accumulate` the old output - of a whole row tile is requested up front (one round trip
    // instead of 16 dependent ones)
    unsigned yo[4][TILES_T];
    cu32x4 pre[4][TILES_T];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int rowb = row0 + wrow0 + 32 * i + 8 * g + 4 * h;
      const bool valid = rowb < p.Mrows;
      const int o = p.s_shift >= 0 ? (rowb >> p.s_shift) : rowb / s;
      const int r = rowb - o * s;
      const int rowy = o * ycs, rowr = o * rcs;
#pragma unroll
      for (int j = 0; j < TILES_T; ++j) {
        const int n = n0 + wcol0 + 32 * j + l31;
        const int u0 = rot4 ? 4 * (n - 1) - (a.pad - rho) : s * (n - ((p.aligned && r >= rho) ? 1 : 0)) + r - a.pad;
        yo[g][j] = valid ? (unsigned)((rowy + u0) * 4) : OOB;
        if (need_pre)
          pre[g][j] = has_res ? __builtin_amdgcn_raw_buffer_load_b128(rr, valid ? (unsigned)((rowr + u0) * 4) : OOB, 0, 0)
                              : __builtin_amdgcn_raw_buffer_load_b128(yr, yo[g][j], 0, 0);
      }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int ro = wrow0 + 32 * i + 8 * g + 4 * h;
      const int rowb = row0 + ro;
      const int o = p.s_shift >= 0 ? (rowb >> p.s_shift) : rowb / s;
      const int r = rowb - o * s;
      const float bo = bias_s[ro];
#pragma unroll
      for (int j = 0; j < TILES_T; ++j) {
        const int n = n0 + wcol0 + 32 * j + l31;
        const int u0 = rot4 ? 4 * (n - 1) - (a.pad - rho) : s * (n - ((p.aligned && r >= rho) ? 1 : 0)) + r - a.pad;
        const float v4[4] = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
        float v[4];
        if (rot == 0) { v[0] = v4[0]; v[1] = v4[1]; v[2] = v4[2]; v[3] = v4[3]; }
        else if (rot == 1) { v[0] = v4[1]; v[1] = v4[2]; v[2] = v4[3]; v[3] = v4[0]; }
        else if (rot == 2) { v[0] = v4[2]; v[1] = v4[3]; v[2] = v4[0]; v[3] = v4[1]; }
        else { v[0] = v4[3]; v[1] = v4[0]; v[2] = v4[1]; v[3] = v4[2]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          v[q] += bo;
          if (has_res) {
            const float rv = __uint_as_float(pre[g][j][q]);
            v[q] = gate ? (rv > 0.f ? v[q] : v[q] * a.slope) : v[q] + rv;
          }
          v[q] = ag_apply_act(v[q], a.act, a.slope);
          if (u0 + q >= lenb) v[q] = 0.f;
          if (!has_res && a.accumulate) v[q] += __uint_as_float(pre[g][j][q]);
        }
        cu32x4 out = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
        if (has_res && a.accumulate) {
          const cu32x4 ov = __builtin_amdgcn_raw_buffer_load_b128(yr, yo[g][j], 0, 0);
#pragma unroll
          for (int q = 0; q < 4; ++q) out[q] = __float_as_uint(__uint_as_float(out[q]) + __uint_as_float(ov[q]));
        }
        __builtin_amdgcn_raw_buffer_store_b128(out, yr, yo[g][j], 0, 0);
      }
    }
  }
  return true;
}

// Element by element, every case (see above).  Deliberately plain: it runs on the few tiles the fast form declines.
template <int TILES_O, int TILES_T>
__device__ __forceinline__ void conv_epilogue_any(const ConvP& p, f32x16 (&acc)[TILES_O][TILES_T], const float* bias_s,
                                                  int b, int row0, int n0, int wrow0, int wcol0, int l31, int h) {
  const ag_conv_args& a = p.a;
  const int64_t lenb = a.lens_i64 ? a.lens_i64[b] : (int64_t)1 << 60;
  float* yb = a.y + (int64_t)b * a.y_bs;
  const float* rb = a.res ? a.res + (int64_t)b * a.res_bs : nullptr;
  const int s = a.stride;
  const int rho = a.pad % s;
#pragma unroll
  for (int i = 0; i < TILES_O; ++i) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int ro = wrow0 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
      const int row = row0 + ro;
      if (row >= p.Mrows) continue;
      const float bo = bias_s[ro];
      const int o = a.mode == 0 ? row : row / s;
      const int r = row - o * s;
#pragma unroll
      for (int j = 0; j < TILES_T; ++j) {
        const int n = n0 + wcol0 + 32 * j + l31;
        const int u = a.mode == 0 ? n : s * (n - ((p.aligned && r >= rho) ? 1 : 0)) + r - a.pad;
        if (u < 0 || u >= a.Lout) continue;
        float v = acc[i][j][e] + bo;
        if (rb) {
          const float rv = rb[(int64_t)o * a.res_cs + u];
          v = a.act == AG_ACT_LEAKY_GATE ? (rv > 0.f ? v : v * a.slope) : v + rv;
        }
        v = ag_apply_act(v, a.act, a.slope);
        if (u >= lenb) v = 0.f;
        float* dst = yb + (int64_t)o * a.y_cs + u;
        if (a.accumulate) v += *dst;
        *dst = v;
      }
    }
  }
}

template <int TILES_O, int TILES_T>
__device__ __forceinline__ void conv_epilogue(const ConvP& p, f32x16 (&acc)[TILES_O][TILES_T], const float* bias_s,
                                              int b, int row0, int n0, int wrow0, int wcol0, int l31, int h) {
  if (!conv_epilogue_fast<TILES_O, TILES_T>(p, acc, bias_s, b, row0, n0, wrow0, wcol0, l31, h))
    conv_epilogue_any<TILES_O, TILES_T>(p, acc, bias_s, b, row0, n0, wrow0, wcol0, l31, h);
}

// TAPS/S0 > 0: taps and polyphase factor known at compile time (tap loop fully unrolled, LDS
// offsets are immediates).  S0 = stride for mode 0, 0 for mode 1.  TAPS == 0: generic runtime loop.
template <int TILES_O, int TILES_T, int WAVES_O, int WAVES_T, int TAPS, int S0>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void conv_engine_kernel(const ConvP p) {
  // 8 waves: waves 0-3 run the MFMAs of chunk i out of LDS buffer i&1 while waves 4-7 stage
  // chunk i+1 (global -> LDS, polyphase de-interleave) into the other buffer; one barrier per
  // chunk.  Staging VALU/VMEM work co-issues with the MFMA pipe of the compute waves.
  static_assert(WAVES_O * WAVES_T == 4, "4 compute waves per workgroup");
  constexpr int OT = 32 * TILES_O * WAVES_O;
  constexpr int TT = 32 * TILES_T * WAVES_T;
  constexpr int SD = S0 > 0 ? S0 : 1;
  extern __shared__ float smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int cw = wid & 3;                          // index inside the role group
  const int wo = cw / WAVES_T, wt = cw % WAVES_T;
  const int wrow0 = wo * (32 * TILES_O), wcol0 = wt * (32 * TILES_T);

  const int b = blockIdx.z;
  const int row0 = blockIdx.y * OT;
  const int n0 = p.n_lo + blockIdx.x * TT;
  const ag_conv_args& a = p.a;
  const int taps = TAPS > 0 ? TAPS : p.taps;
  const int base = (a.mode == 0) ? (a.stride * n0 - a.pad) : (n0 - (taps - 1));
  const float* xb = a.x + (int64_t)b * a.x_bs;

  const int span = p.sp * p.ncols;
  const int rowlen = p.rowlen, chs = p.chs;
  const size_t bufsz = (size_t)p.CC * (chs + taps * OT);   // floats per LDS buffer
  const int nchunk = (p.Cpad + p.CC - 1) / p.CC;

  // stage chunk starting at channel c0 into buffer `buf`, using `nsw` waves (this wave = `sw`)
  auto stage = [&](int c0, int buf, int sw, int nsw) {
    float* xs = smem + buf * bufsz;                  // [CC][sp][rowlen]
    float* ws = xs + (size_t)p.CC * chs;             // [CC][taps][OT]
    // Staging is latency-bound: ALL global loads of a round (UX input samples + UW weight
    // float4 per lane) are issued before the first LDS write, so a chunk costs one or two
    // dependent memory round trips instead of one per loop iteration.
    // Input samples are fetched as 16-byte pieces aligned to a multiple of 4 samples (window start
    // rounded down), then scattered into the polyphase rows.
    constexpr int UX = 2, UW = 6;
    const int step = nsw * 64;
    const int base4 = base - (((base % 4) + 4) % 4);       // floor to a multiple of 4 (base may be < 0)
    const int nq = (base - base4 + span + 3) / 4;          // float4 pieces per channel
    const int xtot = p.CC * nq, wtot = p.CC * taps * (OT / 4);
    int xe = sw * 64 + lane, we = xe;
    while (xe < xtot || we < wtot) {
      f32x4 xv[UX];
      f32x4 wv[UW];
#pragma unroll
      for (int u = 0; u < UX; ++u) {
        const int e = xe + u * step;
        xv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (e < xtot) {
          const int cc = e / nq, i4 = e - cc * nq;
          const int g = base4 + 4 * i4, c = c0 + cc;
          if (c < a.C) {
            const float* src = xb + (int64_t)c * a.x_cs + g;
            if (p.xvec && g >= 0 && g + 3 < a.Lin) {
              xv[u] = *reinterpret_cast<const f32x4*>(src);
            } else {
#pragma unroll
              for (int q = 0; q < 4; ++q)
                if (g + q >= 0 && g + q < a.Lin) xv[u][q] = src[q];
            }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < UW; ++u) {
        const int idx = we + u * step;
        wv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (idx < wtot) {
          const int r4 = idx % (OT / 4), ct = idx / (OT / 4);  // ct = cc * taps + tau
          const int c = c0 + ct / taps, row = row0 + r4 * 4;
          if (c < p.Cpad && row < p.Mpad)
            wv[u] = *reinterpret_cast<const f32x4*>(a.wp + ((int64_t)c0 * taps + ct) * p.Mpad + row);
        }
      }
      if (p.rb) {      // AG_PREC_BF16 (uniform branch): both operands rounded on the way into LDS
#pragma unroll
        for (int u = 0; u < UX; ++u) xv[u] = ag_rbf4_if(xv[u], 1);
#pragma unroll
        for (int u = 0; u < UW; ++u) wv[u] = ag_rbf4_if(wv[u], 1);
      }
#pragma unroll
      for (int u = 0; u < UX; ++u) {
        const int e = xe + u * step;
        if (e < xtot) {
          const int cc = e / nq, i4 = e - cc * nq;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int rem = base4 + 4 * i4 + q - base;
            if (rem < 0 || rem >= span) continue;
            int r, qq;
            if (S0 > 0) {
              r = rem % SD;
              qq = rem / SD;
            } else if (TAPS > 0) {
              r = 0;
              qq = rem;
            } else if (p.sp_shift >= 0) {
              r = rem & (p.sp - 1);
              qq = rem >> p.sp_shift;
            } else {
              qq = rem / p.sp;
              r = rem - qq * p.sp;
            }
            xs[cc * chs + r * rowlen + qq] = xv[u][q];
          }
        }
      }
#pragma unroll
      for (int u = 0; u < UW; ++u) {
        const int idx = we + u * step;
        if (idx < wtot) {
          const int r4 = idx % (OT / 4), ct = idx / (OT / 4);
          *reinterpret_cast<f32x4*>(ws + (size_t)ct * OT + r4 * 4) = wv[u];
        }
      }
      xe += UX * step;
      we += UW * step;
    }
  };

  // ---- LDS-DMA staging (p.dma: solo form of a transposed conv, i.e. no polyphase de-interleave; host guarantees 16-byte
  // aligned rows of x and of the prepared weights, chs % 16 == 0, CC % 16 == 0 and Cpad % CC == 0).  The input window of a chunk
  // is then a plain copy of CC row segments of chs floats starting at base4 = floor4(base) - the reader adds base - base4 -
  // and the weight panel a plain copy of CC * taps segments of OT rows: 1-KiB DMA instructions, lane-linear, no VGPR, no
  // VALU (the register path spends ~40 VALU instructions per 16-byte piece on its index arithmetic, which is what a thin
  // layer's short reduction cannot amortise).  A DMA cannot write zeros: tiles whose window leaves the signal (the first
  // and the last one or two of a row) take the register path; channels beyond C read channel C - 1 (their weights are 0).
  const int base4t = base - (((base % 4) + 4) % 4);
  const bool dma_tile = p.dma && base4t >= 0 && base4t + chs <= a.Lin && row0 + OT <= p.Mpad;      // (uniform)
  const int xshift = dma_tile ? base - base4t : 0;
  auto stage_dma = [&](int c0) {
    float* xs = smem;
    float* ws = xs + (size_t)p.CC * chs;
    const int npc = chs >> 2;                          // 16-byte pieces per channel
    const int xins = (p.CC * npc) >> 6, wins = (p.CC * taps * (OT / 4)) >> 6;      // 64-piece instructions
    for (int g = wid; g < xins; g += 4) {
      const int e = g * 64 + lane, cc = e / npc, i4 = e - cc * npc;
      const float* src = xb + (int64_t)min(c0 + cc, a.C - 1) * a.x_cs + base4t + 4 * i4;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(xs + g * 256), 16, 0, 0);
    }
    for (int g = wid; g < wins; g += 4) {
      const int f = g * 64 + lane, ct = f / (OT / 4), r4 = f - ct * (OT / 4);
      const float* src = a.wp + ((int64_t)c0 * taps + ct) * p.Mpad + row0 + 4 * r4;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(ws + g * 256), 16, 0, 0);
    }
  };

  // the bias of this workgroup's rows goes to LDS once (behind the two staging buffers): read per output group in
  // the epilogue it would be 16-32 dependent global loads per lane
  float* bias_s = smem + (p.solo ? 1 : 2) * bufsz;
  if (tid < OT) {
    const int row = row0 + tid;
    bias_s[tid] = (a.bias && row < p.Mrows) ? a.bias[a.mode == 0 ? row : row / a.stride] : 0.f;
  }
  // p.pipe (host: 16-byte input pieces lie entirely inside or outside the signal, a chunk is one round of loads per
  // staging lane, 31-bit byte offsets): the staging waves run a register pipeline - the loads of chunk i+2 are in
  // flight while chunk i+1 is written to LDS and chunk i is multiplied.  Otherwise: generic staging, chunk 0 by all
  // 8 waves.
  // p.solo (thin layers: a reduction of a few short chunks, where the serial phases of a workgroup - stage, multiply, store -
  // leave the matrix pipes idle more than half the time with only two 8-wave workgroups per CU to overlap): the launch has
  // 4 waves and ONE buffer, every wave stages and multiplies, and FOUR workgroups share a CU.
  if (!p.pipe && !p.solo) {
    stage(0, 0, wid, 8);
    __syncthreads();
  }
  // Both waves of a SIMD share its VALU issue, arbitrated by priority, then age: the staging waves (4-7, the
  // younger half) would only get the slots the MFMA waves leave over and a chunk's staging would take
  // longer than its MFMAs.  Their instruction count is small, so give them priority for the whole loop.
  if (__builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) {
    __builtin_amdgcn_s_setprio(2);
    if (!p.pipe) {
      for (int ci = 0; ci < nchunk; ++ci) {
        if (ci + 1 < nchunk) stage((ci + 1) * p.CC, (ci + 1) & 1, cw, 4);
        __syncthreads();
      }
      return;
    }
    // 16-byte pieces per lane and chunk: input / weights.  Narrow row tiles stage mostly input (wide time tile,
    // stride-s window), wide ones mostly weights (launch_cfg sizes the chunk to these).
    constexpr int FX = OT <= 32 ? 7 : 4, FW = OT <= 32 ? 3 : 8;
    // Buffer loads: a 32-bit offset per piece, and everything that must read as zero (halo outside the signal,
    // channels >= C, rows >= Mpad) carries an offset past num_records.  Every lane's offsets, LDS targets and
    // predicates are the same for every chunk up to a uniform per-chunk increment: computed once.
    const unsigned OOB = 0x80000000u;
    __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, (int)((int64_t)a.C * a.x_cs * 4), 0x00020000);
    __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wp), 0, (int)((int64_t)p.Cpad * taps * p.Mpad * 4), 0x00020000);
    const int sl = cw * 64 + lane;
    const int base4 = base - (((base % 4) + 4) % 4);
    const int nq = (base - base4 + span + 3) / 4;
    const int xtot = p.CC * nq, wtot = p.CC * taps * (OT / 4);
    unsigned xoff[FX], woff[FW];
    unsigned xlp[FX][2];          // LDS targets of a piece's 4 samples, two 16-bit float indices per register (0xffff: no write)
#pragma unroll
    for (int u = 0; u < FX; ++u) {
      const int e = sl + u * 256;
      const bool valid = e < xtot;
      const int cc = valid ? e / nq : 0, i4 = valid ? e - cc * nq : 0;
      const int g = base4 + 4 * i4;
      xoff[u] = (valid && g >= 0 && g + 3 < a.Lin) ? (unsigned)((cc * (int)a.x_cs + g) * 4) : OOB;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int rem = g + q - base;
        int off = -1;
        if (valid && rem >= 0 && rem < span) {
          int r, qq;
          if (S0 > 0) {
            r = rem % SD;
            qq = rem / SD;
          } else if (TAPS > 0) {
            r = 0;
            qq = rem;
          } else if (p.sp_shift >= 0) {
            r = rem & (p.sp - 1);
            qq = rem >> p.sp_shift;
          } else {
            qq = rem / p.sp;
            r = rem - qq * p.sp;
          }
          off = cc * chs + r * rowlen + qq;
        }
        if (q & 1) xlp[u][q >> 1] |= (unsigned)(off & 0xffff) << 16;
        else xlp[u][q >> 1] = (unsigned)(off & 0xffff);
      }
    }
#pragma unroll
    for (int u = 0; u < FW; ++u) {
      const int idx = sl + u * 256;
      const int r4 = idx % (OT / 4), ct = idx / (OT / 4);
      const int row = row0 + r4 * 4;
      woff[u] = (idx < wtot && row < p.Mpad) ? (unsigned)((ct * p.Mpad + row) * 4) : OOB;
    }
    const unsigned xstep = (unsigned)p.CC * (unsigned)a.x_cs * 4u, wstep = (unsigned)(p.CC * taps * p.Mpad) * 4u;
    cu32x4 xv[FX], wv[FW];
    auto load = [&](int ci) {
#pragma unroll
      for (int u = 0; u < FX; ++u) xv[u] = __builtin_amdgcn_raw_buffer_load_b128(xr, xoff[u] + (unsigned)ci * xstep, 0, 0);
#pragma unroll
      for (int u = 0; u < FW; ++u) wv[u] = __builtin_amdgcn_raw_buffer_load_b128(wr, woff[u] + (unsigned)ci * wstep, 0, 0);
    };
    auto write_t = [&](int buf, auto rbtag) {
      constexpr bool RB = decltype(rbtag)::value;
      float* xs = smem + buf * bufsz;
      float* ws = xs + (size_t)p.CC * chs;
      f32x4 xf[FX], wf[FW];
#pragma unroll
      for (int u = 0; u < FX; ++u) xf[u] = __builtin_bit_cast(f32x4, xv[u]);
#pragma unroll
      for (int u = 0; u < FW; ++u) wf[u] = __builtin_bit_cast(f32x4, wv[u]);
      if (RB) {      // AG_PREC_BF16
#pragma unroll
        for (int u = 0; u < FX; ++u) xf[u] = ag_rbf4_if(xf[u], 1);
#pragma unroll
        for (int u = 0; u < FW; ++u) wf[u] = ag_rbf4_if(wf[u], 1);
      }
#pragma unroll
      for (int u = 0; u < FX; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const unsigned off = (xlp[u][q >> 1] >> (16 * (q & 1))) & 0xffffu;
          if (off != 0xffffu) xs[off] = xf[u][q];
        }
#pragma unroll
      for (int u = 0; u < FW; ++u) {
        const int idx = sl + u * 256;                    // LDS target ct * OT + 4 * r4 == 4 * idx
        if (idx < wtot) *reinterpret_cast<f32x4*>(ws + 4 * idx) = wf[u];
      }
    };
    // (two instantiations under one uniform branch: rounding in place under `if (p.rb)` made the allocator keep both
    // versions of every piece live - 48 more registers, one wave per SIMD less)
    auto write = [&](int buf) {
      if (p.rb) write_t(buf, std::true_type{});
      else write_t(buf, std::false_type{});
    };
    load(0);
    write(0);
    __builtin_amdgcn_sched_barrier(0);
    if (nchunk > 1) load(1);
    __syncthreads();
    for (int ci = 0; ci < nchunk; ++ci) {
      if (ci + 1 < nchunk) write((ci + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);      // (the loads below reuse the registers just written out: no hoisting)
      if (ci + 2 < nchunk) load(ci + 2);
      __syncthreads();
    }
    return;
  }
  if (p.pipe) __syncthreads();      // chunk 0 staged
  f32x16 acc[TILES_O][TILES_T];
#pragma unroll
  for (int i = 0; i < TILES_O; ++i)
#pragma unroll
    for (int j = 0; j < TILES_T; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  for (int ci = 0; ci < nchunk; ++ci) {
    if (p.solo) {
      if (dma_tile) {
        stage_dma(ci * p.CC);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        stage(ci * p.CC, 0, wid, 4);
      }
      __syncthreads();
    }
    const float* xs = smem + (p.solo ? 0 : (ci & 1)) * bufsz;
    const float* ws = xs + (size_t)p.CC * chs;
    // ---- MFMA over (channel pair, tap)
    const int npair = p.CC >> 1;
    for (int cp = 0; cp < npair; ++cp) {
      const float* wrow = ws + (size_t)((2 * cp + h) * taps) * OT + wrow0 + l31;
      const float* xrow = xs + (2 * cp + h) * chs + wcol0 + l31 + xshift;
      if (TAPS > 0) {
#pragma unroll
        for (int tau = 0; tau < TAPS; ++tau) {
          const int off = S0 > 0 ? ((tau % SD) * rowlen + tau / SD) : (TAPS - 1 - tau);
          float av[TILES_O], bv[TILES_T];
#pragma unroll
          for (int i = 0; i < TILES_O; ++i) av[i] = wrow[tau * OT + 32 * i];
#pragma unroll
          for (int j = 0; j < TILES_T; ++j) bv[j] = xrow[off + 32 * j];
#pragma unroll
          for (int i = 0; i < TILES_O; ++i)
#pragma unroll
            for (int j = 0; j < TILES_T; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
      } else {
        for (int tau = 0; tau < taps; ++tau) {
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
    __syncthreads();
  }

  conv_epilogue<TILES_O, TILES_T>(p, acc, bias_s, b, row0, n0, wrow0, wcol0, l31, h);
}

// ------------------------------------------------------------------------------------------
// AG_PREC_BF16: the same implicit GEMM on v_mfma_f32_32x32x16_bf16 (8x the fp32 MFMA rate).
//
// One MFMA k-step = 16 input channels at one tap: lane (l31, h) supplies the 8 channels 8h..8h+7, so both LDS images
// keep 8 channels of one (row | position) in one 16-byte slot:
//   XB[group][h][polyphase row][position][8 x bf16]       (input window, de-interleaved like the fp32 kernel's)
//   WB[group][tap][h][row][8 x bf16]                      (weights of the row tile)
// Weights come from the bf16 image behind the prepared fp32 layout (common.h ag_wq_*, written by the weight-norm /
// prep kernels in exactly this order): a straight 16-byte copy.  Activations are fp32 in HBM; a staging lane gathers
// 8 channels x 4 consecutive positions with 8 16-byte loads, rounds (RNE, v_cvt_pk_bf16_f32) and writes 4 slots.
// The MFMAs of a chunk are short (a few hundred cycles), a memory round trip is not: the staging waves keep the loads
// of chunk i+2 in flight (in registers) while chunk i+1 sits in the second LDS buffer and chunk i is being multiplied.
// Results equal the rounding emulation in the fp32 kernel up to fp32 summation order (products of bf16 values are exact
// in fp32; the MFMA accumulates in fp32).
typedef short cbf16x8 __attribute__((ext_vector_type(8)));
// Per staging lane and chunk: NW weight slots and CB_NX input tasks (8 channels x 4 positions).  NW = 16 is the deep-reduction
// variant (>= 8 channel groups, 128-row tiles): one workgroup per CU, two 16-channel groups per chunk - the chunk in
// flight, not a co-resident workgroup, covers the memory latency.
#define CB_NX 2

template <int TILES_O, int TILES_T, int WAVES_O, int WAVES_T, int NW>
__global__ __launch_bounds__(512) void conv_engine_bf16_kernel(const ConvP p) {
  static_assert(WAVES_O * WAVES_T == 4, "4 compute waves per workgroup");
  constexpr int OT = 32 * TILES_O * WAVES_O;
  constexpr int TT = 32 * TILES_T * WAVES_T;
  extern __shared__ float smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int cw = wid & 3;
  const int wo = cw / WAVES_T, wt = cw % WAVES_T;
  const int wrow0 = wo * (32 * TILES_O), wcol0 = wt * (32 * TILES_T);

  const int b = blockIdx.z;
  const int row0 = blockIdx.y * OT;
  const int n0 = p.n_lo + blockIdx.x * TT;
  const ag_conv_args& a = p.a;
  const int taps = p.taps;
  const int base = (a.mode == 0) ? (a.stride * n0 - a.pad) : (n0 - (taps - 1));

  const int span = p.sp * p.ncols;
  const int rowlen = p.rowlen;                     // positions (16-byte slots) per polyphase row
  const int ngr = p.CC >> 4;                       // 16-channel groups per chunk
  const int xslots = 2 * p.chs;                    // slots per group of the input image (chs = sp * rowlen)
  const int wslots = taps * 2 * OT;                // slots per group of the weight image
  const size_t bufsl = (size_t)ngr * (xslots + wslots);   // 16-byte slots per LDS buffer
  const int nchunk = (p.Cpad + p.CC - 1) / p.CC;
  cu32x4* lds = reinterpret_cast<cu32x4*>(smem);

  float* bias_s = smem + p.nbuf * bufsl * 4;
  if (tid < OT) {
    const int row = row0 + tid;
    bias_s[tid] = (a.bias && row < p.Mrows) ? a.bias[a.mode == 0 ? row : row / a.stride] : 0.f;
  }

  if (__builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) {
    // ---------------- staging waves
    __builtin_amdgcn_s_setprio(2);
    const int sl = cw * 64 + lane;                 // 0 .. 255
    // Buffer loads: one 32-bit offset register per load instead of a 64-bit address, and every piece that must read
    // as zero (halo outside the signal, channels >= C, rows >= Mpad, groups beyond the image) simply carries an
    // offset past num_records.  (Host side guarantees: 16-byte pieces lie entirely inside or outside the signal.)
    const unsigned OOB = 0x80000000u;
    __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.x + (int64_t)b * a.x_bs), 0, (int)((int64_t)a.C * a.x_cs * 4), 0x00020000);
    const int ngroups = (p.Cpad + 15) >> 4;        // groups in the weight image
    __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.wp + ag_wq_offset(p.Cpad, taps, p.Mpad)), 0, (int)((int64_t)ngroups * taps * 2 * p.Mpad * 16), 0x00020000);
    const int base4 = base - (((base % 4) + 4) % 4);
    const int nq = (base - base4 + span + 3) / 4;
    const int xtot = ngr * 2 * nq, wtot = ngr * wslots;
    // chunk-invariant part of this lane's tasks
    int xgh[CB_NX], xg[CB_NX];
    unsigned xo[CB_NX], wof[NW];
#pragma unroll
    for (int u = 0; u < CB_NX; ++u) {
      const int e = sl + u * 256;
      const int gh = e < xtot ? e / nq : -1;
      xgh[u] = gh;
      xg[u] = base4 + 4 * (e - max(gh, 0) * nq);
      xo[u] = (gh >= 0 && xg[u] >= 0 && xg[u] + 3 < a.Lin) ? (unsigned)((gh * 8 * (int)a.x_cs + xg[u]) * 4) : OOB;
    }
#pragma unroll
    for (int u = 0; u < NW; ++u) {
      const int idx = sl + u * 256;                // = gth * OT + row
      const int gth = idx / OT, row = row0 + (idx % OT);
      wof[u] = (idx < wtot && row < p.Mpad) ? (unsigned)((gth * p.Mpad + row) * 16) : OOB;
    }
    const unsigned xcs4 = (unsigned)a.x_cs * 4u;
    cu32x4 wv[NW];
    cu32x4 xv[CB_NX][8];
    auto load = [&](int ci) {
      const unsigned wadd = (unsigned)(ci * ngr * taps * 2 * p.Mpad * 16);
      const unsigned xadd = (unsigned)(ci * p.CC) * xcs4;
#pragma unroll
      for (int u = 0; u < NW; ++u) wv[u] = __builtin_amdgcn_raw_buffer_load_b128(wr, wof[u] + wadd, 0, 0);
#pragma unroll
      for (int u = 0; u < CB_NX; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j)
          xv[u][j] = __builtin_amdgcn_raw_buffer_load_b128(xr, xo[u] + xadd + (unsigned)j * xcs4, 0, 0);
    };
    auto write = [&](int buf) {
      cu32x4* xs = lds + buf * bufsl;                  // [ngr][2][sp][rowlen]
      cu32x4* ws = xs + (size_t)ngr * xslots;          // [ngr][taps][2][OT]
#pragma unroll
      for (int u = 0; u < NW; ++u) {
        const int idx = sl + u * 256;
        if (idx < wtot) ws[idx] = wv[u];
      }
#pragma unroll
      for (int u = 0; u < CB_NX; ++u) {
        if (xgh[u] < 0) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int rem = xg[u] + q - base;
          if (rem < 0 || rem >= span) continue;
          int r, qq;
          if (p.sp_shift >= 0) {
            r = rem & (p.sp - 1);
            qq = rem >> p.sp_shift;
          } else {
            qq = rem / p.sp;
            r = rem - qq * p.sp;
          }
#define XF(j) __uint_as_float(xv[u][j][q])
          const cu32x4 w = {ag_pack_bf16(XF(0), XF(1)), ag_pack_bf16(XF(2), XF(3)), ag_pack_bf16(XF(4), XF(5)),
                            ag_pack_bf16(XF(6), XF(7))};
#undef XF
          xs[(size_t)xgh[u] * p.chs + r * rowlen + qq] = w;
        }
      }
    };
    load(0);
    write(0);
    if (nchunk > 1) load(1);
    __syncthreads();
    for (int ci = 0; ci < nchunk; ++ci) {
      if (ci + 1 < nchunk) write((ci + 1) & 1);
      if (ci + 2 < nchunk) load(ci + 2);
      __syncthreads();
    }
    return;
  }
  // ---------------- MFMA waves
  f32x16 acc[TILES_O][TILES_T];
#pragma unroll
  for (int i = 0; i < TILES_O; ++i)
#pragma unroll
    for (int j = 0; j < TILES_T; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  __syncthreads();
  for (int ci = 0; ci < nchunk; ++ci) {
    const cu32x4* xs = lds + (ci & 1) * bufsl;
    const cu32x4* ws = xs + (size_t)ngr * xslots;
    for (int g = 0; g < ngr; ++g) {
      const cu32x4* wg = ws + (size_t)(g * taps * 2 + h) * OT + wrow0 + l31;
      const cu32x4* xg = xs + (size_t)(g * 2 + h) * p.chs + wcol0 + l31;
      for (int tau = 0; tau < taps; ++tau) {
        cbf16x8 av[TILES_O], bv[TILES_T];
#pragma unroll
        for (int i = 0; i < TILES_O; ++i) av[i] = *reinterpret_cast<const cbf16x8*>(wg + (size_t)tau * 2 * OT + 32 * i);
        const int off = p.tapoff[tau];
#pragma unroll
        for (int j = 0; j < TILES_T; ++j) bv[j] = *reinterpret_cast<const cbf16x8*>(xg + off + 32 * j);
#pragma unroll
        for (int i = 0; i < TILES_O; ++i)
#pragma unroll
          for (int j = 0; j < TILES_T; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  conv_epilogue<TILES_O, TILES_T>(p, acc, bias_s, b, row0, n0, wrow0, wcol0, l31, h);
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static inline int ilog2_exact(int v) {
  for (int s = 0; s < 31; ++s)
    if ((1 << s) == v) return s;
  return -1;
}

// fp32 layout + its bf16 image (common.h ag_wq_*)
static const bool g_conv_efast = [] { const char* e = getenv("AG_CONV_EFAST"); return !(e && e[0] == '0'); }();
static const bool g_conv_pipe = [] { const char* e = getenv("AG_CONV_PIPE"); return !(e && e[0] == '0'); }();
// AG_CONV_BF16_MFMA=0 keeps the fp32-MFMA rounding emulation in bf16 mode (A/B measurements)
static const bool g_conv_bf16_mfma = [] { const char* e = getenv("AG_CONV_BF16_MFMA"); return !(e && e[0] == '0'); }();

// AG_CONV_SOLO: -1 (default) = by the heuristic of launch_cfg, 0 = never, 1 = every 128 x 128-tile fp32 launch; AG_CONV_SOLO_K:
// largest reduction length C * taps that takes the solo form under the heuristic.  Measured at batch 64 (tools/prof_layers.py,
// profiles/r04_conv_solo.txt; 8-wave form -> solo -> solo + DMA staging, microseconds): the generator's transposed convs forward
// 92/88/86/72 -> 87/81/80/63 -> 88/83/84/55, its strided convs' backward-data 89/162/172 -> 83/150/139 (DMA: the same); a
// strided conv itself (mode 0, G1.deconv backward-data) LOSES 56 -> 68, so the form is taken for mode 1 only.
static const int g_conv_solo = [] { const char* e = getenv("AG_CONV_SOLO"); return e ? atoi(e) : -1; }();
static const int g_conv_solo_k = [] { const char* e = getenv("AG_CONV_SOLO_K"); return e ? atoi(e) : 512; }();
static const bool g_conv_dma = [] { const char* e = getenv("AG_CONV_DMA"); return !(e && e[0] == '0'); }();

template <int TO, int TTL, int WO, int WT, int TAPS, int S0>
static int launch_one(ConvP& p, size_t lds, dim3 grid, hipStream_t st) {
  auto kern = conv_engine_kernel<TO, TTL, WO, WT, TAPS, S0>;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kern, grid, dim3(p.solo ? 256 : 512), lds, st, p);
  AG_CHECK_LAUNCH("ag_conv1d_engine");
  return AG_OK;
}

// bf16 MFMA variant (AG_PREC_BF16, >= 16 input channels): chunks of whole 16-channel groups
template <int TO, int TTL, int WO, int WT, int NW>
static int launch_bf16_nw(ConvP& p, hipStream_t st) {
  constexpr int OT = 32 * TO * WO, TT = 32 * TTL * WT;
  const ag_conv_args& a = p.a;
  const size_t per_g = (size_t)(2 * p.chs + p.taps * 2 * OT) * 16;     // bytes per 16-channel group
  int ng = (int)(((NW > 8 ? 74 : 36) * 1024) / per_g);
  if (ng < 1) ng = 1;
  if (ng > 4) ng = 4;
  const int groups = ag_cdiv(p.Cpad, 16);
  if (ng > groups) ng = groups;
  // what the 256 staging lanes hold in registers per chunk: NW weight slots and CB_NX input tasks each
  const int nq = (p.sp * p.ncols + 6) / 4 + 1;
  while (ng > 1 && (ng * p.taps * 2 * OT > NW * 256 || ng * 2 * nq > CB_NX * 256)) --ng;
  if (p.taps * 2 * OT > NW * 256 || 2 * nq > CB_NX * 256) return -1;
  p.CC = 16 * ng;
  p.nbuf = ng >= groups ? 1 : 2;                     // a single chunk needs no second buffer
  const size_t lds = p.nbuf * (size_t)ng * per_g + (size_t)OT * sizeof(float);
  if (lds > 160 * 1024) return -1;                   // (many taps x wide polyphase window) -> caller keeps the fp32-MFMA path
  dim3 grid(ag_cdiv(p.n_cnt, TT), ag_cdiv(p.Mrows, OT), a.B);
  auto kern = conv_engine_bf16_kernel<TO, TTL, WO, WT, NW>;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, p);
  AG_CHECK_LAUNCH("ag_conv1d_engine(bf16)");
  return AG_OK;
}

template <int TO, int TTL, int WO, int WT>
static int launch_bf16(ConvP& p, hipStream_t st) {
  constexpr int OT = 32 * TO * WO, TT = 32 * TTL * WT;
  const ag_conv_args& a = p.a;
  const int dmax = (a.mode == 0) ? (a.K - 1) / a.stride : (p.taps - 1);
  p.ncols_tile = TT;
  p.ncols = TT + dmax;
  p.rowlen = p.ncols;
  p.chs = p.sp * p.rowlen;
  for (int t = 0; t < p.taps; ++t)
    p.tapoff[t] = (a.mode == 0) ? ((t % a.stride) * p.rowlen + t / a.stride) : (p.taps - 1 - t);
  // buffer-addressed staging: 16-byte input pieces entirely inside or outside the signal, 31-bit byte offsets
  if (!p.xvec || a.Lin % 4 != 0 || (int64_t)a.C * a.x_cs * 4 >= ((int64_t)1 << 31) ||
      ag_wq_floats(p.Cpad, p.taps, p.Mpad) * 4 >= ((int64_t)1 << 31))
    return -1;
  if (OT >= 128 && ag_cdiv(p.Cpad, 16) * p.taps >= 32) {      // deep reduction
    const int rc = launch_bf16_nw<TO, TTL, WO, WT, (OT >= 128 ? 16 : 8)>(p, st);
    if (rc != -1) return rc;
  }
  return launch_bf16_nw<TO, TTL, WO, WT, 8>(p, st);
}

template <int TO, int TTL, int WO, int WT>
static int launch_cfg(ConvP& p, hipStream_t st) {
  constexpr int OT = 32 * TO * WO, TT = 32 * TTL * WT;
  if (p.rb && p.a.C >= 16 && g_conv_bf16_mfma) {
    const int rc = launch_bf16<TO, TTL, WO, WT>(p, st);
    if (rc != -1) return rc;
  }
  const ag_conv_args& a = p.a;
  // input tile geometry
  const int dmax = (a.mode == 0) ? (a.K - 1) / a.stride : (p.taps - 1);
  p.ncols_tile = TT;
  p.ncols = TT + dmax;
  int rl = p.ncols;
  if (p.sp_shift >= 0 && p.sp > 1 && p.sp <= 32) {
    const int want = 32 / p.sp;  // rowlen == want (mod 32): polyphase rows land on disjoint banks
    rl = p.ncols + ((want - p.ncols) % 32 + 32) % 32;
  }
  if ((p.sp & 1) && (rl & 1)) rl += 1;  // CC (even) * chs must be a multiple of 4 floats: 16-B aligned weight rows
  p.rowlen = rl;
  p.chs = p.sp * p.rowlen;
  for (int t = 0; t < p.taps; ++t)
    p.tapoff[t] = (a.mode == 0) ? ((t % a.stride) * p.rowlen + t / a.stride) : (p.taps - 1 - t);
  // channels per chunk: as many (even, <= 32) as fit in 2 x 36 KiB of LDS (two buffers)
  const size_t per_c = (size_t)(p.chs + p.taps * OT) * sizeof(float);
  int cc = (int)((36 * 1024) / per_c) & ~1;
  if (cc < 2) cc = 2;
  if (cc > 32) cc = 32;
  if (cc > p.Cpad) cc = p.Cpad;
  // Register-pipelined staging: a chunk is one round of loads for the 256 staging lanes (FX input and FW weight
  // 16-byte pieces each: 7/3 for 32-row tiles, else 4/8), pieces lie entirely inside or outside the signal, offsets
  // fit 31 bits.  The chunk is trimmed to that round.
  p.pipe = 0;
  if (p.xvec && a.Lin % 4 == 0 && a.Lin >= 4 && (int64_t)a.C * a.x_cs * 4 < ((int64_t)1 << 31) &&
      (int64_t)p.Cpad * p.taps * p.Mpad * 4 < ((int64_t)1 << 31) && g_conv_pipe) {
    const int nq = (p.sp * p.ncols + 6) / 4 + 1;
    int cf = cc;
    constexpr int FXH = OT <= 32 ? 7 : 4, FWH = OT <= 32 ? 3 : 8;   // = the kernel's FX, FW
    while (cf > 2 && (cf * nq > FXH * 256 || cf * p.taps * (OT / 4) > FWH * 256)) cf -= 2;
    // (a reduction that is a single chunk has nothing to overlap: chunk 0 by all 8 waves is quicker)
    if (cf * nq <= FXH * 256 && cf * p.taps * (OT / 4) <= FWH * 256 && cf < p.Cpad) {
      cc = cf;
      p.pipe = 1;
    }
  }
  // solo form (see the kernel): AG_CONV_SOLO = 1 forces it for every 128 x 128-tile launch, 0 forbids it (A/B runs);
  // AG_CONV_DMA = 0 keeps its staging on the register path
  p.solo = 0;
  p.dma = 0;
  size_t per_cs = per_c;
  if (OT == 128 && (TT == 128 || TT == 64) && g_conv_solo != 0) {
    const int ktot = p.Cpad * p.taps;
    // (half-width tiles: the solo form wins up to C * taps = 128, the 8-wave form above - G1.deconv forward 78 vs 88 us)
    if (g_conv_solo == 1 || (a.mode == 1 && ktot <= (TT == 64 ? g_conv_solo_k / 4 : g_conv_solo_k))) {
      p.solo = 1;
      p.pipe = 0;
      if (a.mode == 1 && !p.rb && g_conv_dma && p.xvec && a.Lin % 4 == 0 && p.Cpad % 16 == 0 && p.Mpad % 4 == 0 &&
          (((uintptr_t)a.wp) & 15) == 0) {
        // DMA staging: row pitch a multiple of 16 floats that covers the window from floor4(base) on
        p.dma = 1;
        p.rowlen = ag_roundup(4 * ((p.ncols + 6) / 4), 16);
        p.chs = p.rowlen;
        per_cs = (size_t)(p.chs + p.taps * OT) * sizeof(float);
        cc = (p.Cpad % 32 == 0 && 32 * per_cs <= 40 * 1024) ? 32 : 16;
      } else {
        cc = (int)((36 * 1024) / per_c) & ~1;          // the generic staging has no per-lane round limit
        if (cc < 2) cc = 2;
        if (cc > 32) cc = 32;
        if (cc > p.Cpad) cc = p.Cpad;
      }
    }
  }
  p.CC = cc;
  const size_t lds = (p.solo ? 1 : 2) * (size_t)cc * per_cs + (size_t)OT * sizeof(float);   // buffers + the bias of the row tile
  if (lds > 160 * 1024) {
    ag_set_error("conv engine: tile needs %zu B of LDS", lds);
    return AG_ERR_UNSUPPORTED;
  }
  dim3 grid(ag_cdiv(p.n_cnt, TT), ag_cdiv(p.Mrows, OT), a.B);
  // (the 64x64 / 128x32 tiles only run the ragged tail columns of transposed convs: no specialisations for them)
  constexpr bool SPEC = !(TO == 1 && TTL == 1);
  const int key = !SPEC ? -1 : (a.mode == 0 ? p.taps * 100 + a.stride : p.taps * 100);
  if constexpr (!SPEC) return launch_one<TO, TTL, WO, WT, 0, 0>(p, lds, grid, st);
  else
  switch (key) {
    case 1708: return launch_one<TO, TTL, WO, WT, 17, 8>(p, lds, grid, st);  // G1.conv
    case 904:  return launch_one<TO, TTL, WO, WT, 9, 4>(p, lds, grid, st);   // G2-4.conv
    case 702:  return launch_one<TO, TTL, WO, WT, 7, 2>(p, lds, grid, st);   // D convs
    case 301:  return launch_one<TO, TTL, WO, WT, 3, 1>(p, lds, grid, st);   // G5.final
    case 1608: return launch_one<TO, TTL, WO, WT, 16, 8>(p, lds, grid, st);  // G1.deconv backward-data
    case 804:  return launch_one<TO, TTL, WO, WT, 8, 4>(p, lds, grid, st);   // G2-4.deconv backward-data
    case 200:  return launch_one<TO, TTL, WO, WT, 2, 0>(p, lds, grid, st);   // deconv fwd (k16 s8, k8 s4)
    case 300:  return launch_one<TO, TTL, WO, WT, 3, 0>(p, lds, grid, st);   // conv bwd-data k17 s8, k9 s4, k3 s1
    case 400:  return launch_one<TO, TTL, WO, WT, 4, 0>(p, lds, grid, st);   // conv bwd-data k7 s2
    default:   return launch_one<TO, TTL, WO, WT, 0, 0>(p, lds, grid, st);
  }
}

