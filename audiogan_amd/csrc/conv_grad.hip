// conv_grad.hip -- weight / bias gradients of the 1-D convolutions and the fused
// LeakyReLU(+mask) backward.  Replaces the cuDNN backward-weight path behind
// loss.backward() (audiogan.py:785,:903) for NN.Conv1d / NN.ConvTranspose1d.
//
//   dw[a, c, k] += sum_{b,t} sh[b, a, t] * lg[b, c, s*t + k - p]
//
// is a GEMM whose reduction axis is (b, t): MFMA row i <-> a, col j <-> flattened
// (c,k), k-pair <-> two consecutive t.  Each workgroup owns one [rows x cols] tile of
// dw and a strided share of the (b, time-chunk) reduction; partial tiles are combined
// with fp32 global atomics (one 128-B row segment per half-wave, the full-rate shape).
#include "common.h"

struct WgP {
  const float* sh;
  const float* lg;
  float* dw;
  int64_t sh_bs, sh_cs, lg_bs, lg_cs;
  int B, A, Lsh, C, Llg, K, s, p;
  int CK;       // C*K
  int TC;       // time chunk (multiple of 4)
  int nchunk;   // chunks per clip
  int lgp;      // LDS pitch of one lg channel row (odd)
  int maxch;    // channel rows staged per tile
  int vec;      // sh rows may be read with 16-byte loads
  float* part;  // two-stage reduction: slab z = blockIdx.z of [A][CK] partial sums (plain stores); never NULL: there is no float-atomic path
  int rb;       // AG_PREC_BF16: both operands rounded to bf16 while staging
};

// 8 waves: waves 0-3 multiply chunk i out of LDS buffer i&1 while waves 4-7 stage chunk i+1.
// LDS: sh tile TRANSPOSED [TC][AT+1] (MFMA A operand = unit-stride read), lg rows [maxch][lgp].
template <int TA, int TN, int WA, int WN>
__global__ __launch_bounds__(512) void conv_wgrad_kernel(const WgP p) {
  // XCD-aware tile order (workgroup ids go round-robin over the 8 XCDs, each with its own L2): XCD i works on a contiguous
  // run of logical tiles, i.e. on all the column tiles of a few (row tile, reduction share) pairs, which read the SAME
  // gradient rows - they then come out of that XCD's L2 instead of crossing the fabric once per column tile (speed only)
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  const int gz_ = gridDim.z;
  {
    const int gx = gridDim.x, gy = gridDim.y, nwg = gx * gy * gz_;
    if (nwg >= 16) {
      // XCD i = lin % 8 holds nwg / 8 (+1 for i < nwg % 8) workgroups: give it that many CONSECUTIVE logical tiles
      const int lin = (bz * gy + by) * gx + bx;
      const int xi = lin & 7, q = nwg >> 3, r = nwg & 7;
      const int t = xi * q + (xi < r ? xi : r) + (lin >> 3);
      bx = t % gx; by = (t / gx) % gy; bz = t / (gx * gy);
    }
  }
  static_assert(WA * WN == 4, "4 compute waves");
  constexpr int AT = 32 * TA * WA, NT = 32 * TN * WN;
  constexpr int SP = AT + 1;
  extern __shared__ float smem[];
  const size_t bufsz = (size_t)p.TC * SP + (size_t)p.maxch * p.lgp;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int cw = wid & 3;
  const int wa = cw / WN, wn = cw % WN;
  const int a0 = by * AT, ck0 = bx * NT;
  const int c_lo = ck0 / p.K;
  int c_hi = (ck0 + NT - 1) / p.K;
  if (c_hi >= p.C) c_hi = p.C - 1;
  const int nch = c_hi - c_lo + 1;
  const int span = p.s * (p.TC - 1) + p.K;
  const int total = p.B * p.nchunk;
  const int nmine = (total - bz + gz_ - 1) / gz_;   // chunks of this block

  auto stage = [&](int ch, int buf, int sw, int nsw) {
    float* shs = smem + buf * bufsz;
    float* lgs = shs + (size_t)p.TC * SP;
    const int b = ch / p.nchunk;
    const int t0 = (ch - b * p.nchunk) * p.TC;
    const float* sb = p.sh + (int64_t)b * p.sh_bs;
    const float* lb = p.lg + (int64_t)b * p.lg_bs;
    const int g0 = p.s * t0 - p.p;
    const int q4 = p.TC >> 2;                 // float4 per sh row
    const int stot = AT * q4, ltot = nch * span, step = nsw * 64;
    constexpr int US = 4, UL = 4;
    int se = sw * 64 + lane, le = se;
    while (se < stot || le < ltot) {
      f32x4 sv[US];
      float lv[UL];
#pragma unroll
      for (int u = 0; u < US; ++u) {
        const int e = se + u * step;
        sv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (e < stot) {
          const int r = e / q4, q = e - r * q4;
          const int a = a0 + r, tg = t0 + 4 * q;
          if (a < p.A) {
            const float* src = sb + (int64_t)a * p.sh_cs + tg;
            if (p.vec && tg + 3 < p.Lsh) {
              sv[u] = *reinterpret_cast<const f32x4*>(src);
            } else {
#pragma unroll
              for (int x = 0; x < 4; ++x)
                if (tg + x < p.Lsh) sv[u][x] = src[x];
            }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < UL; ++u) {
        const int e = le + u * step;
        lv[u] = 0.f;
        if (e < ltot) {
          const int r = e / span, i = e - r * span;
          const int g = g0 + i;
          if (g >= 0 && g < p.Llg) lv[u] = lb[(int64_t)(c_lo + r) * p.lg_cs + g];
        }
      }
      if (p.rb) {      // AG_PREC_BF16 (uniform branch): both operands rounded on the way into LDS
#pragma unroll
        for (int u = 0; u < US; ++u) sv[u] = ag_rbf4_if(sv[u], 1);
#pragma unroll
        for (int u = 0; u < UL; ++u) lv[u] = ag_rbf(lv[u]);
      }
#pragma unroll
      for (int u = 0; u < US; ++u) {
        const int e = se + u * step;
        if (e < stot) {
          const int r = e / q4, q = e - r * q4;
#pragma unroll
          for (int x = 0; x < 4; ++x) shs[(4 * q + x) * SP + r] = sv[u][x];
        }
      }
#pragma unroll
      for (int u = 0; u < UL; ++u) {
        const int e = le + u * step;
        if (e < ltot) {
          const int r = e / span, i = e - r * span;
          lgs[r * p.lgp + i] = lv[u];
        }
      }
      se += US * step;
      le += UL * step;
    }
  };

  if (nmine > 0) stage(bz, 0, wid, 8);
  __syncthreads();

  if (__builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) {
    // ---- staging waves.  They share each SIMD's issue slots with an MFMA wave (arbitrated by priority, then
    // age - they are the younger half), so they get static priority and, on the fast path, a loop without
    // integer divisions: every lane's source offsets and LDS targets are the same for every chunk (pieces are
    // 16-byte aligned in the signal, so a piece is entirely inside or outside it); only two base pointers move.
    __builtin_amdgcn_s_setprio(2);
    constexpr int FS = 4, FL = 5;
    const int shift = (((-p.p) % 4) + 4) % 4;            // (s*t0 - p) mod 4, the same for every chunk
    const int PR = (shift + span + 3) / 4;               // aligned pieces per lg row
    const int q4 = p.TC >> 2;
    const int stot = AT * q4, ltot4 = nch * PR;
    const bool fastp = p.vec && (p.Lsh % p.TC == 0) && ((p.s * p.TC) % 4 == 0) && (p.Llg % 4 == 0) && p.Llg >= 4 &&
                       (p.lg_cs % 4 == 0) && (p.lg_bs % 4 == 0) && (((uintptr_t)p.lg & 15) == 0) &&
                       stot <= FS * 256 && ltot4 <= FL * 256 && (int64_t)AT * p.sh_cs < (1 << 30) &&
                       (int64_t)p.maxch * p.lg_cs < (1 << 30);
    int ssrc[FS], sl[FS], lsrc[FL], li0[FL], lrow[FL];
    if (fastp) {
#pragma unroll
      for (int u = 0; u < FS; ++u) {
        const int e = cw * 64 + lane + u * 256;
        const int r = e / q4, q = e - r * q4;
        const bool ok = e < stot && a0 + r < p.A;
        ssrc[u] = ok ? (int)((a0 + r) * p.sh_cs) + 4 * q : 0;
        sl[u] = e < stot ? (4 * q) * SP + r : -1;          // rows past A are staged as zeros
        if (e < stot && !ok) ssrc[u] = -1;
      }
#pragma unroll
      for (int u = 0; u < FL; ++u) {
        const int e = cw * 64 + lane + u * 256;
        const int r = e / PR, m = e - r * PR;
        const bool ok = e < ltot4;
        lsrc[u] = ok ? (int)((c_lo + r) * p.lg_cs) + 4 * m : 0;
        li0[u] = ok ? 4 * m - shift : -(1 << 20);          // window index of the piece's first element
        lrow[u] = r * p.lgp;
      }
    }
    int ci = 0;
    if (fastp) {
      for (; ci < nmine; ++ci) {
        if (ci + 1 < nmine) {
          const int ch = bz + (ci + 1) * gz_;
          const int b = ch / p.nchunk;
          const int t0 = (ch - b * p.nchunk) * p.TC;
          const int g0 = p.s * t0 - p.p;
          float* shs = smem + ((ci + 1) & 1) * bufsz;
          float* lgs = shs + (size_t)p.TC * SP;
          const float* sbase = p.sh + (int64_t)b * p.sh_bs + t0;
          const float* lrow0 = p.lg + (int64_t)b * p.lg_bs;
          f32x4 sv[FS], lv[FL];
          bool lok[FL];
#pragma unroll
          for (int u = 0; u < FS; ++u) sv[u] = *reinterpret_cast<const f32x4*>(sbase + max(ssrc[u], 0));
#pragma unroll
          for (int u = 0; u < FL; ++u) {
            const int gp = g0 + li0[u];                    // global index of the piece's first element
            lok[u] = gp >= 0 && gp + 3 < p.Llg;
            lv[u] = *reinterpret_cast<const f32x4*>(lrow0 + (lok[u] ? lsrc[u] + (g0 - shift) : 0));
          }
          if (p.rb) {      // AG_PREC_BF16 (uniform branch)
#pragma unroll
            for (int u = 0; u < FS; ++u) sv[u] = ag_rbf4_if(sv[u], 1);
#pragma unroll
            for (int u = 0; u < FL; ++u) lv[u] = ag_rbf4_if(lv[u], 1);
          }
#pragma unroll
          for (int u = 0; u < FS; ++u)
            if (sl[u] >= 0) {
#pragma unroll
              for (int x = 0; x < 4; ++x) shs[sl[u] + x * SP] = ssrc[u] < 0 ? 0.f : sv[u][x];
            }
#pragma unroll
          for (int u = 0; u < FL; ++u)
#pragma unroll
            for (int x = 0; x < 4; ++x) {
              const int i = li0[u] + x;
              if (i >= 0 && i < span) lgs[lrow[u] + i] = lok[u] ? lv[u][x] : 0.f;
            }
        }
        __syncthreads();
      }
    }
    for (; ci < nmine; ++ci) {
      if (ci + 1 < nmine) stage(bz + (ci + 1) * gz_, (ci + 1) & 1, cw, 4);
      __syncthreads();
    }
    return;
  }

  // per-lane column constants
  int joff[TN];
  bool jok[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int ck = ck0 + wn * 32 * TN + 32 * j + l31;
    jok[j] = ck < p.CK;
    const int c = jok[j] ? ck / p.K : c_lo;
    const int k = jok[j] ? ck - c * p.K : 0;
    joff[j] = (c - c_lo) * p.lgp + k + p.s * h;
  }
  f32x16 acc[TA][TN];
#pragma unroll
  for (int i = 0; i < TA; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  for (int ci = 0; ci < nmine; ++ci) {
    const float* shs = smem + (ci & 1) * bufsz;
    const float* lgs = shs + (size_t)p.TC * SP;
    const float* arow = shs + h * SP + wa * 32 * TA + l31;
    for (int t = 0; t < p.TC; t += 2) {
      float av[TA], bv[TN];
#pragma unroll
      for (int i = 0; i < TA; ++i) av[i] = arow[t * SP + 32 * i];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        // (unconditional read of a valid address + select: a conditional load compiles to an exec-masked branch per
        // operand inside the MFMA loop)
        const float bval = lgs[joff[j] + p.s * t];
        bv[j] = jok[j] ? bval : 0.f;
      }
#pragma unroll
      for (int i = 0; i < TA; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

#pragma unroll
  for (int i = 0; i < TA; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int a = a0 + wa * 32 * TA + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
      if (a >= p.A) continue;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int ck = ck0 + wn * 32 * TN + 32 * j + l31;
        if (ck >= p.CK) continue;
        p.part[((int64_t)bz * p.A + a) * p.CK + ck] = acc[i][j][e];
      }
    }
}

// ------------------------------------------------------------------------------------------
// AG_PREC_BF16: the same weight gradient on v_mfma_f32_32x32x16_bf16.  The reduction runs over time, so one MFMA
// k-step = 16 consecutive output positions t: lane (l31, h) supplies t = tb + 8h .. +7.
//   A operand (sh = the gradient side, [a][t] in memory): LDS image [AT rows][TC x bf16] (+16 B pad per row) - staged
//     with plain 16-byte loads, rounded, written 8 bytes at a time; a fragment is ONE ds_read_b128.
//   B operand (lg window): LDS rows [channel][span x bf16]; column (c, k) at step t reads lg[c][s*t + k - p], i.e. 8
//     ds_read_u16 at stride s per fragment (an im2col image would cost K times the LDS).
// Same chunk / slab / fixed-order second stage as the fp32 kernel; staging waves run the register pipeline of the conv
// engine (loads of chunk i+2 in flight while chunk i+1 is written).  Results equal the rounding emulation in the fp32
// kernel up to fp32 summation order.
typedef short wbf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned wu32x4 __attribute__((ext_vector_type(4)));
typedef unsigned wu32x2 __attribute__((ext_vector_type(2)));
#define WB_FS 8   // sh pieces (16 bytes of fp32) per staging lane and chunk
#define WB_FL 5   // lg pieces per staging lane and chunk

template <int TA, int TN, int WA, int WN>
__global__ __launch_bounds__(512) void conv_wgrad_bf16_kernel(const WgP p) {
  // XCD-aware tile order (workgroup ids go round-robin over the 8 XCDs, each with its own L2): XCD i works on a contiguous
  // run of logical tiles, i.e. on all the column tiles of a few (row tile, reduction share) pairs, which read the SAME
  // gradient rows - they then come out of that XCD's L2 instead of crossing the fabric once per column tile (speed only)
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  const int gz_ = gridDim.z;
  {
    const int gx = gridDim.x, gy = gridDim.y, nwg = gx * gy * gz_;
    if (nwg >= 16) {
      // XCD i = lin % 8 holds nwg / 8 (+1 for i < nwg % 8) workgroups: give it that many CONSECUTIVE logical tiles
      const int lin = (bz * gy + by) * gx + bx;
      const int xi = lin & 7, q = nwg >> 3, r = nwg & 7;
      const int t = xi * q + (xi < r ? xi : r) + (lin >> 3);
      bx = t % gx; by = (t / gx) % gy; bz = t / (gx * gy);
    }
  }
  static_assert(WA * WN == 4, "4 compute waves");
  constexpr int AT = 32 * TA * WA, NT = 32 * TN * WN;
  extern __shared__ float smem[];
  char* lds = reinterpret_cast<char*>(smem);
  const int SPB = p.TC * 2 + 16;                       // bytes per sh row
  const int LGPB = p.lgp;                              // bytes per lg channel row (multiple of 4)
  const size_t bufb = ((size_t)AT * SPB + (size_t)p.maxch * LGPB + 15) & ~(size_t)15;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int cw = wid & 3;
  const int wa = cw / WN, wn = cw % WN;
  const int a0 = by * AT, ck0 = bx * NT;
  const int c_lo = ck0 / p.K;
  int c_hi = (ck0 + NT - 1) / p.K;
  if (c_hi >= p.C) c_hi = p.C - 1;
  const int nch = c_hi - c_lo + 1;
  const int span = p.s * (p.TC - 1) + p.K;
  const int total = p.B * p.nchunk;
  const int nmine = (total - bz + gz_ - 1) / gz_;   // chunks of this block

  if (__builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) {
    // ---------------- staging waves
    __builtin_amdgcn_s_setprio(2);
    const int sl_ = cw * 64 + lane;
    const int shift = (((-p.p) % 4) + 4) % 4;            // (s*t0 - p) mod 4, the same for every chunk
    const int PR = (shift + span + 3) / 4;               // aligned pieces per lg row
    const int q4 = p.TC >> 2;
    const int stot = AT * q4, ltot4 = nch * PR;
    int ssrc[WB_FS], sdst[WB_FS], lsrc[WB_FL], li0[WB_FL], lrow[WB_FL];
#pragma unroll
    for (int u = 0; u < WB_FS; ++u) {
      const int e = sl_ + u * 256;
      const int r = e / q4, q = e - r * q4;
      const bool ok = e < stot && a0 + r < p.A;
      ssrc[u] = ok ? (int)((a0 + r) * p.sh_cs) + 4 * q : -1;      // -1: zeros (rows past A)
      sdst[u] = e < stot ? r * SPB + 8 * q : -1;
    }
#pragma unroll
    for (int u = 0; u < WB_FL; ++u) {
      const int e = sl_ + u * 256;
      const int r = e / PR, m = e - r * PR;
      const bool ok = e < ltot4;
      lsrc[u] = ok ? (int)((c_lo + r) * p.lg_cs) + 4 * m : 0;
      li0[u] = ok ? 4 * m - shift : -(1 << 20);          // window index of the piece's first element
      lrow[u] = r * LGPB;
    }
    f32x4 sv[WB_FS], lv[WB_FL];
    bool lok[WB_FL];
    auto loadc = [&](int cix) __attribute__((always_inline)) {
      const int ch = bz + cix * gz_;
      const int b = ch / p.nchunk;
      const int t0 = (ch - b * p.nchunk) * p.TC;
      const int g0 = p.s * t0 - p.p;
      const float* sbase = p.sh + (int64_t)b * p.sh_bs + t0;
      const float* lrow0 = p.lg + (int64_t)b * p.lg_bs;
#pragma unroll
      for (int u = 0; u < WB_FS; ++u) sv[u] = *reinterpret_cast<const f32x4*>(sbase + max(ssrc[u], 0));
#pragma unroll
      for (int u = 0; u < WB_FL; ++u) {
        const int gp = g0 + li0[u];                    // global index of the piece's first element
        lok[u] = gp >= 0 && gp + 3 < p.Llg;
        lv[u] = *reinterpret_cast<const f32x4*>(lrow0 + (lok[u] ? lsrc[u] + (g0 - shift) : 0));
      }
    };
    auto writec = [&](int buf) __attribute__((always_inline)) {
      char* shs = lds + buf * bufb;
      char* lgs = shs + (size_t)AT * SPB;
#pragma unroll
      for (int u = 0; u < WB_FS; ++u)
        if (sdst[u] >= 0) {
          wu32x2 w = {ag_pack_bf16(sv[u][0], sv[u][1]), ag_pack_bf16(sv[u][2], sv[u][3])};
          if (ssrc[u] < 0) w = wu32x2{0u, 0u};
          *reinterpret_cast<wu32x2*>(shs + sdst[u]) = w;
        }
#pragma unroll
      for (int u = 0; u < WB_FL; ++u)
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          const int i = li0[u] + x;
          if (i >= 0 && i < span)
            *reinterpret_cast<unsigned short*>(lgs + lrow[u] + 2 * i) = lok[u] ? (unsigned short)ag_pack_bf16(lv[u][x], lv[u][x]) : (unsigned short)0;
        }
    };
    if (nmine > 0) {
      loadc(0);
      writec(0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (nmine > 1) loadc(1);
    __syncthreads();
    for (int ci = 0; ci < nmine; ++ci) {
      if (ci + 1 < nmine) writec((ci + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);        // (the loads below reuse the registers just written out)
      if (ci + 2 < nmine) loadc(ci + 2);
      __syncthreads();
    }
    return;
  }

  // ---------------- MFMA waves: per-lane column constants (byte offsets into the lg image)
  int joff[TN];
  bool jok[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int ck = ck0 + wn * 32 * TN + 32 * j + l31;
    jok[j] = ck < p.CK;
    const int c = jok[j] ? ck / p.K : c_lo;
    const int k = jok[j] ? ck - c * p.K : 0;
    joff[j] = (c - c_lo) * LGPB + 2 * (k + p.s * 8 * h);
  }
  f32x16 acc[TA][TN];
#pragma unroll
  for (int i = 0; i < TA; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  __syncthreads();      // chunk 0 staged
  const int s2 = 2 * p.s;
  for (int ci = 0; ci < nmine; ++ci) {
    const char* shs = lds + (ci & 1) * bufb;
    const char* lgs = shs + (size_t)AT * SPB;
    const char* arow = shs + (size_t)(wa * 32 * TA + l31) * SPB + 16 * h;
    for (int tb = 0; tb < p.TC; tb += 16) {
      wbf16x8 av[TA], bv[TN];
#pragma unroll
      for (int i = 0; i < TA; ++i) av[i] = *reinterpret_cast<const wbf16x8*>(arow + (size_t)32 * i * SPB + 2 * tb);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const char* bp = lgs + joff[j] + s2 * tb;
        wu32x4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const unsigned lo = *reinterpret_cast<const unsigned short*>(bp + s2 * (2 * e));
          const unsigned hi = *reinterpret_cast<const unsigned short*>(bp + s2 * (2 * e + 1));
          w[e] = jok[j] ? (lo | (hi << 16)) : 0u;
        }
        bv[j] = __builtin_bit_cast(wbf16x8, w);
      }
#pragma unroll
      for (int i = 0; i < TA; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

#pragma unroll
  for (int i = 0; i < TA; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int a = a0 + wa * 32 * TA + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
      if (a >= p.A) continue;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int ck = ck0 + wn * 32 * TN + 32 * j + l31;
        if (ck >= p.CK) continue;
        p.part[((int64_t)bz * p.A + a) * p.CK + ck] = acc[i][j][e];
      }
    }
}

// reduction slices (grid.z) of the MFMA kernel for an [A x CK] gradient tiled AT x NT
static int wgrad_slices(int A, int CK, int AT, int NT, int total) {
  const int gx = ag_cdiv(CK, NT), gy = ag_cdiv(A, AT);
  // ~2 workgroups of 8 waves per CU over the whole grid: every workgroup ends in a full tile of partial sums, so more
  // splits cost more than their shorter chunk loops save (measured over the 15 layers: 256 / 512 / 1024 / 2048
  // workgroups -> 1510 / 1218 / 1307 / 1522 us in total)
  int gz = ag_cdiv(512, gx * gy);
  if (gz > total) gz = total;
  if (gz < 1) gz = 1;
  return gz;
}

// AG_CONV_BF16_MFMA=0 keeps the fp32-MFMA rounding emulation in bf16 mode (A/B measurements)
static const bool g_wgrad_bf16_mfma = [] { const char* e = getenv("AG_CONV_BF16_MFMA"); return !(e && e[0] == '0'); }();

// bf16-MFMA variant; -1 = the shape does not fit its staging scheme (caller takes the fp32 kernel)
template <int TA, int TN, int WA, int WN>
static int launch_wgrad_bf16(WgP& p, hipStream_t st, AgWs ws) {
  constexpr int AT = 32 * TA * WA, NT = 32 * TN * WN;
  p.TC = 64;
  if (p.Lsh % p.TC != 0 || !p.vec || (p.s * p.TC) % 4 != 0 || p.Llg % 4 != 0 || p.Llg < 4 || p.lg_cs % 4 != 0 ||
      p.lg_bs % 4 != 0 || (((uintptr_t)p.lg) & 15) != 0)
    return -1;
  p.nchunk = p.Lsh / p.TC;
  const int span = p.s * (p.TC - 1) + p.K;
  p.lgp = ag_roundup(2 * span, 4) + 4;             // bytes per lg channel row
  p.maxch = (NT - 1) / p.K + 2;
  if (p.maxch > p.C) p.maxch = p.C;
  const int shift = (((-p.p) % 4) + 4) % 4;
  const int PR = (shift + span + 3) / 4;
  if (AT * (p.TC / 4) > WB_FS * 256 || p.maxch * PR > WB_FL * 256) return -1;
  if ((int64_t)AT * p.sh_cs >= (1 << 30) || (int64_t)p.C * p.lg_cs >= (1 << 30)) return -1;
  const size_t lds = 2 * (((size_t)AT * (p.TC * 2 + 16) + (size_t)p.maxch * p.lgp + 15) & ~(size_t)15);
  if (lds > 160 * 1024) return -1;
  const int gx = ag_cdiv(p.CK, NT), gy = ag_cdiv(p.A, AT);
  const int total = p.B * p.nchunk;
  int gz = wgrad_slices(p.A, p.CK, AT, NT, total);
  const int64_t n_out = (int64_t)p.A * p.CK;
  AG_REQUIRE(ws.p && ws.numel >= n_out, "%s: this reduction spans several workgroups and needs a workspace of >= %lld floats bound with ag_bind_workspace (sums are two-stage, in a fixed order; there is no float-atomic accumulation)", "ag_conv1d_wgrad", (long long)n_out);
  if ((int64_t)gz * n_out > ws.numel) gz = (int)(ws.numel / n_out);     // two-stage, fixed-order reduction
  p.part = ws.p;
  auto kern = conv_wgrad_bf16_kernel<TA, TN, WA, WN>;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kern, dim3(gx, gy, gz), dim3(512), lds, st, p);
  AG_CHECK_LAUNCH("ag_conv1d_wgrad(bf16)");
  return ag_slab_reduce(p.part, gz, n_out, p.dw, 1, st);
}

template <int TA, int TN, int WA, int WN>
static int launch_wgrad(WgP& p, hipStream_t st, AgWs ws) {
  constexpr int AT = 32 * TA * WA, NT = 32 * TN * WN;
  if (p.rb && g_wgrad_bf16_mfma) {
    const int rc = launch_wgrad_bf16<TA, TN, WA, WN>(p, st, ws);
    if (rc != -1) return rc;
  }
  p.TC = AT >= 128 ? 32 : 64;   // keeps two LDS buffers of the 128x128 tile under 48 KiB (>= 2 workgroups per CU)
  if (p.Lsh < p.TC) p.TC = ag_roundup(p.Lsh, 4);
  p.nchunk = ag_cdiv(p.Lsh, p.TC);
  const int span = p.s * (p.TC - 1) + p.K;
  p.lgp = span | 1;
  p.maxch = (NT - 1) / p.K + 2;
  if (p.maxch > p.C) p.maxch = p.C;
  const size_t lds = 2 * ((size_t)p.TC * (AT + 1) + (size_t)p.maxch * p.lgp) * sizeof(float);
  if (lds > 160 * 1024) {
    ag_set_error("conv wgrad: tile needs %zu B of LDS", lds);
    return AG_ERR_UNSUPPORTED;
  }
  const int gx = ag_cdiv(p.CK, NT), gy = ag_cdiv(p.A, AT);
  const int total = p.B * p.nchunk;
  int gz = wgrad_slices(p.A, p.CK, AT, NT, total);
  const int64_t n_out = (int64_t)p.A * p.CK;
  AG_REQUIRE(ws.p && ws.numel >= n_out, "%s: this reduction spans several workgroups and needs a workspace of >= %lld floats bound with ag_bind_workspace (sums are two-stage, in a fixed order; there is no float-atomic accumulation)", "ag_conv1d_wgrad", (long long)n_out);
  if ((int64_t)gz * n_out > ws.numel) gz = (int)(ws.numel / n_out);     // two-stage, fixed-order reduction
  p.part = ws.p;
  auto kern = conv_wgrad_kernel<TA, TN, WA, WN>;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
  hipLaunchKernelGGL(kern, dim3(gx, gy, gz), dim3(512), lds, st, p);
  AG_CHECK_LAUNCH("ag_conv1d_wgrad");
  return ag_slab_reduce(p.part, gz, n_out, p.dw, 1, st);
}

// Single-input-channel layer with a short kernel (D1: 1 -> 16 k7 s2): dw is 16 x 7 values, an MFMA tile would be
// > 95 % padding and a thousand workgroups would fight over 112 atomic addresses (100 us).  Plain reduction:
template <int KMAX, int AG>
__global__ __launch_bounds__(256) void conv_c1_wgrad_kernel(const float* __restrict__ sh, int64_t sh_bs, int64_t sh_cs,
                                                            const float* __restrict__ lg, int64_t lg_bs,
                                                            float* __restrict__ dw, int B, int A, int Lsh, int Llg,
                                                            int K, int s, int p, int bper, float* __restrict__ part,
                                                            int rb) {
  // a thread owns ONE time step (consecutive threads = consecutive steps: dy loads coalesce, x loads are s floats
  // apart) and AG output channels: the K-wide x window is loaded once per clip and reused by the AG channels
  __shared__ float red[4][AG * KMAX];
  const int a0 = blockIdx.y * AG;
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int b0 = blockIdx.z * bper, b1 = min(B, b0 + bper);
  float acc[AG][KMAX];
#pragma unroll
  for (int i = 0; i < AG; ++i)
#pragma unroll
    for (int k = 0; k < KMAX; ++k) acc[i][k] = 0.f;
  if (t < Lsh) {
    const int q0 = s * t - p;
    for (int b = b0; b < b1; ++b) {
      const float* xr = lg + (int64_t)b * lg_bs;
      float win[KMAX];
#pragma unroll
      for (int k = 0; k < KMAX; ++k) {
        const int q = q0 + k;
        const float xv = xr[min(max(q, 0), Llg - 1)];          // unconditional load, then select
        win[k] = (k < K && q >= 0 && q < Llg) ? ag_rbf_if(xv, rb) : 0.f;
      }
      const float* dyr = sh + (int64_t)b * sh_bs + t;
#pragma unroll
      for (int i = 0; i < AG; ++i) {
        const float g = ag_rbf_if(dyr[(int64_t)min(a0 + i, A - 1) * sh_cs], rb);
#pragma unroll
        for (int k = 0; k < KMAX; ++k) acc[i][k] += g * win[k];
      }
    }
  }
  // one barrier for all AG*KMAX sums: wave shuffles, per-wave partials in LDS, then one thread per value
#pragma unroll
  for (int i = 0; i < AG; ++i)
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      const float v = ag_wave_sum(acc[i][k]);
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][i * KMAX + k] = v;
    }
  __syncthreads();
  if (threadIdx.x < AG * KMAX) {
    const int i = threadIdx.x / KMAX, k = threadIdx.x % KMAX;
    if (k < K && a0 + i < A) {
      const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
      part[((int64_t)(blockIdx.z * gridDim.x + blockIdx.x) * A + a0 + i) * K + k] = v;
    }
  }
}

extern "C" int ag_conv1d_wgrad(const float* sh, int64_t sh_bs, int64_t sh_cs, const float* lg,
                               int64_t lg_bs, int64_t lg_cs, float* dw, int B, int A, int Lsh, int C,
                               int Llg, int K, int stride, int pad, void* stream) {
  const AgWs ws = ag_ws_take();     // FIRST: an argument error below must not leave a stale binding behind
  AG_REQUIRE(sh && lg && dw, "ag_conv1d_wgrad: null tensor");
  AG_REQUIRE(B > 0 && A > 0 && Lsh > 0 && C > 0 && Llg > 0 && K > 0 && stride > 0 && pad >= 0,
             "ag_conv1d_wgrad: bad shape");
  WgP p;
  p.sh = sh; p.lg = lg; p.dw = dw;
  p.sh_bs = sh_bs; p.sh_cs = sh_cs; p.lg_bs = lg_bs; p.lg_cs = lg_cs;
  p.B = B; p.A = A; p.Lsh = Lsh; p.C = C; p.Llg = Llg; p.K = K; p.s = stride; p.p = pad;
  p.CK = C * K;
  p.rb = ag_precision() == AG_PREC_BF16;
  p.vec = (((uintptr_t)sh & 15) == 0) && (sh_bs % 4 == 0) && (sh_cs % 4 == 0);
  hipStream_t st = (hipStream_t)stream;
  if (C == 1 && ws.p && ag_conv_c1_wgrad_slabs(B, A, Lsh, stride, K, nullptr) > 0 &&
      (((uintptr_t)sh & 15) == 0) && sh_bs % 4 == 0 && sh_cs % 4 == 0 &&
      ws.numel >= (int64_t)ag_conv_c1_wgrad_slabs(B, A, Lsh, stride, K, nullptr) * A * K &&
      (getenv("AG_CONV_C1") == nullptr || getenv("AG_CONV_C1")[0] != '0')) {
    // D1 / G1.conv: streaming kernel, one partial [A][K] per block, fixed-order second stage (conv_c1.hip)
    const int slabs = ag_conv_c1_wgrad_slabs(B, A, Lsh, stride, K, nullptr);
    const int rc = ag_conv_c1_wgrad(sh, sh_bs, sh_cs, lg, lg_bs, ws.p, B, A, Lsh, Llg, stride, K, pad, p.rb, st);
    if (rc != AG_OK) return rc;
    return ag_slab_reduce(ws.p, slabs, (int64_t)A * K, dw, 1, st);
  }
  if (C == 1 && K <= 8) {       // (K = 17, A = 128 - G1.conv - measured faster on the MFMA path: 43 vs 88 us)
    const int ag = 8;
    const int gx = ag_cdiv(Lsh, 256), gy = ag_cdiv(A, ag);
    int gz = ag_cdiv(512, gx * gy);     // ~2 workgroups per CU: each ends in AG*K partial sums
    if (gz > B) gz = B;
    if (gz < 1) gz = 1;
    AG_REQUIRE(ws.p && ws.numel >= (int64_t)gx * A * K, "%s: this reduction spans several workgroups and needs a workspace of >= %lld floats bound with ag_bind_workspace (sums are two-stage, in a fixed order; there is no float-atomic accumulation)", "ag_conv1d_wgrad", (long long)gx * A * K);
    if ((int64_t)gz * gx * A * K > ws.numel) gz = (int)(ws.numel / ((int64_t)gx * A * K));
    float* part = ws.p;
    const int bper = ag_cdiv(B, gz);
    gz = ag_cdiv(B, bper);
    hipLaunchKernelGGL((conv_c1_wgrad_kernel<8, 8>), dim3(gx, gy, gz), dim3(256), 0, st, sh, sh_bs, sh_cs, lg, lg_bs,
                       dw, B, A, Lsh, Llg, K, stride, pad, bper, part, p.rb);
    AG_CHECK_LAUNCH("ag_conv1d_wgrad");
    return ag_slab_reduce(part, gx * gz, (int64_t)A * K, dw, 1, st);
  }
  if (A <= 32) return launch_wgrad<1, 1, 1, 4>(p, st, ws);             // 32 x 128
  if (A <= 64 || p.CK <= 64) return launch_wgrad<1, 1, 2, 2>(p, st, ws);  // 64 x 64
  return launch_wgrad<2, 2, 2, 2>(p, st, ws);                           // 128 x 128
}

// floats of workspace ag_conv1d_wgrad wants bound (ag_bind_workspace) for its two-stage reduction
extern "C" int64_t ag_conv1d_wgrad_ws_numel(int B, int A, int Lsh, int C, int K) {
  if (C == 1 && K == 7) {
    // (the stride is not an argument here: the larger of what the streaming kernel of conv_c1.hip - stride 2 - and the
    // generic single-channel reduction need)
    const int64_t need = (int64_t)ag_conv_c1_wgrad_slabs(B, A, Lsh, 2, K, nullptr) * A * K;
    const int gx = ag_cdiv(Lsh, 256), gy = ag_cdiv(A, 8);
    int gz = ag_cdiv(512, gx * gy);
    if (gz > B) gz = B;
    if (gz < 1) gz = 1;
    const int64_t n2 = (int64_t)gz * gx * A * K;
    return n2 > need ? n2 : need;
  }
  if (C == 1 && K <= 8) {
    const int gx = ag_cdiv(Lsh, 256), gy = ag_cdiv(A, 8);
    int gz = ag_cdiv(512, gx * gy);
    if (gz > B) gz = B;
    if (gz < 1) gz = 1;
    return (int64_t)gz * gx * A * K;
  }
  const int CK = C * K;
  int AT, NT;
  if (A <= 32) { AT = 32; NT = 128; }
  else if (A <= 64 || CK <= 64) { AT = 64; NT = 64; }
  else { AT = 128; NT = 128; }
  int TC = AT >= 128 ? 32 : 64;
  if (Lsh < TC) TC = ag_roundup(Lsh, 4);
  const int total = B * ag_cdiv(Lsh, TC);
  return (int64_t)wgrad_slices(A, CK, AT, NT, total) * A * CK;
}

// ------------------------------------------------------------------------------------------
// db[c] (+)= sum_{b,t} dy[b,c,t]
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void channel_sum_kernel(const float* __restrict__ dy, int64_t bs,
                                                          int64_t cs, float* __restrict__ db, int B,
                                                          int C, int L, int nsplit, float* __restrict__ part, int accumulate) {
  __shared__ float red[17];
  const int c = blockIdx.x;
  const int64_t total = (int64_t)B * L;
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.y * 256 + threadIdx.x; i < total; i += (int64_t)nsplit * 256) {
    const int b = (int)(i / L);
    const int t = (int)(i - (int64_t)b * L);
    s += dy[(int64_t)b * bs + (int64_t)c * cs + t];
  }
  s = ag_block_sum(s, red);
  if (threadIdx.x == 0) {
    if (part) part[(int64_t)blockIdx.y * C + c] = s;
    else db[c] = accumulate ? db[c] + s : s;          // nsplit == 1: this workgroup is the only writer of db[c]
  }
}

extern "C" int ag_channel_sum(const float* dy, int64_t bs, int64_t cs, float* db, int B, int C, int L, int accumulate,
                              void* stream) {
  const AgWs ws = ag_ws_take();     // FIRST: an argument error below must not leave a stale binding behind
  AG_REQUIRE(dy && db && B > 0 && C > 0 && L > 0, "ag_channel_sum: bad args");
  int nsplit = (int)ag_cdiv64((int64_t)B * L, 256 * 16);
  const int cap = ag_cdiv(2048, C);
  if (nsplit > cap) nsplit = cap;
  if (nsplit < 1) nsplit = 1;
  float* part = nullptr;
  if (nsplit > 1) {
    AG_REQUIRE(ws.p && ws.numel >= 2 * (int64_t)C, "%s: this reduction spans several workgroups and needs a workspace of >= %lld floats bound with ag_bind_workspace (sums are two-stage, in a fixed order; there is no float-atomic accumulation)", "ag_channel_sum", (long long)2 * C);
    if ((int64_t)nsplit * C > ws.numel) nsplit = (int)(ws.numel / C);
    part = ws.p;
  }
  hipLaunchKernelGGL(channel_sum_kernel, dim3(C, nsplit), dim3(256), 0, (hipStream_t)stream, dy, bs,
                     cs, db, B, C, L, nsplit, part, accumulate);
  AG_CHECK_LAUNCH("ag_channel_sum");
  if (part) return ag_slab_reduce(part, nsplit, C, db, accumulate ? 1 : 0, (hipStream_t)stream);
  return AG_OK;
}

// ------------------------------------------------------------------------------------------
// dpre = dy * leaky'(y) * mask
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void leaky_bwd_kernel(const float* __restrict__ dy, int64_t dy_bs,
                                                        int64_t dy_cs, const float* __restrict__ y,
                                                        int64_t y_bs, int64_t y_cs,
                                                        float* __restrict__ dp, int64_t dp_bs,
                                                        int64_t dp_cs, float* __restrict__ ad,
                                                        int64_t ad_bs, int64_t ad_cs,
                                                        const int64_t* __restrict__ lens,
                                                        float* __restrict__ bias_grad, int B, int C,
                                                        int L, int bper, float slope, float* __restrict__ part) {
  const int c = blockIdx.y;
  float bsum = 0.f;
  // bper clips per workgroup: with a bias gradient every workgroup ends in ONE atomic, and all channels of a
  // small layer share a cache line - 8192 same-line atomics cost more than the whole pass
  for (int b = blockIdx.z * bper; b < min(B, (int)(blockIdx.z + 1) * bper); ++b) {
    const int64_t lenb = lens ? lens[b] : (int64_t)1 << 60;
    const float* dyr = dy + (int64_t)b * dy_bs + (int64_t)c * dy_cs;
    const float* yr = y + (int64_t)b * y_bs + (int64_t)c * y_cs;
    float* dpr = dp + (int64_t)b * dp_bs + (int64_t)c * dp_cs;
    float* adr = ad ? ad + (int64_t)b * ad_bs + (int64_t)c * ad_cs : nullptr;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < L; t += gridDim.x * 256) {
      float g = dyr[t];
      g = (yr[t] > 0.f) ? g : g * slope;
      if (t >= lenb) g = 0.f;
      dpr[t] = g;
      if (adr) adr[t] += g;
      bsum += g;
    }
  }
  if (bias_grad) {     // uniform branch
    __shared__ float red[4];
    bsum = ag_wave_sum(bsum);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = bsum;
    __syncthreads();
    if (threadIdx.x == 0) {
      const float v = red[0] + red[1] + red[2] + red[3];
      part[(int64_t)(blockIdx.z * gridDim.x + blockIdx.x) * C + c] = v;
    }
  }
}

extern "C" int ag_leaky_bwd(const float* dy, int64_t dy_bs, int64_t dy_cs, const float* y,
                            int64_t y_bs, int64_t y_cs, float* dpre, int64_t dp_bs, int64_t dp_cs,
                            float* add_into, int64_t ad_bs, int64_t ad_cs, const int64_t* lens_i64,
                            float* bias_grad, int B, int C, int L, float slope, void* stream) {
  const AgWs ws = ag_ws_take();     // FIRST: an argument error below must not leave a stale binding behind
  AG_REQUIRE(dy && y && dpre && B > 0 && C > 0 && L > 0, "ag_leaky_bwd: bad args");
  AG_REQUIRE(B <= 65535 && C <= 65535, "ag_leaky_bwd: B or C > 65535");
  int gx = ag_cdiv(L, 256 * 4);
  if (gx < 1) gx = 1;
  int bper = 1;
  float* part = nullptr;
  if (bias_grad) {
    bper = (int)(((int64_t)gx * C * B) / 2048);
    if (bper < 8) bper = 8;
    if (bper > B) bper = B;
    AG_REQUIRE(ws.p && ws.numel >= (int64_t)gx * C, "%s: this reduction spans several workgroups and needs a workspace of >= %lld floats bound with ag_bind_workspace (sums are two-stage, in a fixed order; there is no float-atomic accumulation)", "ag_leaky_bwd", (long long)gx * C);
    const int64_t zmax = ws.numel / ((int64_t)gx * C);      // partial sums per (x, z) block, fixed-order second stage
    if (ag_cdiv(B, bper) > zmax) bper = ag_cdiv(B, (int)zmax);
    part = ws.p;
  }
  const int gz = ag_cdiv(B, bper);
  hipLaunchKernelGGL(leaky_bwd_kernel, dim3(gx, C, gz), dim3(256), 0, (hipStream_t)stream, dy, dy_bs,
                     dy_cs, y, y_bs, y_cs, dpre, dp_bs, dp_cs, add_into, ad_bs, ad_cs, lens_i64, bias_grad, B, C, L,
                     bper, slope, part);
  AG_CHECK_LAUNCH("ag_leaky_bwd");
  if (part) return ag_slab_reduce(part, gx * gz, C, bias_grad, 1, (hipStream_t)stream);
  return AG_OK;
}

// ------------------------------------------------------------------------------------------
// Single-output-channel convolution, stride 1 (the Generator's final Conv1d 113 -> 1, k3,
// audiogan.py:404-407).  One output channel would waste 31/32 of an MFMA tile and the layer is
// HBM-bound anyway (reads C*L, writes L per clip), so it gets plain VALU kernels:
//   fwd   y[b,t]     = bias + sum_{c,k} w[c,k] x[b,c,t+k-p]         (+ activation)
//   bwdx  dx[b,c,t] (+)= sum_k w[c,k] dy[b,t-k+p]
//   wgrad dw[c,k]   += sum_{b,t} dy[b,t] x[b,c,t+k-p]
// ------------------------------------------------------------------------------------------
#define O1_MAXK 9

// A thread owns 4 consecutive time steps and reads each channel row as ONE aligned 16-byte piece; the K-1 halo values
// come from the neighbouring lanes (wave shuffles), only the first / last lane of a wave loads its halo from memory.
// UC channels are in flight per thread before the first FMA: the layer streams the 237 MB slab once and is bound by how
// many bytes are in flight, not by arithmetic.
#define O1_UC 8

template <int K>       // odd K, pad = (K-1)/2 <= 4
__global__ __launch_bounds__(256) void conv_o1_fwd_vec_kernel(const float* __restrict__ x, int64_t x_bs, int64_t x_cs,
                                                              const float* __restrict__ w,
                                                              const float* __restrict__ bias, float* __restrict__ y,
                                                              int64_t y_bs, int C, int L, int act, float slope, int rb) {
  constexpr int P = (K - 1) / 2;
  const int b = blockIdx.y, lane = threadIdx.x & 63;
  const int t0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  const bool live = t0 < L;                               // L % 4 == 0: a piece is entirely inside or outside
  const float* xb = x + (int64_t)b * x_bs + (live ? t0 : 0);
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int c0 = 0; c0 < C; c0 += O1_UC) {
    f32x4 v[O1_UC];
    float hl[O1_UC][P], hr[O1_UC][P];
#pragma unroll
    for (int u = 0; u < O1_UC; ++u) {
      const int c = min(c0 + u, C - 1);
      const float* xc = xb + (int64_t)c * x_cs;
      v[u] = *reinterpret_cast<const f32x4*>(xc);
      // halo of the wave's edge lanes straight from memory (clamped address, zeroed below when outside the clip)
#pragma unroll
      for (int i = 0; i < P; ++i) {
        hl[u][i] = (lane == 0 && live) ? xc[max(-(int)(i + 1), -t0)] : 0.f;
        hr[u][i] = (lane == 63 && live) ? xc[min(4 + i, L - 1 - t0)] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < O1_UC; ++u) {
      if (c0 + u >= C) break;
      float win[4 + 2 * P];
#pragma unroll
      for (int j = 0; j < 4; ++j) win[P + j] = ag_rbf_if(v[u][j], rb);
#pragma unroll
      for (int i = 0; i < P; ++i) {
        // left halo element i+1 positions before t0 = element 3-i of the previous lane; right: element i of the next
        const float l_ = __shfl_up(v[u][3 - i], 1, 64), r_ = __shfl_down(v[u][i], 1, 64);
        const float lv = (lane == 0) ? hl[u][i] : l_, rv = (lane == 63) ? hr[u][i] : r_;
        win[P - 1 - i] = (t0 - 1 - i >= 0) ? ag_rbf_if(lv, rb) : 0.f;
        win[P + 4 + i] = (t0 + 4 + i < L) ? ag_rbf_if(rv, rb) : 0.f;
      }
      const float* wc = w + (c0 + u) * K;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const float wk = ag_rbf_if(wc[k], rb);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += wk * win[j + k];
      }
    }
  }
  if (!live) return;
  const float bo = bias ? bias[0] : 0.f;
  f32x4 o = {ag_apply_act(acc[0] + bo, act, slope), ag_apply_act(acc[1] + bo, act, slope),
             ag_apply_act(acc[2] + bo, act, slope), ag_apply_act(acc[3] + bo, act, slope)};
  *reinterpret_cast<f32x4*>(y + (int64_t)b * y_bs + t0) = o;
}

__global__ __launch_bounds__(256) void conv_o1_fwd_kernel(const float* __restrict__ x, int64_t x_bs, int64_t x_cs,
                                                          const float* __restrict__ w, const float* __restrict__ bias,
                                                          float* __restrict__ y, int64_t y_bs, int C, int L, int K,
                                                          int p, int act, float slope, int rb) {
  const int b = blockIdx.y;
  const int t0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (t0 >= L) return;
  const float* xb = x + (int64_t)b * x_bs;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int c = 0; c < C; ++c) {
    const float* xc = xb + (int64_t)c * x_cs;
    float win[4 + O1_MAXK - 1];
#pragma unroll
    for (int i = 0; i < 4 + O1_MAXK - 1; ++i) {
      const int g = t0 + i - p;
      const float v = (i < 4 + K - 1) ? xc[min(max(g, 0), L - 1)] : 0.f;      // unconditional load, then select
      win[i] = (g >= 0 && g < L) ? ag_rbf_if(v, rb) : 0.f;
    }
#pragma unroll
    for (int k = 0; k < O1_MAXK; ++k) {
      if (k >= K) break;
      const float wk = ag_rbf_if(w[c * K + k], rb);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] += wk * win[j + k];
    }
  }
  const float bo = bias ? bias[0] : 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (t0 + j < L) y[(int64_t)b * y_bs + t0 + j] = ag_apply_act(acc[j] + bo, act, slope);
}

__global__ __launch_bounds__(256) void conv_o1_bwdx_kernel(const float* __restrict__ dy, int64_t dy_bs,
                                                           const float* __restrict__ w, float* __restrict__ dx,
                                                           int64_t dx_bs, int64_t dx_cs, int C, int L, int K, int p,
                                                           int accumulate, int rb) {
  const int b = blockIdx.z, c = blockIdx.y;
  const int t0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (t0 >= L) return;
  const float* dyb = dy + (int64_t)b * dy_bs;
  // dx[t] = sum_k w[c,k] dy[t - k + p]  -> window dy[t0 + p - (K-1) .. t0 + p + 3]
  float win[4 + O1_MAXK - 1];
#pragma unroll
  for (int i = 0; i < 4 + O1_MAXK - 1; ++i) {
    const int g = t0 + p - (K - 1) + i;
    const float v = (i < 4 + K - 1) ? dyb[min(max(g, 0), L - 1)] : 0.f;        // unconditional load, then select
    win[i] = (g >= 0 && g < L) ? ag_rbf_if(v, rb) : 0.f;
  }
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < O1_MAXK; ++k) {
    if (k >= K) break;
    const float wk = ag_rbf_if(w[c * K + k], rb);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] += wk * win[j + (K - 1) - k];
  }
  float* d = dx + (int64_t)b * dx_bs + (int64_t)c * dx_cs;
  if (t0 + 3 < L && (((uintptr_t)(d + t0)) & 15) == 0) {        // one 16-byte store (and load when accumulating)
    f32x4 o = {acc[0], acc[1], acc[2], acc[3]};
    if (accumulate) o += *reinterpret_cast<const f32x4*>(d + t0);
    *reinterpret_cast<f32x4*>(d + t0) = o;
    return;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (t0 + j < L) d[t0 + j] = accumulate ? d[t0 + j] + acc[j] : acc[j];
}

// one workgroup = one channel x 1024 time steps x a slice of the batch; a thread owns 4 consecutive steps
// (its dy values and the 4+K-1 wide x window stay in registers across the K taps)
__global__ __launch_bounds__(256) void conv_o1_wgrad_kernel(const float* __restrict__ dy, int64_t dy_bs,
                                                            const float* __restrict__ x, int64_t x_bs, int64_t x_cs,
                                                            float* __restrict__ dw, int B, int C, int L, int K, int p,
                                                            int bper, float* __restrict__ part, int rb) {
  __shared__ float red[17];
  const int c = blockIdx.y;
  const int t0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  const int b0 = blockIdx.z * bper;
  const int b1 = min(B, b0 + bper);
  float acc[O1_MAXK];
#pragma unroll
  for (int k = 0; k < O1_MAXK; ++k) acc[k] = 0.f;
  if (t0 < L) {
    for (int b = b0; b < b1; ++b) {
      const float* dyb = dy + (int64_t)b * dy_bs;
      const float* xc = x + (int64_t)b * x_bs + (int64_t)c * x_cs;
      float g[4], win[4 + O1_MAXK - 1];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float v = dyb[min(t0 + j, L - 1)];      // unconditional load, then select
        g[j] = (t0 + j < L) ? ag_rbf_if(v, rb) : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 4 + O1_MAXK - 1; ++i) {
        const int q = t0 + i - p;
        const float v = (i < 4 + K - 1) ? xc[min(max(q, 0), L - 1)] : 0.f;
        win[i] = (q >= 0 && q < L) ? ag_rbf_if(v, rb) : 0.f;
      }
#pragma unroll
      for (int k = 0; k < O1_MAXK; ++k) {
        if (k < K) {
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[k] += g[j] * win[j + k];
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < O1_MAXK; ++k) {
    if (k < K) {      // uniform
      const float s = ag_block_sum(acc[k], red);
      if (threadIdx.x == 0) {
        part[((int64_t)(blockIdx.z * gridDim.x + blockIdx.x) * C + c) * K + k] = s;
      }
    }
  }
}

extern "C" int ag_conv1d_o1_fwd(const float* x, int64_t x_bs, int64_t x_cs, const float* w, const float* bias,
                                float* y, int64_t y_bs, int B, int C, int L, int K, int pad, int act, float slope,
                                void* stream) {
  AG_REQUIRE(x && w && y && B > 0 && C > 0 && L > 0 && K > 0 && K <= O1_MAXK && pad >= 0 && B <= 65535,
             "ag_conv1d_o1_fwd: bad args (needs stride 1, K <= 9)");
  AG_REQUIRE(L + 2 * pad - K + 1 == L, "ag_conv1d_o1_fwd: needs a length-preserving ('same') conv");
  const int rb_ = (int)(ag_precision() == AG_PREC_BF16);
  // vector path: 16-byte aligned rows of x and y, L % 4 == 0 (a thread's 4 outputs are one aligned piece)
  if (K == 3 && L % 4 == 0 && L >= 4 && x_bs % 4 == 0 && x_cs % 4 == 0 && y_bs % 4 == 0 &&
      (((uintptr_t)x | (uintptr_t)y) & 15) == 0) {
    hipLaunchKernelGGL((conv_o1_fwd_vec_kernel<3>), dim3(ag_cdiv(L, 1024), B), dim3(256), 0, (hipStream_t)stream, x, x_bs,
                       x_cs, w, bias, y, y_bs, C, L, act, slope, rb_);
    AG_CHECK_LAUNCH("ag_conv1d_o1_fwd");
    return AG_OK;
  }
  hipLaunchKernelGGL(conv_o1_fwd_kernel, dim3(ag_cdiv(L, 1024), B), dim3(256), 0, (hipStream_t)stream, x, x_bs, x_cs,
                     w, bias, y, y_bs, C, L, K, pad, act, slope, (int)(ag_precision() == AG_PREC_BF16));
  AG_CHECK_LAUNCH("ag_conv1d_o1_fwd");
  return AG_OK;
}

// vector form for K = 3: a thread keeps the dy window of its 4 time steps in registers (one 16-byte piece + halo by
// wave shuffles) and walks CPB channels, one 16-byte store each - the layer is a 237 MB write and nothing else
#define O1_CPB 16
__global__ __launch_bounds__(256) void conv_o1_bwdx_vec3_kernel(const float* __restrict__ dy, int64_t dy_bs,
                                                                const float* __restrict__ w, float* __restrict__ dx,
                                                                int64_t dx_bs, int64_t dx_cs, int C, int L,
                                                                int accumulate, int rb) {
  const int b = blockIdx.z, lane = threadIdx.x & 63;
  const int t0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  const bool live = t0 < L;
  const float* dyr = dy + (int64_t)b * dy_bs + (live ? t0 : 0);
  const f32x4 g = *reinterpret_cast<const f32x4*>(dyr);
  const float hl = (lane == 0 && live) ? dyr[max(-1, -t0)] : 0.f;
  const float hr = (lane == 63 && live) ? dyr[min(4, L - 1 - t0)] : 0.f;
  const float l_ = __shfl_up(g[3], 1, 64), r_ = __shfl_down(g[0], 1, 64);
  float win[6];
  win[0] = (t0 - 1 >= 0) ? ag_rbf_if(lane == 0 ? hl : l_, rb) : 0.f;
  win[5] = (t0 + 4 < L) ? ag_rbf_if(lane == 63 ? hr : r_, rb) : 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) win[1 + j] = ag_rbf_if(g[j], rb);
  if (!live) return;
  // dx[c, t] = sum_k w[c, k] dy[t - k + 1]  (pad 1):  k = 0 -> dy[t+1], k = 1 -> dy[t], k = 2 -> dy[t-1]
  const int c0 = blockIdx.y * O1_CPB, c1 = min(C, c0 + O1_CPB);
  float* d = dx + (int64_t)b * dx_bs + (int64_t)c0 * dx_cs + t0;
  for (int c = c0; c < c1; ++c, d += dx_cs) {
    const float w0 = ag_rbf_if(w[c * 3], rb), w1 = ag_rbf_if(w[c * 3 + 1], rb), w2 = ag_rbf_if(w[c * 3 + 2], rb);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = w0 * win[j + 2] + w1 * win[j + 1] + w2 * win[j];
    if (accumulate) o += *reinterpret_cast<const f32x4*>(d);
    *reinterpret_cast<f32x4*>(d) = o;
  }
}

extern "C" int ag_conv1d_o1_bwd_data(const float* dy, int64_t dy_bs, const float* w, float* dx, int64_t dx_bs,
                                     int64_t dx_cs, int B, int C, int L, int K, int pad, int accumulate,
                                     void* stream) {
  AG_REQUIRE(dy && w && dx && B > 0 && C > 0 && L > 0 && K > 0 && K <= O1_MAXK && B <= 65535 && C <= 65535,
             "ag_conv1d_o1_bwd_data: bad args");
  if (K == 3 && pad == 1 && L % 4 == 0 && L >= 4 && dy_bs % 4 == 0 && dx_bs % 4 == 0 && dx_cs % 4 == 0 &&
      (((uintptr_t)dy | (uintptr_t)dx) & 15) == 0) {
    hipLaunchKernelGGL(conv_o1_bwdx_vec3_kernel, dim3(ag_cdiv(L, 1024), ag_cdiv(C, O1_CPB), B), dim3(256), 0,
                       (hipStream_t)stream, dy, dy_bs, w, dx, dx_bs, dx_cs, C, L, accumulate,
                       (int)(ag_precision() == AG_PREC_BF16));
    AG_CHECK_LAUNCH("ag_conv1d_o1_bwd_data");
    return AG_OK;
  }
  hipLaunchKernelGGL(conv_o1_bwdx_kernel, dim3(ag_cdiv(L, 1024), C, B), dim3(256), 0, (hipStream_t)stream, dy, dy_bs,
                     w, dx, dx_bs, dx_cs, C, L, K, pad, accumulate, (int)(ag_precision() == AG_PREC_BF16));
  AG_CHECK_LAUNCH("ag_conv1d_o1_bwd_data");
  return AG_OK;
}

// vector form of conv_o1_wgrad_kernel for K = 3: 16-byte pieces of dy and of the channel row, halo by wave shuffles,
// 4 clips in flight per thread
__global__ __launch_bounds__(256) void conv_o1_wgrad_vec3_kernel(const float* __restrict__ dy, int64_t dy_bs,
                                                                 const float* __restrict__ x, int64_t x_bs, int64_t x_cs,
                                                                 float* __restrict__ dw, int B, int C, int L, int bper,
                                                                 float* __restrict__ part, int rb) {
  __shared__ float red[17];
  const int c = blockIdx.y, lane = threadIdx.x & 63;
  const int t0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  const bool live = t0 < L;
  const int b0 = blockIdx.z * bper, b1 = min(B, b0 + bper);
  const float* xc0 = x + (int64_t)c * x_cs + (live ? t0 : 0);
  const float* dy0 = dy + (live ? t0 : 0);
  float acc[3] = {0.f, 0.f, 0.f};
  for (int bb = b0; bb < b1; bb += 4) {
    f32x4 xv[4], gv[4];
    float hl[4], hr[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int b = min(bb + u, b1 - 1);
      const float* xr = xc0 + (int64_t)b * x_bs;
      xv[u] = *reinterpret_cast<const f32x4*>(xr);
      gv[u] = *reinterpret_cast<const f32x4*>(dy0 + (int64_t)b * dy_bs);
      hl[u] = (lane == 0 && live) ? xr[max(-1, -t0)] : 0.f;
      hr[u] = (lane == 63 && live) ? xr[min(4, L - 1 - t0)] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (bb + u >= b1) break;
      const float l_ = __shfl_up(xv[u][3], 1, 64), r_ = __shfl_down(xv[u][0], 1, 64);
      float win[6];
      win[0] = (t0 - 1 >= 0) ? ag_rbf_if(lane == 0 ? hl[u] : l_, rb) : 0.f;
      win[5] = (t0 + 4 < L) ? ag_rbf_if(lane == 63 ? hr[u] : r_, rb) : 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) win[1 + j] = ag_rbf_if(xv[u][j], rb);
      if (live) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float g = ag_rbf_if(gv[u][j], rb);
#pragma unroll
          for (int k = 0; k < 3; ++k) acc[k] += g * win[j + k];
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float s = ag_block_sum(acc[k], red);
    if (threadIdx.x == 0) {
      part[((int64_t)(blockIdx.z * gridDim.x + blockIdx.x) * C + c) * 3 + k] = s;
    }
  }
}

extern "C" int ag_conv1d_o1_wgrad(const float* dy, int64_t dy_bs, const float* x, int64_t x_bs, int64_t x_cs,
                                  float* dw, int B, int C, int L, int K, int pad, void* stream) {
  const AgWs ws = ag_ws_take();     // FIRST: an argument error below must not leave a stale binding behind
  AG_REQUIRE(dy && x && dw && B > 0 && C > 0 && L > 0 && K > 0 && K <= O1_MAXK, "ag_conv1d_o1_wgrad: bad args");
  AG_REQUIRE(C <= 65535, "ag_conv1d_o1_wgrad: C > 65535");
  // enough workgroups to fill the chip (~8 per CU), the rest of the batch is looped inside
  const int gx = ag_cdiv(L, 1024);
  int gz = ag_cdiv(2048, gx * C);
  if (gz > B) gz = B;
  if (gz < 1) gz = 1;
  AG_REQUIRE(ws.p && ws.numel >= (int64_t)gx * C * K, "%s: this reduction spans several workgroups and needs a workspace of >= %lld floats bound with ag_bind_workspace (sums are two-stage, in a fixed order; there is no float-atomic accumulation)", "ag_conv1d_o1_wgrad", (long long)gx * C * K);
  if ((int64_t)gz * gx * C * K > ws.numel) gz = (int)(ws.numel / ((int64_t)gx * C * K));
  float* part = ws.p;
  const int bper = ag_cdiv(B, gz);
  gz = ag_cdiv(B, bper);
  if (K == 3 && pad == 1 && L % 4 == 0 && L >= 4 && x_bs % 4 == 0 && x_cs % 4 == 0 && dy_bs % 4 == 0 &&
      (((uintptr_t)x | (uintptr_t)dy) & 15) == 0) {
    hipLaunchKernelGGL(conv_o1_wgrad_vec3_kernel, dim3(gx, C, gz), dim3(256), 0, (hipStream_t)stream, dy, dy_bs, x, x_bs,
                       x_cs, dw, B, C, L, bper, part, (int)(ag_precision() == AG_PREC_BF16));
    AG_CHECK_LAUNCH("ag_conv1d_o1_wgrad");
    return ag_slab_reduce(part, gx * gz, (int64_t)C * K, dw, 1, (hipStream_t)stream);
  }
  hipLaunchKernelGGL(conv_o1_wgrad_kernel, dim3(gx, C, gz), dim3(256), 0, (hipStream_t)stream, dy, dy_bs, x, x_bs,
                     x_cs, dw, B, C, L, K, pad, bper, part, (int)(ag_precision() == AG_PREC_BF16));
  AG_CHECK_LAUNCH("ag_conv1d_o1_wgrad");
  return ag_slab_reduce(part, gx * gz, (int64_t)C * K, dw, 1, (hipStream_t)stream);
}
