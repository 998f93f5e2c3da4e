// gemm_bf16_tile.h -- shared definitions of the bf16-operand GEMM (ag_gemm_h: GemmH, the LDS image swizzles) and an
// EXPERIMENT: the same product on tiles larger than 128 x 128.  Included by gemm_bf16s.hip (definitions only: it instantiates no
// tile kernel) and by tools/gemm_lab_h.hip (the bench).  Result (round 4): built, parity-checked in the lab against float64,
// measured, NOT dispatched - through ag_gemm_h over the critic's 11 shapes the 256 x 128 form took 613 us against 605 us of the
// 128 x 128 kernel (tools/prof_gemm_h.py): +7 % on [16384 x 1024 x 1024], -13 % where it leaves one workgroup per CU (N = 512).
//
// v_mfma_f32_32x32x16_bf16 retires 16x the flops of the fp32 MFMA per operand byte, so what bounds the loop is LDS bandwidth:
// a wave tile of TI x TJ blocks reads (TI + TJ) 1-KiB operand fragments per TI * TJ MFMAs of 32 cycles.  At 64 x 64 per wave
// (2 x 2) four waves per CU need 128 B / cycle - all the LDS has - to keep the matrix pipes busy; 64 x 128 needs 96, 128 x 128
// (4 x 4, accumulators in 256 registers, one wave per SIMD) 64.  Same images and swizzles as the 128 x 128 kernel
// (gemm_bf16s.hip): KC [rows][64 k] for k-contiguous operands, KS [64 k][rows] read with ds_read_b64_tr_b16 for k-strided ones.
#pragma once
#include <type_traits>
#include <utility>

#include "common.h"

typedef short hbf16x8 __attribute__((ext_vector_type(8)));
typedef short hs16x4 __attribute__((ext_vector_type(4)));

#define H_LDS_AS(p) ((__attribute__((address_space(3))) void*)(p))
#define H_GLB_AS(p) ((const __attribute__((address_space(1))) void*)(p))

struct GemmH {
  const unsigned short* A;      // TA = 0: [M][K] (lda)   TA = 1: [K][M] (lda)
  const unsigned short* B;      // TB = 1: [N][K] (ldb)   TB = 0: [K][N] (ldb)
  float* C;                     // fp32 output (ldc) or NULL
  unsigned short* C16;          // bf16 output (ldc16) or NULL
  const float* bias;            // [N] fp32
  const float* res;             // fp32 residual / gate source (ldres) or NULL
  const unsigned short* res16;  // bf16 residual / gate source (ldres16) or NULL
  const unsigned short* gate16; // bf16 SAVED OUTPUT of a LeakyReLU (ldgate16) or NULL: the result (after bias / res) is scaled by
                                // that activation's derivative - a residual layer's backward (W^T da + da) gated by the layer below
  float* part;                  // split-K slabs
  int lda, ldb, ldc, ldc16, ldres, ldres16, ldgate16;
  int M, N, K, ksplit, kchunk, act;
  float alpha, beta, slope;
};

// KC image, rows of 2 * BK bytes: BK = 64: chunk c of row r in slot c ^ ((r >> 1) & 7);  BK = 32 (64-byte rows): slot c ^ ((r >> 2) & 3)
template <int BK>
__device__ __forceinline__ int h_kc_slot(int r, int c) {
  return BK == 64 ? r * 128 + ((c ^ ((r >> 1) & 7)) << 4) : r * 64 + ((c ^ ((r >> 2) & 3)) << 4);
}
__device__ __forceinline__ int h_ks_f(int k) { return ((k & 3) << 2) | ((k >> 2) & 3); }
__device__ __forceinline__ float h_bf(unsigned short v) { return __uint_as_float((unsigned)v << 16); }


template <int... I, class F>
__device__ __forceinline__ void ag_static_for_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>()), ...);
}
template <int N, class F>
__device__ __forceinline__ void ag_static_for(F&& f) {
  ag_static_for_impl(std::make_integer_sequence<int, N>(), f);
}

// BM x BN: workgroup tile; TI x TJ: 32 x 32 blocks per wave (TI >= 2); NBUF: LDS stages; WPE: waves per SIMD the registers
// are budgeted for.  Dynamic LDS: max(NBUF x (BM + BN) x 128 B, the epilogue's [64][BN + 4] fp32 image).
template <int TA, int TB, int BM, int BN, int TI, int TJ, int NBUF, int WPE>
__global__ __launch_bounds__((BM / (32 * TI)) * (BN / (32 * TJ)) * 64, WPE) void gemm_bf16t_kernel(const GemmH p) {
  constexpr int WJ = BN / (32 * TJ), NW = (BM / (32 * TI)) * WJ, NT = NW * 64;
  constexpr int GA = BM / 8, GB = BN / 8, IPS = (GA + GB) / NW;       // 1-KiB DMA instructions: per operand image, per wave
  static_assert((GA + GB) % NW == 0 && TI >= 2, "tile shape");
  constexpr int AIMG = BM * 128, STAGE = (BM + BN) * 128;             // bytes
  constexpr bool AKS = TA == 1, BKS = TB == 0;
  extern __shared__ __attribute__((aligned(16))) char ht_sm[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int wr = wid / WJ, wm0 = wr * 32 * TI, wn0 = (wid % WJ) * 32 * TJ;
  int bx, by;
  {   // XCD-aware tile order: each XCD gets a band of consecutive row tiles (tiles sharing an A panel share an L2)
    const int gx = gridDim.x, gy = gridDim.y, lin = blockIdx.y * gx + blockIdx.x, nwg = gx * gy;
    const int q = nwg / 8, r = nwg % 8, xcd = lin & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lin >> 3);
    by = wg / gx;
    bx = wg - by * gx;
  }
  const int bz = blockIdx.z;
  const int m0 = by * BM, n0 = bx * BN;

  // this wave's DMA instructions: the first GA / NW fill the A image, the others the B image
  static_assert(GA % NW == 0 && GB % NW == 0, "each operand image must divide over the waves");
  constexpr int IA = GA / NW;
  const unsigned short* src[IPS];
  int dsto[IPS];
  const int64_t ksa = AKS ? (int64_t)p.lda : 1, ksb = BKS ? (int64_t)p.ldb : 1;     // elements per unit of k
#pragma unroll
  for (int it = 0; it < IPS; ++it) {
    const bool isA = it < IA;
    const int gg = wid + NW * (isA ? it : it - IA);
    const unsigned short* X = isA ? p.A : p.B;
    const int ld = isA ? p.lda : p.ldb, x0 = isA ? m0 : n0, R = isA ? p.M : p.N, BR = isA ? BM : BN;
    const bool ks = isA ? AKS : BKS;
    if (!ks) {      // KC: rows 8 gg .. 8 gg + 7, eight 16-byte chunks each
      const int r = 8 * gg + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
      src[it] = X + (int64_t)min(x0 + r, R - 1) * ld + 8 * c;
    } else {        // KS: the gg-th KiB of the [64 k][BR rows] image
      const int j = gg * 64 + lane, cpr = BR / 8, k = j / cpr, ch = (j - k * cpr) ^ h_ks_f(k);
      src[it] = X + (int64_t)k * ld + min(x0 + 8 * ch, R - 8);
    }
    dsto[it] = (isA ? 0 : AIMG) + gg * 1024;
  }
  auto stage = [&](int k0, int buf) {
    char* S = ht_sm + buf * STAGE;
#pragma unroll
    for (int it = 0; it < IPS; ++it)
      __builtin_amdgcn_global_load_lds(H_GLB_AS(src[it] + (int64_t)k0 * (it < IA ? ksa : ksb)), H_LDS_AS(S + dsto[it]), 16, 0, 0);
  };

  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // transposed-read lane roles (KS operands): lane 4q + p of a 16-lane group addresses block row q, columns 4p .. 4p+3
  const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1;
  auto load_frag = [&](const char* img, bool ks, int BR, int row0, int s_) -> hbf16x8 {
    if (!ks) return *reinterpret_cast<const hbf16x8*>(img + h_kc_slot<64>(row0 + l31, 2 * s_ + h));
    hs16x4 v4[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int row = 16 * s_ + 8 * h + 4 * t + tq, col = row0 + 16 * tg + 4 * tp;
      const int off = 2 * BR * row + 16 * ((col >> 3) ^ ((tq << 2) | ((2 * h + t) & 3))) + 8 * ((col >> 2) & 1);
      v4[t] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) hs16x4*)(img + off));
    }
    return hbf16x8{v4[0][0], v4[0][1], v4[0][2], v4[0][3], v4[1][0], v4[1][1], v4[1][2], v4[1][3]};
  };
  auto mfma_stage = [&](const char* As, const char* Bs) {
    // fragments of k-step s + 1 are requested before the MFMAs of step s (one wave per SIMD has nobody else to hide them)
    hbf16x8 av[2][TI], bv[2][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i) av[0][i] = load_frag(As, AKS, BM, wm0 + 32 * i, 0);
#pragma unroll
    for (int j = 0; j < TJ; ++j) bv[0][j] = load_frag(Bs, BKS, BN, wn0 + 32 * j, 0);
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) {
      const int c = s_ & 1, n = c ^ 1;
      if (s_ < 3) {
#pragma unroll
        for (int i = 0; i < TI; ++i) av[n][i] = load_frag(As, AKS, BM, wm0 + 32 * i, s_ + 1);
#pragma unroll
        for (int j = 0; j < TJ; ++j) bv[n][j] = load_frag(Bs, BKS, BN, wn0 + 32 * j, s_ + 1);
      }
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[c][i], bv[c][j], acc[i][j], 0, 0, 0);
    }
  };

  const int kbeg = bz * p.kchunk;
  const int kend = (kbeg + p.kchunk < p.K) ? kbeg + p.kchunk : p.K;
  if (NBUF == 2) {
    stage(kbeg, 0);
    int buf = 0;
    for (int k0 = kbeg; k0 < kend; k0 += 64) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (k0 + 64 < kend) stage(k0 + 64, buf ^ 1);
      mfma_stage(ht_sm + buf * STAGE, ht_sm + buf * STAGE + AIMG);
      __builtin_amdgcn_sched_barrier(0);
      buf ^= 1;
    }
    __syncthreads();          // (the epilogue reuses the stage memory)
  } else {
    for (int k0 = kbeg; k0 < kend; k0 += 64) {
      stage(k0, 0);
      __syncthreads();                    // vmcnt(0) + barrier: the tile is in LDS
      mfma_stage(ht_sm, ht_sm + AIMG);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();                    // every wave is done reading before the next tile overwrites it
    }
  }

  // ---- epilogue (as gemm_bf16s_kernel): full aligned tiles go through an LDS image, 64 rows at a time, so that every lane
  // stores 4 consecutive columns; bias / residual / gate are read the same way
  const bool fast = p.ksplit == 1 && m0 + BM <= p.M && n0 + BN <= p.N &&
                    (!p.C || ((p.ldc & 3) == 0 && ((uintptr_t)p.C & 15) == 0)) &&
                    (!p.C16 || ((p.ldc16 & 3) == 0 && ((uintptr_t)p.C16 & 7) == 0)) &&
                    (!p.res || ((p.ldres & 3) == 0 && ((uintptr_t)p.res & 15) == 0)) &&
                    (!p.res16 || ((p.ldres16 & 3) == 0 && ((uintptr_t)p.res16 & 7) == 0)) &&
                    (!p.gate16 || ((p.ldgate16 & 3) == 0 && ((uintptr_t)p.gate16 & 7) == 0)) &&
                    (!p.bias || ((uintptr_t)p.bias & 15) == 0);
  if (fast) {
    typedef unsigned u32x2e __attribute__((ext_vector_type(2)));
    constexpr int CP = BN + 4;                                  // row pitch (floats): rows 4 apart on different banks
    float* Ct = reinterpret_cast<float*>(ht_sm);                // [64][CP] fp32
    ag_static_for<BM / 64>([&](auto psc) {
      constexpr int ps = decltype(psc)::value;
      constexpr int i0 = ((64 * ps) % (32 * TI)) / 32;
      if (wr == (64 * ps) / (32 * TI)) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
          for (int e = 0; e < 16; ++e)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
              Ct[(32 * ii + (e & 3) + 8 * (e >> 2) + 4 * h) * CP + wn0 + 32 * j + l31] = acc[i0 + ii][j][e];
      }
      __syncthreads();
#pragma unroll
      for (int it = 0; it < 64 * (BN / 4) / NT; ++it) {
        const int idx = tid + NT * it;
        const int r = idx / (BN / 4), c4 = (idx - r * (BN / 4)) * 4;
        const int64_t row = m0 + 64 * ps + r;
        const int col = n0 + c4;
        f32x4 v = *reinterpret_cast<const f32x4*>(Ct + r * CP + c4);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] *= p.alpha;
        if (p.beta != 0.f) {
          const f32x4 o = *reinterpret_cast<const f32x4*>(p.C + row * p.ldc + col);
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] += p.beta * o[q];
        }
        if (p.bias) {
          const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + col);
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] += b[q];
        }
        f32x4 rr = {0.f, 0.f, 0.f, 0.f};
        const bool hr = p.res != nullptr || p.res16 != nullptr;
        if (p.res) rr = *reinterpret_cast<const f32x4*>(p.res + row * p.ldres + col);
        if (p.res16) {
          const u32x2e w = *reinterpret_cast<const u32x2e*>(p.res16 + row * p.ldres16 + col);
          rr = f32x4{__uint_as_float(w[0] << 16), __uint_as_float(w[0] & 0xFFFF0000u), __uint_as_float(w[1] << 16),
                     __uint_as_float(w[1] & 0xFFFF0000u)};
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = ag_res_act(v[q], hr, rr[q], p.act, p.slope);
        if (p.gate16) {
          const u32x2e w = *reinterpret_cast<const u32x2e*>(p.gate16 + row * p.ldgate16 + col);
          const float g0 = __uint_as_float(w[0] << 16), g1 = __uint_as_float(w[0] & 0xFFFF0000u),
                      g2 = __uint_as_float(w[1] << 16), g3 = __uint_as_float(w[1] & 0xFFFF0000u);
          if (!(g0 > 0.f)) v[0] *= p.slope;
          if (!(g1 > 0.f)) v[1] *= p.slope;
          if (!(g2 > 0.f)) v[2] *= p.slope;
          if (!(g3 > 0.f)) v[3] *= p.slope;
        }
        if (p.C) *reinterpret_cast<f32x4*>(p.C + row * p.ldc + col) = v;
        if (p.C16) *reinterpret_cast<u32x2e*>(p.C16 + row * p.ldc16 + col) = u32x2e{ag_pack_bf16(v[0], v[1]), ag_pack_bf16(v[2], v[3])};
      }
      __syncthreads();
    });
    return;
  }

  ag_static_for<TI>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = m0 + wm0 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
      if (row >= p.M) continue;
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        const int col = n0 + wn0 + 32 * j + l31;
        if (col >= p.N) continue;
        float v = p.alpha * acc[i][j][e];
        if (p.ksplit > 1) {      // K slice: a partial tile into its slab (fixed-order second stage)
          p.part[((int64_t)bz * p.M + row) * p.N + col] = v;
          continue;
        }
        if (p.beta != 0.f) v += p.beta * p.C[(int64_t)row * p.ldc + col];
        if (p.bias) v += p.bias[col];
        const bool hr = p.res != nullptr || p.res16 != nullptr;
        const float r = p.res ? p.res[(int64_t)row * p.ldres + col]
                              : (p.res16 ? h_bf(p.res16[(int64_t)row * p.ldres16 + col]) : 0.f);
        v = ag_res_act(v, hr, r, p.act, p.slope);
        if (p.gate16 && !(h_bf(p.gate16[(int64_t)row * p.ldgate16 + col]) > 0.f)) v *= p.slope;
        if (p.C) p.C[(int64_t)row * p.ldc + col] = v;
        if (p.C16) p.C16[(int64_t)row * p.ldc16 + col] = (unsigned short)(ag_pack_bf16(v, v) & 0xFFFFu);
      }
    }
  });
}

// tile shapes: index -> (BM, BN, TI, TJ, NBUF, WPE).  Measured on [16384 x 1024] x [1024 x 1024] (tools/gemm_lab_h.hip,
// fp32 / bf16 output; the 128 x 128 kernel of gemm_bf16s.hip: 617-644 TF):
//   1: 256 x 128, 8 waves of 64 x 64,   one stage  (48 KiB, 2 workgroups per CU)                    684 / 736 TF
//   2: 256 x 256, 8 waves of 64 x 128,  two stages (128 KiB, 1 per CU)                              587 / 617
//   3: 256 x 256, 4 waves of 128 x 128, two stages (128 KiB, 1 per CU, one wave per SIMD; spills)   419 / 435
//   4: 256 x 256, 8 waves of 64 x 128,  one stage  (67 KiB, 1 per CU)                               560 / 580
// With K = 512 .. 1024 a tile is 8-16 stages long: its prologue (first DMA round trip) and its epilogue (128 KiB of stores)
// weigh as much as its MFMAs (14 us at peak for 34 GFLOP), and what helps is a second workgroup on the CU to run under
// them - not the lower LDS traffic of a larger wave tile, which is what decided the fp32 kernel.
#ifndef AG_GEMMH_TILE_CASES
#define AG_GEMMH_TILE_CASES(F) F(1, 256, 128, 2, 2, 1, 4)
#endif
#define AG_GEMMH_TILE_CASES_ALL(F) F(1, 256, 128, 2, 2, 1, 4) F(2, 256, 256, 2, 4, 2, 2) F(3, 256, 256, 4, 4, 2, 1) F(4, 256, 256, 2, 4, 1, 2)

template <int TA, int TB, int BM, int BN, int TI, int TJ, int NBUF, int WPE>
static inline int gemm_bf16t_launch_one(const GemmH& p, hipStream_t st) {
  constexpr int NT = (BM / (32 * TI)) * (BN / (32 * TJ)) * 64;
  constexpr int stage = NBUF * (BM + BN) * 128, epi = 64 * (BN + 4) * 4;
  constexpr int lds = stage > epi ? stage : epi;
  auto k = gemm_bf16t_kernel<TA, TB, BM, BN, TI, TJ, NBUF, WPE>;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return AG_ERR_LAUNCH;
    attr = true;
  }
  dim3 grid(ag_cdiv(p.N, BN), ag_cdiv(p.M, BM), p.ksplit);
  hipLaunchKernelGGL(k, grid, dim3(NT), lds, st, p);
  return AG_OK;
}

template <int BM, int BN, int TI, int TJ, int NBUF, int WPE>
static inline int gemm_bf16t_launch_layout(const GemmH& p, int ta, int tb, hipStream_t st) {
  if (ta == 0 && tb == 0) return gemm_bf16t_launch_one<0, 0, BM, BN, TI, TJ, NBUF, WPE>(p, st);
  if (ta == 0 && tb == 1) return gemm_bf16t_launch_one<0, 1, BM, BN, TI, TJ, NBUF, WPE>(p, st);
  if (ta == 1 && tb == 0) return gemm_bf16t_launch_one<1, 0, BM, BN, TI, TJ, NBUF, WPE>(p, st);
  return gemm_bf16t_launch_one<1, 1, BM, BN, TI, TJ, NBUF, WPE>(p, st);
}

static inline int gemm_bf16t_launch(const GemmH& p, int ta, int tb, int shape, hipStream_t st) {
#define AG_HT_CASE(I, BM_, BN_, TI_, TJ_, NB_, WPE_) \
  if (shape == I) return gemm_bf16t_launch_layout<BM_, BN_, TI_, TJ_, NB_, WPE_>(p, ta, tb, st);
  AG_GEMMH_TILE_CASES(AG_HT_CASE)
#undef AG_HT_CASE
  return AG_ERR_ARG;
}

static inline void gemm_bf16t_dims(int shape, int& bm, int& bn) {
  bm = 256;
  bn = shape == 1 ? 128 : 256;
}
