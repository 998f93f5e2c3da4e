// gemm_tile.h -- the fp32 GEMM of the large products: BM x BN output tiles fed by LDS-DMA, v_mfma_f32_32x32x2_f32.
// Included by gemm.hip (the library) and by tools/gemm_lab.hip (the bench the tile shapes were chosen on).
//
// Why tiles larger than 128 x 128 (round 4, tools/gemm_lab.hip on [16384 x 1024] = [16384 x 1024] x [1024 x 1024]^T):
//   matrix pipes alone (MFMA from registers, 2 waves per SIMD)     148 TF   (the practical ceiling; 157.3 is 2.4 GHz x 256 CUs)
//   128 x 128 tile, MFMA + barriers only (no LDS reads, no DMA)     147 TF
//   128 x 128 tile, MFMA + LDS reads + barriers (no DMA)            127 TF
//   128 x 128 tile, complete                                        120 TF   (4 workgroups per CU)
//   256 x 128 tile, 8 waves                                         133 TF   (2 workgroups per CU)
//   256 x 256 tile, 8 waves of 64 x 128                             137 TF   (1 workgroup per CU)
// The loop is bound by the bytes that go L2 -> LDS -> registers per MFMA, not by MFMA issue: a 256 x 256 tile moves half the
// bytes per flop of a 128 x 128 one (32 -> 64 flop per byte of LDS traffic), and that buys more than the occupancy it costs.
// More LDS stages (3, 4) or fewer waves with larger register tiles (4 waves of 128 x 128) changed nothing at equal tile size.
//
//   * a DMA instruction of a wave moves 1 KiB: global (lane-dependent address, 16 B per lane) -> LDS (wave-uniform base +
//     lane * 16 B).  The LDS image of a stage is therefore in lane order:
//       k-contiguous operand  (A [M][K], B [N][K]):  [rows][16 k]  64-byte rows; the 16-byte piece kq of row r sits in slot
//                                                    kq ^ ((r >> 2) & 3) (applied to the SOURCE address and by the reader):
//                                                    one ds_read_b128 per 8 k and row, conflict free
//       row-contiguous operand (A [K][M], B [K][N]): [16 k][rows]  read with ds_read_b32, unit stride over the lanes
//   * the MFMA group of 8 k uses them as k = 8q + 4h + e (h = lane >> 5, e = 0..3) for BOTH operands
//   * two stages; per iteration: wait for the stage, barrier, request the next stage, multiply.  hipcc puts no vmcnt wait
//     between a DMA and a later ds_read (checked in the ISA), the explicit s_waitcnt + s_barrier at the top is what orders them
// Needs K % 16 == 0 (per K slice), 16-byte aligned rows, row counts % 4 == 0 for row-contiguous operands.
#pragma once
#include <type_traits>

#include "common.h"

struct GemmP {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  const float* res;
  int lda, ldb, ldc, ldres;
  int M, N, K;
  float alpha, beta, slope;
  int act;
  int vecA, vecB;  // 16-B vector loads allowed for A / B
  int rb;          // AG_PREC_BF16: operands are rounded to bf16 on the way into LDS (fp32 MFMA on rounded values)
  int ksplit;      // > 1: grid.z K slices; partial tiles go to `part` (two-stage, fixed-order reduction)
  int kchunk;      // K per slice (multiple of GBK)
  float* part;     // [ksplit][M][N] partial products (alpha applied), or NULL
};

#define AG_LDS_AS(p) ((__attribute__((address_space(3))) void*)(p))
#define AG_GLB_AS(p) ((const __attribute__((address_space(1))) void*)(p))

// XCD-aware placement.  Workgroup ids go round-robin over the 8 XCDs, each with its own L2.  In plain order the column
// tiles of one row band land on 8 different XCDs and every XCD pulls ALL of A through the fabric; with K split over
// grid.z every XCD additionally pulls every K slice of B.  Remapped: with 8 | ksplit one XCD works on whole K slices (its
// L2 fetches that slice of A and of B once); with ksplit in {1, 2, 4} a slice is shared by 8 / ksplit XCDs, each taking a
// contiguous band of the slice's tiles.  Speed only: the result does not depend on placement (slices are indexed by bz).
__device__ __forceinline__ void xcd_place(int& bx, int& by, int& bz) {
  const int gx = gridDim.x, nwg = gx * gridDim.y, ks = gridDim.z;
  const int lin = (blockIdx.z * gridDim.y + blockIdx.y) * gx + blockIdx.x;
  const int x = lin & 7, q = lin >> 3;
  bx = blockIdx.x; by = blockIdx.y; bz = blockIdx.z;
  int tile;
  if ((ks & 7) == 0) {
    bz = x + 8 * (q / nwg);
    tile = q % nwg;
  } else if (ks <= 4 && (8 % ks) == 0 && nwg % (8 / ks) == 0) {
    bz = x % ks;
    tile = (x / ks) * (nwg / (8 / ks)) + q;
  } else {
    return;
  }
  by = tile / gx;
  bx = tile - by * gx;
}

// TA / TB: 0 = A stored [M][K] / B stored [K][N], 1 = A stored [K][M] / B stored [N][K]   (as ag_gemm's ta / tb)
// BM x BN: workgroup tile; TI x TJ: blocks of 32 x 32 per wave.  Dynamic LDS: 2 stages x (BM + BN) x 16 floats.
// DBG: ablations of tools/gemm_lab.hip (LAB_DBG) - 1 = the lab's 1-D tile placement, 2 = plain-store epilogue; the library
// instantiates DBG = 0 only.
template <int TA, int TB, int BM, int BN, int TI, int TJ, int DBG = 0>
__global__ __launch_bounds__((BM / (32 * TI)) * (BN / (32 * TJ)) * 64, (BM * BN >= 65536 ? 2 : 4))      // (threads, waves per SIMD)
void gemm_tile_kernel(const GemmP p) {
  constexpr int WJ = BN / (32 * TJ), NW = (BM / (32 * TI)) * WJ;
  constexpr int GA = BM / 16, GB = BN / 16, IPS = (GA + GB) / NW;     // 1-KiB DMA instructions: per operand, per wave
  static_assert((GA + GB) % NW == 0, "the DMA instructions of a stage must divide over the waves");
  constexpr int STAGE = (BM + BN) * 16;                              // floats (A image, then B image)
  constexpr bool AK = TA == 0, BK = TB == 1;                          // operand is k-contiguous
  extern __shared__ __attribute__((aligned(16))) float gt_sm[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int wm0 = (wid / WJ) * 32 * TI, wn0 = (wid % WJ) * 32 * TJ;
  int bx, by, bz;
  if (DBG & 1) {
    const int nb = gridDim.x * gridDim.y;
    int id = blockIdx.y * gridDim.x + blockIdx.x;
    id = (id & 7) * (nb >> 3) + (id >> 3);
    by = id / gridDim.x; bx = id - by * gridDim.x; bz = 0;
  } else {
    xcd_place(bx, by, bz);
  }
  const int m0 = by * BM, n0 = bx * BN;

  // this wave's DMA instructions: source at k = 0 of the operand, k step, wave-uniform LDS offset in the stage
  const float* src[IPS];
  int kst[IPS], dsto[IPS];
#pragma unroll
  for (int it = 0; it < IPS; ++it) {
    const int g = wid + NW * it;
    const bool isA = g < GA;
    const int gg = isA ? g : g - GA;
    const float* X = isA ? p.A : p.B;
    const int ld = isA ? p.lda : p.ldb, x0 = isA ? m0 : n0, R = isA ? p.M : p.N, BR = isA ? BM : BN;
    const bool kc = isA ? AK : BK;
    if (kc) {       // rows 16 gg .. 16 gg + 15, four 16-byte pieces each
      const int r = 16 * gg + (lane >> 2), sl = lane & 3;
      src[it] = X + (int64_t)min(x0 + r, R - 1) * ld + 4 * (sl ^ ((r >> 2) & 3));
      kst[it] = 1;
    } else {        // the gg-th KiB of the [16 k][BR rows] image
      const int o = gg * 256 + lane * 4, k = o / BR, r = o - k * BR;
      src[it] = X + (int64_t)k * ld + min(x0 + r, R - 4);
      kst[it] = ld;
    }
    dsto[it] = (isA ? 0 : BM * 16) + gg * 256;
  }
  auto stage = [&](int k0, int buf) {
    float* S = gt_sm + buf * STAGE;
#pragma unroll
    for (int it = 0; it < IPS; ++it)
      __builtin_amdgcn_global_load_lds(AG_GLB_AS(src[it] + (int64_t)k0 * kst[it]), AG_LDS_AS(S + dsto[it]), 16, 0, 0);
  };

  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int kbeg = bz * p.kchunk;
  const int kend = (kbeg + p.kchunk < p.K) ? kbeg + p.kchunk : p.K;
  stage(kbeg, 0);
  int buf = 0;
  for (int k0 = kbeg; k0 < kend; k0 += 16) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (k0 + 16 < kend) stage(k0 + 16, buf ^ 1);
    const float* As = gt_sm + buf * STAGE;
    const float* Bs = As + BM * 16;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      f32x4 av[TI], bv[TJ];
#pragma unroll
      for (int i = 0; i < TI; ++i) {
        const int m = wm0 + 32 * i + l31;
        if (AK) {
          av[i] = *reinterpret_cast<const f32x4*>(As + m * 16 + 4 * ((2 * q + h) ^ ((m >> 2) & 3)));
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) av[i][e] = As[(8 * q + 4 * h + e) * BM + m];
        }
      }
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        const int n = wn0 + 32 * j + l31;
        if (BK) {
          bv[j] = *reinterpret_cast<const f32x4*>(Bs + n * 16 + 4 * ((2 * q + h) ^ ((n >> 2) & 3)));
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) bv[j][e] = Bs[(8 * q + 4 * h + e) * BN + n];
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][e], bv[j][e], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    buf ^= 1;
  }

  if (DBG & 2) {
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm0 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
        float* dst = p.C + (int64_t)row * p.ldc + n0 + wn0 + l31;
#pragma unroll
        for (int j = 0; j < TJ; ++j) dst[32 * j] = acc[i][j][e];
      }
    return;
  }
  if (p.ksplit > 1 && m0 + BM <= p.M && n0 + BN <= p.N) {
    // interior tile of a K slice: the partial tile goes to its slab as it is (alpha applied), no tests per element
    float* slab = p.part + (int64_t)bz * p.M * p.N;
    const float alpha = p.alpha;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm0 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
        float* dst = slab + (int64_t)row * p.N + n0 + wn0 + l31;
#pragma unroll
        for (int j = 0; j < TJ; ++j) dst[32 * j] = alpha * acc[i][j][e];
      }
    return;
  }
  if (p.ksplit == 1 && m0 + BM <= p.M && n0 + BN <= p.N) {
    // interior tile: no bounds tests.  The options are uniform: the reads of `res` / of C (beta) are issued EG rows at a time
    // (all in flight together) under one branch each, the activation is chosen once around the arithmetic + stores - an
    // epilogue that tested the options per element cost 7 % of a 256 x 256 tile's time (tools/gemm_lab.hip, LAB_DBG).
    const bool hb = p.bias != nullptr, hr = p.res != nullptr, hbeta = p.beta != 0.f;
    constexpr int EG = 4;                        // rows per group: EG x TJ values of res and of C in registers
    float bj[TJ];
#pragma unroll
    for (int j = 0; j < TJ; ++j) bj[j] = hb ? p.bias[n0 + wn0 + 32 * j + l31] : 0.f;
    const float alpha = p.alpha, beta = p.beta, slope = p.slope;
    const int act = p.act;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int e0 = 0; e0 < 16; e0 += EG) {
        float rv[EG][TJ], cv[EG][TJ];
        float* dst[EG];
#pragma unroll
        for (int u = 0; u < EG; ++u) {
          const int e = e0 + u, row = m0 + wm0 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
          dst[u] = p.C + (int64_t)row * p.ldc + n0 + wn0 + l31;
#pragma unroll
          for (int j = 0; j < TJ; ++j) rv[u][j] = cv[u][j] = 0.f;
        }
        if (hr) {
#pragma unroll
          for (int u = 0; u < EG; ++u) {
            const int e = e0 + u, row = m0 + wm0 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
            const float* rs = p.res + (int64_t)row * p.ldres + n0 + wn0 + l31;
#pragma unroll
            for (int j = 0; j < TJ; ++j) rv[u][j] = rs[32 * j];
          }
        }
        if (hbeta) {
#pragma unroll
          for (int u = 0; u < EG; ++u)
#pragma unroll
            for (int j = 0; j < TJ; ++j) cv[u][j] = dst[u][32 * j];
        }
        auto finish = [&](auto actc) {
          constexpr int ACT = decltype(actc)::value;
#pragma unroll
          for (int u = 0; u < EG; ++u)
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
              float v = alpha * acc[i][j][e0 + u] + bj[j];
              v = __builtin_fmaf(beta, cv[u][j], v);          // (beta == 0: cv is 0, not C)
              if (ACT == AG_ACT_LEAKY_GATE) {
                v = rv[u][j] > 0.f ? v : v * slope;
              } else {
                v += rv[u][j];
                if (ACT == AG_ACT_LEAKY) v = v > 0.f ? v : v * slope;
                if (ACT == AG_ACT_TANH) v = tanhf(v);
              }
              dst[u][32 * j] = v;
            }
        };
        if (act == AG_ACT_NONE) finish(std::integral_constant<int, AG_ACT_NONE>());
        else if (act == AG_ACT_LEAKY) finish(std::integral_constant<int, AG_ACT_LEAKY>());
        else if (act == AG_ACT_LEAKY_GATE) finish(std::integral_constant<int, AG_ACT_LEAKY_GATE>());
        else finish(std::integral_constant<int, AG_ACT_TANH>());
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = m0 + wm0 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
      if (row >= p.M) continue;
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        const int col = n0 + wn0 + 32 * j + l31;
        if (col >= p.N) continue;
        float v = p.alpha * acc[i][j][e];
        float* dst = p.C + (int64_t)row * p.ldc + col;
        if (p.ksplit > 1) {      // K slice: a partial tile into its slab (the host never splits K without one)
          p.part[((int64_t)bz * p.M + row) * p.N + col] = v;
          continue;
        }
        if (p.beta != 0.f) v += p.beta * *dst;
        if (p.bias) v += p.bias[col];
        *dst = ag_res_act(v, p.res != nullptr, p.res ? p.res[(int64_t)row * p.ldres + col] : 0.f, p.act, p.slope);
      }
    }
}

// tile shapes built into the library: index -> (BM, BN, TI, TJ)
//   0: 128 x 128, 4 waves of 64 x 64   32 KiB LDS, 4 workgroups per CU
//   1: 256 x 128, 8 waves of 64 x 64   48 KiB, 2 per CU
//   2: 128 x 256, 8 waves of 64 x 64   48 KiB, 2 per CU
//   3: 256 x 256, 8 waves of 64 x 128  64 KiB, 1 per CU
#define AG_GEMM_TILE_CASES(F) F(0, 128, 128, 2, 2) F(1, 256, 128, 2, 2) F(2, 128, 256, 2, 2) F(3, 256, 256, 2, 4)

template <int TA, int TB, int BM, int BN, int TI, int TJ>
static inline int gemm_tile_launch_one(const GemmP& p, hipStream_t st) {
  constexpr int NT = (BM / (32 * TI)) * (BN / (32 * TJ)) * 64;
  constexpr int lds = 2 * (BM + BN) * 16 * 4;
  auto k = gemm_tile_kernel<TA, TB, BM, BN, TI, TJ>;
  static bool attr = false;        // (> 64 KiB of dynamic LDS needs the attribute; set once per instantiation)
  if (!attr) {
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return AG_ERR_LAUNCH;
    attr = true;
  }
  dim3 grid(ag_cdiv(p.N, BN), ag_cdiv(p.M, BM), p.ksplit);
  hipLaunchKernelGGL(k, grid, dim3(NT), lds, st, p);
  return AG_OK;
}

template <int BM, int BN, int TI, int TJ>
static inline int gemm_tile_launch_layout(const GemmP& p, int ta, int tb, hipStream_t st) {
  if (ta == 0 && tb == 0) return gemm_tile_launch_one<0, 0, BM, BN, TI, TJ>(p, st);
  if (ta == 0 && tb == 1) return gemm_tile_launch_one<0, 1, BM, BN, TI, TJ>(p, st);
  if (ta == 1 && tb == 0) return gemm_tile_launch_one<1, 0, BM, BN, TI, TJ>(p, st);
  return gemm_tile_launch_one<1, 1, BM, BN, TI, TJ>(p, st);
}

static inline int gemm_tile_launch(const GemmP& p, int ta, int tb, int shape, hipStream_t st) {
#define AG_GT_CASE(I, BM_, BN_, TI_, TJ_) \
  if (shape == I) return gemm_tile_launch_layout<BM_, BN_, TI_, TJ_>(p, ta, tb, st);
  AG_GEMM_TILE_CASES(AG_GT_CASE)
#undef AG_GT_CASE
  return AG_ERR_ARG;
}

static inline void gemm_tile_dims(int shape, int& bm, int& bn) {
  bm = (shape == 1 || shape == 3) ? 256 : 128;
  bn = (shape == 2 || shape == 3) ? 256 : 128;
}
