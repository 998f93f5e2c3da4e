// gemm_bf16s.hip -- GEMM on operands that are STORED as bfloat16 (BASELINE configs[2]; round 4).
//
// ag_gemm in AG_PREC_BF16 mode reads fp32 operands and rounds them while it stages them: 4 bytes per element cross the
// fabric and a convert sits in front of every LDS write (190-250 TFLOP/s, bound by the bytes in flight).  Here the
// operands already ARE bfloat16 in HBM - activations are written as bf16 by the kernel that produces them, weights get a
// bf16 image when they are materialised - so a tile is half the bytes and goes global -> LDS by LDS-DMA
// (global_load_lds_dwordx4) without touching a VGPR.
//
//   C[M,N] (fp32 and / or bf16) = act(alpha * op(A) op(B) + beta * C + bias + res)
//
// 128 x 128 tile, BK = 64, 4 waves x (2 x 2) v_mfma_f32_32x32x16_bf16, two LDS buffers of 32 KiB, the DMA of tile i+1 in
// flight under the MFMAs of tile i.  Two operand images, chosen per operand by how it is stored:
//   KC  k contiguous ([rows][K]: activations x weights^T):  image [128 rows][64 k], 128-byte rows; 16-byte chunk c of row
//       r sits in slot c ^ ((r >> 1) & 7); one ds_read_b128 per MFMA operand (the image of gemm_bf16_kernel).
//   KS  k strided ([K][rows]: both operands of a weight gradient dW = dY^T X, the weight of a data gradient dX = dY W):
//       image [64 k][128 rows], 256-byte rows; chunk ch of row k sits in slot ch ^ (((k & 3) << 2) | ((k >> 2) & 3)); an
//       MFMA operand (8 consecutive k of one row) is two ds_read_b64_tr_b16 - the hardware transpose read of gfx950 (a
//       16-lane group reads a 4 k x 16 rows block and each lane receives one row's 4 k).
// LDS-DMA writes LDS in lane order, so both swizzles are applied to the SOURCE address of a DMA lane and to the reader.
// Needs K (and a K slice) % 64 == 0, 16-byte aligned rows, and row counts % 8 == 0 for KS operands.
#include "common.h"
// (GemmH, the image swizzles - and the larger-tile kernel that tools/gemm_lab_h.hip measured and that did NOT beat this one
// on the critic's shapes: see the table in the header)
#define AG_GEMMH_TILE_CASES(F)
#include "gemm_bf16_tile.h"

// BK: k per stage (64 or 32);  NBUF: LDS stages (2: the DMA of tile i+1 under the MFMAs of tile i;  1: one stage, 32 KiB (BK
// 64) per workgroup - latency is hidden by co-resident workgroups instead, four per CU)
template <int TA, int TB, int BK, int NBUF, int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void gemm_bf16s_kernel(const GemmH p) {
  constexpr int BM = 128, BN = 128, IMG = 128 * BK * 2;            // bytes per operand image (either form)
  constexpr int NCH = IMG / 16 / 256;                               // 16-byte chunks per thread, operand and stage
  constexpr int KCC = BK / 8;                                       // chunks per row of the KC image
  constexpr bool AKS = TA == 1, BKS = TB == 0;
  extern __shared__ __attribute__((aligned(16))) char sm[];        // NBUF x (A image + B image)
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int wm0 = (wid >> 1) * 64, wn0 = (wid & 1) * 64;
  // XCD-aware tile order: workgroup ids go round-robin over the 8 XCDs; give each XCD a band of consecutive row tiles so
  // that the tiles sharing an A panel share an L2
  int bx, by;
  {
    const int gx = gridDim.x, gy = gridDim.y, lin = blockIdx.y * gx + blockIdx.x, nwg = gx * gy;
    const int q = nwg / 8, r = nwg % 8, xcd = lin & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lin >> 3);
    by = wg / gx;
    bx = wg - by * gx;
  }
  const int bz = blockIdx.z;
  const int m0 = by * BM, n0 = bx * BN;

  // per-lane DMA sources at k = 0
  const unsigned short* asrc[NCH];
  const unsigned short* bsrc[NCH];
#pragma unroll
  for (int it = 0; it < NCH; ++it) {
    const int j = tid + 256 * it;
    if (!AKS) {
      const int r = j / KCC, c = (j % KCC) ^ (BK == 64 ? ((r >> 1) & 7) : ((r >> 2) & 3));
      asrc[it] = p.A + (int64_t)min(m0 + r, p.M - 1) * p.lda + 8 * c;
    } else {
      const int k = j >> 4, ch = (j & 15) ^ h_ks_f(k);
      asrc[it] = p.A + (int64_t)k * p.lda + min(m0 + 8 * ch, p.M - 8);
    }
    if (!BKS) {
      const int r = j / KCC, c = (j % KCC) ^ (BK == 64 ? ((r >> 1) & 7) : ((r >> 2) & 3));
      bsrc[it] = p.B + (int64_t)min(n0 + r, p.N - 1) * p.ldb + 8 * c;
    } else {
      const int k = j >> 4, ch = (j & 15) ^ h_ks_f(k);
      bsrc[it] = p.B + (int64_t)k * p.ldb + min(n0 + 8 * ch, p.N - 8);
    }
  }
  const int64_t astep = AKS ? (int64_t)p.lda : 1, bstep = BKS ? (int64_t)p.ldb : 1;     // elements per unit of k

  auto stage = [&](int k0, int buf) {
    char* As = sm + buf * 2 * IMG;
    char* Bs = As + IMG;
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
      const int wbase = (wid * 64 + 256 * it) * 16;          // wave-uniform LDS byte offset of this instruction
      __builtin_amdgcn_global_load_lds(H_GLB_AS(asrc[it] + (int64_t)k0 * astep), H_LDS_AS(As + wbase), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(H_GLB_AS(bsrc[it] + (int64_t)k0 * bstep), H_LDS_AS(Bs + wbase), 16, 0, 0);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // transposed-read lane roles (KS operands): lane 4q + p of a 16-lane group addresses block row q, columns 4p .. 4p+3
  const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1;

  auto mfma_tile = [&](const char* As, const char* Bs) {
#pragma unroll
    for (int s_ = 0; s_ < BK / 16; ++s_) {
      hbf16x8 av[2], bv[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (!AKS) {
          av[i] = *reinterpret_cast<const hbf16x8*>(As + h_kc_slot<BK>(wm0 + 32 * i + l31, 2 * s_ + h));
        } else {
          hs16x4 v4[2];
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const int row = 16 * s_ + 8 * h + 4 * t + tq, col = wm0 + 32 * i + 16 * tg + 4 * tp;
            const int off = 256 * row + 16 * ((col >> 3) ^ ((tq << 2) | ((2 * h + t) & 3))) + 8 * ((col >> 2) & 1);
            v4[t] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) hs16x4*)(As + off));
          }
          av[i] = hbf16x8{v4[0][0], v4[0][1], v4[0][2], v4[0][3], v4[1][0], v4[1][1], v4[1][2], v4[1][3]};
        }
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (!BKS) {
          bv[j] = *reinterpret_cast<const hbf16x8*>(Bs + h_kc_slot<BK>(wn0 + 32 * j + l31, 2 * s_ + h));
        } else {
          hs16x4 v4[2];
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const int row = 16 * s_ + 8 * h + 4 * t + tq, col = wn0 + 32 * j + 16 * tg + 4 * tp;
            const int off = 256 * row + 16 * ((col >> 3) ^ ((tq << 2) | ((2 * h + t) & 3))) + 8 * ((col >> 2) & 1);
            v4[t] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) hs16x4*)(Bs + off));
          }
          bv[j] = hbf16x8{v4[0][0], v4[0][1], v4[0][2], v4[0][3], v4[1][0], v4[1][1], v4[1][2], v4[1][3]};
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
  };

  const int kbeg = bz * p.kchunk;
  const int kend = (kbeg + p.kchunk < p.K) ? kbeg + p.kchunk : p.K;
  if (NBUF == 2) {
    stage(kbeg, 0);
    __syncthreads();
    int buf = 0;
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
      if (k0 + BK < kend) stage(k0 + BK, buf ^ 1);
      mfma_tile(sm + buf * 2 * IMG, sm + buf * 2 * IMG + IMG);
      // (keeps the barrier - and its vmcnt(0) on the next tile's DMA - behind ALL of this tile's MFMAs)
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
      buf ^= 1;
    }
  } else {
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
      stage(k0, 0);
      __syncthreads();                    // vmcnt(0) + barrier: the tile is in LDS
      mfma_tile(sm, sm + IMG);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();                    // every wave is done reading before the next tile overwrites it
    }
  }

  // ---- epilogue.  Full tiles with 16-byte aligned outputs go through LDS: the accumulators (column per lane) are written as a
  // [64 rows][128] fp32 image, 64 rows at a time, and read back row-wise, so that every lane stores 4 consecutive columns -
  // 256 contiguous bytes of bf16 (512 of fp32) per 32 lanes instead of 2-byte (4-byte) pieces; bias / residual / gate are
  // read the same way.  (A 16384 x 2048 x 512 projection writes 4x the bytes it reads: its store pattern IS its speed.)
  const bool fast = p.ksplit == 1 && m0 + BM <= p.M && n0 + BN <= p.N &&
                    (!p.C || ((p.ldc & 3) == 0 && ((uintptr_t)p.C & 15) == 0)) &&
                    (!p.C16 || ((p.ldc16 & 3) == 0 && ((uintptr_t)p.C16 & 7) == 0)) &&
                    (!p.res || ((p.ldres & 3) == 0 && ((uintptr_t)p.res & 15) == 0)) &&
                    (!p.res16 || ((p.ldres16 & 3) == 0 && ((uintptr_t)p.res16 & 7) == 0)) &&
                    (!p.gate16 || ((p.ldgate16 & 3) == 0 && ((uintptr_t)p.gate16 & 7) == 0)) &&
                    (!p.bias || ((uintptr_t)p.bias & 15) == 0);
  if (fast) {
    typedef unsigned u32x2e __attribute__((ext_vector_type(2)));
    float* Ct = reinterpret_cast<float*>(sm);                 // [64][128] fp32 = 32 KiB
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if ((wid >> 1) == half) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int e = 0; e < 16; ++e)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              Ct[(32 * i + (e & 3) + 8 * (e >> 2) + 4 * h) * 128 + wn0 + 32 * j + l31] = acc[i][j][e];
      }
      __syncthreads();
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int idx = tid + 256 * it;
        const int r = idx >> 5, c4 = (idx & 31) * 4;
        const int64_t row = m0 + 64 * half + r;
        const int col = n0 + c4;
        f32x4 v = *reinterpret_cast<const f32x4*>(Ct + r * 128 + c4);
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
    }
    return;
  }

  // C layout of a 32x32 tile: col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * h
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = m0 + wm0 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
      if (row >= p.M) continue;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
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
}

// variant of the staging structure (AG_GEMMH_VARIANT, for A/B runs): 1 (default) = BK 64, ONE stage (32 KiB) and registers
// capped at 128: four workgroups per CU hide each other's DMA latency;  0 = BK 64, two stages (64 KiB: 2 workgroups per CU,
// the DMA of tile i+1 under the MFMAs of tile i - 16 MFMAs of 32 cycles do not cover an HBM round trip);  2 = BK 32, two
// stages (32 KiB), 4 per CU;  3 / 4 = as 1 / 2 with the registers uncapped (3 per CU).
// Measured over the critic's 11 product shapes (tools/prof_gemm_h.py, sum of the launch times): 0: 830 us, 1: 686, 2: 764,
// 3: 807, 4: 859 (the fp32-operand bf16 kernel: 885).
static const int g_h_variant = [] { const char* e = getenv("AG_GEMMH_VARIANT"); return e ? atoi(e) : 1; }();

template <int TA, int TB, int BK, int NBUF, int WPE>
static void launch_h2(const GemmH& p, dim3 grid, hipStream_t st) {
  auto kern = gemm_bf16s_kernel<TA, TB, BK, NBUF, WPE>;
  const int stage_bytes = NBUF * 2 * 128 * BK * 2;
  const int lds = stage_bytes < 32 * 1024 ? 32 * 1024 : stage_bytes;       // (the epilogue's [64][128] fp32 image)
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, p);
}

template <int TA, int TB>
static void launch_h(const GemmH& p, dim3 grid, hipStream_t st) {
  if (g_h_variant == 1) launch_h2<TA, TB, 64, 1, 4>(p, grid, st);
  else if (g_h_variant == 2) launch_h2<TA, TB, 32, 2, 4>(p, grid, st);
  else if (g_h_variant == 3) launch_h2<TA, TB, 64, 1, 3>(p, grid, st);
  else if (g_h_variant == 4) launch_h2<TA, TB, 32, 2, 3>(p, grid, st);
  else launch_h2<TA, TB, 64, 2, 2>(p, grid, st);
}

static int h_pick_ksplit(int64_t tiles, int64_t mn, int K) {
  static const int cand[] = {32, 24, 16, 8, 4, 2};
  const int64_t want = (512 * 4 / tiles + 2) / 3;
  for (int c : cand)
    if (c <= want && c <= K / 256 && (int64_t)c * mn <= ((int64_t)12 << 20)) return c;
  return 1;
}

extern "C" int ag_gemm_h_ok(int M, int N, int K, int ta, int tb, int lda, int ldb) {
  if (M <= 0 || N <= 0 || K < 64 || K % 64 != 0) return 0;
  if (lda % 8 != 0 || ldb % 8 != 0) return 0;
  if (ta && (M % 8 != 0 || M < 8)) return 0;
  if (!tb && (N % 8 != 0 || N < 8)) return 0;
  return 1;
}

// floats of workspace ag_gemm_h wants bound (ag_bind_workspace) so that a long reduction with few output tiles is split
// over K and summed in two stages (0: no split)
extern "C" int64_t ag_gemm_h_ws_numel(int M, int N, int K, int act, int has_c16) {
  const int64_t tiles = (int64_t)ag_cdiv(M, 128) * ag_cdiv(N, 128);
  if (!(tiles < 192 && K >= 1024 && act == AG_ACT_NONE && !has_c16)) return 0;
  const int ks = h_pick_ksplit(tiles, (int64_t)M * N, K);
  return ks >= 2 ? (int64_t)ks * M * N : 0;
}

extern "C" int ag_gemm_h(const uint16_t* A, int lda, int ta, const uint16_t* B, int ldb, int tb, float* C, int ldc,
                         uint16_t* C16, int ldc16, int M, int N, int K, float alpha, float beta, const float* bias,
                         const float* res, int ldres, const uint16_t* res16, int ldres16, const uint16_t* gate16,
                         int ldgate16, int act, float slope, void* stream) {
  const AgWs ws = ag_ws_take();     // FIRST: an argument error below must not leave a stale binding behind
  AG_REQUIRE(A && B && (C || C16), "ag_gemm_h: null tensor");
  AG_REQUIRE((ta == 0 || ta == 1) && (tb == 0 || tb == 1), "ag_gemm_h: bad transpose flag");
  AG_REQUIRE(ag_gemm_h_ok(M, N, K, ta, tb, lda, ldb),
             "ag_gemm_h: needs K %% 64 == 0, leading dimensions %% 8 == 0 and row counts %% 8 == 0 for k-strided operands "
             "(M %d N %d K %d ta %d tb %d lda %d ldb %d)", M, N, K, ta, tb, lda, ldb);
  AG_REQUIRE((((uintptr_t)A | (uintptr_t)B) & 15) == 0, "ag_gemm_h: operands must be 16-byte aligned");
  AG_REQUIRE(lda >= (ta ? M : K) && ldb >= (tb ? K : N) && (!C || ldc >= N) && (!C16 || ldc16 >= N), "ag_gemm_h: bad leading dim");
  AG_REQUIRE(beta == 0.f || C, "ag_gemm_h: beta needs the fp32 output");
  AG_REQUIRE(!(res && res16), "ag_gemm_h: one residual / gate source");
  AG_REQUIRE(act != AG_ACT_LEAKY_GATE || res || res16, "ag_gemm_h: AG_ACT_LEAKY_GATE needs the saved activation");
  AG_REQUIRE(ag_cdiv(M, 128) <= 65535, "ag_gemm_h: M too large");
  GemmH p;
  p.A = A; p.B = B; p.C = C; p.C16 = C16; p.bias = bias; p.res = res; p.res16 = res16; p.gate16 = gate16; p.part = nullptr;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldc16 = ldc16; p.ldres = ldres; p.ldres16 = ldres16; p.ldgate16 = ldgate16;
  p.M = M; p.N = N; p.K = K; p.act = act; p.alpha = alpha; p.beta = beta; p.slope = slope;
  p.ksplit = 1;
  p.kchunk = K;
  hipStream_t st = (hipStream_t)stream;
  const int64_t tiles = (int64_t)ag_cdiv(M, 128) * ag_cdiv(N, 128), mn = (int64_t)M * N;
  if (tiles < 192 && K >= 1024 && act == AG_ACT_NONE && !C16 && !gate16 && ws.p && ws.numel >= 2 * mn) {
    int ks = h_pick_ksplit(tiles, mn, K);
    if ((int64_t)ks * mn > ws.numel) ks = (int)(ws.numel / mn);
    if (ks >= 2) {
      p.kchunk = ag_roundup(ag_cdiv(K, ks), 64);
      p.ksplit = ag_cdiv(K, p.kchunk);
      if (p.ksplit > 1) p.part = ws.p;
      else p.kchunk = K;
    }
  }
  dim3 grid(ag_cdiv(N, 128), ag_cdiv(M, 128), p.ksplit);
  if (ta == 0 && tb == 0) launch_h<0, 0>(p, grid, st);
  if (ta == 0 && tb == 1) launch_h<0, 1>(p, grid, st);
  if (ta == 1 && tb == 0) launch_h<1, 0>(p, grid, st);
  if (ta == 1 && tb == 1) launch_h<1, 1>(p, grid, st);
  AG_CHECK_LAUNCH("ag_gemm_h");
  if (!p.part) return AG_OK;
  if (ag_reduces_deferred() && !bias && !res && !res16 && (beta == 0.f || beta == 1.f))
    return ag_slab_defer_2d(p.part, p.ksplit, M, N, C, ldc, beta == 1.f ? 1 : 0, st);
  AG_REQUIRE(!res16, "ag_gemm_h: a split-K product takes its residual in fp32");
  return ag_splitk_reduce(p.part, p.ksplit, mn, M, N, C, ldc, beta, bias, res, ldres, st);
}

// ------------------------------------------------------------------------------------------
// fp32 -> bf16 images (round to nearest even): what a producer that still writes fp32 hands to ag_gemm_h
//   ag_to_bf16_2d: dst[r, c] = bf16(src[r, c]) for a [rows, cols] block with row pitches (a column block of a weight)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void to_bf16_2d_kernel(const float* __restrict__ src, int64_t lds_, unsigned short* __restrict__ dst,
                                                         int64_t ldd, int rows, int cols) {
  const int r = blockIdx.y;
  const float* s = src + (int64_t)r * lds_;
  unsigned short* d = dst + (int64_t)r * ldd;
  const bool vec = (cols & 3) == 0 && (((uintptr_t)s & 15) == 0) && (((uintptr_t)d & 7) == 0);
  if (vec) {
    for (int c = (blockIdx.x * 256 + threadIdx.x) * 4; c < cols; c += gridDim.x * 1024) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(s + c);
      typedef unsigned u32x2h __attribute__((ext_vector_type(2)));
      *reinterpret_cast<u32x2h*>(d + c) = u32x2h{ag_pack_bf16(v[0], v[1]), ag_pack_bf16(v[2], v[3])};
    }
  } else {
    for (int c = blockIdx.x * 256 + threadIdx.x; c < cols; c += gridDim.x * 256)
      d[c] = (unsigned short)(ag_pack_bf16(s[c], s[c]) & 0xFFFFu);
  }
}

extern "C" int ag_to_bf16_2d(const float* src, int64_t ld_src, uint16_t* dst, int64_t ld_dst, int rows, int cols, void* stream) {
  AG_REQUIRE(src && dst && rows > 0 && cols > 0 && ld_src >= cols && ld_dst >= cols && rows <= 65535 * 16, "ag_to_bf16_2d: bad args");
  // (rows beyond 65535 are folded into the column loop by viewing contiguous data as fewer, longer rows - the caller's job)
  AG_REQUIRE(rows <= 65535, "ag_to_bf16_2d: more than 65535 rows (reshape contiguous data to fewer, longer rows)");
  int gx = ag_cdiv(cols, 1024);
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(to_bf16_2d_kernel, dim3(gx, rows), dim3(256), 0, (hipStream_t)stream, src, ld_src, dst, ld_dst, rows, cols);
  AG_CHECK_LAUNCH("ag_to_bf16_2d");
  return AG_OK;
}
