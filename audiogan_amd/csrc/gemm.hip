// gemm.hip -- dense fp32 GEMM on v_mfma_f32_32x32x2_f32 with a fused epilogue.
// Stands in for the cuBLAS calls behind NN.Linear / NN.LSTMCell / NN.LSTM gate products
// (audiogan.py:260,380,385,409,410,498,509,511) and their backward.
//
//   C = act(alpha * op(A) op(B) + beta * C + bias[n] + res)
//
// LDS tiles are k-major (As[k][m], Bs[k][n]) so that both MFMA operands are unit-stride,
// conflict-free ds_read_b32; the loaders transpose k-contiguous operands on the way in.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "gemm_tile.h"

#ifndef GBK
#define GBK 16
#endif
#define GKQ (GBK / 4)


// Tile loaders, split into "global -> registers" and "registers -> LDS" so that the loads of
// tile i+1 are in flight while tile i is being multiplied (one barrier per k-step, two LDS buffers).
// operand stored [rows][K] (k contiguous): tile -> S[k][r]
template <int ROWS>
struct KContig {
  static constexpr int ITER = ROWS * (GBK / 4) / 256;
  static_assert(ITER >= 1, "tile too small");
  f32x4 v[ITER];
  __device__ __forceinline__ void load(const float* __restrict__ G, int ld, int r0, int nrows, int k0, int K,
                                       bool vec, int tid) {
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int idx = tid + it * 256;
      const int r = idx / GKQ, kq = idx % GKQ;
      const int gr = r0 + r, gk = k0 + kq * 4;
      f32x4 q = {0.f, 0.f, 0.f, 0.f};
      if (gr < nrows) {
        const float* src = G + (int64_t)gr * ld + gk;
        if (vec && gk + 3 < K) {
          q = *reinterpret_cast<const f32x4*>(src);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (gk + e < K) q[e] = src[e];
        }
      }
      v[it] = q;
    }
  }
  // whole k-tile inside K, 16-B aligned rows: unconditional loads (rows past the end are clamped - they only
  // feed output rows the epilogue drops).  A predicated load costs a branch and serialises the requests.
  __device__ __forceinline__ void load_fast(const float* __restrict__ G, int ld, int r0, int nrows, int k0, int tid) {
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int idx = tid + it * 256;
      const int r = idx / GKQ, kq = idx % GKQ;
      const int gr = min(r0 + r, nrows - 1);
      v[it] = *reinterpret_cast<const f32x4*>(G + (int64_t)gr * ld + k0 + kq * 4);
    }
  }
  static __device__ __forceinline__ bool fast_ok(int r0, int nrows) { (void)r0; (void)nrows; return true; }
  // per-lane source pointers at k = 0, computed once: the k loop then only adds the (uniform) k offset
  const float* base[ITER];
  __device__ __forceinline__ void init_fast(const float* __restrict__ G, int ld, int r0, int nrows, int tid) {
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int idx = tid + it * 256;
      const int r = idx / GKQ, kq = idx % GKQ;
      base[it] = G + (int64_t)min(r0 + r, nrows - 1) * ld + kq * 4;
    }
  }
  __device__ __forceinline__ void load_hoisted(int ld, int k0) {
    (void)ld;
#pragma unroll
    for (int it = 0; it < ITER; ++it) v[it] = *reinterpret_cast<const f32x4*>(base[it] + k0);
  }
  template <int PITCH>
  __device__ __forceinline__ void store(float* S, int tid, int rb = 0) {
    if (rb) {      // AG_PREC_BF16 (uniform branch)
#pragma unroll
      for (int it = 0; it < ITER; ++it) v[it] = ag_rbf4_if(v[it], 1);
    }
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int idx = tid + it * 256;
      const int r = idx / GKQ, kq = idx % GKQ;
#pragma unroll
      for (int e = 0; e < 4; ++e) S[(kq * 4 + e) * PITCH + r] = v[it][e];
    }
  }
};

// operand stored [K][rows] (row index contiguous): tile -> S[k][r]
template <int ROWS>
struct RContig {
  static constexpr int R4 = ROWS / 4;
  static constexpr int ITER = GBK * R4 / 256;
  static_assert(ITER >= 1, "tile too small");
  f32x4 v[ITER];
  __device__ __forceinline__ void load(const float* __restrict__ G, int ld, int r0, int nrows, int k0, int K,
                                       bool vec, int tid) {
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int idx = tid + it * 256;
      const int k = idx / R4, r = (idx % R4) * 4;
      const int gk = k0 + k, gr = r0 + r;
      f32x4 q = {0.f, 0.f, 0.f, 0.f};
      if (gk < K) {
        const float* src = G + (int64_t)gk * ld + gr;
        if (vec && gr + 3 < nrows) {
          q = *reinterpret_cast<const f32x4*>(src);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (gr + e < nrows) q[e] = src[e];
        }
      }
      v[it] = q;
    }
  }
  __device__ __forceinline__ void load_fast(const float* __restrict__ G, int ld, int r0, int nrows, int k0, int tid) {
    // nrows % 4 == 0: a 16-byte piece is entirely inside or entirely outside; outside pieces are clamped
    // onto the last valid piece (they only feed outputs the epilogue drops)
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int idx = tid + it * 256;
      const int k = idx / R4, r = (idx % R4) * 4;
      v[it] = *reinterpret_cast<const f32x4*>(G + (int64_t)(k0 + k) * ld + min(r0 + r, nrows - 4));
    }
  }
  static __device__ __forceinline__ bool fast_ok(int r0, int nrows) { (void)r0; return (nrows & 3) == 0; }
  const float* base[ITER];
  __device__ __forceinline__ void init_fast(const float* __restrict__ G, int ld, int r0, int nrows, int tid) {
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int idx = tid + it * 256;
      const int k = idx / R4, r = (idx % R4) * 4;
      base[it] = G + (int64_t)k * ld + min(r0 + r, nrows - 4);
    }
  }
  __device__ __forceinline__ void load_hoisted(int ld, int k0) {
    const int64_t koff = (int64_t)k0 * ld;      // uniform
#pragma unroll
    for (int it = 0; it < ITER; ++it) v[it] = *reinterpret_cast<const f32x4*>(base[it] + koff);
  }
  template <int PITCH>
  __device__ __forceinline__ void store(float* S, int tid, int rb = 0) {
    if (rb) {      // AG_PREC_BF16 (uniform branch)
#pragma unroll
      for (int it = 0; it < ITER; ++it) v[it] = ag_rbf4_if(v[it], 1);
    }
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int idx = tid + it * 256;
      const int k = idx / R4, r = (idx % R4) * 4;
      *reinterpret_cast<f32x4*>(S + k * PITCH + r) = v[it];
    }
  }
};

template <int TM, int TN, int WM, int WN, int TA, int TB>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmP p) {
  static_assert(WM * WN == 4, "4 waves");
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
  constexpr int PA = BM + 4, PB = BN + 4;
  __shared__ __attribute__((aligned(16))) float As[2][GBK * PA];
  __shared__ __attribute__((aligned(16))) float Bs[2][GBK * PB];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int wm0 = (wid / WN) * 32 * TM, wn0 = (wid % WN) * 32 * TN;
  int bx, by, bz;
  xcd_place(bx, by, bz);
  const int m0 = by * BM, n0 = bx * BN;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  typename std::conditional<TA == 0, KContig<BM>, RContig<BM>>::type la;
  typename std::conditional<TB == 1, KContig<BN>, RContig<BN>>::type lb;

  const int kbeg = bz * p.kchunk;
  const int kend = (kbeg + p.kchunk < p.K) ? kbeg + p.kchunk : p.K;
  const bool fast = p.vecA && p.vecB && la.fast_ok(m0, p.M) && lb.fast_ok(n0, p.N);
  if (fast) {
    la.init_fast(p.A, p.lda, m0, p.M, tid);
    lb.init_fast(p.B, p.ldb, n0, p.N, tid);
  }
  la.load(p.A, p.lda, m0, p.M, kbeg, p.K, p.vecA, tid);
  lb.load(p.B, p.ldb, n0, p.N, kbeg, p.K, p.vecB, tid);
  la.template store<PA>(As[0], tid, p.rb);
  lb.template store<PB>(Bs[0], tid, p.rb);
  __syncthreads();
  int buf = 0;
  for (int k0 = kbeg; k0 < kend; k0 += GBK) {
    const bool more = k0 + GBK < kend;
    if (more) {
      if (fast && k0 + 2 * GBK <= p.K) {
        la.load_hoisted(p.lda, k0 + GBK);
        lb.load_hoisted(p.ldb, k0 + GBK);
      } else {
        la.load(p.A, p.lda, m0, p.M, k0 + GBK, p.K, p.vecA, tid);
        lb.load(p.B, p.ldb, n0, p.N, k0 + GBK, p.K, p.vecB, tid);
      }
    }
    const float* Ac = As[buf];
    const float* Bc = Bs[buf];
#pragma unroll
    for (int kk = 0; kk < GBK; kk += 2) {
      float av[TM], bv[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) av[i] = Ac[(kk + h) * PA + wm0 + 32 * i + l31];
#pragma unroll
      for (int j = 0; j < TN; ++j) bv[j] = Bc[(kk + h) * PB + wn0 + 32 * j + l31];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
    if (more) {
      la.template store<PA>(As[buf ^ 1], tid, p.rb);
      lb.template store<PB>(Bs[buf ^ 1], tid, p.rb);
    }
    __syncthreads();
    buf ^= 1;
  }

  if (p.ksplit == 1 && m0 + BM <= p.M && n0 + BN <= p.N) {
    // interior tile: no bounds tests; the uniform epilogue options are tested once per element by scalar branches
    const bool hb = p.bias != nullptr, hr = p.res != nullptr, hbeta = p.beta != 0.f;
    float bj[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) bj[j] = hb ? p.bias[n0 + wn0 + 32 * j + l31] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm0 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
        float* dst = p.C + (int64_t)row * p.ldc + n0 + wn0 + l31;
        const float* rs = p.res + (int64_t)row * p.ldres + n0 + wn0 + l31;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          float v = p.alpha * acc[i][j][e] + bj[j];
          if (hbeta) v += p.beta * dst[32 * j];
          dst[32 * j] = ag_res_act(v, hr, hr ? rs[32 * j] : 0.f, p.act, p.slope);
        }
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = m0 + wm0 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
      if (row >= p.M) continue;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
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

// ------------------------------------------------------------------------------------------
// AG_PREC_BF16: 128x128 tile on v_mfma_f32_32x32x16_bf16.  Operands are fp32 in memory; the loaders round them to bf16
// (RNE) on the way into LDS, so a k-tile of 32 costs half the LDS bytes and 1/8 of the MFMA instructions of the fp32
// kernels.  LDS image per operand: [128 rows][32 k] bf16 (64-byte rows); the 16-byte chunk c of row r sits in slot
// c ^ ((r >> 2) & 3) so that the ds_read_b128 of a 16-lane group (16 consecutive rows, one chunk) is conflict free.
// MFMA operand of lane (row l & 31, half h = l >> 5) for k-step s: chunk 2s + h = k 8h .. 8h+7 of that step.
//   k-contiguous operands ([rows][K]): a lane loads one float4 (4 k) -> 8 bytes of bf16 -> ds_write_b64
//   row-contiguous operands ([K][rows]): a lane loads a 4 k x 4 rows micro-tile (4 float4, each coalesced along the
//   rows), transposes it in registers and writes 4 x 8 bytes
// Needs K % 4 == 0, 16-byte aligned rows and (row-contiguous operands) rows % 4 == 0; other shapes take gemm_kernel with
// operands rounded in registers (same sums).
// ------------------------------------------------------------------------------------------
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2g __attribute__((ext_vector_type(2)));

#define BF_BK 64       // k per tile: 16 float4 loads in flight per lane and operand pair (the kernel is bound by how many
                       // bytes it keeps in flight, not by the MFMA pipe)
// LDS image per operand: [128 rows][64 k] bf16 = 128-byte rows of eight 16-byte chunks; chunk c of row r sits in slot
// c ^ ((r >> 1) & 7): the 16 rows of a ds_read_b128 lane group then cover 16 distinct (slot, 128-B half) pairs.
__device__ __forceinline__ int bf_slot(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }

// AG_PREC_F32X3 (an EXPERIMENT, never the headline precision): x = hi + lo with hi = bf16(x), lo = bf16(x - hi); the product
// a b ~ a_hi b_hi + a_hi b_lo + a_lo b_hi on three bf16 MFMAs with fp32 accumulation drops only a_lo b_lo (2^-16 of the
// product).  The lo image sits LO_OFF bytes behind the hi image in LDS.
__device__ __forceinline__ void bf_split(float x0, float x1, unsigned& hi, unsigned& lo) {
  hi = ag_pack_bf16(x0, x1);
  lo = ag_pack_bf16(x0 - __uint_as_float(hi << 16), x1 - __uint_as_float(hi & 0xFFFF0000u));
}

template <int T_>      // 0: stored [rows][K] (k contiguous)   1: stored [K][rows]
struct Bf16Loader {
  f32x4 v[8];
  // k-contiguous: 128 rows x 16 float4 per tile = 8 per thread;  row-contiguous: two 4 k x 4 rows micro-tiles per thread
  __device__ __forceinline__ void load(const float* __restrict__ G, int ld, int r0, int nrows, int k0, int K, int tid) {
    if (T_ == 0) {
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int idx = tid + 256 * it;
        const int r = idx >> 4, q = idx & 15;
        const int k = k0 + 4 * q;
        // (loads only: no arithmetic on the loaded registers here, or the compiler waits for them before the MFMA loop
        // of the current tile instead of after it; the k >= K zeroing happens in store())
        v[it] = *reinterpret_cast<const f32x4*>(G + (int64_t)min(r0 + r, nrows - 1) * ld + min(k, K - 4));
      }
    } else {
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int idx = tid + 256 * it;
        const int kq = idx >> 5, r4 = idx & 31;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int k = k0 + 4 * kq + i;
          v[4 * it + i] = *reinterpret_cast<const f32x4*>(G + (int64_t)min(k, K - 1) * ld + min(r0 + 4 * r4, nrows - 4));
        }
      }
    }
  }
  template <int LO_OFF>
  __device__ __forceinline__ void store_split(char* S, int tid, int k0, int K) const {
    if (T_ == 0) {
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int idx = tid + 256 * it;
        const int r = idx >> 4, q = idx & 15;
        const unsigned msk = (k0 + 4 * q < K) ? 0xFFFFFFFFu : 0u;
        unsigned h0, l0, h1, l1;
        bf_split(v[it][0], v[it][1], h0, l0);
        bf_split(v[it][2], v[it][3], h1, l1);
        char* d = S + bf_slot(r, q >> 1) + ((q & 1) << 3);
        *reinterpret_cast<u32x2g*>(d) = u32x2g{h0 & msk, h1 & msk};
        *reinterpret_cast<u32x2g*>(d + LO_OFF) = u32x2g{l0 & msk, l1 & msk};
      }
    } else {
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int idx = tid + 256 * it;
        const int kq = idx >> 5, r4 = idx & 31;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int r = 4 * r4 + rr;
          const unsigned msk = (k0 + 4 * kq < K) ? 0xFFFFFFFFu : 0u;
          unsigned h0, l0, h1, l1;
          bf_split(v[4 * it][rr], v[4 * it + 1][rr], h0, l0);
          bf_split(v[4 * it + 2][rr], v[4 * it + 3][rr], h1, l1);
          char* d = S + bf_slot(r, kq >> 1) + ((kq & 1) << 3);
          *reinterpret_cast<u32x2g*>(d) = u32x2g{h0 & msk, h1 & msk};
          *reinterpret_cast<u32x2g*>(d + LO_OFF) = u32x2g{l0 & msk, l1 & msk};
        }
      }
    }
  }
  __device__ __forceinline__ void store(char* S, int tid, int k0, int K) const {
    if (T_ == 0) {
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int idx = tid + 256 * it;
        const int r = idx >> 4, q = idx & 15;
        const unsigned msk = (k0 + 4 * q < K) ? 0xFFFFFFFFu : 0u;      // (K % 4 == 0: a float4 is entirely in or out)
        u32x2g w = {ag_pack_bf16(v[it][0], v[it][1]) & msk, ag_pack_bf16(v[it][2], v[it][3]) & msk};
        *reinterpret_cast<u32x2g*>(S + bf_slot(r, q >> 1) + ((q & 1) << 3)) = w;
      }
    } else {
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int idx = tid + 256 * it;
        const int kq = idx >> 5, r4 = idx & 31;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int r = 4 * r4 + rr;
          const unsigned msk = (k0 + 4 * kq < K) ? 0xFFFFFFFFu : 0u;   // (K % 4 == 0: a 4-k group is entirely in or out)
          u32x2g w = {ag_pack_bf16(v[4 * it][rr], v[4 * it + 1][rr]) & msk,
                      ag_pack_bf16(v[4 * it + 2][rr], v[4 * it + 3][rr]) & msk};
          *reinterpret_cast<u32x2g*>(S + bf_slot(r, kq >> 1) + ((kq & 1) << 3)) = w;
        }
      }
    }
  }
};

template <int TA, int TB, int X3 = 0>      // X3: the split-bf16 form (hi + lo images, 3 MFMAs per product)
__global__ __launch_bounds__(256) void gemm_bf16_kernel(const GemmP p) {
  constexpr int BM = 128, BN = 128, IMG = 128 * 128;        // bytes per operand image
  constexpr int TILEB = X3 ? 2 * IMG : IMG;                  // bytes per operand tile (X3: hi image, then lo image)
  extern __shared__ __attribute__((aligned(16))) char sm[];   // 2 buffers x (A tile + B tile) = 64 KiB (X3: 128 KiB)
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int wm0 = (wid >> 1) * 64, wn0 = (wid & 1) * 64;
  int bx, by, bz;
  xcd_place(bx, by, bz);
  const int m0 = by * BM, n0 = bx * BN;
  Bf16Loader<TA> la;
  Bf16Loader<(TB == 1 ? 0 : 1)> lb;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int kbeg = bz * p.kchunk;
  const int kend = (kbeg + p.kchunk < p.K) ? kbeg + p.kchunk : p.K;
  la.load(p.A, p.lda, m0, p.M, kbeg, kend, tid);
  lb.load(p.B, p.ldb, n0, p.N, kbeg, kend, tid);
  if (X3) {
    la.template store_split<IMG>(sm, tid, kbeg, kend);
    lb.template store_split<IMG>(sm + TILEB, tid, kbeg, kend);
  } else {
    la.store(sm, tid, kbeg, kend);
    lb.store(sm + TILEB, tid, kbeg, kend);
  }
  __syncthreads();
  int buf = 0;
  for (int k0 = kbeg; k0 < kend; k0 += BF_BK) {
    const bool more = k0 + BF_BK < kend;
    if (more) {
      la.load(p.A, p.lda, m0, p.M, k0 + BF_BK, kend, tid);
      lb.load(p.B, p.ldb, n0, p.N, k0 + BF_BK, kend, tid);
    }
    const char* As = sm + buf * 2 * TILEB;
    const char* Bs = As + TILEB;
#pragma unroll
    for (int s_ = 0; s_ < BF_BK / 16; ++s_) {
      bf16x8 av[2], bv[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) av[i] = *reinterpret_cast<const bf16x8*>(As + bf_slot(wm0 + 32 * i + l31, 2 * s_ + h));
#pragma unroll
      for (int j = 0; j < 2; ++j) bv[j] = *reinterpret_cast<const bf16x8*>(Bs + bf_slot(wn0 + 32 * j + l31, 2 * s_ + h));
      if (X3) {
        bf16x8 al[2], bl[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) al[i] = *reinterpret_cast<const bf16x8*>(As + IMG + bf_slot(wm0 + 32 * i + l31, 2 * s_ + h));
#pragma unroll
        for (int j = 0; j < 2; ++j) bl[j] = *reinterpret_cast<const bf16x8*>(Bs + IMG + bf_slot(wn0 + 32 * j + l31, 2 * s_ + h));
        // small terms first, the large one last
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bv[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bl[j], acc[i][j], 0, 0, 0);
          }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
    if (more) {
      if (X3) {
        la.template store_split<IMG>(sm + (buf ^ 1) * 2 * TILEB, tid, k0 + BF_BK, kend);
        lb.template store_split<IMG>(sm + (buf ^ 1) * 2 * TILEB + TILEB, tid, k0 + BF_BK, kend);
      } else {
        la.store(sm + (buf ^ 1) * 2 * TILEB, tid, k0 + BF_BK, kend);
        lb.store(sm + (buf ^ 1) * 2 * TILEB + TILEB, tid, k0 + BF_BK, kend);
      }
    }
    __syncthreads();
    buf ^= 1;
  }
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

template <int TA, int TB, int X3>
static void launch_bf16_one(const GemmP& p, dim3 grid, hipStream_t st) {
  auto kern = gemm_bf16_kernel<TA, TB, X3>;
  const int lds = (X3 ? 128 : 64) * 1024;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, p);
}

static int launch_gemm_bf16(const GemmP& p, int ta, int tb, hipStream_t st, bool x3 = false) {
  dim3 grid(ag_cdiv(p.N, 128), ag_cdiv(p.M, 128), p.ksplit);
  if (x3) {
    if (ta == 0 && tb == 0) launch_bf16_one<0, 0, 1>(p, grid, st);
    if (ta == 0 && tb == 1) launch_bf16_one<0, 1, 1>(p, grid, st);
    if (ta == 1 && tb == 0) launch_bf16_one<1, 0, 1>(p, grid, st);
    if (ta == 1 && tb == 1) launch_bf16_one<1, 1, 1>(p, grid, st);
  } else {
    if (ta == 0 && tb == 0) launch_bf16_one<0, 0, 0>(p, grid, st);
    if (ta == 0 && tb == 1) launch_bf16_one<0, 1, 0>(p, grid, st);
    if (ta == 1 && tb == 0) launch_bf16_one<1, 0, 0>(p, grid, st);
    if (ta == 1 && tb == 1) launch_bf16_one<1, 1, 0>(p, grid, st);
  }
  AG_CHECK_LAUNCH("ag_gemm(bf16)");
  return AG_OK;
}

// second stage of a split-K product: C = beta*C + sum_z part[z] + bias + res.  8 threads per output element: thread zq
// sums slabs z = zq, zq + 8, ... ascending, then the 8 partial sums are added in the order zq = 0..7 (fixed order).
__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(const float* __restrict__ part, int Z, int64_t pitch,
                                                                 int M, int N, float* __restrict__ C, int ldc,
                                                                 float beta, const float* __restrict__ bias,
                                                                 const float* __restrict__ res, int ldres) {
  __shared__ float sh[256];
  const int64_t mn = (int64_t)M * N;
  const int64_t i = (int64_t)blockIdx.x * 32 + (threadIdx.x & 31);
  float s = 0.f;
  if (i < mn)
    for (int z = threadIdx.x >> 5; z < Z; z += 8) s += part[(int64_t)z * pitch + i];
  sh[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x >= 32 || i >= mn) return;
  s = sh[threadIdx.x];
#pragma unroll
  for (int q = 1; q < 8; ++q) s += sh[q * 32 + threadIdx.x];
  const int row = (int)(i / N), col = (int)(i - (int64_t)row * N);
  float* dst = C + (int64_t)row * ldc + col;
  if (beta != 0.f) s += beta * *dst;
  if (bias) s += bias[col];
  if (res) s += res[(int64_t)row * ldres + col];
  *dst = s;
}

int ag_splitk_reduce(const float* part, int Z, int64_t pitch, int M, int N, float* C, int ldc, float beta,
                     const float* bias, const float* res, int ldres, hipStream_t st) {
  const int64_t mn = (int64_t)M * N;
  hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((unsigned)ag_cdiv64(mn, 32)), dim3(256), 0, st, part, Z, pitch, M, N,
                     C, ldc, beta, bias, res, ldres);
  AG_CHECK_LAUNCH("ag_splitk_reduce");
  return AG_OK;
}

// Tile shape of the LDS-DMA kernel (gemm_tile.h) for an [M x N] output in `ksplit` K slices.  A CU's matrix pipes are shared
// by the workgroups resident on it, so a launch takes about ceil(workgroups / 256) x (BM x BN) / rate(shape): prefer the
// largest tile that still gives every CU a workgroup.  Rates: TF measured by tools/gemm_lab.hip (profiles/r04_gemm_lab.txt).
// AG_GEMM_TILE=0..3 in the environment forces a shape (A/B switch of tools/prof_gemm.py).
static const int g_gemm_tile_force = [] { const char* e = getenv("AG_GEMM_TILE"); return e ? atoi(e) : -1; }();
static int gemm_pick_tile(int M, int N, int ksplit) {
  if (g_gemm_tile_force >= 0 && g_gemm_tile_force <= 3) return g_gemm_tile_force;
  static const int order[4] = {3, 1, 2, 0};
  static const double rate[4] = {120., 133., 133., 137.};
  int best = 0;
  double best_t = 0.;
  for (int s : order) {
    int bm, bn;
    gemm_tile_dims(s, bm, bn);
    const int64_t wgs = (int64_t)ag_cdiv(M, bm) * ag_cdiv(N, bn) * ksplit;
    // (ties: the shape that computes fewer padded elements)
    const double t = (double)ag_cdiv64(wgs, 256) * bm * bn / rate[s] + 1e-9 * (double)wgs * bm * bn;
    if (best_t == 0. || t < best_t * 0.98) { best = s; best_t = t; }
  }
  return best;
}

static int launch_gemm_dma(const GemmP& p, int ta, int tb, hipStream_t st) {
  const int rc = gemm_tile_launch(p, ta, tb, gemm_pick_tile(p.M, p.N, p.ksplit), st);
  if (rc != AG_OK) {
    ag_set_error("ag_gemm: tile kernel set-up failed");
    return rc;
  }
  AG_CHECK_LAUNCH("ag_gemm");
  return AG_OK;
}

template <int TM, int TN, int WM, int WN>
static int launch_gemm(const GemmP& p, int ta, int tb, hipStream_t st) {
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
  dim3 grid(ag_cdiv(p.N, BN), ag_cdiv(p.M, BM), p.ksplit);
  if (ta == 0 && tb == 0) hipLaunchKernelGGL((gemm_kernel<TM, TN, WM, WN, 0, 0>), grid, dim3(256), 0, st, p);
  if (ta == 0 && tb == 1) hipLaunchKernelGGL((gemm_kernel<TM, TN, WM, WN, 0, 1>), grid, dim3(256), 0, st, p);
  if (ta == 1 && tb == 0) hipLaunchKernelGGL((gemm_kernel<TM, TN, WM, WN, 1, 0>), grid, dim3(256), 0, st, p);
  if (ta == 1 && tb == 1) hipLaunchKernelGGL((gemm_kernel<TM, TN, WM, WN, 1, 1>), grid, dim3(256), 0, st, p);
  AG_CHECK_LAUNCH("ag_gemm");
  return AG_OK;
}

static bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// K slices for a product with few output tiles and a long reduction: about 512 workgroups, slices of at least 256 k,
// slab traffic (2 * ks * M * N * 4 B) capped at 12 M floats, and a count xcd_place can put on whole XCDs (2, 4 or a
// multiple of 8).  0 / 1 = do not split.
static int gemm_pick_ksplit(int64_t tiles, int64_t mn, int K) {
  static const int cand[] = {32, 24, 16, 8, 4, 2};
  const int64_t want = (512 * 4 / tiles + 2) / 3;            // up to a third above 512 / tiles
  for (int c : cand)
    if (c <= want && c <= K / 256 && (int64_t)c * mn <= ((int64_t)12 << 20)) return c;
  return 1;
}

extern "C" int ag_gemm(const float* A, int lda, int ta, const float* B, int ldb, int tb, float* C,
                       int ldc, int M, int N, int K, float alpha, float beta, const float* bias,
                       const float* res, int ldres, int act, float slope, void* stream) {
  const AgWs ws = ag_ws_take();     // FIRST: an argument error below must not leave a stale binding behind
  AG_REQUIRE(A && B && C, "ag_gemm: null tensor");
  AG_REQUIRE(M > 0 && N > 0 && K > 0, "ag_gemm: bad shape %d %d %d", M, N, K);
  AG_REQUIRE((ta == 0 || ta == 1) && (tb == 0 || tb == 1), "ag_gemm: bad transpose flag");
  AG_REQUIRE(lda >= (ta ? M : K) && ldb >= (tb ? K : N) && ldc >= N, "ag_gemm: bad leading dim");
  AG_REQUIRE(ag_cdiv(M, 64) <= 65535, "ag_gemm: M too large");
  AG_REQUIRE(act != AG_ACT_LEAKY_GATE || res, "ag_gemm: AG_ACT_LEAKY_GATE needs the saved activation in `res`");
  GemmP p;
  p.A = A; p.B = B; p.C = C; p.bias = bias; p.res = res;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldres = ldres;
  p.M = M; p.N = N; p.K = K;
  p.alpha = alpha; p.beta = beta; p.slope = slope; p.act = act;
  p.vecA = aligned16(A) && (lda % 4 == 0);
  p.vecB = aligned16(B) && (ldb % 4 == 0);
  p.rb = ag_precision() == AG_PREC_BF16;
  hipStream_t st = (hipStream_t)stream;
  p.ksplit = 1;
  p.kchunk = ag_roundup(K, 64);
  const int64_t big = (int64_t)ag_cdiv(M, 128) * ag_cdiv(N, 128);
  const bool use128 = M > 64 && N > 64 && (big >= 192 || K >= 2048);
  const int64_t tiles = use128 ? big : (int64_t)ag_cdiv(M, 64) * ag_cdiv(N, 64);
  // few output tiles but a long reduction (weight gradients over all frames): slice K over grid.z; the slices' partial
  // tiles go to the bound workspace and are summed in a fixed order by a second kernel (deterministic).  Needs a linear
  // epilogue.
  p.part = nullptr;
  if (tiles < 192 && K >= 1024 && act == AG_ACT_NONE) {
    const int64_t mn = (int64_t)M * N;
    int ks = gemm_pick_ksplit(tiles, mn, K);
    // (no bound workspace, or one too small for two slices: the product runs unsplit - slower, same result class; the
    // float-atomic combination of round 1 is gone)
    const bool slabs = ws.p && ws.numel >= 2 * mn;
    if (slabs && (int64_t)ks * mn > ws.numel) ks = (int)(ws.numel / mn);
    if (ks >= 2 && slabs) {
      p.ksplit = ks;
      p.kchunk = ag_roundup(ag_cdiv(K, ks), 64);
      p.ksplit = ag_cdiv(K, p.kchunk);
      p.part = p.ksplit > 1 ? ws.p : nullptr;
    }
  }
  int rc;
  // bf16 mode: the bf16-MFMA kernel takes every shape with more than one row tile's worth of work whose operands
  // can be read 16 bytes at a time; the rest runs the fp32 kernels on operands rounded in registers
  // AG_PREC_F32X3: the same kernel in its split-bf16 form for the large products only (the others stay exact fp32)
  const bool x3 = ag_precision() == AG_PREC_F32X3 && use128;
  const bool bf16k = (p.rb || x3) && M > 32 && N > 32 && p.vecA && p.vecB && K % 4 == 0 && p.kchunk % 64 == 0 &&
                     (ta == 0 || (M % 4 == 0 && M >= 4)) && (tb == 1 || (N % 4 == 0 && N >= 4));
  if (bf16k) {
    rc = launch_gemm_bf16(p, ta, tb, st, x3);
  } else if (use128) {
    // LDS-DMA variant: whole 16-k tiles only, 16-byte aligned rows, row-contiguous operands with rows % 4 == 0
    // (AG_GEMM_NODMA=1 in the environment forces the register-staged kernel: A/B switch for tools/prof_gemm.py)
    const bool dma = p.vecA && p.vecB && K % 16 == 0 && p.kchunk % 16 == 0 && (ta == 0 || M % 4 == 0) &&
                     (tb == 1 || N % 4 == 0) && M >= 4 && N >= 4 && !p.rb && getenv("AG_GEMM_NODMA") == nullptr;
    rc = dma ? launch_gemm_dma(p, ta, tb, st) : launch_gemm<2, 2, 2, 2>(p, ta, tb, st);  // 128x128
  } else {
    rc = launch_gemm<1, 1, 2, 2>(p, ta, tb, st);              // 64x64
  }
  if (rc != AG_OK || !p.part) return rc;
  // a weight gradient's second stage inside a deferral scope joins the scope's ONE launch: slab z of a contiguous C is
  // exactly ag_slab_reduce's layout and both kernels sum z = zq, zq + 8, ... then zq = 0..7 (bitwise the same result)
  if (ag_reduces_deferred() && !bias && !res && (beta == 0.f || beta == 1.f))
    return ag_slab_defer_2d(p.part, p.ksplit, M, N, C, ldc, beta == 1.f ? 1 : 0, st);
  return ag_splitk_reduce(p.part, p.ksplit, (int64_t)M * N, M, N, C, ldc, beta, bias, res, ldres, st);
}

// floats of workspace ag_gemm wants bound (ag_bind_workspace) so that a split-K product is reduced in two stages
extern "C" int64_t ag_gemm_ws_numel(int M, int N, int K, int act) {
  const int64_t big = (int64_t)ag_cdiv(M, 128) * ag_cdiv(N, 128);
  const bool use128 = M > 64 && N > 64 && (big >= 192 || K >= 2048);
  const int64_t tiles = use128 ? big : (int64_t)ag_cdiv(M, 64) * ag_cdiv(N, 64);
  if (!(tiles < 192 && K >= 1024 && act == AG_ACT_NONE)) return 0;
  const int ks = gemm_pick_ksplit(tiles, (int64_t)M * N, K);
  return ks >= 2 ? (int64_t)ks * M * N : 0;
}

// out[n] (+)= sum_m X[m, n]
__global__ __launch_bounds__(256) void col_sum_kernel(const float* __restrict__ X, int ldx,
                                                      float* __restrict__ out, int M, int N,
                                                      int rows_per, float* __restrict__ part, int accumulate, int x16) {
  const int n = blockIdx.x * 64 + (threadIdx.x & 63);
  const int sub = threadIdx.x >> 6;  // 4 row-phases per block
  const int mbeg = blockIdx.y * rows_per;
  int mend = mbeg + rows_per;
  if (mend > M) mend = M;
  float s = 0.f;
  if (n < N) {
    if (x16) {        // X stored as bfloat16 (ldx in 2-byte elements)
      const unsigned short* X16 = reinterpret_cast<const unsigned short*>(X);
      for (int m = mbeg + sub; m < mend; m += 4) s += __uint_as_float((unsigned)X16[(int64_t)m * ldx + n] << 16);
    } else {
      for (int m = mbeg + sub; m < mend; m += 4) s += X[(int64_t)m * ldx + n];
    }
  }
  __shared__ float red[4][64];
  red[sub][threadIdx.x & 63] = s;
  __syncthreads();
  if (sub == 0 && n < N) {
    s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    if (part) part[(int64_t)blockIdx.y * N + n] = s;
    else out[n] = accumulate ? out[n] + s : s;        // one row block: this thread is the only writer of out[n]
  }
}

extern "C" int ag_col_sum(const void* X, int x_bf16, int ldx, float* out, int M, int N, int accumulate, void* stream) {
  const AgWs ws = ag_ws_take();     // FIRST: an argument error below must not leave a stale binding behind
  AG_REQUIRE(X && out && M > 0 && N > 0 && ldx >= N, "ag_col_sum: bad args");
  const int gx = ag_cdiv(N, 64);
  int gy = ag_cdiv(1024, gx);
  if (gy > ag_cdiv(M, 16)) gy = ag_cdiv(M, 16);
  if (gy < 1) gy = 1;
  float* part = nullptr;
  if (gy > 1 && ws.p && ws.numel >= 2 * (int64_t)N) {
    if ((int64_t)gy * N > ws.numel) gy = (int)(ws.numel / N);
    part = ws.p;
  } else {
    gy = 1;       // one row block: a single writer per column, nothing to order (and no workspace needed)
  }
  const int rows_per = ag_cdiv(M, gy);
  gy = ag_cdiv(M, rows_per);
  if (gy == 1) part = nullptr;
  hipLaunchKernelGGL(col_sum_kernel, dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, (const float*)X, ldx, out, M,
                     N, rows_per, part, accumulate, x_bf16 ? 1 : 0);
  AG_CHECK_LAUNCH("ag_col_sum");
  if (part) return ag_slab_reduce(part, gy, N, out, accumulate ? 1 : 0, (hipStream_t)stream);
  return AG_OK;
}
