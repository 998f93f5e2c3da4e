// conv_engine.hip -- host dispatch of the 1-D convolution engine (device code: conv_engine_impl.h; one translation unit
// per tile configuration: conv_engine_t*.hip)
#include "conv_engine_impl.h"

// tile configurations <TILES_O, TILES_T, WAVES_O, WAVES_T>, each compiled in its own translation unit
int ag_conv_cfg_1214(ConvP& p, hipStream_t st);
int ag_conv_cfg_2114(ConvP& p, hipStream_t st);
int ag_conv_cfg_1122(ConvP& p, hipStream_t st);
int ag_conv_cfg_2122(ConvP& p, hipStream_t st);
int ag_conv_cfg_2222(ConvP& p, hipStream_t st);
int ag_conv_cfg_1141(ConvP& p, hipStream_t st);

extern "C" int64_t ag_wpa_numel(int d0, int d1, int K) {
  const int cp2 = ag_roundup(d1, 2), mp = ag_roundup(d0, 32);
  return ag_wq_offset(cp2, K, mp) + ag_wq_floats(cp2, K, mp);
}
extern "C" int64_t ag_wpb_numel(int d0, int d1, int K, int stride) {
  const int cp2 = ag_roundup(d0, 2), mt = ag_cdiv(K, stride), mp = ag_roundup(d1 * stride, 32);
  return ag_wq_offset(cp2, mt, mp) + ag_wq_floats(cp2, mt, mp);
}

// The ragged last column(s) of a transposed conv go to a second, narrow-tile launch after the main one.
// (Running it beside the main launch on a side stream - fork / join events, a parallel branch under graph
// capture - was measured: slower, both eagerly and in the replayed step.)
template <typename MainFn, typename TailFn>
static int main_and_tail(hipStream_t st, MainFn mainf, TailFn tailf) {
  const int rc = mainf(st);
  return rc != AG_OK ? rc : tailf(st);
}

extern "C" int ag_conv1d_engine(const ag_conv_args* args, void* stream) {
  AG_REQUIRE(args != nullptr, "ag_conv1d_engine: null args");
  ConvP p;
  p.a = *args;
  const ag_conv_args& a = p.a;
  p.aligned = 0;
  p.rb = ag_precision() == AG_PREC_BF16;
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
    if (a.wp_pad == a.pad && ag_scatter_aligned(a.K, a.stride, a.pad)) {
      p.aligned = 1;                              // every phase on columns m = 0 .. ceil(Lout/s) - 1
      p.n_lo = ag_cdiv(a.pad, a.stride);
      p.n_cnt = ag_cdiv(a.Lout, a.stride);
    } else {
      p.n_lo = a.pad / a.stride;
      const int n_hi = (a.Lout - 1 + a.pad) / a.stride;
      p.n_cnt = n_hi - p.n_lo + 1;
    }
  }
  if (getenv("AG_CONV_C1") == nullptr || getenv("AG_CONV_C1")[0] != '0') {
    // single-input-channel layers (D1, G1.conv) and their backward-data: streaming kernels (conv_c1.hip)
    int rc = AG_OK;
    if (ag_conv_c1_try_fwd(a, p.rb, (hipStream_t)stream, &rc)) return rc;
    if (ag_conv_c1_try_bwdx(a, p.rb, (hipStream_t)stream, &rc)) return rc;
  }
  AG_REQUIRE(p.taps <= MAX_TAPS, "ag_conv1d_engine: more than %d taps", MAX_TAPS);
  p.sp_shift = ilog2_exact(p.sp);
  p.s_shift = ilog2_exact(a.stride);
  p.efast = g_conv_efast && (int64_t)a.O * a.y_cs * 4 < ((int64_t)1 << 31) && (!a.res || (int64_t)a.O * a.res_cs * 4 < ((int64_t)1 << 31));
  p.xvec = (((uintptr_t)a.x & 15) == 0) && (a.x_bs % 4 == 0) && (a.x_cs % 4 == 0);
  p.Cpad = ag_roundup(a.C, 2);
  p.Mpad = ag_roundup(p.Mrows, 32);
  hipStream_t st = (hipStream_t)stream;
  // tile choice: wide in rows for fat layers, wide in time for thin ones
  if (p.Mrows <= 32) return ag_conv_cfg_1214(p, st);   // 32 x 256
  // A transposed conv with pad % stride != 0 has n_cnt = L/s + 1 columns: give the ragged last
  // column(s) to a narrow tile instead of a whole extra 128-wide one.
  // Inside a replayed graph that second launch costs ~20 us of its own (a node never takes less than ~5, and each of its
  // workgroups stages a full weight panel for one column) while one more column of tiles costs 1 / (number of column
  // tiles) of the main launch: measured at batch 64, G2-G4.deconv forward 88 / 88 / 76 -> 81 / 81 / 68 us without the
  // tail, G1.deconv (8 column tiles) 82 -> 84.  So: a tail only when the main launch has fewer than 8 column tiles
  // (AG_CONV_TAIL=1: always, for A/B runs).
  static const bool g_tail_always = [] { const char* e = getenv("AG_CONV_TAIL"); return e && e[0] == '1'; }();
  const int tail0 = p.n_cnt % 128;
  const int tail = (g_tail_always || (p.n_cnt - tail0) / 128 < 8) ? tail0 : 0, main_cols = p.n_cnt - tail;
  if (p.Mrows <= 64) {
    if (tail > 0 && tail <= 32 && main_cols > 0) {
      ConvP qm = p, qt = p;
      qm.n_cnt = main_cols;
      qt.n_lo += main_cols;
      qt.n_cnt = tail;
      return main_and_tail(st, [&](hipStream_t s_) { return ag_conv_cfg_2114(qm, s_); },
                           [&](hipStream_t s_) { return ag_conv_cfg_1122(qt, s_); });   // 64 x 64
    }
    return ag_conv_cfg_2114(p, st);                     // 64 x 128
  }
  if (p.n_cnt <= 64) return ag_conv_cfg_2122(p, st);   // 128 x 64
  // Transposed convs with two tap slots (the generator's k = 2s deconvs forward: a reduction of C * 2, one or two short
  // chunks): 128 x 64 tiles.  A launch of 128 x 128 tiles is ONE round of workgroups (1088 for 1024 slots) that all store at
  // the same time after they all multiplied; twice as many half-width workgroups run in two rounds, and the second round's
  // staging and MFMAs overlap the first one's stores.  Measured (profiles/r04_conv_solo.txt, microseconds, 128 x 128 solo ->
  // 128 x 64): G1-G4.deconv forward 88 / 84 / 84 / 54 -> 78 (8-wave form) / 73 / 73 / 49 (solo); three tap slots (the strided
  // convs' backward-data) lose 151 -> 165.  AG_CONV_HALF = 1 / 0 forces / forbids it (A/B runs).
  static const int g_half = [] { const char* e = getenv("AG_CONV_HALF"); return e ? atoi(e) : -1; }();
  // (fp32 kernels only: the bf16-MFMA kernel's launches are 40-60 us of mostly stores and lose with half tiles, 51 -> 59 us)
  if (a.mode == 1 && (g_half == 1 || (g_half != 0 && p.taps == 2 && !p.rb))) return ag_conv_cfg_2122(p, st);
  // fewer than two 128x128 workgroups per CU: one MFMA wave per SIMD cannot keep the matrix pipe fed, take
  // half-width tiles (twice the workgroups, two co-resident per CU)
  // (the bf16 kernel runs one workgroup per CU and wants the full tile's reuse of the staged weights)
  const bool bfp = p.rb && ag_cdiv(p.Cpad, 16) * p.taps >= 32 && g_conv_bf16_mfma;      // (the deep-reduction variant)
  if (!bfp && (int64_t)ag_cdiv(p.n_cnt, 128) * ag_cdiv(p.Mrows, 128) * a.B < 512) return ag_conv_cfg_2122(p, st);
  if (tail > 0 && tail <= 32 && main_cols > 0) {
    ConvP qm = p, qt = p;
    qm.n_cnt = main_cols;
    qt.n_lo += main_cols;
    qt.n_cnt = tail;
    // (main: half-width tiles when there would be fewer than two 128x128 workgroups per CU, as above)
    const bool half = !bfp && (int64_t)(main_cols / 128) * ag_cdiv(p.Mrows, 128) * a.B < 512;
    return main_and_tail(st, [&](hipStream_t s_) { return half ? ag_conv_cfg_2122(qm, s_)
                                                                : ag_conv_cfg_2222(qm, s_); },
                         [&](hipStream_t s_) { return ag_conv_cfg_1141(qt, s_); });     // 128 x 32
  }
  return ag_conv_cfg_2222(p, st);                       // 128 x 128
}

// ------------------------------------------------------------------------------------------
// weight layouts for plain (not weight-normed) tensors
// ------------------------------------------------------------------------------------------
__global__ void prep_conv_weight_kernel(const float* __restrict__ w, float* __restrict__ wpa,
                                        float* __restrict__ wpb, int d0, int d1, int K, int s, int pad) {
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
    const float val = (a0 < d0 && c < d1) ? w[((int64_t)a0 * d1 + c) * K + k] : 0.f;
    wpa[i] = val;
    reinterpret_cast<unsigned short*>(wpa + na)[ag_wq_index(c, k, a0, K, d0p32)] = (unsigned short)ag_pack_bf16(val, val);
  }
  if (wpb && i < nb) {
    const int row = (int)(i % mp);
    const int64_t am = i / mp;
    const int m = (int)(am % mt);
    const int a0 = (int)(am / mt);
    const int o = row / s, r = row - o * s;
    const int mm = m - (ag_scatter_aligned(K, s, pad) ? ag_scatter_shift(s, pad, r) : 0);   // slot -> tap
    const int k = r + s * mm;
    const float val = (a0 < d0 && o < d1 && mm >= 0 && k < K) ? w[((int64_t)a0 * d1 + o) * K + k] : 0.f;
    wpb[i] = val;
    reinterpret_cast<unsigned short*>(wpb + nb)[ag_wq_index(a0, m, row, mt, mp)] = (unsigned short)ag_pack_bf16(val, val);
  }
}

extern "C" int ag_prep_conv_weight(const float* w, float* wpa, float* wpb, int d0, int d1, int K,
                                   int stride, int pad, void* stream) {
  AG_REQUIRE(w && (wpa || wpb), "ag_prep_conv_weight: null tensor");
  AG_REQUIRE(d0 > 0 && d1 > 0 && K > 0 && stride > 0, "ag_prep_conv_weight: bad shape");
  int64_t n = 0;     // threads: one per fp32 layout element (each also writes its bf16 image element)
  if (wpa) n = ag_wq_offset(ag_roundup(d1, 2), K, ag_roundup(d0, 32));
  const int64_t nbw = ag_wq_offset(ag_roundup(d0, 2), ag_cdiv(K, stride), ag_roundup(d1 * stride, 32));
  if (wpb && nbw > n) n = nbw;
  hipLaunchKernelGGL(prep_conv_weight_kernel, dim3((unsigned)ag_cdiv64(n, 256)), dim3(256), 0,
                     (hipStream_t)stream, w, wpa, wpb, d0, d1, K, stride, pad);
  AG_CHECK_LAUNCH("ag_prep_conv_weight");
  return AG_OK;
}
