// lstm_persist.hip -- a whole NN.LSTM layer pass (audiogan.py:498-503, :543; Embedder :315) as ONE persistent launch
// with the recurrent weights resident in LDS.
//
// The per-step launches of lstm_step.hip re-stream W_hh from the fabric on every one of the T steps (L2 is not kept
// across launches): 8.4 MB per step for the critic's biLSTM, more than the step's own operands.  Here every workgroup
// keeps ITS slice of W_hh in LDS for the whole sequence and only the hidden state crosses workgroups:
//
//   workgroup  = (direction d, batch tile of 32*RT clips, unit tile of 8 hidden units)      1 per CU, all co-resident
//   LDS        = W_hh rows of its 4 gates x 8 units  [32 gate columns][H]  (128 B * H)  + the K-slice reduction buffer
//   per step   : wait until the 64 unit tiles of its (d, batch tile) GROUP have published h_{k-1}  (one flag each)
//                -> A = h_{k-1} rows of the batch tile, straight from L2 into registers (sc1 loads: L1 is never
//                   refreshed by other CUs' stores, MI355X_MICROARCH.md "inter-workgroup visibility")
//                -> gates = pre_k + A * W_slice^T on v_mfma_f32_32x32x2_f32 (K split over the waves, LDS reduction)
//                -> cell non-linearity; c stays in a register for the whole sequence
//                -> publish h_k: write-through (sc1) stores, every wave drains, barrier, one flag store.
//
// Forward exchange buffer layout [parity][group][unit tile q][row][8 units]: a producer writes 1-2 KB contiguous, a
// consumer lane reads one float4 = 4 consecutive k of its row (k = 8q + 4*(lane>>5) + e, the k-slot order of the MFMA),
// 64 lanes covering 1 KB contiguous.  Two parities suffice: h_{k+1} is written only after every group member has
// finished step k, i.e. after it has read h_{k-1}.
//
// Synchronisation is placement independent (agent-scope flags + write-through payload, Guideline 16 R1); every spin is
// bounded by wall clock (s_memrealtime) and reports through a status word instead of hanging.  ONE such launch may be
// in flight per device (all its workgroups must be co-resident).
#include "common.h"

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// Workspace layout: [sticky area PS_STICKY_BYTES][header PS_HDR_BYTES][exchange buffers].
//   sticky word 0 : OR of every timeout since the workspace was allocated; NEVER cleared by a launch (a step issues many
//                   persistent launches on one workspace, the host reads this word once per step / bench run)
//   header word 0 : status of THIS launch ("somebody gave up": the other workgroups stop waiting too)
//   header words PS_FLAG_OFF.. : flags.  The header is zeroed by a memset node in front of every launch.
// A workgroup that gave up poisons everything it writes from then on with NaN, so the failure also reaches the loss and
// the optimiser's check_grad flags instead of leaving plausible garbage.
#define PS_STICKY_BYTES 256
#define PS_FLAG_OFF 16
#define PS_HDR_BYTES 8192
#define PS_TIMEOUT_TICKS 300000000ull   // 3 s of the 100 MHz realtime counter

// test hook (ag_persist_debug): a shorter timeout and one workgroup that never publishes its flags, to exercise the
// give-up path on purpose.  ONE-SHOT and per thread: it arms the NEXT persistent launch issued by the calling thread and
// is consumed by it (like ag_bind_workspace), so a test that dies between arming and launching cannot leave a 2-ms
// timeout behind for the rest of the process.
static thread_local unsigned long long g_ps_timeout = PS_TIMEOUT_TICKS;
static thread_local int g_ps_mute = -1;

extern "C" int ag_persist_debug(int64_t timeout_ticks, int mute_block) {
  g_ps_timeout = timeout_ticks > 0 ? (unsigned long long)timeout_ticks : PS_TIMEOUT_TICKS;
  g_ps_mute = mute_block;
  return AG_OK;
}

struct PersistCtl {
  unsigned* hdr;              // status + flags of this launch
  unsigned* sticky;           // sticky status word
  unsigned long long timeout; // ticks
  int mute;                   // block that never publishes (-1: none)
};

static PersistCtl ps_ctl(void* ws) {
  PersistCtl c;
  c.sticky = (unsigned*)ws;
  c.hdr = (unsigned*)((char*)ws + PS_STICKY_BYTES);
  c.timeout = g_ps_timeout;
  c.mute = g_ps_mute;
  g_ps_timeout = PS_TIMEOUT_TICKS;      // consumed: the hook arms one launch
  g_ps_mute = -1;
  return c;
}

struct PersistDir {
  float* pre;          // [T,B,4H] in: x-projection (+ biases); out: activated gates
  const float* whh;    // [4H,H]
  float* c_all;        // [T+1,B,H]  (c_all[0] is written 0 by the launch)
  const float* cb;     // optional [B,4H] time-invariant part of the pre-activations
};

struct PersistFwdP {
  PersistDir d[2];
  float* y;            // [T,B,ndir*H]  (y16 != 0: the same tensor as bfloat16 - the pointer then addresses 2-byte elements)
  int y16;
  const int64_t* valid;
  float* xbuf;         // exchange buffer
  PersistCtl ctl;
  int T, B, H, ndir;
  int nbt;             // batch tiles
  int ntile;           // H / 8
  int rb;              // AG_PREC_BF16: both operands of the recurrent product rounded to bf16
};

__device__ __forceinline__ bool ps_wait_flags(const PersistCtl& ctl, const unsigned* flags, int n, unsigned want, int lane) {
  unsigned* hdr = ctl.hdr;
  // ONE wave polls the group's flags (lane i <-> flags i, i + 64, ...), relaxed agent-scope loads
  unsigned long long t0 = 0;
  for (unsigned spins = 0;; ++spins) {
    bool ok = true;
    for (int i = lane; i < n; i += 64)
      ok &= __hip_atomic_load(flags + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want;
    if (__all(ok)) return true;
    if ((spins & 63) == 63) {
      if (__hip_atomic_load(hdr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;   // somebody gave up
      const unsigned long long now = __builtin_amdgcn_s_memrealtime();
      if (t0 == 0) t0 = now;
      else if (now - t0 > ctl.timeout) {
        if (lane == 0) {
          __hip_atomic_store(hdr, 0x80000000u | want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_fetch_or(ctl.sticky, 0x80000000u | want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return false;
      }
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

typedef short ps_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned ps_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ ps_bf16x8 ps_pack8(u32x4 lo, u32x4 hi) {      // 8 floats (bit patterns) -> 8 bf16, RNE
  ps_u32x4 r = {ag_pack_bf16(__uint_as_float(lo[0]), __uint_as_float(lo[1])), ag_pack_bf16(__uint_as_float(lo[2]), __uint_as_float(lo[3])),
                ag_pack_bf16(__uint_as_float(hi[0]), __uint_as_float(hi[1])), ag_pack_bf16(__uint_as_float(hi[2]), __uint_as_float(hi[3]))};
  return __builtin_bit_cast(ps_bf16x8, r);
}

// PM (precision mode of the recurrent product): 0 = fp32 operands on the fp32 MFMA; 1 = AG_PREC_BF16 with operands
// rounded in registers in front of the fp32 MFMA (shapes whose K slices are not multiples of 16); 2 = AG_PREC_BF16 on
// v_mfma_f32_32x32x16_bf16: W_hh slice in LDS as bf16 (8 k per 16-byte piece), h rows packed on load.
template <int RT, int PM>      // 32-clip row tiles per workgroup
__global__ __launch_bounds__(512) void lstm_persist_fwd_kernel(const PersistFwdP p) {
  constexpr bool RB = PM == 1;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NKS = 8 / RT;                 // K slices (waves per row tile)
  constexpr int ROWS = 32 * RT;
  const int H = p.H, B = p.B, T = p.T, ntile = p.ntile;
  float* wl = smem;                           // [ntile][2][32][4]
  float* red = smem + (size_t)32 * H;         // [8 waves][1024]
  __shared__ int s_dead;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  if (tid == 0) s_dead = 0;
  const int ngroups = p.ndir * p.nbt;
  const int grp = blockIdx.x % ngroups, ut = blockIdx.x / ngroups;    // a group's blocks share blockIdx % ngroups
  const int dir = grp / p.nbt, bt = grp % p.nbt;
  const PersistDir& D = p.d[dir];
  const int u0 = ut * 8, row0 = bt * ROWS;

  // ---- W_hh slice -> LDS, once: column j = gate (j>>3), unit u0 + (j&7);  element (q, hslot, j, e) = W[row(j)][8q+4hslot+e]
  if (PM == 2) {      // bf16 image: piece (q, j) = W[row(j)][8q .. 8q+7] as 8 bf16
    const int k8n = H >> 3;
    for (int idx = tid; idx < 32 * k8n; idx += 512) {
      const int j = idx / k8n, q = idx - j * k8n;
      const float* src = D.whh + (int64_t)((j >> 3) * H + u0 + (j & 7)) * H + 8 * q;
      const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
      ps_u32x4 r = {ag_pack_bf16(a[0], a[1]), ag_pack_bf16(a[2], a[3]), ag_pack_bf16(b[0], b[1]), ag_pack_bf16(b[2], b[3])};
      *reinterpret_cast<ps_u32x4*>(wl + ((size_t)q * 32 + j) * 4) = r;
    }
  } else {
    const int k4n = H >> 2;
    for (int idx = tid; idx < 32 * k4n; idx += 512) {
      const int j = idx / k4n, k4 = idx - j * k4n;
      const f32x4 v = ag_rbf4_if(*reinterpret_cast<const f32x4*>(D.whh + (int64_t)((j >> 3) * H + u0 + (j & 7)) * H + 4 * k4), RB);
      *reinterpret_cast<f32x4*>(wl + ((size_t)((k4 >> 1) * 2 + (k4 & 1)) * 32 + j) * 4) = v;
    }
  }
  __syncthreads();

  unsigned* flags = p.ctl.hdr + PS_FLAG_OFF + grp * ntile;
  const int64_t xg = (int64_t)ntile * ROWS * 8;                       // floats per (parity, group)
  __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(p.xbuf, 0, (int)(2 * (int64_t)ngroups * xg * 4), 0x00020000);

  // epilogue role: thread <-> (clip row, unit) for the whole sequence; c and h live in registers
  const int erow = tid >> 3, euu = tid & 7;
  const bool ethread = tid < ROWS * 8;
  const int em = row0 + erow, eu = u0 + euu;
  const bool epi = ethread && em < B;
  const int64_t vlen = (epi && p.valid) ? p.valid[em] : ((int64_t)1 << 60);
  float creg = 0.f, hreg = 0.f;
  // accumulator element of (row i, column j) in a 32x32 tile: lane = j + 32*((i>>2)&1), e = (i&3) + 4*(i>>3)
  const int ei = erow & 31, ert = erow >> 5;
  const int ee = (ei & 3) + 4 * (ei >> 3), ehq = (ei >> 2) & 1;

  // MFMA role
  const int rt = RT == 2 ? (wid & 1) : 0, ks = RT == 2 ? (wid >> 1) : wid;
  const int QW = ntile / NKS, q0w = ks * QW;
  bool alive = true;

  // The epilogue's operands.  The pre-activations of step k + 1 are requested at the END of step k, after the flag and BEFORE the
  // trailing gate / cell / output stores, and nothing is written at the top of the loop: the ISA of the previous form began every
  // step with s_waitcnt vmcnt(3) ... vmcnt(0) in front of the re-initialisation of these registers, i.e. with a wait for the
  // previous step's trailing stores to COMPLETE - the very stores that were moved behind the flag so that nobody would wait for
  // them (profiles/r04_lstm_fwd_isa_skeleton.txt).  The time-invariant part is read once.
  bool sv_pad = false;
  float sv_ig = 0.f, sv_fg = 0.f, sv_gg = 0.f, sv_og = 0.f, sv_y = 0.f;
  float pre4[4] = {0.f, 0.f, 0.f, 0.f}, cb4[4] = {0.f, 0.f, 0.f, 0.f};
  auto load_pre = [&](int kk) {
    if (epi) {
      const int tt = dir == 0 ? kk : T - 1 - kk;
      const float* pr = D.pre + ((int64_t)tt * B + em) * 4 * H + eu;
      pre4[0] = pr[0]; pre4[1] = pr[H]; pre4[2] = pr[2 * H]; pre4[3] = pr[3 * H];
    }
  };
  if (epi && D.cb) {
    const float* cb = D.cb + (int64_t)em * 4 * H + eu;
    cb4[0] = cb[0]; cb4[1] = cb[H]; cb4[2] = cb[2 * H]; cb4[3] = cb[3 * H];
  }
  load_pre(0);
  for (int k = 0; k < T; ++k) {
    const int t = dir == 0 ? k : T - 1 - k;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    if (k > 0) {
      // (polling from wave 7, which with 32-clip tiles has no epilogue loads or stores in flight for the poll's in-order
      // s_waitcnt vmcnt(0) to retire behind, measured the same: 7.9 / 5.6 vs 8.1 / 5.5 us per step)
      if (wid == 0 && alive) {
        alive = ps_wait_flags(p.ctl, flags, ntile, (unsigned)k, lane);
        if (!alive) s_dead = 1;
      }
      __syncthreads();
      const int par = (k - 1) & 1;
      // byte offset of (q, row, 4*hh) in the exchange buffer
      const unsigned abase = (unsigned)((((int64_t)(par * ngroups + grp) * xg) + (int64_t)(rt * 32 + l31) * 8 + 4 * hh) * 4);
      const float* wrow = wl + ((size_t)hh * 32 + l31) * 4;
      if (PM == 2) {
        // one MFMA per 16 k: lane half hh takes units 8*(2Q+hh) .. +7 of its row = one 32-byte row of tile 2Q+hh
        const unsigned ab2 = (unsigned)((((int64_t)(par * ngroups + grp) * xg) + (int64_t)(rt * 32 + l31) * 8) * 4);
        const float* w2 = wl + (size_t)l31 * 4;
        for (int qb = 0; qb < QW; qb += 16) {
          u32x4 a0[8], a1[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int q = q0w + min(qb + 2 * i, QW - 2) + hh;
            a0[i] = __builtin_amdgcn_raw_buffer_load_b128(xr, ab2 + (unsigned)(q * ROWS * 32), 0, 16);
            a1[i] = __builtin_amdgcn_raw_buffer_load_b128(xr, ab2 + (unsigned)(q * ROWS * 32) + 16u, 0, 16);
          }
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            if (qb + 2 * i < QW) {
              const int q = q0w + qb + 2 * i + hh;
              const ps_bf16x8 b = *reinterpret_cast<const ps_bf16x8*>(w2 + (size_t)q * 128);
              acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ps_pack8(a0[i], a1[i]), b, acc, 0, 0, 0);
            }
          }
        }
      } else
      for (int qb = 0; qb < QW; qb += 8) {
        u32x4 a[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int q = q0w + min(qb + i, QW - 1);
          a[i] = __builtin_amdgcn_raw_buffer_load_b128(xr, abase + (unsigned)(q * ROWS * 32), 0, 16 /* sc1 */);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (qb + i < QW) {
            const int q = q0w + qb + i;
            const f32x4 b = *reinterpret_cast<const f32x4*>(wrow + (size_t)q * 256);
#pragma unroll
            for (int e = 0; e < 4; ++e)
              acc = __builtin_amdgcn_mfma_f32_32x32x2f32(RB ? ag_rbf(__uint_as_float(a[i][e])) : __uint_as_float(a[i][e]), b[e], acc, 0, 0, 0);
          }
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) red[wid * 1024 + e * 64 + lane] = acc[e];
    __syncthreads();
    if (epi) {
      float g4[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float s = pre4[g] + cb4[g];
        if (k > 0) {
#pragma unroll
          for (int s_ = 0; s_ < NKS; ++s_) {
            const int w = RT == 2 ? (ert + 2 * s_) : s_;
            s += red[w * 1024 + ee * 64 + 32 * ehq + 8 * g + euu];
          }
        }
        g4[g] = s;
      }
      const bool padded = t >= vlen;
      float yv = 0.f, ig = 0.f, fg = 0.f, gg = 0.f, og = 0.f;
      if (!padded) {
        ig = ag_sigmoid(g4[0]); fg = ag_sigmoid(g4[1]); gg = tanhf(g4[2]); og = ag_sigmoid(g4[3]);
        creg = fg * creg + ig * gg;
        hreg = og * tanhf(creg);
        yv = hreg;
      }
      if (s_dead) { creg = hreg = yv = ig = __builtin_nanf(""); }     // a wait timed out: poison instead of garbage
      // publish h_k FIRST (write-through); the stores nobody waits for (gates, cell, output) go out after the flag
      if (k + 1 < T) {
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(hreg), xr,
            (unsigned)(((int64_t)((k & 1) * ngroups + grp) * xg + ((int64_t)ut * ROWS + erow) * 8 + euu) * 4), 0, 16 /* sc1 */);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // EVERY storing wave drains its write-through stores
      }
      sv_pad = padded; sv_ig = ig; sv_fg = fg; sv_gg = gg; sv_og = og; sv_y = yv;
    } else if (ethread && k + 1 < T) {
      // rows past the batch: keep the exchange buffer defined (the consumers' MFMA rows that read it are never stored)
      __builtin_amdgcn_raw_buffer_store_b32(0u, xr,
          (unsigned)(((int64_t)((k & 1) * ngroups + grp) * xg + ((int64_t)ut * ROWS + erow) * 8 + euu) * 4), 0, 16);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (k + 1 < T) {
      __syncthreads();
      if (tid == 0 && (int)blockIdx.x != p.ctl.mute)
        __hip_atomic_store(flags + ut, (unsigned)(k + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      load_pre(k + 1);
    }
    if (epi) {
      if (!sv_pad) {
        float* pr = D.pre + ((int64_t)t * B + em) * 4 * H + eu;
        pr[0] = sv_ig; pr[H] = sv_fg; pr[2 * H] = sv_gg; pr[3 * H] = sv_og;
      }
      if (k == 0) D.c_all[(int64_t)em * H + eu] = 0.f;         // c_0 = 0: written here, so the caller need not fill it
      D.c_all[((int64_t)(k + 1) * B + em) * H + eu] = creg;
      const int64_t yi = ((int64_t)t * B + em) * p.ndir * H + (int64_t)dir * H + eu;
      if (p.y16) reinterpret_cast<unsigned short*>(p.y)[yi] = (unsigned short)(ag_pack_bf16(sv_y, sv_y) & 0xFFFFu);
      else p.y[yi] = sv_y;
    }
  }
}

static bool persist_shape_ok(int B, int H, int ndir, int n_cu, int* rt_out, int* nbt_out) {
  if (H % 64 != 0 || H < 64 || H > 768 || B < 1) return false;
  if (n_cu > 256) n_cu = 256;
  const int ntile = H / 8;
  // one workgroup per CU, all co-resident: take 64-clip tiles when 32-clip tiles would not fit the chip
  for (int rt = 1; rt <= 2; ++rt) {
    const int nbt = ag_cdiv(B, 32 * rt);
    if ((int64_t)ndir * nbt * ntile <= n_cu && ndir * nbt * ntile <= (PS_HDR_BYTES / 4 - PS_FLAG_OFF)) {
      *rt_out = rt;
      *nbt_out = nbt;
      return true;
    }
  }
  return false;
}

extern "C" int ag_lstm_persist_ok(int B, int H, int ndir, int n_cu) {
  int rt, nbt;
  return persist_shape_ok(B, H, ndir, n_cu, &rt, &nbt) ? 1 : 0;
}

extern "C" int64_t ag_lstm_persist_ws_bytes(int B, int H, int ndir) {
  // header + 2 parities of the padded hidden state, both directions (64-clip padding covers both tilings)
  return PS_STICKY_BYTES + PS_HDR_BYTES + (int64_t)2 * ndir * ag_roundup(B, 64) * H * 4;
}

// Whole (bi)directional layer forward in ONE launch.  Same tensors as ag_lstm_seq_fwd (lstm_step.hip) minus the
// hidden-state scratch; `ws` = ag_lstm_persist_ws_bytes() bytes of device memory, 16-byte aligned, used by no other
// launch in flight.  `n_cu`: compute units of the device (the grid must be co-resident).
extern "C" int ag_lstm_seq_fwd_persist(float* const* pre, const float* const* whh, float* const* c_all, void* y, int y_bf16,
                                       const int64_t* valid_i64, const float* const* static_pre, void* ws,
                                       int64_t ws_bytes, int T, int B, int H, int ndir, int n_cu, void* stream) {
  AG_REQUIRE(pre && whh && c_all && y && ws, "ag_lstm_seq_fwd_persist: null tensor");
  AG_REQUIRE(ndir == 1 || ndir == 2, "ag_lstm_seq_fwd_persist: ndir must be 1 or 2");
  AG_REQUIRE(T > 0, "ag_lstm_seq_fwd_persist: T must be positive");
  int rt = 0, nbt = 0;
  if (!persist_shape_ok(B, H, ndir, n_cu, &rt, &nbt)) {
    ag_set_error("ag_lstm_seq_fwd_persist: shape B=%d H=%d ndir=%d does not fit %d CUs", B, H, ndir, n_cu);
    return AG_ERR_UNSUPPORTED;
  }
  AG_REQUIRE(ws_bytes >= ag_lstm_persist_ws_bytes(B, H, ndir) && ((uintptr_t)ws & 15) == 0,
             "ag_lstm_seq_fwd_persist: workspace too small or misaligned");
  for (int d = 0; d < ndir; ++d) AG_REQUIRE(((uintptr_t)whh[d] & 15) == 0, "ag_lstm_seq_fwd_persist: W_hh must be 16-B aligned");
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync((char*)ws + PS_STICKY_BYTES, 0, PS_HDR_BYTES, st) != hipSuccess) {
    ag_set_error("ag_lstm_seq_fwd_persist: memset failed");
    return AG_ERR_LAUNCH;
  }
  PersistFwdP p;
  for (int d = 0; d < 2; ++d) {
    const int s = d < ndir ? d : 0;
    p.d[d].pre = pre[s]; p.d[d].whh = whh[s]; p.d[d].c_all = c_all[s];
    p.d[d].cb = static_pre ? static_pre[s] : nullptr;
  }
  p.y = (float*)y; p.y16 = y_bf16 ? 1 : 0; p.valid = valid_i64;
  p.ctl = ps_ctl(ws);
  p.xbuf = (float*)((char*)ws + PS_STICKY_BYTES + PS_HDR_BYTES);
  p.T = T; p.B = B; p.H = H; p.ndir = ndir; p.nbt = nbt; p.ntile = H / 8;
  p.rb = ag_precision() == AG_PREC_BF16;
  const size_t lds = ((size_t)32 * H + 8 * 1024) * sizeof(float);
  const int grid = ndir * nbt * p.ntile;
  // bf16 mode: the bf16 MFMA needs 16-k steps inside a wave's K slice (H/8 unit tiles over 8/rt waves, 2 tiles a step)
  const int pm = !p.rb ? 0 : ((p.ntile / (8 / rt)) % 2 == 0 ? 2 : 1);
  void (*kern)(const PersistFwdP) =
      rt == 1 ? (pm == 0 ? lstm_persist_fwd_kernel<1, 0> : pm == 1 ? lstm_persist_fwd_kernel<1, 1> : lstm_persist_fwd_kernel<1, 2>)
              : (pm == 0 ? lstm_persist_fwd_kernel<2, 0> : pm == 1 ? lstm_persist_fwd_kernel<2, 1> : lstm_persist_fwd_kernel<2, 2>);
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, p);
  AG_CHECK_LAUNCH("ag_lstm_seq_fwd_persist");
  return AG_OK;
}

// ------------------------------------------------------------------------------------------
// Backward through time as ONE persistent launch.
//
//   dh_k[m,u] = sum_kk dgates_{k+1}[m,kk] * W_hh[kk,u]  (+ pass-through of padded rows),   K = 4H
//   (dgates_k, dc, pass-through) = cell backward of step k
//
// The contraction runs over ALL 4H gate columns, so a workgroup that owns 32 hidden units needs the W_hh slice
// [4H x 32] = 256 KB at H = 512: more than LDS, but not more than the REGISTER FILE (512 KB per CU).  Each of the
// 8 waves keeps its K slice of that panel in VGPRs for the whole sequence (4H/8 k-values x 32 units = 128 registers
// per lane at H = 512) as ready-made MFMA B operands; nothing but dgates is read per step.
//
//   workgroup = (direction, 16-clip tile, 32-unit tile); group = the H/32 unit tiles of one (direction, clip tile)
//   per step  : wait for the group's flags -> A = dgates_{k+1}[16 clips, K slice] (sc1 loads, straight from L2)
//               -> v_mfma_f32_16x16x4_f32 against the resident panel -> LDS sum of the 8 K slices
//               -> cell backward for its (clip, unit) pairs; dc and the pass-through stay in registers
//               -> dgates_k written THROUGH (sc1) into the output tensor itself, which is also the exchange
//                  buffer (one slot per step: no parity), every wave drains, barrier, flag.
// ------------------------------------------------------------------------------------------
struct PersistBwdDir {
  const float* ga;      // [T,B,4H] activated gates
  const float* whh;     // [4H,H]
  const float* c_all;   // [T+1,B,H]
  float* dgates;        // [T,B,4H] out (and exchange)
  float* dgsum;         // optional [B,4H] out: sum over time of dgates (what the biases and a time-invariant input see)
  unsigned short* dg16; // optional [T,B,4H] out (row pitch dg16_ld): dgates as bfloat16, the operand of the gradient products
};

struct PersistBwdP {
  PersistBwdDir d[2];
  const float* dy;      // [T,B,ndir*H]  (dy16 != 0: stored as bfloat16, the pointer addresses 2-byte elements)
  int dy16;
  int dg16_ld;          // row pitch of dg16 in elements (>= 4H: both directions may share one [T,B,ndir*4H] tensor)
  const int64_t* valid;
  PersistCtl ctl;
  int T, B, H, ndir;
  int nbt;              // 16-clip tiles
  int ntile;            // H / 32
  int rb;               // AG_PREC_BF16: both operands of the recurrent product rounded to bf16
};

typedef short ps_bf16x8b __attribute__((ext_vector_type(8)));

// PM: 0 = fp32; 2 = AG_PREC_BF16 on v_mfma_f32_16x16x32_bf16 (the W_hh panel in registers as bf16: half the VGPRs)
template <int NU, int PM>       // 16-k units per wave: 4H = 8 waves * NU * 16
__global__ __launch_bounds__(512) void lstm_persist_bwd_kernel(const PersistBwdP p) {
  constexpr bool RB = false;
  __shared__ float red[8 * 512];
  __shared__ int s_dead;
  const int H = p.H, B = p.B, T = p.T, ntile = p.ntile;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  if (tid == 0) s_dead = 0;
  __syncthreads();
  const int ngroups = p.ndir * p.nbt;
  const int grp = blockIdx.x % ngroups, ut = blockIdx.x / ngroups;
  const int dir = grp / p.nbt, bt = grp % p.nbt;
  const PersistBwdDir& D = p.d[dir];
  const int n0 = ut * 32, m0 = bt * 16;
  const int64_t BG = (int64_t)B * 4 * H, BH = (int64_t)B * H;

  // ---- resident weight panel: wreg[u][e][c] = W_hh[(wid*NU + u)*16 + 4g + e][n0 + 16c + li]
  constexpr int NW = PM == 2 ? 1 : NU;        // (the fp32 panel is not allocated in bf16 mode)
  float wreg[NW][4][2];
  if (PM != 2) {
#pragma unroll
    for (int u = 0; u < NW; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int c = 0; c < 2; ++c)
          wreg[u][e][c] = ag_rbf_if(D.whh[(int64_t)((wid * NU + u) * 16 + 4 * g + e) * H + n0 + 16 * c + li], RB);
  }
  // bf16 panel: 32-k steps; wbf[U][c] = W_hh[(wid*NU/2 + U)*32 + 8g + 0..7][n0 + 16c + li]
  constexpr int NU2 = PM == 2 ? NU / 2 : 1;
  ps_bf16x8b wbf[NU2][2];
  if (PM == 2) {
#pragma unroll
    for (int U = 0; U < NU2; ++U)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const float* src = D.whh + (int64_t)((wid * NU2 + U) * 32 + 8 * g) * H + n0 + 16 * c + li;
        ps_u32x4 r = {ag_pack_bf16(src[0], src[H]), ag_pack_bf16(src[2 * (int64_t)H], src[3 * (int64_t)H]),
                      ag_pack_bf16(src[4 * (int64_t)H], src[5 * (int64_t)H]), ag_pack_bf16(src[6 * (int64_t)H], src[7 * (int64_t)H])};
        wbf[U][c] = __builtin_bit_cast(ps_bf16x8b, r);
      }
  }

  unsigned* flags = p.ctl.hdr + PS_FLAG_OFF + grp * ntile;
  // epilogue role: thread <-> (clip row, unit)
  const int erow = tid >> 5, eun = tid & 31;
  const int em = m0 + erow, eu = n0 + eun;
  const bool epi = em < B;
  const int64_t vlen = (epi && p.valid) ? p.valid[em] : ((int64_t)1 << 60);
  float dcn = 0.f, dpass = 0.f;
  float sum0 = 0.f, sum1 = 0.f, sum2 = 0.f, sum3 = 0.f;      // time sums of this thread's four dgates (fixed order: k = T-1 .. 0)
  // A operand row of this lane (clamped: rows past the batch only feed their own, unwritten outputs)
  const int arow = min(m0 + li, B - 1);
  const unsigned aoff = (unsigned)(((int64_t)arow * 4 * H + (wid * NU) * 16 + 4 * g) * 4);
  bool alive = true;

  for (int k = T - 1; k >= 0; --k) {
    const int t = dir == 0 ? k : T - 1 - k;
    float ig = 0.f, fg = 0.f, gg = 0.f, og = 0.f, cp = 0.f, cn = 0.f, dyv = 0.f;
    if (epi) {
      const float* gr = D.ga + (int64_t)t * BG + (int64_t)em * 4 * H + eu;
      ig = gr[0]; fg = gr[H]; gg = gr[2 * H]; og = gr[3 * H];
      cp = D.c_all[(int64_t)k * BH + (int64_t)em * H + eu];
      cn = D.c_all[(int64_t)(k + 1) * BH + (int64_t)em * H + eu];
      const int64_t yi = ((int64_t)t * B + em) * p.ndir * H + (int64_t)dir * H + eu;
      dyv = p.dy16 ? __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(p.dy)[yi] << 16) : p.dy[yi];
    }
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    if (k < T - 1) {
      if (wid == 0 && alive) {
        alive = ps_wait_flags(p.ctl, flags, ntile, (unsigned)(T - 1 - k), lane);
        if (!alive) s_dead = 1;
      }
      __syncthreads();
      const int tn = dir == 0 ? k + 1 : T - 2 - k;
      __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc(D.dgates + (int64_t)tn * BG, 0, (int)(BG * 4), 0x00020000);
      if (PM == 2) {
        // lane (clip li, k slot g) takes k = 32U + 8g .. +7 of its dgates row (32 contiguous bytes)
        const unsigned aoff2 = (unsigned)(((int64_t)arow * 4 * H + (wid * NU2) * 32 + 8 * g) * 4);
        u32x4 a0[NU2], a1[NU2];
#pragma unroll
        for (int U = 0; U < NU2; ++U) {
          a0[U] = __builtin_amdgcn_raw_buffer_load_b128(ar, aoff2 + (unsigned)(U * 128), 0, 16);
          a1[U] = __builtin_amdgcn_raw_buffer_load_b128(ar, aoff2 + (unsigned)(U * 128) + 16u, 0, 16);
        }
#pragma unroll
        for (int U = 0; U < NU2; ++U) {
          const ps_bf16x8b av = __builtin_bit_cast(ps_bf16x8b, ps_pack8(a0[U], a1[U]));
#pragma unroll
          for (int c = 0; c < 2; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, wbf[U][c], acc[c], 0, 0, 0);
        }
      }
      constexpr int UB = NU < 8 ? NU : 8;
#pragma unroll
      for (int ub = 0; ub < (PM == 2 ? 0 : NU); ub += UB) {
        u32x4 a[UB];
#pragma unroll
        for (int i = 0; i < UB; ++i) a[i] = __builtin_amdgcn_raw_buffer_load_b128(ar, aoff + (unsigned)((ub + i) * 64), 0, 16);
#pragma unroll
        for (int i = 0; i < UB; ++i) {
          float av[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) av[e] = RB ? ag_rbf(__uint_as_float(a[i][e])) : __uint_as_float(a[i][e]);
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int c = 0; c < 2; ++c)
              acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], wreg[PM == 2 ? 0 : ub + i][e][c], acc[c], 0, 0, 0);
        }
      }
    }
    // C layout of a 16x16 tile: col = lane & 15, row = 4 * (lane >> 4) + e
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) red[wid * 512 + (4 * g + e) * 32 + 16 * c + li] = acc[c][e];
    __syncthreads();
    if (epi) {
      float dhf = dpass;
      if (k < T - 1) {
#pragma unroll
        for (int w = 0; w < 8; ++w) dhf += red[w * 512 + tid];
      }
      float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
      if (t >= vlen) {
        dpass = dhf;          // padded step: gradient passes through h and c unchanged
      } else {
        const float dhv = dhf + dyv;
        const float tc = tanhf(cn);
        const float dc = dcn + dhv * og * (1.f - tc * tc);
        d0 = dc * gg * ig * (1.f - ig);
        d1 = dc * cp * fg * (1.f - fg);
        d2 = dc * ig * (1.f - gg * gg);
        d3 = dhv * tc * og * (1.f - og);
        dcn = dc * fg;
        dpass = 0.f;
      }
      if (s_dead) { d0 = d1 = d2 = d3 = dcn = __builtin_nanf(""); }   // a wait timed out: poison instead of garbage
      sum0 += d0; sum1 += d1; sum2 += d2; sum3 += d3;
      if (D.dg16) {
        unsigned short* q16 = D.dg16 + ((int64_t)t * B + em) * p.dg16_ld + eu;
        q16[0] = (unsigned short)(ag_pack_bf16(d0, d0) & 0xFFFFu); q16[H] = (unsigned short)(ag_pack_bf16(d1, d1) & 0xFFFFu);
        q16[2 * H] = (unsigned short)(ag_pack_bf16(d2, d2) & 0xFFFFu); q16[3 * H] = (unsigned short)(ag_pack_bf16(d3, d3) & 0xFFFFu);
      }
      __amdgpu_buffer_rsrc_t orr = __builtin_amdgcn_make_buffer_rsrc(D.dgates + (int64_t)t * BG, 0, (int)(BG * 4), 0x00020000);
      const unsigned o = (unsigned)(((int64_t)em * 4 * H + eu) * 4);
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(d0), orr, o, 0, 16);
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(d1), orr, o + (unsigned)(H * 4), 0, 16);
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(d2), orr, o + (unsigned)(2 * H * 4), 0, 16);
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(d3), orr, o + (unsigned)(3 * H * 4), 0, 16);
    }
    if (k > 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // EVERY storing wave drains its write-through stores
      __syncthreads();
      if (tid == 0 && (int)blockIdx.x != p.ctl.mute)
        __hip_atomic_store(flags + ut, (unsigned)(T - k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (epi && D.dgsum) {
    float* q = D.dgsum + (int64_t)em * 4 * H + eu;
    q[0] = sum0; q[H] = sum1; q[2 * H] = sum2; q[3 * H] = sum3;
  }
}

static bool persist_bwd_shape_ok(int B, int H, int ndir, int n_cu) {
  if (!(H == 64 || H == 128 || H == 256 || H == 512) || B < 1) return false;
  if (n_cu > 256) n_cu = 256;
  const int64_t grid = (int64_t)ndir * ag_cdiv(B, 16) * (H / 32);
  return grid <= n_cu && grid <= (PS_HDR_BYTES / 4 - PS_FLAG_OFF);
}

extern "C" int ag_lstm_persist_bwd_ok(int B, int H, int ndir, int n_cu) {
  return persist_bwd_shape_ok(B, H, ndir, n_cu) ? 1 : 0;
}

// Whole layer backward through time in ONE launch; tensors as for ag_lstm_seq_bwd (no scratch state: dc and the
// pass-through term stay in registers).  `ws`: >= 8 KiB (status + flags), zeroed by a memset node in front.
extern "C" int ag_lstm_seq_bwd_persist(const float* const* gates, const float* const* whh, const float* const* c_all,
                                       const void* dy, int dy_bf16, float* const* dgates, float* const* dgsum,
                                       uint16_t* const* dg16, int dg16_ld, const int64_t* valid_i64, void* ws,
                                       int64_t ws_bytes, int T, int B, int H, int ndir, int n_cu, void* stream) {
  AG_REQUIRE(gates && whh && c_all && dy && dgates && ws, "ag_lstm_seq_bwd_persist: null tensor");
  AG_REQUIRE(ndir == 1 || ndir == 2, "ag_lstm_seq_bwd_persist: ndir must be 1 or 2");
  AG_REQUIRE(T > 0, "ag_lstm_seq_bwd_persist: T must be positive");
  if (!persist_bwd_shape_ok(B, H, ndir, n_cu)) {
    ag_set_error("ag_lstm_seq_bwd_persist: shape B=%d H=%d ndir=%d does not fit %d CUs", B, H, ndir, n_cu);
    return AG_ERR_UNSUPPORTED;
  }
  AG_REQUIRE(ws_bytes >= PS_STICKY_BYTES + PS_HDR_BYTES && ((uintptr_t)ws & 15) == 0, "ag_lstm_seq_bwd_persist: workspace too small");
  AG_REQUIRE((int64_t)B * 4 * H * 4 < ((int64_t)1 << 31), "ag_lstm_seq_bwd_persist: step slab too large");
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync((char*)ws + PS_STICKY_BYTES, 0, PS_HDR_BYTES, st) != hipSuccess) {
    ag_set_error("ag_lstm_seq_bwd_persist: memset failed");
    return AG_ERR_LAUNCH;
  }
  PersistBwdP p;
  for (int d = 0; d < 2; ++d) {
    const int s = d < ndir ? d : 0;
    p.d[d].ga = gates[s]; p.d[d].whh = whh[s]; p.d[d].c_all = c_all[s]; p.d[d].dgates = dgates[s];
    p.d[d].dgsum = dgsum ? dgsum[s] : nullptr;
    p.d[d].dg16 = dg16 ? dg16[s] : nullptr;
  }
  AG_REQUIRE(!dg16 || dg16_ld >= 4 * H, "ag_lstm_seq_bwd_persist: dg16 row pitch smaller than a row");
  p.dy = (const float*)dy; p.dy16 = dy_bf16 ? 1 : 0; p.dg16_ld = dg16_ld; p.valid = valid_i64; p.ctl = ps_ctl(ws);
  p.T = T; p.B = B; p.H = H; p.ndir = ndir; p.nbt = ag_cdiv(B, 16); p.ntile = H / 32;
  p.rb = ag_precision() == AG_PREC_BF16;
  const int grid = ndir * p.nbt * p.ntile;
  void (*kern)(const PersistBwdP);
  switch (H) {
    case 512: kern = p.rb ? lstm_persist_bwd_kernel<16, 2> : lstm_persist_bwd_kernel<16, 0>; break;
    case 256: kern = p.rb ? lstm_persist_bwd_kernel<8, 2> : lstm_persist_bwd_kernel<8, 0>; break;
    case 128: kern = p.rb ? lstm_persist_bwd_kernel<4, 2> : lstm_persist_bwd_kernel<4, 0>; break;
    default:  kern = p.rb ? lstm_persist_bwd_kernel<2, 2> : lstm_persist_bwd_kernel<2, 0>; break;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 0, st, p);
  AG_CHECK_LAUNCH("ag_lstm_seq_bwd_persist");
  return AG_OK;
}

// ------------------------------------------------------------------------------------------
// The Generator's recurrent front (audiogan.py:428-460, one LSTMCell layer) as ONE persistent launch:
//   frame t:  gates = pre_t + [x_{t-1} | h_{t-1}] [W_x | W_hh]^T ;  (c_t, h_t) = cell(gates) ;  x_t = tanh(h_t W_p^T + b_p)
// (pre_t = the z/c columns of W_ih and both biases, one GEMM over all frames beforehand; the stop head is one GEMM over
// all frames afterwards).  Per frame the launch-per-op path costs a fused step kernel plus a projection kernel, each
// re-streaming its weights (21 MB + 1 MB); here every weight stays in REGISTERS for all T frames:
//
//   workgroup (rt, ut): 32 clips x 8 hidden units (32 gate columns); its 8 waves split K: each holds S/8 rows of the
//   [W_hh] panel and fs/8 rows of the [W_x] panel as MFMA B operands (80 VGPRs at S = 1024, fs = 256).
//   phase A, frame t: h_{t-1} part of the product as soon as the group's h flags are up, x_{t-1} part when the
//   projection tiles are up, LDS sum over the waves, cell, h_t published (write-through) + flag.
//   phase B, frame t (workgroups ut < 2*fs/16 only): one [16 clips x 16 frame samples] tile of x_t = tanh(h_t W_p^T + b):
//   waits for all h_t of its clips, v_mfma_f32_16x16x4_f32 against its resident W_p panel, publishes x_t + flag.
//   The h part of frame t+1 (80 % of the MFMAs) overlaps phase B of frame t on the other workgroups.
// Exchange buffers ([parity][rt][8-unit tile][32 clips][8], as in the layer kernels) hold h and x; two parities are
// enough (h_{t+1} / x_{t+1} are written only after every reader of h_{t-1} / x_{t-1} has finished frame t).
// ------------------------------------------------------------------------------------------
struct FrontFwdP {
  float* gates;        // [T,B,4S] in: pre; out: activated gates
  const float* wx;     // W_ih[:, :fs]   [4S, ldwx]
  const float* whh;    // [4S, S]
  const float* wp;     // [fs, S]
  const float* bp;     // [fs]
  float* hs;           // [T,B,S]
  float* cs;           // [T+1,B,S]  (cs[0] is written 0 by the launch)
  float* x;            // [B, T*fs], row pitch ldx (the caller may hand over channel 0 of the conv trunk's slab)
  float* xt;           // optional [T,B,fs]: the same frames time-major (what the weight-gradient products read)
  int64_t ldx;
  float* gh;           // GRU cell only: [T,B,3S], the n slot receives W_hn h + b_hn (what the backward needs)
  const float* bhn;    // GRU cell only: b_hh[2S:3S]
  float* hx;           // exchange: h   [2][nrt][S/8][32][8]
  float* xx;           // exchange: x   [2][nrt][fs/8][32][8]
  PersistCtl ctl;
  int T, B, ldwx, nrt, rb;
};

// PM: 0 = as stored / operands rounded in registers when p.rb (fp32 MFMA); 2 = AG_PREC_BF16 on the bf16 MFMAs (panels
// in registers as bf16: 8 k per 4 VGPRs)
// CELL: 0 = LSTMCell (4 gate columns per unit: i f g o).  1 = GRU cell (BASELINE configs[3]; PM = 0 only): the 4 column
// blocks of a unit tile are  r | z | n_h | n_x : the h panel carries W_hr, W_hz, W_hn and zeros, the x panel W_xr, W_xz,
// zeros and W_xn, so that ONE accumulator ends up with  r, z pre-activations complete and the two halves of n apart
// (n = tanh(n_x + r * (n_h + b_hn)));  h_t = (1 - z) n + z h_{t-1} stays in a register like the LSTM's cell state.
template <int S, int FS, int PM, int CELL = 0>
__global__ __launch_bounds__(512) void gfront_persist_fwd_kernel(const FrontFwdP p) {
  constexpr int NUT = S / 8;                 // unit tiles = workgroups per row tile
  constexpr int QH = S / 64, QX = FS / 64;   // 8-k groups of the h / x panel per wave
  constexpr int NB = 2 * (FS / 16);          // projection tiles per row tile (2 x 16-clip subtiles)
  constexpr int UP = S / 128;                // 16-k units of W_p per wave
  static_assert(S % 128 == 0 && FS % 64 == 0 && NB <= NUT, "unsupported front shape");
  __shared__ float red[8 * 1024];
  __shared__ int s_dead;
  const int T = p.T, B = p.B;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5, li = lane & 15, g = lane >> 4;
  if (tid == 0) s_dead = 0;
  __syncthreads();
  const int rt = blockIdx.x % p.nrt, ut = blockIdx.x / p.nrt;
  const int u0 = ut * 8, row0 = rt * 32;

  // ---- resident panels (B operands).  Gate column j = gate (j>>3), unit u0 + (j&7).
  constexpr int QHf = PM == 2 ? 1 : QH, QXf = PM == 2 ? 1 : QX, QH2 = PM == 2 ? QH / 2 : 1, QX2 = PM == 2 ? QX / 2 : 1;
  float wh[QHf][4], wxr[QXf][4];
  ps_bf16x8 whb[QH2], wxb[QX2];
  if (PM == 2) {
    const int wrow = (l31 >> 3) * S + u0 + (l31 & 7);
#pragma unroll
    for (int Q = 0; Q < QH2; ++Q) {
      const float* src = p.whh + (int64_t)wrow * S + (wid * QH + 2 * Q + hh) * 8;
      const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
      ps_u32x4 r = {ag_pack_bf16(a[0], a[1]), ag_pack_bf16(a[2], a[3]), ag_pack_bf16(b[0], b[1]), ag_pack_bf16(b[2], b[3])};
      whb[Q] = __builtin_bit_cast(ps_bf16x8, r);
    }
#pragma unroll
    for (int Q = 0; Q < QX2; ++Q) {
      const float* src = p.wx + (int64_t)wrow * p.ldwx + (wid * QX + 2 * Q + hh) * 8;
      const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
      ps_u32x4 r = {ag_pack_bf16(a[0], a[1]), ag_pack_bf16(a[2], a[3]), ag_pack_bf16(b[0], b[1]), ag_pack_bf16(b[2], b[3])};
      wxb[Q] = __builtin_bit_cast(ps_bf16x8, r);
    }
  } else {
    const int cb = l31 >> 3;
    // (GRU: column block 3 of the h panel and block 2 of the x panel are zero; block 3 of the x panel is W_xn)
    const bool hz = CELL == 1 && cb == 3, xz = CELL == 1 && cb == 2;
    const int wrow = cb * S + u0 + (l31 & 7), wrowx = (CELL == 1 && cb == 3 ? 2 : cb) * S + u0 + (l31 & 7);
#pragma unroll
    for (int q = 0; q < QH; ++q) {
      const f32x4 v = hz ? f32x4{0.f, 0.f, 0.f, 0.f}
                         : ag_rbf4_if(*reinterpret_cast<const f32x4*>(p.whh + (int64_t)wrow * S + (wid * QH + q) * 8 + 4 * hh), p.rb);
#pragma unroll
      for (int e = 0; e < 4; ++e) wh[q][e] = v[e];
    }
#pragma unroll
    for (int q = 0; q < QX; ++q) {
      const f32x4 v = xz ? f32x4{0.f, 0.f, 0.f, 0.f}
                         : ag_rbf4_if(*reinterpret_cast<const f32x4*>(p.wx + (int64_t)wrowx * p.ldwx + (wid * QX + q) * 8 + 4 * hh), p.rb);
#pragma unroll
      for (int e = 0; e < 4; ++e) wxr[q][e] = v[e];
    }
  }
  const bool bwg = ut < NB;                  // this workgroup also owns a projection tile
  const int bsub = ut & 1, bcol0 = (ut >> 1) * 16;
  constexpr int UPf = PM == 2 ? 1 : UP, UP2 = PM == 2 ? UP / 2 : 1;
  float wpr[UPf][4];
  ps_bf16x8b wpb[UP2];
  if (PM == 2) {
#pragma unroll
    for (int U = 0; U < UP2; ++U) {
      ps_u32x4 r = {0u, 0u, 0u, 0u};
      if (bwg) {
        const float* src = p.wp + (int64_t)(bcol0 + li) * S + (wid * UP2 + U) * 32 + 8 * g;
        const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
        r = ps_u32x4{ag_pack_bf16(a[0], a[1]), ag_pack_bf16(a[2], a[3]), ag_pack_bf16(b[0], b[1]), ag_pack_bf16(b[2], b[3])};
      }
      wpb[U] = __builtin_bit_cast(ps_bf16x8b, r);
    }
  } else {
#pragma unroll
    for (int u = 0; u < UPf; ++u) {
      const f32x4 v = bwg ? ag_rbf4_if(*reinterpret_cast<const f32x4*>(p.wp + (int64_t)(bcol0 + li) * S + (wid * UP + u) * 16 + 4 * g), p.rb)
                          : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 4; ++e) wpr[u][e] = v[e];
    }
  }

  unsigned* flag_h = p.ctl.hdr + PS_FLAG_OFF + rt * NUT;
  unsigned* flag_x = p.ctl.hdr + PS_FLAG_OFF + p.nrt * NUT + rt * NB;
  const int64_t hgs = (int64_t)NUT * 32 * 8, xgs = (int64_t)(FS / 8) * 32 * 8;     // floats per (parity, rt)
  __amdgpu_buffer_rsrc_t hr = __builtin_amdgcn_make_buffer_rsrc(p.hx, 0, (int)(2 * p.nrt * hgs * 4), 0x00020000);
  __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(p.xx, 0, (int)(2 * p.nrt * xgs * 4), 0x00020000);

  // phase A epilogue role: thread (clip row, unit), tid < 256
  const int erow = tid >> 3, euu = tid & 7;
  const int em = row0 + erow, eu = u0 + euu;
  const bool epi = tid < 256 && em < B;
  const int ee = (erow & 3) + 4 * (erow >> 3), ehq = (erow >> 2) & 1;
  float creg = 0.f;            // LSTM: cell state; GRU: h_{t-1}
  const float ebhn = (CELL == 1 && epi) ? p.bhn[eu] : 0.f;
  // phase B epilogue role: thread (clip row within the 16-row subtile, column), tid < 256
  const int brow = tid >> 4, bcl = tid & 15;
  const int bm = row0 + 16 * bsub + brow;
  const float bbias = (bwg && tid < 256) ? p.bp[bcol0 + bcl] : 0.f;
  bool alive = true;

  // (the pre-activations of frame t + 1 are requested at the end of frame t, after the flag and before the trailing stores, and
  // nothing is written at the top of the loop - as in lstm_persist_fwd_kernel, where the re-initialisation of these registers made
  // every step begin with a wait for the previous step's trailing stores)
  float pre4[4] = {0.f, 0.f, 0.f, 0.f};
  auto load_pre = [&](int tt) {
    if (epi) {
      if (CELL == 1) {
        const float* pr = p.gates + ((int64_t)tt * B + em) * 3 * S + eu;
        pre4[0] = pr[0]; pre4[1] = pr[S]; pre4[2] = pr[2 * S];
      } else {
        const float* pr = p.gates + ((int64_t)tt * B + em) * 4 * S + eu;
        pre4[0] = pr[0]; pre4[1] = pr[S]; pre4[2] = pr[2 * S]; pre4[3] = pr[3 * S];
      }
    }
  };
  load_pre(0);
  for (int t = 0; t < T; ++t) {
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    if (t > 0) {
      const int par = (t - 1) & 1;
      // ---- h part (its flags were already waited for by the projection phase of frame t-1 on projection workgroups)
      if (wid == 0 && alive) {
        alive = ps_wait_flags(p.ctl, flag_h, NUT, (unsigned)t, lane);
        if (!alive) s_dead = 1;
      }
      __syncthreads();
      if (PM == 2) {
        const unsigned ab = (unsigned)(((int64_t)(par * p.nrt + rt) * hgs + (int64_t)l31 * 8) * 4);
        u32x4 a0[QH2], a1[QH2];
#pragma unroll
        for (int Q = 0; Q < QH2; ++Q) {
          const unsigned o = ab + (unsigned)((wid * QH + 2 * Q + hh) * 32 * 32);
          a0[Q] = __builtin_amdgcn_raw_buffer_load_b128(hr, o, 0, 16);
          a1[Q] = __builtin_amdgcn_raw_buffer_load_b128(hr, o + 16u, 0, 16);
        }
#pragma unroll
        for (int Q = 0; Q < QH2; ++Q) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ps_pack8(a0[Q], a1[Q]), whb[Q], acc, 0, 0, 0);
      } else {
        const unsigned ab = (unsigned)(((int64_t)(par * p.nrt + rt) * hgs + (int64_t)l31 * 8 + 4 * hh) * 4);
        u32x4 a[QH];
#pragma unroll
        for (int q = 0; q < QH; ++q) a[q] = __builtin_amdgcn_raw_buffer_load_b128(hr, ab + (unsigned)((wid * QH + q) * 32 * 32), 0, 16);
        if (p.rb) {
#pragma unroll
          for (int q = 0; q < QH; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) a[q][e] = __float_as_uint(ag_rbf(__uint_as_float(a[q][e])));
        }
#pragma unroll
        for (int q = 0; q < QH; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[q][e]), wh[PM == 2 ? 0 : q][e], acc, 0, 0, 0);
      }
      // ---- x part
      if (wid == 0 && alive) {
        alive = ps_wait_flags(p.ctl, flag_x, NB, (unsigned)t, lane);
        if (!alive) s_dead = 1;
      }
      __syncthreads();
      if (PM == 2) {
        const unsigned ab = (unsigned)(((int64_t)(par * p.nrt + rt) * xgs + (int64_t)l31 * 8) * 4);
        u32x4 a0[QX2], a1[QX2];
#pragma unroll
        for (int Q = 0; Q < QX2; ++Q) {
          const unsigned o = ab + (unsigned)((wid * QX + 2 * Q + hh) * 32 * 32);
          a0[Q] = __builtin_amdgcn_raw_buffer_load_b128(xr, o, 0, 16);
          a1[Q] = __builtin_amdgcn_raw_buffer_load_b128(xr, o + 16u, 0, 16);
        }
#pragma unroll
        for (int Q = 0; Q < QX2; ++Q) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ps_pack8(a0[Q], a1[Q]), wxb[Q], acc, 0, 0, 0);
      } else {
        const unsigned ab = (unsigned)(((int64_t)(par * p.nrt + rt) * xgs + (int64_t)l31 * 8 + 4 * hh) * 4);
        u32x4 a[QX];
#pragma unroll
        for (int q = 0; q < QX; ++q) a[q] = __builtin_amdgcn_raw_buffer_load_b128(xr, ab + (unsigned)((wid * QX + q) * 32 * 32), 0, 16);
        if (p.rb) {
#pragma unroll
          for (int q = 0; q < QX; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) a[q][e] = __float_as_uint(ag_rbf(__uint_as_float(a[q][e])));
        }
#pragma unroll
        for (int q = 0; q < QX; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[q][e]), wxr[PM == 2 ? 0 : q][e], acc, 0, 0, 0);
      }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) red[wid * 1024 + e * 64 + lane] = acc[e];
    __syncthreads();
    float hreg = 0.f, ig = 0.f, fg = 0.f, gg = 0.f, og = 0.f;
    if (epi) {
      float g4[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float s = CELL == 1 ? 0.f : pre4[q];
        if (t > 0) {
#pragma unroll
          for (int w = 0; w < 8; ++w) s += red[w * 1024 + ee * 64 + 32 * ehq + 8 * q + euu];
        }
        g4[q] = s;
      }
      if (CELL == 1) {
        // ig / fg / gg = r / z / n;  og = the h half of n incl. its bias (saved for the backward)
        og = g4[2] + ebhn;
        ig = ag_sigmoid(pre4[0] + g4[0]); fg = ag_sigmoid(pre4[1] + g4[1]); gg = tanhf(pre4[2] + g4[3] + ig * og);
        hreg = (1.f - fg) * gg + fg * creg;
        creg = hreg;
      } else {
        ig = ag_sigmoid(g4[0]); fg = ag_sigmoid(g4[1]); gg = tanhf(g4[2]); og = ag_sigmoid(g4[3]);
        creg = fg * creg + ig * gg;
        hreg = og * tanhf(creg);
      }
      if (s_dead) creg = hreg = ig = __builtin_nanf("");             // a wait timed out: poison instead of garbage
    }
    if (tid < 256) {
      // publish h_t (rows past the batch: zeros, so that the exchange buffer stays defined)
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(hreg), hr,
          (unsigned)(((int64_t)((t & 1) * p.nrt + rt) * hgs + ((int64_t)ut * 32 + erow) * 8 + euu) * 4), 0, 16);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (tid == 0 && (int)blockIdx.x != p.ctl.mute)
      __hip_atomic_store(flag_h + ut, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // this frame's pre-activations were consumed above (ig .. og hold what goes back into their place)
    if (t + 1 < T) load_pre(t + 1);
    if (epi) {
      if (CELL == 1) {
        float* pr = p.gates + ((int64_t)t * B + em) * 3 * S + eu;
        pr[0] = ig; pr[S] = fg; pr[2 * S] = gg;
        p.gh[((int64_t)t * B + em) * 3 * S + 2 * S + eu] = og;
      } else {
        float* pr = p.gates + ((int64_t)t * B + em) * 4 * S + eu;
        pr[0] = ig; pr[S] = fg; pr[2 * S] = gg; pr[3 * S] = og;
        if (t == 0) p.cs[(int64_t)em * S + eu] = 0.f;               // c_0 = 0: written here, so the caller need not fill it
        p.cs[((int64_t)(t + 1) * B + em) * S + eu] = creg;
      }
      p.hs[((int64_t)t * B + em) * S + eu] = hreg;
    }
    // ---- phase B: this workgroup's tile of x_t = tanh(h_t W_p^T + b)
    if (bwg) {
      if (wid == 0 && alive) {
        alive = ps_wait_flags(p.ctl, flag_h, NUT, (unsigned)(t + 1), lane);
        if (!alive) s_dead = 1;
      }
      __syncthreads();
      f32x4 pacc = {0.f, 0.f, 0.f, 0.f};
      if (PM == 2) {
        // lane (clip li, k slot g) takes k = 32U + 8g .. +7 = the whole row of unit tile 4U + g
        const unsigned ab = (unsigned)(((int64_t)((t & 1) * p.nrt + rt) * hgs + (int64_t)(16 * bsub + li) * 8) * 4);
        u32x4 a0[UP2], a1[UP2];
#pragma unroll
        for (int U = 0; U < UP2; ++U) {
          const unsigned o = ab + (unsigned)((4 * (wid * UP2 + U) + g) * 32 * 32);
          a0[U] = __builtin_amdgcn_raw_buffer_load_b128(hr, o, 0, 16);
          a1[U] = __builtin_amdgcn_raw_buffer_load_b128(hr, o + 16u, 0, 16);
        }
#pragma unroll
        for (int U = 0; U < UP2; ++U)
          pacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(ps_bf16x8b, ps_pack8(a0[U], a1[U])), wpb[U], pacc, 0, 0, 0);
      } else {
        // A = h_t rows of the 16-clip subtile: lane (clip li, k slot g) takes k = 16u + 4g .. +3 = unit tile 2u + (g>>1), half g&1
        const unsigned ab = (unsigned)(((int64_t)((t & 1) * p.nrt + rt) * hgs + (int64_t)(16 * bsub + li) * 8 + 4 * (g & 1)) * 4);
        u32x4 a[UP];
#pragma unroll
        for (int u = 0; u < UP; ++u)
          a[u] = __builtin_amdgcn_raw_buffer_load_b128(hr, ab + (unsigned)((2 * (wid * UP + u) + (g >> 1)) * 32 * 32), 0, 16);
        if (p.rb) {
#pragma unroll
          for (int u = 0; u < UP; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) a[u][e] = __float_as_uint(ag_rbf(__uint_as_float(a[u][e])));
        }
#pragma unroll
        for (int u = 0; u < UP; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e) pacc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[u][e]), wpr[PM == 2 ? 0 : u][e], pacc, 0, 0, 0);
      }
      // 16x16 tile: col = lane & 15, row = 4 * (lane >> 4) + e   (red is free again: the gate sums were consumed above)
#pragma unroll
      for (int e = 0; e < 4; ++e) red[wid * 256 + (4 * g + e) * 16 + li] = pacc[e];
      __syncthreads();
      if (tid < 256) {
        float v = bbias;
#pragma unroll
        for (int w = 0; w < 8; ++w) v += red[w * 256 + tid];
        v = s_dead ? __builtin_nanf("") : tanhf(v);
        const int col = bcol0 + bcl;
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(bm < B ? v : 0.f), xr,
            (unsigned)(((int64_t)((t & 1) * p.nrt + rt) * xgs + ((int64_t)(col >> 3) * 32 + 16 * bsub + brow) * 8 + (col & 7)) * 4), 0, 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (bm < B) {
          p.x[(int64_t)bm * p.ldx + (int64_t)t * FS + col] = v;
          if (p.xt) p.xt[((int64_t)t * B + bm) * FS + col] = v;
        }
      }
      __syncthreads();
      if (tid == 0) __hip_atomic_store(flag_x + ut, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

static bool front_shape_ok(int B, int S, int fs, int n_cu) {
  if (!((S == 1024 && fs == 256) || (S == 128 && fs == 64))) return false;
  if (B < 1 || B > 64) return false;
  if (n_cu > 256) n_cu = 256;
  return ag_cdiv(B, 32) * (S / 8) <= n_cu;
}

extern "C" int ag_gfront_persist_ok(int B, int S, int fs, int n_cu) { return front_shape_ok(B, S, fs, n_cu) ? 1 : 0; }

extern "C" int64_t ag_gfront_persist_ws_bytes(int B, int S, int fs) {
  return PS_STICKY_BYTES + PS_HDR_BYTES + (int64_t)2 * ag_cdiv(B, 32) * 32 * (S + fs) * 4;
}

// One launch for the whole frame loop of the Generator front (one LSTMCell layer).  gates [T,B,4S]: in = the z/c
// part of the pre-activations + both biases, out = activated gates; w_x = W_ih[:, :fs] (row pitch ldwx), w_hh [4S,S],
// w_p [fs,S], b_p [fs]; outputs hs [T,B,S], cs [T+1,B,S] (cs[0] is written 0 by the launch), x [B,T*fs].  Shapes: ag_gfront_persist_ok.
extern "C" int ag_gfront_fwd_persist(float* gates, const float* w_x, int ldwx, const float* w_hh, const float* w_p,
                                     const float* b_p, float* hs, float* cs, float* x, int64_t ldx, float* xt, void* ws,
                                     int64_t ws_bytes, int T, int B, int S, int fs, int n_cu, void* stream) {
  AG_REQUIRE(gates && w_x && w_hh && w_p && b_p && hs && cs && x && ws, "ag_gfront_fwd_persist: null tensor");
  AG_REQUIRE(T > 0, "ag_gfront_fwd_persist: T must be positive");
  AG_REQUIRE(ldx >= (int64_t)T * fs, "ag_gfront_fwd_persist: x row pitch smaller than a row");
  if (!front_shape_ok(B, S, fs, n_cu)) {
    ag_set_error("ag_gfront_fwd_persist: shape B=%d S=%d fs=%d is not supported on %d CUs", B, S, fs, n_cu);
    return AG_ERR_UNSUPPORTED;
  }
  AG_REQUIRE(ws_bytes >= ag_gfront_persist_ws_bytes(B, S, fs) && ((uintptr_t)ws & 15) == 0,
             "ag_gfront_fwd_persist: workspace too small or misaligned");
  AG_REQUIRE(ldwx % 4 == 0 && (((uintptr_t)w_x | (uintptr_t)w_hh | (uintptr_t)w_p) & 15) == 0,
             "ag_gfront_fwd_persist: weights must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync((char*)ws + PS_STICKY_BYTES, 0, PS_HDR_BYTES, st) != hipSuccess) {
    ag_set_error("ag_gfront_fwd_persist: memset failed");
    return AG_ERR_LAUNCH;
  }
  FrontFwdP p;
  p.gates = gates; p.gh = nullptr; p.bhn = nullptr; p.wx = w_x; p.whh = w_hh; p.wp = w_p; p.bp = b_p; p.hs = hs; p.cs = cs; p.x = x;
  p.ldx = ldx; p.xt = xt;
  p.ctl = ps_ctl(ws);
  p.nrt = ag_cdiv(B, 32);
  p.hx = (float*)((char*)ws + PS_STICKY_BYTES + PS_HDR_BYTES);
  p.xx = p.hx + (int64_t)2 * p.nrt * 32 * S;
  p.T = T; p.B = B; p.ldwx = ldwx; p.rb = ag_precision() == AG_PREC_BF16;
  const int grid = p.nrt * (S / 8);
  if (S == 1024) {
    if (p.rb) hipLaunchKernelGGL((gfront_persist_fwd_kernel<1024, 256, 2>), dim3(grid), dim3(512), 0, st, p);
    else hipLaunchKernelGGL((gfront_persist_fwd_kernel<1024, 256, 0>), dim3(grid), dim3(512), 0, st, p);
  } else {
    hipLaunchKernelGGL((gfront_persist_fwd_kernel<128, 64, 0>), dim3(grid), dim3(512), 0, st, p);   // (FS/8 per wave is odd)
  }
  AG_CHECK_LAUNCH("ag_gfront_fwd_persist");
  return AG_OK;
}

// The GRU-front generator's frame loop (BASELINE configs[3]; the feedback loop of audiogan.py:428-460 with a GRU cell, gate
// order r z n as torch.nn.GRUCell) as ONE persistent launch: the same kernel with the GRU column layout.
//   gates [T,B,3S]  in: W_ih[:, fs:] zc_t + b_ih + (b_hr, b_hz, 0)  (one GEMM over all frames); out: activated (r, z, n)
//   gh    [T,B,3S]  out: only the n slot, W_hn h_{t-1} + b_hn (what ag_gru_cell_bwd reads)
//   w_x = W_ih[:, :fs] (row pitch ldwx), w_hh [3S,S], b_hn [S] = b_hh[2S:], w_p [fs,S], b_p [fs]
//   hs [T,B,S] (h_t), x [B,T*fs].  Shapes as ag_gfront_persist_ok; workspace as ag_gfront_fwd_persist.
extern "C" int ag_grufront_fwd_persist(float* gates, float* gh, const float* w_x, int ldwx, const float* w_hh,
                                       const float* b_hn, const float* w_p, const float* b_p, float* hs, float* x, int64_t ldx,
                                       float* xt, void* ws, int64_t ws_bytes, int T, int B, int S, int fs, int n_cu,
                                       void* stream) {
  AG_REQUIRE(gates && gh && w_x && w_hh && b_hn && w_p && b_p && hs && x && ws, "ag_grufront_fwd_persist: null tensor");
  AG_REQUIRE(T > 0, "ag_grufront_fwd_persist: T must be positive");
  AG_REQUIRE(ldx >= (int64_t)T * fs, "ag_grufront_fwd_persist: x row pitch smaller than a row");
  if (!front_shape_ok(B, S, fs, n_cu)) {
    ag_set_error("ag_grufront_fwd_persist: shape B=%d S=%d fs=%d is not supported on %d CUs", B, S, fs, n_cu);
    return AG_ERR_UNSUPPORTED;
  }
  AG_REQUIRE(ws_bytes >= ag_gfront_persist_ws_bytes(B, S, fs) && ((uintptr_t)ws & 15) == 0,
             "ag_grufront_fwd_persist: workspace too small or misaligned");
  AG_REQUIRE(ldwx % 4 == 0 && (((uintptr_t)w_x | (uintptr_t)w_hh | (uintptr_t)w_p) & 15) == 0,
             "ag_grufront_fwd_persist: weights must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync((char*)ws + PS_STICKY_BYTES, 0, PS_HDR_BYTES, st) != hipSuccess) {
    ag_set_error("ag_grufront_fwd_persist: memset failed");
    return AG_ERR_LAUNCH;
  }
  FrontFwdP p;
  p.gates = gates; p.gh = gh; p.bhn = b_hn; p.wx = w_x; p.whh = w_hh; p.wp = w_p; p.bp = b_p; p.hs = hs; p.cs = nullptr; p.x = x;
  p.ldx = ldx; p.xt = xt;
  p.ctl = ps_ctl(ws);
  p.nrt = ag_cdiv(B, 32);
  p.hx = (float*)((char*)ws + PS_STICKY_BYTES + PS_HDR_BYTES);
  p.xx = p.hx + (int64_t)2 * p.nrt * 32 * S;
  p.T = T; p.B = B; p.ldwx = ldwx; p.rb = ag_precision() == AG_PREC_BF16;
  const int grid = p.nrt * (S / 8);
  if (S == 1024) hipLaunchKernelGGL((gfront_persist_fwd_kernel<1024, 256, 0, 1>), dim3(grid), dim3(512), 0, st, p);
  else hipLaunchKernelGGL((gfront_persist_fwd_kernel<128, 64, 0, 1>), dim3(grid), dim3(512), 0, st, p);
  AG_CHECK_LAUNCH("ag_grufront_fwd_persist");
  return AG_OK;
}

// ------------------------------------------------------------------------------------------
// Backward through time of the Generator's recurrent front (audiogan.py:428-460 under .backward() :903) as ONE
// persistent launch.  Per frame t (T-1 .. 0), with dacc[t] = [dL/dh_t | dL/dx_t] (the stop head's and the conv trunk's
// gradients, filled in by the host) as the external part:
//   gx_t  = (dacc_x[t] + dgates_{t+1} W_x) * (1 - x_t^2)                       d(pre-tanh) of the projection
//   dh_t  = dacc_h[t] + dgates_{t+1} W_hh + gx_t W_p
//   dgates_t = cell backward(dh_t, dc_{t+1})
// The launch-per-op path costs three launches per frame (fused cell step, a [B,4S] x [4S,S+fs] product that re-streams
// its 21 MB of weights from the fabric, the second stage of its K split).  Here every weight stays in REGISTERS:
//   workgroup (clip tile of 32, column tile of 16): one 16-column tile of [W_hh | W_x] = a [4S x 16] panel, K split
//   over the 8 waves (128 VGPRs per lane at S = 1024); S/16 "h tiles" + fs/16 "x tiles" per clip tile.
//   x tile, frame t: dx_t -> gx_t, published (write-through into dxt[t], which is also an output) + flag
//   h tile, frame t: waits for gx_t -> gx_t W_p against its resident [fs x 16] W_p panel -> cell backward of its
//                    (clip, unit) pairs (dc stays in a register) -> dgates_t published (into dgs[t]) + flag
//   both, t > 0    : wait for all dgates_t of the clip tile -> [32 x 4S] x panel on v_mfma_f32_16x16x4_f32 -> LDS sum of
//                    the 8 K slices -> the recurrent term of frame t-1, kept in a register.
// Two hand-offs per frame; one slot per frame in dgs / dxt (no parity); flags count frames.
// ------------------------------------------------------------------------------------------
struct FrontBwdP {
  const float* ga;     // [T,B,4S] activated gates (i, f, g, o)
  const float* c_all;  // [T+1,B,S]
  const float* x;      // [B,T*fs] the front's output (after tanh), row pitch ldx
  const float* dh_ext; // [T,B,S] external gradient dL/dh_t (the stop head's), or NULL = none
  const float* dx_ext; // [B,T*fs] external gradient dL/dx_t (the conv trunk's), row pitch lddx, or NULL = none
  int64_t ldx, lddx;
  const float* w_hh;   // [4S,S]
  const float* w_x;    // [4S,ldwx]: W_ih[:, :fs]
  const float* w_p;    // [fs,S]
  float* dgs;          // [T,B,4S] out (and exchange)
  float* dxt;          // [T,B,fs] out (and exchange)
  // GRU cell (CELL = 1): ga = activated (r, z, n) [T,B,3S], c_all = hs [T+1,B,S] (hs[t] = h_{t-1}), gh [T,B,3S] (its n
  // slot: W_hn h_{t-1} + b_hn); dgs = dgi [T,B,3S] (what the x tiles multiply by W_x), dgh [T,B,3S] (what the h tiles
  // multiply by W_hh: the n slot times r).  LSTM: gh unused, dgh = dgs.
  const float* gh;
  float* dgh;
  PersistCtl ctl;
  int T, B, ldwx;
};

template <int S, int FS, bool RB, int CELL = 0>
__global__ __launch_bounds__(512) void gfront_persist_bwd_kernel(const FrontBwdP p) {
  constexpr int NG = CELL == 1 ? 3 : 4;                     // gate blocks
  constexpr int K4 = NG * S, NU = K4 / 128;                 // 16-k units of the big product per wave
  constexpr int NHT = S / 16, NXT = FS / 16, NT = NHT + NXT;
  constexpr int NP = FS >= 128 ? FS / 128 : 1;              // 16-k units of the projection product per wave
  static_assert(NU * 128 == K4 && FS % 16 == 0 && S % 16 == 0, "shape");
  __shared__ float red[8 * 512];
  __shared__ int s_dead;
  const int T = p.T, B = p.B;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  if (tid == 0) s_dead = 0;
  __syncthreads();
  // placement: workgroup ids go round-robin over the 8 XCDs; with nbt clip tiles (8 % nbt == 0) a clip tile's NT workgroups
  // are put on 8 / nbt XCDs, so a frame's dgates rows are pulled into that many L2s instead of all 8 (speed only)
  int bt = blockIdx.x / NT, ct = blockIdx.x % NT;
  {
    const int nbt = gridDim.x / NT, per = nbt <= 8 ? 8 / nbt : 0;
    if (per > 0 && per * nbt == 8 && NT % per == 0) {
      const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;       // q < NT / per
      bt = xcd / per;
      ct = (xcd % per) * (NT / per) + q;
    }
  }
  const bool isx = ct >= NHT;
  const int m0 = bt * 32;
  const int n0 = (isx ? ct - NHT : ct) * 16;                // first column inside W_hh / W_x
  const int64_t BG = (int64_t)B * K4, BH = (int64_t)B * S, BX = (int64_t)B * FS;

  // resident panel: wreg[u][e] = W[(wid*NU + u)*16 + 4g + e][n0 + li]
  float wreg[NU][4];
  {
    const float* W = isx ? p.w_x : p.w_hh;
    const int ldw = isx ? p.ldwx : S;
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        wreg[u][e] = ag_rbf_if(W[(int64_t)((wid * NU + u) * 16 + 4 * g + e) * ldw + n0 + li], RB);
  }
  // h tiles: wp[u][e] = W_p[(wid*NP + u)*16 + 4g + e][n0 + li]
  const bool pw_on = !isx && (wid * NP * 16 < FS);
  float wp[NP][4];
#pragma unroll
  for (int u = 0; u < NP; ++u)
#pragma unroll
    for (int e = 0; e < 4; ++e)
      wp[u][e] = pw_on ? ag_rbf_if(p.w_p[(int64_t)((wid * NP + u) * 16 + 4 * g + e) * S + n0 + li], RB) : 0.f;

  unsigned* flags_h = p.ctl.hdr + PS_FLAG_OFF + bt * NT;
  unsigned* flags_x = flags_h + NHT;
  unsigned* my_flag = flags_h + ct;
  // epilogue role: thread <-> (clip row, column of the tile)
  const int erow = tid >> 4, ecol = tid & 15;
  const int em = m0 + erow;
  const bool epi = em < B;
  // A operand rows of this lane for the two 16-clip halves (clamped: rows past the batch feed only their own, unwritten outputs)
  const int ar0 = min(m0 + li, B - 1), ar1 = min(m0 + 16 + li, B - 1);
  float carry = 0.f, dcn = 0.f, direct = 0.f;
  bool alive = true;

  for (int t = T - 1; t >= 0; --t) {
    const unsigned seq = (unsigned)(T - t);
    if (isx) {
      if (epi) {
        const float dx = (p.dx_ext ? p.dx_ext[(int64_t)em * p.lddx + (int64_t)t * FS + n0 + ecol] : 0.f) + carry;
        const float xv = p.x[(int64_t)em * p.ldx + (int64_t)t * FS + n0 + ecol];
        float gx = dx * (1.f - xv * xv);
        if (s_dead) gx = __builtin_nanf("");
        __amdgpu_buffer_rsrc_t orr = __builtin_amdgcn_make_buffer_rsrc(p.dxt + (int64_t)t * BX, 0, (int)(BX * 4), 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(gx), orr, (unsigned)(((int64_t)em * FS + n0 + ecol) * 4), 0, 16);
      }
    } else {
      float ig = 0.f, fg = 0.f, gg = 0.f, og = 0.f, cp = 0.f, cn = 0.f, ext = 0.f;
      if (epi) {
        const float* gr = p.ga + (int64_t)t * BG + (int64_t)em * K4 + n0 + ecol;
        ig = gr[0]; fg = gr[S]; gg = gr[2 * S];            // LSTM: i, f, g (, o)   GRU: r, z, n
        cp = p.c_all[(int64_t)t * BH + (int64_t)em * S + n0 + ecol];              // c_{t-1}  /  h_{t-1}
        if (CELL == 1) {
          og = p.gh[(int64_t)t * BG + (int64_t)em * K4 + 2 * S + n0 + ecol];      // W_hn h_{t-1} + b_hn
        } else {
          og = gr[3 * S];
          cn = p.c_all[(int64_t)(t + 1) * BH + (int64_t)em * S + n0 + ecol];
        }
        ext = p.dh_ext ? p.dh_ext[(int64_t)t * BH + (int64_t)em * S + n0 + ecol] : 0.f;
      }
      if (wid == 0 && alive) {
        alive = ps_wait_flags(p.ctl, flags_x, NXT, seq, lane);
        if (!alive) s_dead = 1;
      }
      __syncthreads();
      // gx_t [32 clips, fs] x W_p panel
      f32x4 pa[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      if (pw_on) {
        __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dxt) + (int64_t)t * BX, 0, (int)(BX * 4), 0x00020000);
        u32x4 a[2][NP];
#pragma unroll
        for (int u = 0; u < NP; ++u) {
          const unsigned ko = (unsigned)(((wid * NP + u) * 16 + 4 * g) * 4);
          a[0][u] = __builtin_amdgcn_raw_buffer_load_b128(xr, (unsigned)(ar0 * FS * 4) + ko, 0, 16);
          a[1][u] = __builtin_amdgcn_raw_buffer_load_b128(xr, (unsigned)(ar1 * FS * 4) + ko, 0, 16);
        }
#pragma unroll
        for (int u = 0; u < NP; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int r = 0; r < 2; ++r)
              pa[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(ag_rbf_if(__uint_as_float(a[r][u][e]), RB), wp[u][e], pa[r], 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[wid * 512 + (16 * r + 4 * g + e) * 16 + li] = pa[r][e];
      __syncthreads();
      if (epi) {
        float dhv = ext + carry;
#pragma unroll
        for (int w = 0; w < 8; ++w) dhv += red[w * 512 + tid];
        __amdgpu_buffer_rsrc_t orr = __builtin_amdgcn_make_buffer_rsrc(p.dgs + (int64_t)t * BG, 0, (int)(BG * 4), 0x00020000);
        const unsigned o = (unsigned)(((int64_t)em * K4 + n0 + ecol) * 4);
        if (CELL == 1) {
          // torch.nn.GRUCell backward: r = ig, z = fg, n = gg, hn = og, h_{t-1} = cp
          float dn = dhv * (1.f - fg) * (1.f - gg * gg);
          float dz = dhv * (cp - gg) * fg * (1.f - fg);
          float dr = dn * og * ig * (1.f - ig);
          float dnr = dn * ig;
          direct = dhv * fg;                                  // the direct path dh * z, added to frame t-1's dh
          if (s_dead) { dn = dz = dr = dnr = direct = __builtin_nanf(""); }
          __amdgpu_buffer_rsrc_t hrr = __builtin_amdgcn_make_buffer_rsrc(p.dgh + (int64_t)t * BG, 0, (int)(BG * 4), 0x00020000);
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dr), orr, o, 0, 16);
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dz), orr, o + (unsigned)(S * 4), 0, 16);
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dn), orr, o + (unsigned)(2 * S * 4), 0, 16);
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dr), hrr, o, 0, 16);
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dz), hrr, o + (unsigned)(S * 4), 0, 16);
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dnr), hrr, o + (unsigned)(2 * S * 4), 0, 16);
        } else {
          const float tc = tanhf(cn);
          const float dc = dcn + dhv * og * (1.f - tc * tc);
          float d0 = dc * gg * ig * (1.f - ig);
          float d1 = dc * cp * fg * (1.f - fg);
          float d2 = dc * ig * (1.f - gg * gg);
          float d3 = dhv * tc * og * (1.f - og);
          dcn = dc * fg;
          if (s_dead) { d0 = d1 = d2 = d3 = dcn = __builtin_nanf(""); }   // a wait timed out: poison instead of garbage
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(d0), orr, o, 0, 16);
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(d1), orr, o + (unsigned)(S * 4), 0, 16);
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(d2), orr, o + (unsigned)(2 * S * 4), 0, 16);
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(d3), orr, o + (unsigned)(3 * S * 4), 0, 16);
        }
      }
    }
    if (t == 0 && !isx) break;                                // dgates_0 has no reader inside the launch
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // EVERY storing wave drains its write-through stores
    __syncthreads();
    if (tid == 0 && (int)blockIdx.x != p.ctl.mute)
      __hip_atomic_store(my_flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == 0) break;
    // ---- the recurrent term of frame t-1: dgates_t [32 clips, 4S] x panel
    if (wid == 0 && alive) {
      alive = ps_wait_flags(p.ctl, flags_h, NHT, seq, lane);
      if (!alive) s_dead = 1;
    }
    __syncthreads();
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    {
      __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc((isx ? p.dgs : p.dgh) + (int64_t)t * BG, 0, (int)(BG * 4), 0x00020000);
      const unsigned o0 = (unsigned)(((int64_t)ar0 * K4 + wid * NU * 16 + 4 * g) * 4);
      const unsigned o1 = (unsigned)(((int64_t)ar1 * K4 + wid * NU * 16 + 4 * g) * 4);
      constexpr int UB = NU < 4 ? NU : 4;
#pragma unroll
      for (int ub = 0; ub < NU; ub += UB) {
        u32x4 a[2][UB];
#pragma unroll
        for (int i = 0; i < UB; ++i) {
          a[0][i] = __builtin_amdgcn_raw_buffer_load_b128(gr, o0 + (unsigned)((ub + i) * 64), 0, 16);
          a[1][i] = __builtin_amdgcn_raw_buffer_load_b128(gr, o1 + (unsigned)((ub + i) * 64), 0, 16);
        }
#pragma unroll
        for (int i = 0; i < UB; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int r = 0; r < 2; ++r)
              acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(ag_rbf_if(__uint_as_float(a[r][i][e]), RB), wreg[ub + i][e], acc[r], 0, 0, 0);
      }
    }
    // C layout of a 16x16 tile: col = lane & 15, row = 4 * (lane >> 4) + e
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int e = 0; e < 4; ++e) red[wid * 512 + (16 * r + 4 * g + e) * 16 + li] = acc[r][e];
    __syncthreads();
    carry = direct;                                           // (GRU h tiles: dh_t * z_t; otherwise 0)
#pragma unroll
    for (int w = 0; w < 8; ++w) carry += red[w * 512 + tid];
    __syncthreads();                                          // red is written again in the next frame
  }
}

static bool front_bwd_shape_ok(int B, int S, int fs, int n_cu) {
  if (!((S == 1024 && fs == 256) || (S == 128 && fs == 64))) return false;
  if (B < 1) return false;
  if (n_cu > 256) n_cu = 256;
  const int64_t grid = (int64_t)ag_cdiv(B, 32) * ((S + fs) / 16);
  return grid <= n_cu && grid <= (PS_HDR_BYTES / 4 - PS_FLAG_OFF);
}

extern "C" int ag_gfront_bwd_persist_ok(int B, int S, int fs, int n_cu) { return front_bwd_shape_ok(B, S, fs, n_cu) ? 1 : 0; }

// The frame loop of the Generator front's backward in ONE launch.  ga [T,B,4S] activated gates and c_all [T+1,B,S] as
// saved by the forward, x [B,T*fs] the front's output (row pitch ldx), the external gradients dh_ext [T,B,S] = dL/dh_t
// (the stop head's) and dx_ext [B,T*fs] = dL/dx_t (the conv trunk's, row pitch lddx: it may be channel 0 of the trunk's
// gradient slab) - each read only, each may be NULL = zero -, w_hh [4S,S], w_x = W_ih[:, :fs] (row pitch ldwx), w_p [fs,S]; outputs dgs [T,B,4S] (d gate pre-activations) and
// dxt [T,B,fs] (d pre-tanh of the projection).  `ws`: >= PS_STICKY_BYTES + 8 KiB (status + flags).
static int front_bwd_launch(int cell, const float* ga, const float* c_all, const float* gh, const float* x, int64_t ldx,
                            const float* dh_ext, const float* dx_ext, int64_t lddx, const float* w_hh, const float* w_x,
                            int ldwx, const float* w_p, float* dgs, float* dgh, float* dxt, void* ws, int64_t ws_bytes, int T,
                            int B, int S, int fs, int n_cu, void* stream);

extern "C" int ag_gfront_bwd_persist(const float* ga, const float* c_all, const float* x, int64_t ldx, const float* dh_ext,
                                     const float* dx_ext, int64_t lddx, const float* w_hh, const float* w_x, int ldwx,
                                     const float* w_p, float* dgs, float* dxt, void* ws, int64_t ws_bytes, int T, int B,
                                     int S, int fs, int n_cu, void* stream) {
  return front_bwd_launch(0, ga, c_all, nullptr, x, ldx, dh_ext, dx_ext, lddx, w_hh, w_x, ldwx, w_p, dgs, dgs, dxt, ws,
                          ws_bytes, T, B, S, fs, n_cu, stream);
}

// The same for the GRU front (BASELINE configs[3]; torch.nn.GRUCell, gate order r z n): ga [T,B,3S] activated gates, hs
// [T+1,B,S] (hs[t] = h_{t-1}, hs[0] = 0) and gh [T,B,3S] (n slot) as saved by ag_grufront_fwd_persist; outputs dgi
// [T,B,3S] (d of the input-side pre-activations), dgh [T,B,3S] (hidden side: the n slot times r) and dxt [T,B,fs].
extern "C" int ag_grufront_bwd_persist(const float* ga, const float* hs, const float* gh, const float* x, int64_t ldx,
                                       const float* dh_ext, const float* dx_ext, int64_t lddx, const float* w_hh,
                                       const float* w_x, int ldwx, const float* w_p, float* dgi, float* dgh, float* dxt,
                                       void* ws, int64_t ws_bytes, int T, int B, int S, int fs, int n_cu, void* stream) {
  AG_REQUIRE(gh && dgh, "ag_grufront_bwd_persist: null tensor");
  return front_bwd_launch(1, ga, hs, gh, x, ldx, dh_ext, dx_ext, lddx, w_hh, w_x, ldwx, w_p, dgi, dgh, dxt, ws, ws_bytes, T,
                          B, S, fs, n_cu, stream);
}

static int front_bwd_launch(int cell, const float* ga, const float* c_all, const float* gh, const float* x, int64_t ldx,
                            const float* dh_ext, const float* dx_ext, int64_t lddx, const float* w_hh, const float* w_x,
                            int ldwx, const float* w_p, float* dgs, float* dgh, float* dxt, void* ws, int64_t ws_bytes, int T,
                            int B, int S, int fs, int n_cu, void* stream) {
  AG_REQUIRE(ga && c_all && x && w_hh && w_x && w_p && dgs && dxt && ws, "ag_gfront_bwd_persist: null tensor");
  AG_REQUIRE(T > 0, "ag_gfront_bwd_persist: T must be positive");
  AG_REQUIRE(ldx >= (int64_t)T * fs && (!dx_ext || lddx >= (int64_t)T * fs), "ag_gfront_bwd_persist: row pitch smaller than a row");
  if (!front_bwd_shape_ok(B, S, fs, n_cu)) {
    ag_set_error("ag_gfront_bwd_persist: shape B=%d S=%d fs=%d is not supported on %d CUs", B, S, fs, n_cu);
    return AG_ERR_UNSUPPORTED;
  }
  AG_REQUIRE(ws_bytes >= PS_STICKY_BYTES + PS_HDR_BYTES && ((uintptr_t)ws & 15) == 0, "ag_gfront_bwd_persist: workspace too small");
  AG_REQUIRE(ldwx >= fs, "ag_gfront_bwd_persist: bad row pitch");
  AG_REQUIRE((int64_t)B * 4 * S * 4 < ((int64_t)1 << 31), "ag_gfront_bwd_persist: frame slab too large");
  AG_REQUIRE((((uintptr_t)dgs | (uintptr_t)dxt) & 15) == 0, "ag_gfront_bwd_persist: outputs must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync((char*)ws + PS_STICKY_BYTES, 0, PS_HDR_BYTES, st) != hipSuccess) {
    ag_set_error("ag_gfront_bwd_persist: memset failed");
    return AG_ERR_LAUNCH;
  }
  FrontBwdP p;
  p.ga = ga; p.c_all = c_all; p.x = x; p.ldx = ldx; p.dh_ext = dh_ext; p.dx_ext = dx_ext; p.lddx = lddx; p.w_hh = w_hh; p.w_x = w_x; p.w_p = w_p; p.dgs = dgs; p.dxt = dxt;
  p.gh = gh; p.dgh = dgh;
  p.ctl = ps_ctl(ws);
  p.T = T; p.B = B; p.ldwx = ldwx;
  const bool rb = ag_precision() == AG_PREC_BF16;
  const int grid = ag_cdiv(B, 32) * ((S + fs) / 16);
  void (*kern)(const FrontBwdP);
  if (cell == 0)
    kern = S == 1024 ? (rb ? gfront_persist_bwd_kernel<1024, 256, true, 0> : gfront_persist_bwd_kernel<1024, 256, false, 0>)
                     : (rb ? gfront_persist_bwd_kernel<128, 64, true, 0> : gfront_persist_bwd_kernel<128, 64, false, 0>);
  else
    kern = S == 1024 ? (rb ? gfront_persist_bwd_kernel<1024, 256, true, 1> : gfront_persist_bwd_kernel<1024, 256, false, 1>)
                     : (rb ? gfront_persist_bwd_kernel<128, 64, true, 1> : gfront_persist_bwd_kernel<128, 64, false, 1>);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 0, st, p);
  AG_CHECK_LAUNCH("ag_gfront_bwd_persist");
  return AG_OK;
}
