/* audiogan_hip.h  --  C ABI of libaudiogan_hip.so (gfx950 / MI355X only).
 *
 * The reference (BarclayII/audiogan) has no FFI layer: its hot path bottoms out in
 * third-party PyTorch ops called from audiogan.py.  Each entry point below replaces
 * one of those implicit device ops; the comment on each names the reference call
 * site(s) it stands in for (paths relative to /root/reference).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 data unless its name ends in _i64
 *     (int64) or says otherwise; the library borrows them for the call and owns
 *     nothing except the descriptor tables it is handed.
 *   - tensors are dense along their last (time / feature) axis; batch and channel
 *     strides are passed explicitly (in elements) so a layer can read from / write
 *     into a slice of a pre-allocated channel slab (this is what removes the
 *     T.cat at audiogan.py:467).
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued, never
 *     synchronised.  No allocation, no host sync inside any call (graph-capturable).
 *   - return value: AG_OK (0) or a negative AG_ERR_* code; never aborts.
 */
#ifndef AUDIOGAN_HIP_H
#define AUDIOGAN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AG_OK 0
#define AG_ERR_ARG (-1)         /* bad shape / null pointer / unsupported size     */
#define AG_ERR_LAUNCH (-2)      /* hipGetLastError() != hipSuccess after a launch  */
#define AG_ERR_UNSUPPORTED (-3) /* valid request this build has no kernel for      */

#define AG_ACT_NONE 0
#define AG_ACT_LEAKY 1 /* LeakyReLU(slope): audiogan.py:261,277,532 (slope 0.01) */
#define AG_ACT_TANH 2  /* audiogan.py:443 */
/* ag_conv1d_engine only: `res` is not a residual but the SAVED OUTPUT y of a LeakyReLU; the result is scaled by that
 * activation's derivative (1 where y > 0, slope elsewhere) - the LeakyReLU backward of the layer below folded into this
 * layer's backward-data pass (the .backward() of audiogan.py:277, :532); bias, length mask and accumulate as usual */
#define AG_ACT_LEAKY_GATE 3

/* library / build identification (ABI version, gfx arch string) */
int ag_abi_version(void);
const char* ag_arch(void);
const char* ag_last_error(void);

/* ---------------------------------------------------------------------------
 * Weight norm  (audiogan.py:77-80; torch.nn.utils.weight_norm, dim 0)
 *   w[r, :] = g[r] * v[r, :] / ||v[r, :]||_2      rows = size of dim 0, cols = rest
 * A table of `n` descriptors (device memory, see ag_wn_desc) is processed by ONE
 * launch.  For 3-D conv weights the kernel can also emit the two MFMA-friendly
 * layouts the conv engine consumes (see ag_conv1d_engine).
 * ------------------------------------------------------------------------- */
typedef struct ag_wn_desc {
  const float* v;  /* [rows, cols]                                      */
  const float* g;  /* [rows]                                            */
  float* w;        /* [rows, cols] standard layout (may be NULL)        */
  float* wpa;      /* gather layout  [pad2(d1)][K][pad32(d0)]  or NULL  */
  float* wpb;      /* scatter layout [d0][ceil(K/s)][pad32(d1*s)] or NULL */
                   /* Each layout [Cp2][taps][rows] is followed IN THE SAME BUFFER by its bf16 image
                    * [ceil(Cp2/16)][taps][2][rows][8 x bf16] (what the bf16-MFMA conv kernel stages in AG_PREC_BF16
                    * mode); ag_wpa_numel / ag_wpb_numel return the size of both parts, the buffers must start
                    * zero-filled (padding is never written).                                                  */
  float* inv_norm; /* [rows] 1/||v||, saved for backward (may be NULL)  */
  int32_t rows;    /* d0                                                */
  int32_t cols;    /* d1*K                                              */
  int32_t d1;      /* second dim (1 for 2-D / 1-D tensors)              */
  int32_t K;       /* taps (1 for 2-D / 1-D tensors)                    */
  int32_t stride;  /* conv stride used for the wpb layout               */
  int32_t pad;     /* conv padding the wpb layout is prepared for: when pad % stride != 0 and it costs no tap slot,
                    * phase r's taps sit ceil(pad/s) - ceil((pad-r)/s) slots later ("aligned" layout, see
                    * ag_conv_args.wp_pad); 0 = plain layout */
} ag_wn_desc;

int ag_weight_norm_fwd(const ag_wn_desc* descs_dev, int n, int max_rows, void* stream);

typedef struct ag_wn_bwd_desc {
  const float* v;
  const float* g;
  const float* dw; /* [rows, cols] gradient wrt w (standard layout) */
  float* dv;       /* [rows, cols] */
  float* dg;       /* [rows] */
  int32_t rows;
  int32_t cols;
  int32_t accumulate; /* 1: dv += ..., dg += ... (gradient accumulation straight into .grad) */
  int32_t pad_;
} ag_wn_bwd_desc;

int ag_weight_norm_bwd(const ag_wn_bwd_desc* descs_dev, int n, int max_rows, void* stream);

/* ---------------------------------------------------------------------------
 * 1-D convolution engine (fp32 MFMA implicit GEMM).  One kernel family covers
 *   mode 0 "gather":  y[b,o,t]      = sum_{c,k} W[o,c,k] x[b,c, s*t+k-p]
 *        = NN.Conv1d forward            (audiogan.py:272,406,490)
 *        = NN.ConvTranspose1d backward-data
 *   mode 1 "scatter": y[b,o,s*t+k-p] += sum_c W[c,o,k] x[b,c,t]
 *        = NN.ConvTranspose1d forward   (audiogan.py:275)
 *        = NN.Conv1d backward-data      (loss.backward(), audiogan.py:785,903)
 * followed by the fused epilogue
 *   v = acc + bias[o] + res[b,o,t];  v = act(v);  v *= (t < lens_i64[b]);
 *   y = accumulate ? y + v : v
 * (bias/res/lens optional).  `wp` is the prepared weight layout written by
 * ag_weight_norm_fwd (wpa for mode 0, wpb for mode 1) or by ag_prep_conv_weight.
 * ------------------------------------------------------------------------- */
typedef struct ag_conv_args {
  const float* x;
  const float* wp;
  const float* bias;
  const float* res;
  float* y;
  const int64_t* lens_i64;
  int64_t x_bs, x_cs;     /* input batch / channel stride (elements)  */
  int64_t y_bs, y_cs;     /* output strides                            */
  int64_t res_bs, res_cs; /* residual strides                          */
  int32_t B, C, Lin, O, Lout;
  int32_t K, stride, pad; /* parameters of the underlying conv         */
  int32_t mode;           /* 0 gather, 1 scatter                       */
  int32_t act;            /* AG_ACT_*                                  */
  float slope;
  int32_t accumulate;
  int32_t wp_pad;         /* mode 1: the padding `wp` was prepared for (ag_wn_desc.pad / ag_prep_conv_weight's pad).
                           * Equal to `pad` with pad % stride != 0: the aligned scatter layout is assumed and all
                           * phases share one column range (no ragged last column); anything else: plain layout */
} ag_conv_args;

int ag_conv1d_engine(const ag_conv_args* args, void* stream);

/* plain (non weight-normed) weights -> engine layouts; w is [d0][d1][K] */
int ag_prep_conv_weight(const float* w, float* wpa, float* wpb, int d0, int d1, int K, int stride, int pad,
                        void* stream);
/* sizes (in floats) of the two layouts - fp32 part + bf16 image, see ag_wn_desc - so the host can allocate them */
int64_t ag_wpa_numel(int d0, int d1, int K);
int64_t ag_wpb_numel(int d0, int d1, int K, int stride);

/* Weight gradient of a strided conv (conv backward-weight, convT backward-weight):
 *   dw[a,c,k] (+)= sum_{b,t} sh[b,a,t] * lg[b,c, s*t+k-p]
 * Conv1d:          sh = dY (a = out ch),  lg = x  (c = in ch)   -> dw[O][C][K]
 * ConvTranspose1d: sh = x  (a = in ch),   lg = dY (c = out ch)  -> dw[Ci][Co][K]
 * dw must be zero-filled (or hold a value to accumulate into); fp32 atomics. */
int ag_conv1d_wgrad(const float* sh, int64_t sh_bs, int64_t sh_cs, const float* lg, int64_t lg_bs,
                    int64_t lg_cs, float* dw, int B, int A, int Lsh, int C, int Llg, int K,
                    int stride, int pad, void* stream);

/* Single-output-channel stride-1 'same' convolution (the Generator's final Conv1d 113 -> 1, k3,
 * audiogan.py:404-407): one output channel cannot fill an MFMA tile and the layer is HBM-bound, so it
 * has plain vector kernels.  w is the standard [1, C, K] weight (= [C, K]); K <= 9.
 *   fwd      y[b,t]      = act(bias + sum_{c,k} w[c,k] x[b,c,t+k-pad])
 *   bwd_data dx[b,c,t] (+)= sum_k w[c,k] dy[b,t-k+pad]
 *   wgrad    dw[c,k]    += sum_{b,t} dy[b,t] x[b,c,t+k-pad]        (two-stage, needs a bound workspace; dw pre-zeroed) */
int ag_conv1d_o1_fwd(const float* x, int64_t x_bs, int64_t x_cs, const float* w, const float* bias, float* y,
                     int64_t y_bs, int B, int C, int L, int K, int pad, int act, float slope, void* stream);
int ag_conv1d_o1_bwd_data(const float* dy, int64_t dy_bs, const float* w, float* dx, int64_t dx_bs,
                          int64_t dx_cs, int B, int C, int L, int K, int pad, int accumulate, void* stream);
int ag_conv1d_o1_wgrad(const float* dy, int64_t dy_bs, const float* x, int64_t x_bs, int64_t x_cs, float* dw,
                       int B, int C, int L, int K, int pad, void* stream);

/* db[c] = (accumulate ? db[c] : 0) + sum_{b,t} dy[b,c,t]   (bias gradient; two-stage through the bound workspace) */
int ag_channel_sum(const float* dy, int64_t bs, int64_t cs, float* db, int B, int C, int L, int accumulate,
                   void* stream);

/* dpre = dy * (y > 0 ? 1 : slope) * (t < lens[b])   elementwise on [B,C,L] views.
 * LeakyReLU keeps the sign, so the saved OUTPUT y is enough (no pre-activation).
 * add_into (optional, same shape): add_into += dpre  -- the gradient of the
 * "act += x[:, -out:, :]" skip connection of dense_res_bottleneck (audiogan.py:281-282).
 * bias_grad (optional, [C]): bias_grad[c] += sum_{b,t} dpre[b,c,t] -- the bias gradient of the conv that
 * produced y, taken in the same pass (atomics; pre-zeroed by the caller). */
int ag_leaky_bwd(const float* dy, int64_t dy_bs, int64_t dy_cs, const float* y, int64_t y_bs,
                 int64_t y_cs, float* dpre, int64_t dp_bs, int64_t dp_cs, float* add_into,
                 int64_t ad_bs, int64_t ad_cs, const int64_t* lens_i64, float* bias_grad, int B, int C,
                 int L, float slope, void* stream);

/* ---------------------------------------------------------------------------
 * Dense fp32 GEMM on MFMA (NN.Linear / LSTM gate products: audiogan.py:260,380,
 * 385,409,410,498,509,511 and their backward):
 *   C[M,N] = act( alpha * op(A)[M,K] * op(B)[K,N] + beta * C + bias[N] + res[M,N] )
 * ta: 0 -> A is [M,K] row-major (lda), 1 -> A is stored [K,M]
 * tb: 0 -> B is [K,N] row-major (ldb), 1 -> B is stored [N,K]   (Linear weight)
 * act = AG_ACT_LEAKY_GATE: `res` is not added; it is the saved OUTPUT of a LeakyReLU and the result is scaled by that
 * LeakyReLU's derivative (1 where res > 0, slope elsewhere): the activation backward of the heads (audiogan.py:261,:510
 * under .backward()) rides in the epilogue of the backward-data product.
 * ------------------------------------------------------------------------- */
int ag_gemm(const float* A, int lda, int ta, const float* B, int ldb, int tb, float* C, int ldc,
            int M, int N, int K, float alpha, float beta, const float* bias, const float* res,
            int ldres, int act, float slope, void* stream);

/* column sums: out[n] = (accumulate ? out[n] : 0) + sum_m X[m, n]   (Linear bias gradient; two-stage through the bound
 * workspace, or one row block per column when none is bound) */
int ag_col_sum(const void* X, int x_bf16 /* X stored as bfloat16, ldx in 2-byte elements */, int ldx, float* out, int M, int N,
               int accumulate, void* stream);

/* ---------------------------------------------------------------------------
 * LSTM cell pointwise step (NN.LSTMCell / NN.LSTM, audiogan.py:380,440-442,498):
 * gates[B,4H] hold i|f|g|o pre-activations (PyTorch order) INCLUDING biases.
 *   i,f,o = sigmoid, g = tanh;  c' = f*c + i*g;  h' = o*tanh(c')
 * fwd overwrites gates with the ACTIVATED values (saved for backward).
 * `valid_i64`/`t`: rows with t >= valid[b] keep (h,c) unchanged and output 0
 * (packed-sequence semantics of dynamic_rnn, audiogan.py:214-229); NULL = all valid.
 * ------------------------------------------------------------------------- */
int ag_lstm_cell_fwd(float* gates, int ldg, const float* c_prev, int ldcp, float* h_out, int ldh,
                     float* c_out, int ldc, float* y_out, int ldy, const float* h_prev, int ldhp,
                     const int64_t* valid_i64, int t, int B, int H, void* stream);
/* backward: dh = gradient reaching the state h' from later steps (NULL = 0), dy = gradient
 * of the step's output (NULL = 0), dc_next (NULL = 0) -> dgates (pre-activation), dc_prev.
 * Valid rows use dh + dy and write dh_pass = 0; padded rows write dgates = 0,
 * dc_prev = dc_next and pass dh through (dh_pass = dh; their output was the constant 0). */
int ag_lstm_cell_bwd(const float* gates_act, int ldg, const float* c_prev, int ldcp,
                     const float* c_new, int ldc, const float* dh, int lddh, const float* dy,
                     int lddy, const float* dc_next, int lddcn, float* dgates, int lddg,
                     float* dc_prev, int lddcp, float* dh_pass, int lddhp, const int64_t* valid_i64,
                     int t, int B, int H, void* stream);

/* GRU cell pointwise step (BASELINE config C4; the reference has no GRU -- SURVEY.md F5 -- so the
 * contract is torch.nn.GRUCell's: gate order r|z|n, n = tanh(gi_n + r * gh_n)).  gi [B,3H] and
 * gh [B,3H] (contiguous) hold the complete input / hidden products incl. biases; fwd overwrites gi
 * with the activated (r,z,n), gh keeps gh_n.  bwd writes dgi, dgh and dh_prev = dh * z. */
int ag_gru_cell_fwd(float* gi, const float* gh, const float* h_prev, int ldhp, float* h_out, int ldh,
                    int B, int H, void* stream);
int ag_gru_cell_bwd(const float* gates_act, const float* gh, const float* h_prev, int ldhp,
                    const float* dh, int lddh, float* dgi, float* dgh, float* dh_prev, int lddhp, int B,
                    int H, void* stream);

/* Precision mode of every contraction launched by this PROCESS from now on (default 0 = fp32; process-wide because
 * an autograd engine runs backward passes on a thread of its own).
 *   1 = bf16: each contraction (conv / transposed conv / linear / recurrent products, in their forward,
 *   backward-data and backward-weight forms) rounds BOTH operands to bfloat16 (round-to-nearest-even) and accumulates in
 *   fp32; tensors in memory, epilogues, losses and the optimiser stay fp32.  ag_gemm and the persistent recurrent
 *   kernels run v_mfma_f32_32x32x16_bf16 in this mode; the other kernels round their operands in registers in front of
 *   the fp32 MFMA (same sums up to the order of the fp32 additions).  Spec: BASELINE.json configs[2] (bf16, 8 GPUs);
 *   the reference itself is fp32 only. */
#define AG_PREC_F32 0
#define AG_PREC_BF16 1
/*   2 = f32x3, an EXPERIMENT (bench.py --dtype f32x3; never the headline precision): the large ag_gemm products (128x128-tile
 *   class) split both operands into bfloat16 hi + lo parts and sum hi*hi + hi*lo + lo*hi on the bf16 MFMA with fp32
 *   accumulation (~2^-16 relative per product); every other kernel computes exactly as in mode 0. */
#define AG_PREC_F32X3 2
int ag_set_precision(int mode);
int ag_get_precision(void);

/* Deterministic cross-workgroup reductions.  Entry points that sum over workgroups (ag_conv1d_wgrad,
 * ag_conv1d_o1_wgrad, ag_channel_sum, ag_leaky_bwd's bias gradient, ag_gemm's split-K products, ag_col_sum,
 * ag_skinny_gemm in accumulate mode, ag_lstm_seq_bwd's unfused fallback, ag_grad_norms) do so in TWO STAGES through a
 * bound workspace: partial results with plain stores, then a sum in a fixed order - bitwise reproducible.  There is NO
 * float-atomic path any more (round 4): a call whose sum spans workgroups returns AG_ERR_ARG when no (or too small a)
 * workspace is bound; ag_gemm and ag_col_sum then run unsplit instead.  ag_bind_workspace() binds `numel` floats (16-byte
 * aligned device memory on the launch stream's device) for the NEXT such call of this host thread; the call consumes
 * the binding.  The ag_*_ws_numel() functions return the size a call wants (0: none needed); a smaller workspace (down to
 * the minimum each call states in its error text) reduces the number of partial slabs. */
int ag_bind_workspace(float* ws, int64_t numel);
int64_t ag_conv1d_wgrad_ws_numel(int B, int A, int Lsh, int C, int K);
int64_t ag_gemm_ws_numel(int M, int N, int K, int act);

/* Batched 2-D transpose through LDS: out[b][j][i] = in[b][i][j] for i < R, j < Cc; in element (b,i,j) at in + b*ibs + i*irs + j,
 * out element (b,j,i) at out + b*obs + j*ors + i (inner index contiguous on both sides).  The critic's conv features
 * [B,C,T'] -> time-major [T',B,C] for the biLSTM (audiogan.py:542) and the gradient's way back. */
/* (round 4) in_bf16 / out_bf16: that side is stored as bfloat16 (strides in elements of its own type; fp32 -> bf16 rounds to
 * nearest even): in AG_PREC_BF16 mode with bf16 storage the time-major features are written as bf16, the operand type of
 * ag_gemm_h, and their gradient comes back as bf16 */
int ag_transpose_batched(const void* in, int in_bf16, int64_t ibs, int64_t irs, void* out, int out_bf16, int64_t obs,
                         int64_t ors, int B, int R, int Cc, void* stream);

/* Input assembly, one launch each (round 4; replaces T.cat / expand / transpose / + on the iteration path).
 * ag_build_zc: zc[t,b,:] = [z[b,t,:ns] | c[b,:es]], the non-recurrent part of the Generator front's LSTM input
 * (audiogan.py:433-439: z [B,T,ns], c [B,es] -> zc [T,B,ns+es] contiguous).
 * ag_critic_batch: the critic's minibatch (audiogan.py:724-728, :749-751, :844): rows [0,nA) = xa + na, rows [nA,nA+nB) =
 * xb + nb (na / nb = instance noise, NULL = none; row pitches in elements) -> x_out [nA+nB, L] contiguous; the rows'
 * lengths after every conv layer, lens_out[i, row] = ceil(len / prods[i]) (audiogan.py:533; lenA / lenB NULL = L; prods =
 * the nl <= 8 cumulative stride products, a HOST array) and the conditioning rows c_out = [cA ; cB] ([.,E]; NULL = skip). */
/* A Linear layer with ONE output (the critic's last layer 512 -> 1, audiogan.py:511,:549; the Generator's stop head
 * 1024 -> 1, :410,:445).  ag_rowdot_fwd: y[m*ldy] = x[m,:K] . w + bias[0] (bias may be NULL).  ag_rowdot_bwd, one pass over
 * x: dx[m,k] = dy[m*lddy] * w[k] (times LeakyReLU'(x[m,k]) when gate: x is then the SAVED OUTPUT of the LeakyReLU below;
 * dx may be NULL), and when dw != NULL: dw[k] (+)= sum_m dy[m] x[m,k], db[0] (+)= sum_m dy[m] with db == dw + K (one
 * [K+1] gradient row), two-stage through a bound workspace of ag_rowdot_bwd_ws_numel floats (deferrable).
 * x_bf16 / h16: x (and dx) are stored as bfloat16, pitches in 2-byte elements (the bf16-storage path of AG_PREC_BF16). */
int ag_rowdot_fwd(const void* x, int x_bf16, int ldx, const float* w, const float* bias, float* y, int64_t ldy, int M, int K,
                  void* stream);
int64_t ag_rowdot_bwd_ws_numel(int M, int K);
int ag_rowdot_bwd(const float* dy, int64_t lddy, const void* x, int ldx, const float* w, void* dx, int lddx, int h16, float* dw,
                  float* db, int accumulate, int M, int K, int gate, float slope, void* stream);
int ag_build_zc(const float* z, const float* c, float* zc, int B, int T, int ns, int es, void* stream);

/* ---------------------------------------------------------------------------
 * GEMM on operands STORED as bfloat16 (BASELINE configs[2]; csrc/gemm_bf16s.hip):
 *   C[M,N] = act(alpha * op(A) op(B) + beta * C + bias + res), fp32 accumulation, output as fp32 (C) and / or bf16 (C16)
 * A / B: bfloat16 bit patterns; ta / tb as ag_gemm.  Tiles go global -> LDS by LDS-DMA (no convert, half the bytes of the
 * fp32-operand kernel); k-strided operands (ta = 1, tb = 0: weight gradients, the weight of a data gradient) are read
 * with the transposed LDS read of gfx950.  res (fp32) or res16 (bf16): residual, or with AG_ACT_LEAKY_GATE the saved
 * activation whose derivative scales the result; gate16 (bf16, optional): a saved LeakyReLU output applied AFTER bias /
 * res (a residual layer's backward: (W^T da + da) gated by the layer below).  Shapes: ag_gemm_h_ok (K % 64 == 0, leading
 * dimensions % 8 == 0, row counts % 8 == 0 for k-strided operands, 16-byte aligned operands).  Split-K (fp32 output, plain
 * epilogue) through a bound workspace of ag_gemm_h_ws_numel floats; its second stage is deferrable.
 * ag_to_bf16_2d: dst[r,c] = bf16(src[r,c]) for a pitched [rows <= 65535, cols] block (weights -> their bf16 image).
 * ------------------------------------------------------------------------- */
int ag_gemm_h_ok(int M, int N, int K, int ta, int tb, int lda, int ldb);
int64_t ag_gemm_h_ws_numel(int M, int N, int K, int act, int has_c16);
int ag_gemm_h(const uint16_t* A, int lda, int ta, const uint16_t* B, int ldb, int tb, float* C, int ldc, uint16_t* C16,
              int ldc16, int M, int N, int K, float alpha, float beta, const float* bias, const float* res, int ldres,
              const uint16_t* res16, int ldres16, const uint16_t* gate16, int ldgate16, int act, float slope, void* stream);
int ag_to_bf16_2d(const float* src, int64_t ld_src, uint16_t* dst, int64_t ld_dst, int rows, int cols, void* stream);
int ag_critic_batch(const float* xa, int64_t xa_ld, const float* na, int64_t na_ld, int nA, const float* xb,
                    int64_t xb_ld, const float* nb, int64_t nb_ld, int nB, int L, float* x_out,
                    const int64_t* lenA_i64, const int64_t* lenB_i64, const int32_t* prods_host, int nl,
                    int64_t* lens_out_i64, const float* cA, const float* cB, int E, float* c_out, void* stream);

/* Deferred second stages.  Between ag_defer_reduces(1) and ag_flush_reduces() every two-stage reduction (conv weight
 * gradients, bias / channel / column sums, and a split-K ag_gemm whose second stage has a plain epilogue: contiguous C, no
 * bias / res, beta 0 or 1) only records its second stage; the flush sums all recorded outputs in ONE launch, each in the
 * order its own launch would use (bitwise the same results).  The caller keeps every bound workspace alive until the flush
 * and turns deferral off again with ag_defer_reduces(0) (an error if recorded stages were never flushed).
 * Modes: 1 = start recording, 0 = stop, 2 = pause (a call issued now finishes at once; recorded stages stay), 3 = resume.
 * Process-wide since round 4 (a scope is opened by the thread that calls backward(), the recording calls come from
 * torch's autograd thread; never concurrently), unlike ag_bind_workspace, which stays per thread. */
int ag_defer_reduces(int mode);
int ag_flush_reduces(void* stream);
int64_t ag_skinny_ws_numel(int M, int N, int K);

/* Skinny product for the sequential part of the recurrent layers (M = clips per call <= 256):
 *   C[M,N] = act(A[M,K] * op(B) + beta*C + bias)         (accumulate_atomic == 0)
 *   C[M,N] += A[M,K] * op(B) (+ bias)   K split over workgroups, fp32 atomics  (== 1)
 * tb: 1 -> B stored [N,K] (Linear weight), 0 -> B stored [K,N].  Needs K % 8 == 0 and 16-byte
 * aligned rows of A (and of B when tb == 1). */
int ag_skinny_gemm(const float* A, int lda, const float* B, int ldb, int tb, float* C, int ldc,
                   int M, int N, int K, float beta, const float* bias, int act, float slope,
                   int accumulate_atomic, void* stream);

/* One fused LSTMCell step (audiogan.py:439-442): gates_pre [B,4H] holds the part of the gate
 * pre-activations that does not depend on the recurrence (z/c columns of W_ih + both biases);
 * the kernel adds x[B,Kx]*wx^T (fed-back frame, columns [0,Kx) of W_ih) and h_prev*whh^T,
 * applies the cell and overwrites gates_pre with the activated gates. first_step: x and h_prev
 * are zero and skipped. */
int ag_lstm_step_fwd(float* gates_pre, const float* x, int ldx, const float* wx, int ldwx, int Kx,
                     const float* h_prev, const float* whh, const float* c_prev, float* c_out,
                     float* h_out, int B, int H, int first_step, void* stream);

/* A whole (bi)directional NN.LSTM layer over a padded batch (audiogan.py:498-503, :543 and the
 * pack/unpack of :214-229 expressed as a per-clip valid length): T fused steps enqueued by one
 * call.  Tables are HOST arrays of device pointers, one entry per direction.
 *   pre[d]   [T,B,4H] x-projection + biases; overwritten with the activated gates
 *   whh[d]   [4H,H];  c_all[d] [T+1,B,H] with c_all[d][0] = 0;  hbuf[d] [2,B,H] scratch
 *   y        [T,B,ndir*H]; direction 1 walks the sequence backwards
 *   static_pre  NULL, or a table of [B,4H] tensors added to the pre-activations of EVERY step: the projection of
 *            a time-invariant part of the input (the Discriminator concatenates the conditioning vector c to
 *            every frame, audiogan.py:541) plus the biases - computed once per clip instead of once per frame
 * Processing steps [k_begin, k_end) are enqueued (0, T for the whole layer; the state buffers
 * carry over between calls, forward in increasing and backward in decreasing step order).
 * ag_lstm_seq_bwd's `phases` is 3 in normal use; 1 enqueues only the pointwise cell backward and
 * 2 only the recurrent product of each step (lets a profiler time the two kernels separately). */
int ag_lstm_seq_fwd(float* const* pre, const float* const* whh, float* const* c_all,
                    float* const* hbuf, float* y, const int64_t* valid_i64, const float* const* static_pre,
                    int T, int B, int H, int ndir, int k_begin, int k_end, void* stream);
int ag_lstm_seq_bwd(const float* const* gates, const float* const* whh, const float* const* c_all,
                    const float* dy, float* const* dgates, float* const* dhbuf, float* const* dcbuf,
                    const int64_t* valid_i64, int T, int B, int H, int ndir, int k_begin, int k_end,
                    int phases, void* stream);

/* The same layer forward as ONE persistent launch with W_hh resident in LDS (csrc/lstm_persist.hip): each
 * workgroup owns 8 hidden units x 32 or 64 clips of one direction for the whole sequence; only the hidden state
 * crosses workgroups (write-through stores + one agent-scope flag per workgroup and step).  Replaces T launches
 * that re-stream W_hh from the fabric each.  Needs every workgroup co-resident: ag_lstm_persist_ok() says whether
 * (B, H, ndir) fits `n_cu` compute units (H % 64 == 0, H <= 768, ndir * ceil(B/32 or 64) * H/8 <= n_cu); `ws` is
 * ag_lstm_persist_ws_bytes() bytes of 16-byte aligned device memory owned by this launch while it is in flight.
 * Workspace layout: [256 B sticky area][8 KiB header: launch status + flags][exchange buffers].  The header is zeroed
 * by a memset node enqueued in front of the kernel.  The STICKY word (word 0 of the workspace, zero when the caller
 * allocates it) is never cleared by a launch: every bounded spin that times out ORs 0x80000000|step into it, so one
 * host read after any number of launches tells whether all of them completed.  A
 * workgroup that gave up writes NaN into everything it produces from then on, so the failure also reaches the loss.
 * ONE persistent launch per device at a time.
 * Tensors as for ag_lstm_seq_fwd (no hbuf: the state stays in registers). */
/* test hook, ONE-SHOT and per thread: timeout of the bounded spins in ticks of the 100 MHz realtime counter (<= 0: the
 * default 3 s) and one block index that never publishes its flags (-1: none), applied to the NEXT persistent launch the
 * calling thread enqueues and consumed by it (every later launch runs with the defaults again) */
int ag_persist_debug(int64_t timeout_ticks, int mute_block);
int ag_lstm_persist_ok(int B, int H, int ndir, int n_cu);
int64_t ag_lstm_persist_ws_bytes(int B, int H, int ndir);
/* (round 4) y_bf16: the layer output y is WRITTEN as bfloat16 (y then addresses 2-byte elements) - the operand type of the
 * products that consume it (ag_gemm_h) */
int ag_lstm_seq_fwd_persist(float* const* pre, const float* const* whh, float* const* c_all, void* y, int y_bf16,
                            const int64_t* valid_i64, const float* const* static_pre, void* ws, int64_t ws_bytes,
                            int T, int B, int H, int ndir, int n_cu, void* stream);

/* The layer's backward through time as ONE persistent launch: each workgroup (16 clips x 32 hidden units of one
 * direction) keeps its [4H x 32] panel of W_hh in REGISTERS for the whole sequence; dgates_k is written through
 * into the output tensor, which doubles as the exchange buffer.  H in {64,128,256,512} and
 * ndir * ceil(B/16) * H/32 <= n_cu (ag_lstm_persist_bwd_ok); `ws` >= 256 B + 8 KiB, laid out as above.  Tensors as for
 * ag_lstm_seq_bwd (no dhbuf/dcbuf: the state stays in registers). */
int ag_lstm_persist_bwd_ok(int B, int H, int ndir, int n_cu);
/* (round 4) dy_bf16: dy is stored as bfloat16; dg16: optional table of ndir [T,B,4H] bfloat16 outputs, dgates rounded - the
 * operand type of ag_gemm_h for the weight / input gradient products (dgates itself stays fp32: it is the exchange buffer).
 * dgsum: optional table of ndir [B,4H] outputs, the sum over time of dgates (what the biases and a time-invariant
 * input see), accumulated in registers by the thread that produces each (clip, gate) pair - no separate pass over dgates */
int ag_lstm_seq_bwd_persist(const float* const* gates, const float* const* whh, const float* const* c_all,
                            const void* dy, int dy_bf16, float* const* dgates, float* const* dgsum, uint16_t* const* dg16,
                            int dg16_ld /* row pitch of dg16 in elements, >= 4H */, const int64_t* valid_i64, void* ws,
                            int64_t ws_bytes, int T, int B, int H, int ndir, int n_cu, void* stream);

/* The Generator front's whole frame loop (audiogan.py:428-460, one LSTMCell layer + tanh(proj) fed back) as ONE
 * persistent launch with every weight resident in registers (csrc/lstm_persist.hip).  gates [T,B,4S]: in = the z / c
 * part of the gate pre-activations + both biases (one GEMM over all frames), out = activated gates; w_x = W_ih[:, :fs]
 * (row pitch ldwx), w_hh [4S,S], w_p [fs,S], b_p [fs]; outputs hs [T,B,S], cs [T+1,B,S] (cs[0] is written 0 by the launch) and the
 * frames x [B,T*fs].  Supported: (S, fs) in {(1024, 256), (128, 64)}, B <= 64 (ag_gfront_persist_ok); `ws` =
 * ag_gfront_persist_ws_bytes() bytes, used as for ag_lstm_seq_fwd_persist. */
int ag_gfront_persist_ok(int B, int S, int fs, int n_cu);
int64_t ag_gfront_persist_ws_bytes(int B, int S, int fs);
/* (round 4) x [B,T*fs] has a row pitch `ldx`: the caller may hand over channel 0 of the conv trunk's activation slab, so the
 * frames land where the trunk reads them (no copy); `xt` (optional) receives the same frames time-major [T,B,fs], the layout
 * the weight-gradient products over all frames read. */
int ag_gfront_fwd_persist(float* gates, const float* w_x, int ldwx, const float* w_hh, const float* w_p,
                          const float* b_p, float* hs, float* cs, float* x, int64_t ldx, float* xt, void* ws,
                          int64_t ws_bytes, int T, int B, int S, int fs, int n_cu, void* stream);

/* The frame loop of the Generator front's BACKWARD through time (audiogan.py:437-443 under .backward() :903) in one
 * persistent launch: per frame gx_t = (dx_ext[t] + dgates_{t+1} W_x)(1 - x_t^2), dh_t = dh_ext[t] + dgates_{t+1} W_hh +
 * gx_t W_p, dgates_t = cell backward.  ga [T,B,4S] activated gates and c_all [T+1,B,S] as saved by ag_gfront_fwd_persist,
 * x [B,T*fs] (row pitch ldx); the external gradients dh_ext [T,B,S] = dL/dh_t (the stop head's) and dx_ext [B,T*fs] =
 * dL/dx_t (the conv trunk's; row pitch lddx - it may be channel 0 of the trunk's gradient slab), each read only and each
 * NULL = zero; w_hh [4S,S], w_x = W_ih[:, :fs] (row pitch ldwx), w_p [fs,S]; outputs dgs [T,B,4S] and dxt [T,B,fs] (d pre-tanh of the projection), which the weight-gradient
 * GEMMs over all frames read.  Supported: (S, fs) as above and ceil(B/32) * (S+fs)/16 <= n_cu (ag_gfront_bwd_persist_ok);
 * `ws`: the sticky word + 8 KiB header (see ag_lstm_seq_fwd_persist). */
int ag_gfront_bwd_persist_ok(int B, int S, int fs, int n_cu);
int ag_gfront_bwd_persist(const float* ga, const float* c_all, const float* x, int64_t ldx, const float* dh_ext,
                          const float* dx_ext, int64_t lddx, const float* w_hh, const float* w_x, int ldwx,
                          const float* w_p, float* dgs, float* dxt, void* ws, int64_t ws_bytes, int T, int B, int S, int fs,
                          int n_cu, void* stream);
/* The same for the GRU-front generator (BASELINE configs[3]; torch.nn.GRUCell backward, gate order r z n): ga [T,B,3S]
 * activated gates, hs [T+1,B,S] (hs[t] = h_{t-1}, hs[0] = 0) and gh [T,B,3S] (n slot = W_hn h_{t-1} + b_hn) as saved by
 * ag_grufront_fwd_persist; outputs dgi [T,B,3S] (d of the input-side pre-activations), dgh [T,B,3S] (hidden side: the n slot
 * times r) and dxt [T,B,fs].  Shapes and workspace as ag_gfront_bwd_persist. */
int ag_grufront_bwd_persist(const float* ga, const float* hs, const float* gh, const float* x, int64_t ldx,
                            const float* dh_ext, const float* dx_ext, int64_t lddx, const float* w_hh, const float* w_x,
                            int ldwx, const float* w_p, float* dgi, float* dgh, float* dxt, void* ws, int64_t ws_bytes, int T,
                            int B, int S, int fs, int n_cu, void* stream);

/* The same frame loop with a GRU cell (BASELINE configs[3]: the audiogan.py Generator with the LSTMCell of :380-386 replaced
 * by a GRU cell, gate order r z n as torch.nn.GRUCell) as ONE persistent launch.  gates [T,B,3S]: in = W_ih[:, fs:] zc_t +
 * b_ih + (b_hr, b_hz, 0), out = activated (r, z, n); gh [T,B,3S]: only its n slot is written (W_hn h_{t-1} + b_hn, what
 * ag_gru_cell_bwd reads); w_x = W_ih[:, :fs] (row pitch ldwx), w_hh [3S,S], b_hn [S] = b_hh[2S:], w_p [fs,S], b_p [fs];
 * outputs hs [T,B,S] and the frames x [B,T*fs].  Shapes and workspace as for ag_gfront_fwd_persist. */
int ag_grufront_fwd_persist(float* gates, float* gh, const float* w_x, int ldwx, const float* w_hh, const float* b_hn,
                            const float* w_p, const float* b_p, float* hs, float* x, int64_t ldx, float* xt, void* ws,
                            int64_t ws_bytes, int T, int B, int S, int fs, int n_cu, void* stream);

/* One fused backward step of the Generator front (audiogan.py:428-460: LSTMCell -> tanh(Linear) fed back), frame t:
 *   gx     = dxa * (1 - x_t^2)                        d(pre-tanh) of the projection, stored to gx_out [B,Kp]
 *   dh     = dh_acc + gx * w_proj                      w_proj [Kp = frame size, H]; dh_acc [B,H] rows, pitch lddh
 *   (dgates, dc_prev) = LSTMCell backward(dh, dc_next; gates, c_prev, c_new)
 * ONE launch instead of tanh-backward + product + cell-backward.  dxa / x / gx_out are [B,Kp] row views (pitches in
 * floats, multiples of 4, 16-byte aligned); Kp % 16 == 0, H % 16 == 0; dc_next may be NULL (last frame). */
int ag_lstm_front_bwd_step(const float* dxa, int lddxa, const float* x, int ldx, float* gx_out, int ldgx, int Kp,
                           const float* w_proj, const float* dh_acc, int lddh, const float* gates, const float* c_prev,
                           const float* c_new, const float* dc_next, float* dgates, float* dc_prev, int B, int H,
                           void* stream);

/* ---------------------------------------------------------------------------
 * Masked BCE-with-logits per sample (audiogan.py:187-197 + :204-211 + the
 * "/ nframes ... .mean()" at :739-740,766,780,864,897), fused:
 *   l[b,t]  = x - x*target + m + log(exp(-m) + exp(-x-m)),  m = max(-x, 0)
 *   per[b]  = sum_{t < n[b]} l[b,t]               n = nframes_i64 (NULL: all T)
 *   loss   += scale * sum_b per[b] / n[b]         (scale = 1/B gives the mean)
 * bwd: dx[b,t] = gscale * scale / n[b] * (sigmoid(x) - target) * (t < n[b])
 * ------------------------------------------------------------------------- */
int ag_bce_logits_fwd(const float* x, int ldx, float target, const int64_t* nframes_i64,
                      float* per_sample, float* loss, float scale, int B, int T, void* stream);
int ag_bce_logits_bwd(const float* x, int ldx, float target, const int64_t* nframes_i64,
                      const float* gscale_dev, float scale, float* dx, int lddx, int B, int T,
                      void* stream);

/* The same loss on logits of any (row, column) pitch (element strides sxb, sxt), an optional target per row (target_rows,
 * NULL = `target` for every row) and loss[0] WRITTEN (= scale * sum_b per[b] / n[b]) by one launch; per_sample may be NULL.
 * The critic iteration scores real and fake clips in one pass (audiogan.py:739-740 + :766 + :780: targets 0.9 and 0). */
int ag_bce_logits_fwd_strided(const float* x, int64_t sxb, int64_t sxt, float target, const float* target_rows,
                              const int64_t* nframes_i64, float* per_sample, float* loss, float scale, int B, int T,
                              void* stream);
int ag_bce_logits_bwd_strided(const float* x, int64_t sxb, int64_t sxt, float target, const float* target_rows,
                              const int64_t* nframes_i64, const float* gscale_dev, float scale, float* dx, int64_t sdb,
                              int64_t sdt, int B, int T, void* stream);

/* ---------------------------------------------------------------------------
 * Elementwise helpers
 * ------------------------------------------------------------------------- */
/* y = act(x) on n contiguous floats (in place allowed) */
int ag_act_fwd(const float* x, float* y, int64_t n, int act, float slope, void* stream);
/* dx = dy * act'(.) using the saved OUTPUT y */
int ag_act_bwd(const float* dy, const float* y, float* dx, int64_t n, int act, float slope,
               void* stream);
/* same on row-strided 2-D views [rows, cols] (frame slices of the generator's [B, T*frame] output) */
int ag_act_bwd2d(const float* dy, int lddy, const float* y, int ldy, float* dx, int lddx, int rows,
                 int cols, int act, float slope, void* stream);
/* y = a*x + b*y  on n contiguous floats */
int ag_axpby(const float* x, float* y, int64_t n, float a, float b, void* stream);

/* Feature-matching statistics over time of one activation h [B,C,L] (calc_dists, audiogan.py:341-350):
 *   m[b,c] = sum_t h / len[b];  cen_t = h_t - m * [t < len[b]];
 *   s[b,c] = sqrt(sum_t cen^2) / len[b];  f[b,c] = (sum_t cen^4)^(1/4) / len[b]
 * (sums over ALL t: the critic has zeroed padded steps).  bwd: dh from the gradients of m, s, f (any may be NULL). */
int ag_time_moments_fwd(const float* h, int64_t bs, int64_t cs, const int64_t* lens_i64, float* m, float* s, float* f,
                        int B, int C, int L, void* stream);
int ag_time_moments_bwd(const float* h, int64_t bs, int64_t cs, const int64_t* lens_i64, const float* gm,
                        const float* gs, const float* gf, float* dh, int64_t dbs, int64_t dcs, int B, int C, int L,
                        void* stream);

/* ---------------------------------------------------------------------------
 * Fused optimiser: check_grad + per-PARAMETER clip + update for a whole network
 * in two launches (audiogan.py:232-253, 693-694, 786-788, 909-921).
 *   pass 1: norm[i] = ||grad_i||_2 ; flags |= NaN / |g|>1e5
 *   pass 2: g = grad * min(1, clip/norm[i]) (clip == 0: no clip), then
 *     RMSprop (torch defaults): sq = a*sq + (1-a) g^2 ; p -= lr * g / (sqrt(sq)+eps)
 *     Adam (TF/torch defaults): m,v moments with bias correction from `step`
 * `norm_sum` receives sum_i norm[i] (the value clip_grad returns, :253).
 * ------------------------------------------------------------------------- */
typedef struct ag_opt_desc {
  float* p;
  const float* grad;
  float* s1; /* RMSprop: square_avg ; Adam: exp_avg    */
  float* s2; /* Adam: exp_avg_sq (unused for RMSprop)  */
  int64_t n;
} ag_opt_desc;

#define AG_OPT_RMSPROP 0
#define AG_OPT_ADAM 1
#define AG_FLAG_NAN 1
#define AG_FLAG_BIG 2

/* step_dev (optional): device-side optimiser step counter, incremented by ag_grad_norms and
 * read by ag_opt_step for Adam's bias correction, so a captured hipGraph of the train step
 * advances it on every replay.  When NULL the host value `step` is used.
 * ag_grad_norms needs a bound workspace of 2 * n * 256 floats (ag_bind_workspace): per-chunk sums of squares and flag
 * words, every slot written by its workgroup (nothing is zeroed in front of the launch, `flags` is WRITTEN).  finish = 1:
 * a second, one-workgroup launch turns the partials into norms / norm_sum / flags.  finish = 0 (round 4): that launch is
 * left out and the caller hands the same workspace to ag_opt_step as `part`: every workgroup of the update then sums its
 * tensor's 256 partials itself (same order, same value) and workgroup (0,0) writes norms_out / norm_sum / flags - two
 * launches per network and step instead of three plus two memset nodes.  part = NULL: `norms` as computed before. */
int ag_grad_norms(const ag_opt_desc* descs_dev, int n, float* norms, float* norm_sum,
                  int32_t* flags, float grad_scale, int32_t* step_dev, int finish, void* stream);
int ag_opt_step(const ag_opt_desc* descs_dev, int n, const float* norms, int kind, float lr,
                float clip, float grad_scale, float alpha_or_beta1, float beta2, float eps, int step,
                const int32_t* step_dev, const float* part, float* norms_out, float* norm_sum, int32_t* flags,
                void* stream);

/* ---- Conv2DLSTMCell (reference cells.py:4-103: convolutional LSTM with peepholes and TF layer normalisation) ----------
 * Pointwise / normalisation pieces (csrc/convlstm.hip); the convolution runs on ag_conv1d_engine, one launch per kernel row.
 * Every map is [H, B, C, W] contiguous (rows x batch = the 1-D engine's batch axis, W = its time axis); peephole weights
 * [H, F, W]; gate blocks of the convolution output y [H,B,4F,W] along C in TF.split order j | i | f | o (cells.py:66).
 *   peephole_fwd  j, i_pre = i + W_ci c, f_pre = f + W_cf c, o_raw  (cells.py:66-70; w_ci / w_cf may be NULL: no peepholes)
 *   cell_fwd      c' = c sigmoid(f + forget_bias) + sigmoid(i) tanh(j);  o_pre = o_raw + W_co c'   (:77-82)
 *   out_fwd       h = sigmoid(o) tanh(c)                                                               (:88-89)
 *   layer_norm    tf.contrib.layers.layer_norm: per sample over (H, W, F), gamma / beta per feature, eps inside the sqrt
 * The backward forms take the forward's inputs again (nothing is saved besides mean / rstd); dc of peephole_bwd is
 * ACCUMULATED into (it also receives the cell path's dc); weight gradients are written (summed over the batch in a fixed
 * order); layer_norm_bwd writes per-sample partials [B,F] of dgamma / dbeta for a fixed-order column sum by the caller. */
int ag_convlstm_peephole_fwd(const float* y, const float* c, const float* w_ci, const float* w_cf, float* j, float* i_pre,
                             float* f_pre, float* o_raw, int H, int B, int F, int W, void* stream);
int ag_convlstm_peephole_bwd(const float* dj, const float* di, const float* df, const float* d_o, const float* c,
                             const float* w_ci, const float* w_cf, float* dy, float* dc, float* dw_ci, float* dw_cf, int H,
                             int B, int F, int W, void* stream);
int ag_convlstm_cell_fwd(const float* j, const float* i_, const float* f_, const float* c, const float* o_raw,
                         const float* w_co, float forget_bias, float* c_new, float* o_pre, int H, int B, int F, int W,
                         void* stream);
int ag_convlstm_cell_bwd(const float* j, const float* i_, const float* f_, const float* c, const float* c_new,
                         const float* w_co, float forget_bias, float* dc_new, const float* do_pre, float* dj, float* di,
                         float* df, float* dc, float* dw_co, int H, int B, int F, int W, void* stream);
int ag_convlstm_out_fwd(const float* o, const float* c, float* h, int64_t n, void* stream);
int ag_convlstm_out_bwd(const float* o, const float* c, const float* dh, float* d_o, float* dc, int64_t n, void* stream);
int ag_layer_norm_hbfw_fwd(const float* x, const float* gamma, const float* beta, float eps, float* y, float* mean, float* rstd,
                           int H, int B, int F, int W, void* stream);
int ag_layer_norm_hbfw_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd, float* dx,
                           float* dgamma_part, float* dbeta_part, int H, int B, int F, int W, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AUDIOGAN_HIP_H */
