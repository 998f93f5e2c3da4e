"""Recurrent blocks of the hot path.

  LSTMSeqFn  NN.LSTM layer over a padded batch      audiogan.py:498-503, :543, :214-229, :315
  GFrontFn   LSTMCell stack + tanh(proj) feedback   audiogan.py:377-386, :409-410, :428-460

The per-time-step work is a product with only B (clips per GPU) rows.  When shapes allow
(B <= 64, sizes multiples of 8, 16-byte aligned rows -- true for every BASELINE config) the
steps run on the fused / skinny kernels of lstm_step.hip, a whole NN.LSTM layer being enqueued
by ONE C call; other shapes take the generic GEMM + pointwise path.  Everything that does not
depend on the recurrence (input projections, weight gradients, stopper head) is batched over
all time steps into a few large GEMMs.
"""
import torch

from . import kernels as K
from .kernels import ACT_NONE, ACT_TANH
from .common import WNGroup, _zeros_like_list, grad_target


# set inside losses.only_stopper_trains: the backward that runs there is the REINFORCE update of the stop head, whose
# gradient is wanted in the stopper's parameters ONLY (audiogan.py:897-903) - not in the block's input either
STOPPER_ONLY = [False]


def _gh(*a, **k):
    from .ops import _gh as f
    return f(*a, **k)


def _small(A, B, Cm, tb=False, beta=0.0, bias=None, act=ACT_NONE, res=None):
    """Cm = act(A @ op(B) + beta*Cm + bias + res) for a product with few rows"""
    if res is None and K.skinny_ok(A, B, tb):
        K.skinny_gemm(A, B, Cm, tb=tb, beta=beta, bias=bias, act=act)
    else:
        K.gemm(A, B, Cm, tb=tb, beta=beta, bias=bias, res=res, act=act)


def _linear1(x, w, b, y):
    """y [M,1] = x @ w^T + b for a Linear with ONE output (the stop head): one wave per row (ag_rowdot_fwd) instead of a
    64-wide MFMA tile that is 98 % padding plus a split-K second stage"""
    if K.rowdot_ok(x, w):
        K.rowdot_fwd(x, w, b, y)
    else:
        K.gemm(x, w, y, tb=True, bias=b)


def _bcast_rows(v, rows):
    """[n] -> a [rows, n] view whose rows are all `v` (row pitch 0): a second bias for K.gemm's `res`"""
    return v.view(1, -1).expand(rows, v.numel())


def _small_acc(A, B, Cm, tb=False):
    """Cm += A @ op(B)"""
    if K.skinny_ok(A, B, tb):
        K.skinny_gemm(A, B, Cm, tb=tb, atomic=True)
    else:
        K.gemm(A, B, Cm, tb=tb, beta=1.0)


# --------------------------------------------------------------------------------------
# LSTM layer over a padded sequence (uni- or bidirectional), NN.LSTM parameter layout
# --------------------------------------------------------------------------------------
_LSTM_IMAGES = {}      # id(first weight Parameter of a layer) -> common.Bf16Images (bf16 images of that layer's weights)


def _lstm_images(w):
    from .common import Bf16Images
    key = id(w[0])
    im = _LSTM_IMAGES.get(key)
    if im is None:
        if len(_LSTM_IMAGES) > 64:
            _LSTM_IMAGES.clear()
        im = _LSTM_IMAGES[key] = Bf16Images()
    return im


def _wih16(w, ndir, Fx):
    """[ndir * 4H, Fx] bfloat16: the x-columns of every direction's W_ih stacked - the B operand of the input projections
    (per direction: a row block, k contiguous) and of the input gradient dx = [dg_0 | dg_1] @ stack (k strided)"""
    im = _lstm_images(w)
    H4 = w[0].size(0)
    stack = getattr(im, 'stack', None)
    if stack is None or stack.shape != (ndir * H4, Fx) or stack.device != w[0].device:
        stack = im.stack = torch.empty(ndir * H4, Fx, device=w[0].device, dtype=torch.bfloat16)
        im.keys = [None] * ndir
    from .common import capture_tag, param_epoch
    for d in range(ndir):
        p_ = w[4 * d]
        key = (capture_tag(p_.device), p_.data_ptr(), p_._version, param_epoch(p_))
        if im.keys[d] != key:
            K.to_bf16(p_.data[:, :Fx], out=stack[d * H4:(d + 1) * H4])
            im.keys[d] = key
    return stack


class LSTMSeqFn(torch.autograd.Function):
    """x: [T,B,Fx]; lengths: int64 [B] on device or None; ``static``: None or [B,Fc], a time-invariant part
    of the input (the layer sees cat([x_t, static]) at every step, audiogan.py:541); weights per direction:
    (w_ih [4H,Fx+Fc], w_hh [4H,H], b_ih [4H], b_hh [4H]).  Returns y [T,B,D*H] with zeros at padded steps
    (pad_packed_sequence semantics, audiogan.py:214-229).

    The static part is projected ONCE per clip (c @ W_ih[:, Fx:]^T + biases -> [B,4H]) and added inside the
    step kernel, instead of being concatenated to all T frames and multiplied T times; its weight / input
    gradients come from the time sum of dgates (the same pass that gives the bias gradient)."""

    @staticmethod
    def forward(ctx, x, lengths, ndir, static, *w):
        T, B, Fx = x.shape
        H = w[1].size(1)
        dev = x.device
        x2 = x.contiguous().view(T * B, Fx)
        Fc = static.size(1) if static is not None else 0
        assert w[0].size(1) == Fx + Fc
        st = static.contiguous() if static is not None else None
        # bf16 storage (x arrives as bfloat16): x, the layer output y, dgates' operand copy and dx are bfloat16 in HBM and
        # every large product runs on ag_gemm_h; gate pre-activations, cell states and the persistent kernels' exchange stay
        # fp32.  Needs the persistent launches (they write y / read dy as bf16).
        s16 = x.dtype == torch.bfloat16
        if s16:
            assert K.lstm_step_ok(B, H) and K.lstm_persist_ok(B, H, ndir, dev) and K.lstm_persist_bwd_ok(B, H, ndir, dev), \
                'bf16 storage needs the persistent LSTM launches (modules.Discriminator.classify checks this)'
            wst = _wih16(w, ndir, Fx)
        y = torch.empty(T, B, ndir * H, device=dev, dtype=torch.bfloat16 if s16 else torch.float32)
        gates_all, c_all, whh, cbs = [], [], [], []
        fused = K.lstm_step_ok(B, H)
        for d in range(ndir if s16 else 0):
            w_ih, w_hh, b_ih, b_hh = w[4 * d:4 * d + 4]
            g = torch.empty(T, B, 4 * H, device=dev)
            wx16 = wst[d * 4 * H:(d + 1) * 4 * H]
            if st is not None:
                cb = torch.empty(B, 4 * H, device=dev)
                K.gemm(st, w_ih.data[:, Fx:], cb, tb=True, bias=b_ih.data, res=_bcast_rows(b_hh.data, B))
                _gh(x2, wx16, C=g.view(T * B, 4 * H), tb=True)
                cbs.append(cb)
            else:
                _gh(x2, wx16, C=g.view(T * B, 4 * H), tb=True, bias=b_ih.data, res=_bcast_rows(b_hh.data, T * B))
            gates_all.append(g)
            c_all.append(torch.empty(T + 1, B, H, device=dev))
            whh.append(w_hh.data.contiguous())
        for d in range(0 if s16 else ndir):
            w_ih, w_hh, b_ih, b_hh = w[4 * d:4 * d + 4]
            g = torch.empty(T, B, 4 * H, device=dev)
            # b_ih + b_hh: one rides as the GEMM's bias, the other as a `res` whose row pitch is 0 (the same row for every
            # output row) - no launch to add the two vectors first
            if st is not None:
                cb = torch.empty(B, 4 * H, device=dev)
                K.gemm(st, w_ih.data[:, Fx:], cb, tb=True, bias=b_ih.data, res=_bcast_rows(b_hh.data, B))
                if fused:
                    K.gemm(x2, w_ih.data[:, :Fx], g.view(T * B, 4 * H), tb=True)
                    cbs.append(cb)
                else:       # generic per-step path: fold the static term into the pre-activations up front
                    g.copy_(cb.unsqueeze(0).expand(T, B, 4 * H))
                    K.gemm(x2, w_ih.data[:, :Fx], g.view(T * B, 4 * H), tb=True, beta=1.0)
            else:
                K.gemm(x2, w_ih.data, g.view(T * B, 4 * H), tb=True, bias=b_ih.data, res=_bcast_rows(b_hh.data, T * B))
            c = torch.empty(T + 1, B, H, device=dev)   # c[k+1] = cell after the k-th processed step
            if not (fused and K.lstm_persist_ok(B, H, ndir, dev)):
                c[0].zero_()                           # (the persistent launch writes c_0 = 0 itself)
            gates_all.append(g)
            c_all.append(c)
            whh.append(w_hh.data.contiguous())
        hbuf = [torch.empty(2, B, H, device=dev) for _ in range(ndir)]
        if fused:
            K.lstm_seq_fwd(gates_all, whh, c_all, hbuf, y, lengths, static=cbs if cbs else None)
        else:
            for d in range(ndir):
                g, c = gates_all[d], c_all[d]
                hbuf[d][0].zero_()
                for k in range(T):
                    t = k if d == 0 else T - 1 - k
                    hp, hn = hbuf[d][k & 1], hbuf[d][(k + 1) & 1]
                    if k > 0:
                        K.gemm(hp, whh[d], g[t], tb=True, beta=1.0)
                    K.lstm_cell_fwd(g[t], c[k], c[k + 1], h_out=hn, y_out=y[t, :, d * H:(d + 1) * H],
                                    h_prev=hp, valid=lengths, t=t)
        ctx.ndir, ctx.has_len, ctx.has_static, ctx.s16 = ndir, lengths is not None, st is not None, s16
        ctx.params = w
        ctx.save_for_backward(x2, y, lengths if lengths is not None else x2.new_empty(0),
                              st if st is not None else x2.new_empty(0),
                              *(gates_all + c_all + [t_.data for t_ in w]))
        ctx.shape = (T, B, Fx, H)
        return y

    @staticmethod
    def backward(ctx, dy):
        T, B, Fx, H = ctx.shape
        ndir = ctx.ndir
        sv = ctx.saved_tensors
        x2, y = sv[0], sv[1]
        lengths = sv[2] if ctx.has_len else None
        st = sv[3] if ctx.has_static else None
        gates_all, c_all, w = list(sv[4:4 + ndir]), list(sv[4 + ndir:4 + 2 * ndir]), sv[4 + 2 * ndir:]
        dev = x2.device
        dy = dy.contiguous()
        if ctx.s16:
            return LSTMSeqFn._backward16(ctx, dy, x2, y, lengths, st, gates_all, c_all, w)
        whh = [w[4 * d + 1].contiguous() for d in range(ndir)]
        dgs = [torch.empty(T, B, 4 * H, device=dev) for _ in range(ndir)]
        dhb = [torch.empty(2, B, H, device=dev) for _ in range(ndir)]
        dcb = [torch.empty(2, B, H, device=dev) for _ in range(ndir)]
        wg = any(ctx.needs_input_grad[4:])
        need_ds = st is not None and ctx.needs_input_grad[3]
        # sum over time of dgates (the static input and the biases see every step's gradient): the persistent launch sums it
        # in the registers of the threads that produce dgates; other paths leave it to a pass over dgates below
        want_sum = st is not None and (wg or need_ds)
        dgsums = [torch.empty(B, 4 * H, device=dev) for _ in range(ndir)] if want_sum else None
        have_sum = False
        if B <= 256 and H % 2 == 0:
            have_sum = K.lstm_seq_bwd(gates_all, whh, c_all, dy, dgs, dhb, dcb, lengths, dgsum=dgsums)
        else:
            for d in range(ndir):
                g, c, dg = gates_all[d], c_all[d], dgs[d]
                for k in reversed(range(T)):
                    t = k if d == 0 else T - 1 - k
                    dpass = dhb[d][(k + 1) & 1]
                    K.lstm_cell_bwd(g[t], c[k], c[k + 1], None if k == T - 1 else dhb[d][k & 1],
                                    dy[t, :, d * H:(d + 1) * H], None if k == T - 1 else dcb[d][(k + 1) & 1],
                                    dg[t], dcb[d][k & 1], dh_pass=dpass, valid=lengths, t=t)
                    if k > 0:
                        K.gemm(dg[t], whh[d], dpass, beta=1.0)
        dx2 = torch.empty(T * B, Fx, device=dev)
        outs = []
        dstatic = torch.zeros_like(st) if need_ds else None
        if dgsums is None:
            dgsums = [None] * ndir
        for d in range(ndir):
            w_ih = w[4 * d]
            wx = w_ih[:, :Fx] if st is not None else w_ih
            dg2 = dgs[d].view(T * B, 4 * H)
            K.gemm(dg2, wx, dx2, beta=0.0 if d == 0 else 1.0)
            dgsum = dgsums[d]
            if want_sum:
                if not have_sum:
                    # (read right below and by the weight-gradient products: complete at once, also inside a deferral scope)
                    K.col_sum(dgs[d].view(T, B * 4 * H), dgsum.view(-1), accumulate=False, defer=False)
                if need_ds:
                    K.gemm(dgsum, w_ih[:, Fx:], dstatic, beta=1.0)
        if not wg:
            outs = [None] * (4 * ndir)
        tgs = [[grad_target(p_) for p_ in ctx.params[4 * d:4 * d + 4]] for d in range(ndir)] if wg else []
        # every parameter gradient goes straight into ``.grad``: nothing reads it before the optimiser, so the second
        # stages of the split-K products and column sums below may wait for the enclosing scope's ONE launch
        direct_all = wg and all(t_ is not None for tg in tgs for t_ in tg)
        import contextlib
        with (K.deferred_reduces() if direct_all else contextlib.nullcontext()):
            for d in range(ndir if wg else 0):
                w_ih, w_hh = w[4 * d], w[4 * d + 1]
                dg2 = dgs[d].view(T * B, 4 * H)
                dgsum = dgsums[d]
                tg = tgs[d]
                direct = all(t_ is not None for t_ in tg)
                # h_prev of processing step k is the layer output of step k-1 (zero at padded steps,
                # where dgates is zero as well)
                if direct:      # accumulate straight into .grad
                    dw_ih, dw_hh = tg[0], tg[1]
                else:
                    dw_ih = torch.zeros_like(w_ih)
                    dw_hh = torch.zeros_like(w_hh)
                K.gemm(dg2, x2, dw_ih[:, :Fx] if st is not None else dw_ih, ta=True, beta=1.0, defer=direct_all)
                if st is not None:
                    K.gemm(dgsum, st, dw_ih[:, Fx:], ta=True, beta=1.0, defer=direct_all)
                if T > 1:
                    if d == 0:
                        K.gemm(dgs[d][1:].view((T - 1) * B, 4 * H), y[:-1].view((T - 1) * B, ndir * H)[:, :H],
                               dw_hh, ta=True, beta=1.0, defer=direct_all)
                    else:
                        K.gemm(dgs[d][:-1].view((T - 1) * B, 4 * H),
                               y[1:].view((T - 1) * B, ndir * H)[:, H:2 * H], dw_hh, ta=True, beta=1.0, defer=direct_all)
                src, rows = (dgsum, B) if dgsum is not None else (dg2, T * B)
                if direct:
                    K.col_sum(src.view(rows, 4 * H), tg[2])
                    K.col_sum(src.view(rows, 4 * H), tg[3])
                    outs += [None, None, None, None]
                    continue
                db = torch.zeros(4 * H, device=dev)
                K.col_sum(src.view(rows, 4 * H), db)
                outs += [dw_ih, dw_hh, db, db.clone()]
        dx = dx2.view(T, B, Fx) if ctx.needs_input_grad[0] else None
        return (dx, None, None, dstatic) + tuple(outs)


def _lstm_backward16(ctx, dy, x2, y, lengths, st, gates_all, c_all, w):
    """LSTMSeqFn.backward on bf16 storage: dy, x2, y are bfloat16; the persistent launch leaves dgates in fp32 (its exchange
    buffer) AND as one bfloat16 tensor [T,B,ndir*4H], the operand of the three gradient products (ag_gemm_h)"""
    T, B, Fx, H = ctx.shape
    ndir = ctx.ndir
    dev = x2.device
    H4 = 4 * H
    whh = [w[4 * d + 1].contiguous() for d in range(ndir)]
    dgs = [torch.empty(T, B, H4, device=dev) for _ in range(ndir)]
    dg16 = torch.empty(T, B, ndir * H4, device=dev, dtype=torch.bfloat16)
    wg = any(ctx.needs_input_grad[4:])
    need_ds = st is not None and ctx.needs_input_grad[3]
    want_sum = st is not None and (wg or need_ds)
    dgsums = [torch.empty(B, H4, device=dev) for _ in range(ndir)] if want_sum else None
    have_sum = K.lstm_seq_bwd(gates_all, whh, c_all, dy, dgs, None, None, lengths, dgsum=dgsums,
                              dg16=[dg16[:, :, d * H4:(d + 1) * H4] for d in range(ndir)])
    assert have_sum or not want_sum
    dg2 = dg16.view(T * B, ndir * H4)
    dx2 = None
    if ctx.needs_input_grad[0]:
        dx2 = torch.empty(T * B, Fx, device=dev, dtype=torch.bfloat16)
        _gh(dg2, _wih16(w, ndir, Fx), C16=dx2)            # both directions in ONE product (K = ndir * 4H)
    dstatic = torch.zeros_like(st) if need_ds else None
    for d in range(ndir if need_ds else 0):
        K.gemm(dgsums[d], w[4 * d][:, Fx:], dstatic, beta=1.0)
    outs = []
    if not wg:
        outs = [None] * (4 * ndir)
    tgs = [[grad_target(p_) for p_ in ctx.params[4 * d:4 * d + 4]] for d in range(ndir)] if wg else []
    direct_all = wg and all(t_ is not None for tg in tgs for t_ in tg)
    import contextlib
    with (K.deferred_reduces() if direct_all else contextlib.nullcontext()):
        for d in range(ndir if wg else 0):
            w_ih, w_hh = w[4 * d], w[4 * d + 1]
            tg = tgs[d]
            direct = all(t_ is not None for t_ in tg)
            dw_ih, dw_hh = (tg[0], tg[1]) if direct else (torch.zeros_like(w_ih), torch.zeros_like(w_hh))
            dgd = dg2[:, d * H4:(d + 1) * H4]
            _gh(dgd, x2, C=dw_ih[:, :Fx] if st is not None else dw_ih, ta=True, beta=1.0, defer=direct_all)
            if st is not None:
                K.gemm(dgsums[d], st, dw_ih[:, Fx:], ta=True, beta=1.0, defer=direct_all)
            if T > 1:
                y2 = y.view(T * B, ndir * H)
                if d == 0:
                    _gh(dgd[B:], y2[:(T - 1) * B, :H], C=dw_hh, ta=True, beta=1.0, defer=direct_all)
                else:
                    _gh(dgd[:(T - 1) * B], y2[B:, H:2 * H], C=dw_hh, ta=True, beta=1.0, defer=direct_all)
            src, rows = (dgsums[d], B) if want_sum else (dgs[d].view(T * B, H4), T * B)
            if direct:
                K.col_sum(src.view(rows, H4), tg[2])
                K.col_sum(src.view(rows, H4), tg[3])
                outs += [None, None, None, None]
                continue
            db = torch.zeros(H4, device=dev)
            K.col_sum(src.view(rows, H4), db)
            outs += [dw_ih, dw_hh, db, db.clone()]
    dx = dx2.view(T, B, Fx) if dx2 is not None else None
    return (dx, None, None, dstatic) + tuple(outs)


LSTMSeqFn._backward16 = staticmethod(_lstm_backward16)


# --------------------------------------------------------------------------------------
# Generator recurrent front: T x [LSTMCell stack -> tanh(proj) fed back, stopper logit]
# --------------------------------------------------------------------------------------
def _frames_buffer(front, B, n, dev):
    """where the front writes its frames x [B, n].  When the block feeds a conv trunk that keeps its activations in one
    [B, ctot, n] slab (``front.slab_channels = ctot``, set by the Generator), x IS channel 0 of a fresh slab - the trunk
    finds its input in place (ops.GTrunkFn) instead of copying 2 MB per forward."""
    ctot = getattr(front, 'slab_channels', 0)
    # (device tensors only: the kernels write through raw pointers; under the CPU kernel model of the tests torch's version
    # counter of the shared storage would see the trunk's writes as in-place edits of the saved frames)
    if ctot and ctot > 1 and torch.device(dev).type == 'cuda':
        return torch.empty(B, ctot, n, device=dev)[:, 0, :]
    return torch.empty(B, n, device=dev)


class GFront(object):
    """WN items: per layer [w_ih, w_hh, b_ih, b_hh] * num_layers, then [proj.w, proj.b, stop.w, stop.b]"""

    def __init__(self, frame_size, num_layers, state_size):
        self.fs, self.nl, self.ss = frame_size, num_layers, state_size
        self.slab_channels = 0
        self.group = WNGroup()


class GFrontFn(torch.autograd.Function):
    """zc: [T,B,noise+embed] contiguous.  Returns x [B,T*fs], s [B,T].  Frame t's LSTM input is
    [x_{t-1}, zc_t] (audiogan.py:439); the zc part of every frame's gate product is done in one
    GEMM up front, only the fed-back x_{t-1} and h products stay in the sequential loop."""

    @staticmethod
    def forward(ctx, zc, front, *params):
        T, B, Fz = zc.shape
        fs, nl, S = front.fs, front.nl, front.ss
        dev = zc.device
        ctx.set_materialize_grads(False)
        prep = front.group.prepare()
        lw = [[p.w for p in prep[4 * l:4 * l + 4]] for l in range(nl)]
        pw, pb, sw, sb = [p.w for p in prep[4 * nl:4 * nl + 4]]
        x = _frames_buffer(front, B, T * fs, dev)
        gates = [torch.empty(T, B, 4 * S, device=dev) for _ in range(nl)]
        hs = [torch.empty(T, B, S, device=dev) for _ in range(nl)]
        cs = [torch.empty(T + 1, B, S, device=dev) for _ in range(nl)]
        w_ih0 = lw[0][0]
        wx, wz = w_ih0[:, :fs], w_ih0[:, fs:]
        # all frames at once: zc_t @ W_ih[:, fs:]^T + b_ih + b_hh (the second bias as a `res` of row pitch 0: no launch to
        # add the two vectors first)
        K.gemm(zc.contiguous().view(T * B, Fz), wz, gates[0].view(T * B, 4 * S), tb=True, bias=lw[0][2],
               res=_bcast_rows(lw[0][3], T * B))
        fused0 = K.lstm_step_ok(B, S, x[:, :fs], wx)
        persist = nl == 1 and T > 0 and wx.stride(1) == 1 and K.gfront_persist_ok(B, S, fs, dev)
        xt = None
        if not persist:
            for l in range(nl):
                cs[l][0].zero_()                       # (the persistent launch writes c_0 = 0 itself)
        h0 = None if persist else torch.zeros(B, S, device=dev)      # (only the per-frame path reads it)
        bsum = None if persist else [lw[l][2] + lw[l][3] for l in range(nl)]
        if persist:
            # the whole frame loop (LSTMCell step + projection, fed back) in ONE launch, weights resident in registers; the
            # frames are also written time-major (xt): what the weight gradient of W_ih[:, :fs] reads in backward
            xt = torch.empty(T, B, fs, device=dev)
            K.gfront_fwd_persist(gates[0], wx, lw[0][1], pw, pb, hs[0], cs[0], x, xt)
        for t in range(0 if persist else T):
            xprev = x[:, (t - 1) * fs:t * fs] if t > 0 else x[:, :fs]
            if fused0:
                K.lstm_step_fwd(gates[0][t], xprev, wx, hs[0][t - 1] if t > 0 else h0, lw[0][1], cs[0][t],
                                cs[0][t + 1], hs[0][t], first_step=(t == 0))
            else:
                if t > 0:
                    K.gemm(xprev, wx, gates[0][t], tb=True, beta=1.0)
                    K.gemm(hs[0][t - 1], lw[0][1], gates[0][t], tb=True, beta=1.0)
                K.lstm_cell_fwd(gates[0][t], cs[0][t], cs[0][t + 1], h_out=hs[0][t])
            for l in range(1, nl):
                _small(hs[l - 1][t], lw[l][0], gates[l][t], tb=True, bias=bsum[l])
                if K.lstm_step_ok(B, S):
                    K.lstm_step_fwd(gates[l][t], None, None, hs[l][t - 1] if t > 0 else h0, lw[l][1],
                                    cs[l][t], cs[l][t + 1], hs[l][t], first_step=(t == 0))
                else:
                    if t > 0:
                        K.gemm(hs[l][t - 1], lw[l][1], gates[l][t], tb=True, beta=1.0)
                    K.lstm_cell_fwd(gates[l][t], cs[l][t], cs[l][t + 1], h_out=hs[l][t])
            _small(hs[-1][t], pw, x[:, t * fs:(t + 1) * fs], tb=True, bias=pb, act=ACT_TANH)
        s = torch.empty(T * B, 1, device=dev)
        _linear1(hs[-1].view(T * B, S), sw, sb, s)
        ctx.front, ctx.key = front, front.group._key[1:]
        ctx.dims = (T, B, Fz)
        ctx.has_xt = xt is not None
        ctx.save_for_backward(zc, x, *(gates + hs + cs + ([xt] if xt is not None else [])))
        return x, s.view(T, B).t()

    @staticmethod
    def _bwd_frames(T, B, fs, S, nl, x, dx, ds, gates, cs, lw, wx, pw, sw, hs, dws):
        """per-frame chain, one launch per operation (any number of layers)"""
        dev = x.device
        # dxa[:, frame t] accumulates dL/dx_t: output gradient + feedback from frame t+1
        dxa = dx.contiguous().clone() if dx is not None else torch.zeros(B, T * fs, device=dev)
        # dha[l][t] accumulates dL/dh_l[t]: recurrent term from frame t+1, the layer above / the
        # projection at frame t, and (top layer) the stopper head
        dha = [torch.zeros(T, B, S, device=dev) for _ in range(nl)]
        if ds is not None:
            ds_tb = ds.t().contiguous().view(T * B, 1)
            if dws is not None:
                K.gemm(ds_tb, hs[-1].view(T * B, S), dws[4 * nl + 2], ta=True)
                K.col_sum(ds_tb, dws[4 * nl + 3])
            K.gemm(ds_tb, sw, dha[-1].view(T * B, S))
        dgs = [torch.empty(T, B, 4 * S, device=dev) for _ in range(nl)]
        dxt = torch.empty(T, B, fs, device=dev)      # d(pre-tanh) of the projection, per frame
        dcs = [[torch.zeros(B, S, device=dev), torch.empty(B, S, device=dev)] for _ in range(nl)]
        for t in reversed(range(T)):
            gx = dxt[t]
            K.act_bwd2d(dxa[:, t * fs:(t + 1) * fs], x[:, t * fs:(t + 1) * fs], gx, ACT_TANH)
            _small_acc(gx, pw, dha[-1][t])                                   # through the projection
            for l in reversed(range(nl)):
                K.lstm_cell_bwd(gates[l][t], cs[l][t], cs[l][t + 1], dha[l][t], None,
                                dcs[l][(t + 1) & 1] if t < T - 1 else None, dgs[l][t], dcs[l][t & 1])
                if t > 0:
                    _small_acc(dgs[l][t], lw[l][1], dha[l][t - 1])           # into h_l[t-1]
                if l > 0:
                    _small_acc(dgs[l][t], lw[l][0], dha[l - 1][t])           # into h_{l-1}[t]
                elif t > 0:
                    _small_acc(dgs[0][t], wx, dxa[:, (t - 1) * fs:t * fs])   # into x_{t-1}
        return dgs, dxt

    @staticmethod
    def _bwd_frames_fused(T, B, fs, S, x, dx, ds, gates, cs, w_hh, wx, pw, sw, hs, dws, nl):
        """single-layer front: ONE persistent launch where the shape fits (ag_gfront_bwd_persist), else TWO launches per
        frame: dacc[t] = [dL/dh_t | dL/dx_t] lives in one [B, S+fs] row so that  dacc[t-1] += dgates_t @ [W_hh |
        W_ih[:, :fs]]  is ONE product, and the tanh backward of the projection, its product and the cell backward are one
        fused step (ag_lstm_front_bwd_step)."""
        dev = x.device
        persist = T > 0 and wx.stride(1) == 1 and K.gfront_bwd_persist_ok(B, S, fs, dev)
        ds_tb = None
        if ds is not None:
            ds_tb = ds.t().contiguous().view(T * B, 1)
            if dws is not None:
                K.gemm(ds_tb, hs.view(T * B, S), dws[4 * nl + 2], ta=True)
                K.col_sum(ds_tb, dws[4 * nl + 3], defer=False)
        dgs = torch.empty(T, B, 4 * S, device=dev)
        dxt = torch.empty(T, B, fs, device=dev)
        if persist:
            # the whole loop in ONE launch, [W_hh | W_x] and W_p resident in registers (ag_gfront_bwd_persist).  The external
            # gradients are read where they are: dL/dx_t from the trunk's gradient (any row pitch: channel 0 of its slab),
            # dL/dh_t (the stop head's, if any) from a [T,B,S] product - no [T,B,S+fs] staging tensor to fill and copy into
            dh_ext = None
            if ds_tb is not None:
                dh_ext = torch.empty(T, B, S, device=dev)
                K.gemm(ds_tb, sw, dh_ext.view(T * B, S))
            dx_ext = dx if (dx is None or dx.stride(1) == 1) else dx.contiguous()
            K.gfront_bwd_persist(gates, cs, x, dh_ext, dx_ext, w_hh, wx, pw, dgs, dxt)
            return [dgs], dxt
        dacc = torch.zeros(T, B, S + fs, device=dev)
        if dx is not None:
            dacc[:, :, S:].copy_(dx.view(B, T, fs).transpose(0, 1))
        if ds_tb is not None:
            K.gemm(ds_tb, sw, dacc.view(T * B, S + fs)[:, :S])
        wcat = torch.cat([w_hh, wx], 1)                # [4S, S+fs]
        dcs = [torch.empty(B, S, device=dev), torch.empty(B, S, device=dev)]
        for t in reversed(range(T)):
            K.lstm_front_bwd_step(dacc[t, :, S:], x[:, t * fs:(t + 1) * fs], dxt[t], pw, dacc[t, :, :S], gates[t], cs[t],
                                  cs[t + 1], dcs[(t + 1) & 1] if t < T - 1 else None, dgs[t], dcs[t & 1])
            if t > 0:
                K.skinny_gemm(dgs[t], wcat, dacc[t - 1], atomic=True)
        return [dgs], dxt

    @staticmethod
    def backward(ctx, dx, ds):
        front = ctx.front
        fs, nl, S = front.fs, front.nl, front.ss
        T, B, Fz = ctx.dims
        prep = front.group.prepare()
        assert front.group._key[1:] == ctx.key, 'parameters changed between forward and backward'
        sv = ctx.saved_tensors
        zc, x = sv[0], sv[1]
        gates, hs, cs = sv[2:2 + nl], sv[2 + nl:2 + 2 * nl], sv[2 + 2 * nl:2 + 3 * nl]
        xt = sv[2 + 3 * nl] if ctx.has_xt else None
        dev = zc.device
        lw = [[p.w for p in prep[4 * l:4 * l + 4]] for l in range(nl)]
        pw, sw = prep[4 * nl].w, prep[4 * nl + 2].w
        wx = lw[0][0][:, :fs]
        # no parameter of the block needs a gradient (FGSM-style input-gradient passes): skip every weight-gradient
        # product and leave ``.grad`` untouched
        wg = any(ctx.needs_input_grad[2:])
        items = front.group.items
        if dx is None and ds is not None and (STOPPER_ONLY[0] or not ctx.needs_input_grad[0]) and \
                not any(it['v'].requires_grad or it['g'].requires_grad for it in items[:4 * nl + 2]):
            # REINFORCE backward of the stop head alone (audiogan.py:897-903: every other generator parameter is
            # frozen): s_t = h_t W_s^T + b_s, so only dW_s = ds^T h and db_s = sum ds are wanted - no frame loop
            dws = [None] * len(items)
            dws[4 * nl + 2], dws[4 * nl + 3] = _zeros_like_list([items[4 * nl + 2]['v'], items[4 * nl + 3]['v']])
            ds_tb = ds.t().contiguous().view(T * B, 1)
            K.gemm(ds_tb, hs[-1].view(T * B, S), dws[4 * nl + 2], ta=True)
            K.col_sum(ds_tb, dws[4 * nl + 3])
            return (None, None) + tuple(front.group.backward(dws))
        dws = front.group.zero_dws() if wg else None
        fused = (nl == 1 and T > 0 and (S + fs) % 4 == 0
                 and K.lstm_front_bwd_ok(B, S, fs, x[:, :fs], x[:, :fs]) and K.skinny_ok(gates[0][0], lw[0][1], False))
        if fused:
            dgs, dxt = GFrontFn._bwd_frames_fused(T, B, fs, S, x, dx, ds, gates[0], cs[0], lw[0][1], wx, pw, sw,
                                                  hs[0], dws, nl)
        else:
            dgs, dxt = GFrontFn._bwd_frames(T, B, fs, S, nl, x, dx, ds, gates, cs, lw, wx, pw, sw, hs, dws)
        # parameter gradients, one GEMM per tensor over all frames
        dxt2 = dxt.view(T * B, fs)
        with K.deferred_reduces():      # the bias sums' second stages in ONE launch (read only after the block)
            if wg:
                K.gemm(dxt2, hs[-1].view(T * B, S), dws[4 * nl], ta=True, defer=True)
                K.col_sum(dxt2, dws[4 * nl + 1])
            for l in range(nl if wg else 0):
                dg2 = dgs[l].view(T * B, 4 * S)
                dwih = dws[4 * l]
                if l == 0:
                    K.gemm(dg2, zc.contiguous().view(T * B, Fz), dwih[:, fs:], ta=True, defer=True)
                    if T > 1:
                        # x_{t-1} rows in (t, b) order: the persistent forward left them time-major (xt)
                        xprev = xt[:T - 1].view((T - 1) * B, fs) if xt is not None else \
                            x.view(B, T, fs)[:, :T - 1].transpose(0, 1).contiguous().view((T - 1) * B, fs)
                        K.gemm(dgs[0][1:].view((T - 1) * B, 4 * S), xprev, dwih[:, :fs], ta=True, defer=True)
                else:
                    K.gemm(dg2, hs[l - 1].view(T * B, S), dwih, ta=True, defer=True)
                if T > 1:
                    K.gemm(dgs[l][1:].view((T - 1) * B, 4 * S), hs[l][:T - 1].view((T - 1) * B, S),
                           dws[4 * l + 1], ta=True, defer=True)
                K.col_sum(dg2, dws[4 * l + 2])
        for l in range(nl if wg else 0):
            dws[4 * l + 3] = dws[4 * l + 2]       # b_hh sees the same gradient as b_ih: the weight-norm backward reads one buffer twice
        dzc = None
        if ctx.needs_input_grad[0]:
            dzc = torch.empty(T * B, Fz, device=dev)
            K.gemm(dgs[0].view(T * B, 4 * S), lw[0][0][:, fs:], dzc)
            dzc = dzc.view(T, B, Fz)
        grads = front.group.backward(dws) if wg else [None] * (2 * len(front.group.items))
        return (dzc, None) + tuple(grads)


# --------------------------------------------------------------------------------------
# GRU variant of the generator front (BASELINE config C4; oracle = torch.nn.GRUCell because the
# reference has no GRU).  One layer; same frame feedback as GFrontFn.
# --------------------------------------------------------------------------------------
class GRUFront(object):
    """WN items: [w_ih, w_hh, b_ih, b_hh, proj.w, proj.b, stop.w, stop.b]"""

    def __init__(self, frame_size, state_size):
        self.fs, self.ss = frame_size, state_size
        self.slab_channels = 0
        self.group = WNGroup()


class GRUFrontFn(torch.autograd.Function):
    """zc [T,B,Fz] -> x [B,T*fs], s [B,T];  frame t: h_t = GRUCell([x_{t-1}, zc_t], h_{t-1}),
    x_t = tanh(proj(h_t)), s_t = stopper(h_t)."""

    @staticmethod
    def forward(ctx, zc, front, *params):
        T, B, Fz = zc.shape
        fs, S = front.fs, front.ss
        dev = zc.device
        ctx.set_materialize_grads(False)
        w_ih, w_hh, b_ih, b_hh, pw, pb, sw, sb = [p.w for p in front.group.prepare()]
        wx, wz = w_ih[:, :fs], w_ih[:, fs:]
        x = _frames_buffer(front, B, T * fs, dev)
        xt = None
        gi = torch.empty(T, B, 3 * S, device=dev)      # -> activated (r, z, n)
        gh = torch.empty(T, B, 3 * S, device=dev)      # h-part incl. b_hh (n slot needed for backward)
        hs = torch.empty(T + 1, B, S, device=dev)      # hs[t+1] = h_t, hs[0] = 0
        hs[0].zero_()
        persist = T > 0 and wx.stride(1) == 1 and K.gfront_persist_ok(B, S, fs, dev)
        if persist:
            # the whole frame loop (GRU step + projection, fed back) in ONE launch, weights resident in registers: the r / z
            # halves of b_hh ride with the precomputed input part, b_hn stays apart (it sits inside r * (...))
            bias = b_ih.clone()
            bias[:2 * S] += b_hh[:2 * S]
            K.gemm(zc.contiguous().view(T * B, Fz), wz, gi.view(T * B, 3 * S), tb=True, bias=bias)
            xt = torch.empty(T, B, fs, device=dev)
            K.grufront_fwd_persist(gi, gh, wx, w_hh, b_hh[2 * S:].contiguous(), pw, pb, hs[1:], x, xt)
        else:
            K.gemm(zc.contiguous().view(T * B, Fz), wz, gi.view(T * B, 3 * S), tb=True, bias=b_ih)
        for t in range(0 if persist else T):
            if t > 0:
                _small_acc(x[:, (t - 1) * fs:t * fs], wx, gi[t], tb=True)
            _small(hs[t], w_hh, gh[t], tb=True, bias=b_hh)
            K.gru_cell_fwd(gi[t], gh[t], hs[t], hs[t + 1])
            _small(hs[t + 1], pw, x[:, t * fs:(t + 1) * fs], tb=True, bias=pb, act=ACT_TANH)
        s = torch.empty(T * B, 1, device=dev)
        _linear1(hs[1:].view(T * B, S), sw, sb, s)
        ctx.front, ctx.key, ctx.dims = front, front.group._key[1:], (T, B, Fz)
        ctx.has_xt = xt is not None
        ctx.save_for_backward(zc, x, gi, gh, hs, *([xt] if xt is not None else []))
        return x, s.view(T, B).t()

    @staticmethod
    def backward(ctx, dx, ds):
        front = ctx.front
        fs, S = front.fs, front.ss
        T, B, Fz = ctx.dims
        prep = front.group.prepare()
        assert front.group._key[1:] == ctx.key, 'parameters changed between forward and backward'
        w_ih, w_hh, _, _, pw, _, sw, _ = [p.w for p in prep]
        wx = w_ih[:, :fs]
        zc, x, gi, gh, hs = ctx.saved_tensors[:5]
        xt = ctx.saved_tensors[5] if ctx.has_xt else None
        dev = zc.device
        wg = any(ctx.needs_input_grad[2:])
        dws = front.group.zero_dws() if wg else None
        persist = T > 0 and K.gfront_bwd_persist_ok(B, S, fs, dev) and wx.stride(1) == 1
        dgi = torch.empty(T, B, 3 * S, device=dev)
        dgh = torch.empty(T, B, 3 * S, device=dev)
        dxt = torch.empty(T, B, fs, device=dev)
        dh_ext = None
        if not persist:
            dxa = dx.contiguous().clone() if dx is not None else torch.zeros(B, T * fs, device=dev)
            dha = torch.zeros(T + 1, B, S, device=dev)     # dha[t+1] accumulates dL/dh_t
        if ds is not None:
            ds_tb = ds.t().contiguous().view(T * B, 1)
            if wg:
                K.gemm(ds_tb, hs[1:].view(T * B, S), dws[6], ta=True)
                K.col_sum(ds_tb, dws[7], defer=False)
            if persist:
                dh_ext = torch.empty(T, B, S, device=dev)
            K.gemm(ds_tb, sw, dh_ext.view(T * B, S) if persist else dha[1:].view(T * B, S))
        if persist:
            # the whole loop in ONE launch (ag_grufront_bwd_persist); the external gradients dL/dh_t, dL/dx_t are read in place
            dx_ext = dx if (dx is None or dx.stride(1) == 1) else dx.contiguous()
            K.grufront_bwd_persist(gi, hs, gh, x, dh_ext, dx_ext, w_hh, wx, pw, dgi, dgh, dxt)
        dh_dir = None if persist else torch.empty(B, S, device=dev)
        for t in reversed(range(0 if persist else T)):
            gx = dxt[t]
            K.act_bwd2d(dxa[:, t * fs:(t + 1) * fs], x[:, t * fs:(t + 1) * fs], gx, ACT_TANH)
            _small_acc(gx, pw, dha[t + 1])
            K.gru_cell_bwd(gi[t], gh[t], hs[t], dha[t + 1], dgi[t], dgh[t], dh_dir)
            K.axpby(dh_dir, dha[t], 1.0, 1.0)                     # direct path  dh * z
            _small_acc(dgh[t], w_hh, dha[t])                      # through the hidden product
            if t > 0:
                _small_acc(dgi[t], wx, dxa[:, (t - 1) * fs:t * fs])
        dxt2, dgi2, dgh2 = dxt.view(T * B, fs), dgi.view(T * B, 3 * S), dgh.view(T * B, 3 * S)
        if wg:
            K.gemm(dxt2, hs[1:].view(T * B, S), dws[4], ta=True)
            K.col_sum(dxt2, dws[5])
            K.gemm(dgi2, zc.contiguous().view(T * B, Fz), dws[0][:, fs:], ta=True)
            if T > 1:
                xprev = xt[:T - 1].view((T - 1) * B, fs) if xt is not None else \
                    x.view(B, T, fs)[:, :T - 1].transpose(0, 1).contiguous().view((T - 1) * B, fs)
                K.gemm(dgi[1:].view((T - 1) * B, 3 * S), xprev, dws[0][:, :fs], ta=True)
            K.gemm(dgh2, hs[:T].view(T * B, S), dws[1], ta=True)
            K.col_sum(dgi2, dws[2])
            K.col_sum(dgh2, dws[3])
        dzc = None
        if ctx.needs_input_grad[0]:
            dzc = torch.empty(T * B, Fz, device=dev)
            K.gemm(dgi2, w_ih[:, fs:], dzc)
            dzc = dzc.view(T, B, Fz)
        return (dzc, None) + tuple(front.group.backward(dws) if wg else [None] * (2 * len(front.group.items)))
