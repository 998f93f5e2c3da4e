"""Autograd blocks of the hot path, built on the HIP kernels (audiogan_amd.kernels).

Each block is ONE ``torch.autograd.Function`` whose forward and backward enqueue the
hand-written kernels; torch is used for memory, views and the autograd graph only.

  GTrunkFn     dense_res_gen of Generator      audiogan.py:387-407, 462-468, 266-283
  DConvStackFn cnn of Discriminator            audiogan.py:483-494, 529-536
  LSTMSeqFn    NN.LSTM (bi)directional layer   audiogan.py:498-503, 543 (+ :214-229)
  DHeadFn      residual_net + classifier       audiogan.py:256-264, 504-512, 547-549
  GFrontFn     LSTMCell/proj/stopper loop      audiogan.py:377-386, 409-410, 428-460
  BCEFn        masked BCE / nframes, mean      audiogan.py:187-197, 204-211, 739-740
"""
import torch

from . import kernels as K
from .kernels import ACT_NONE, ACT_LEAKY, ACT_TANH, ACT_LEAKY_GATE

from .common import PARAM_EPOCH, Prepared, WNGroup, _zeros_like_list  # noqa: F401


class ConvSpec(object):
    """kind 'conv': weight [Cout,Cin,K] (NN.Conv1d); 'convT': weight [Cin,Cout,K] (NN.ConvTranspose1d)."""
    __slots__ = ('kind', 'cin', 'cout', 'K', 'stride', 'pad', 'out_pad')

    def __init__(self, kind, cin, cout, K_, stride, pad, out_pad=0):
        self.kind, self.cin, self.cout, self.K, self.stride, self.pad = kind, cin, cout, K_, stride, pad
        self.out_pad = out_pad      # ConvTranspose1d output_padding (extra positions on the right)

    def out_len(self, lin):
        if self.kind == 'conv':
            return (lin + 2 * self.pad - self.K) // self.stride + 1
        return (lin - 1) * self.stride - 2 * self.pad + self.K + self.out_pad


# --------------------------------------------------------------------------------------
# conv helpers (layout bookkeeping between NN.Conv1d / NN.ConvTranspose1d and the engine)
# --------------------------------------------------------------------------------------
def _o1(spec):
    return K.conv_o1_ok(spec.kind, spec.cout, spec.K, spec.stride, spec.pad)


def conv_fwd(spec, prep, x, y, bias=None, res=None, lens=None, act=ACT_NONE):
    if prep.w is not None and res is None and lens is None and _o1(spec):
        K.conv_o1_fwd(x, prep.w, bias, y, spec.K, spec.pad, act)
    elif spec.kind == 'conv':
        K.conv_engine(x, prep.wpa, y, spec.K, spec.stride, spec.pad, 0, bias, res, lens, act)
    else:
        K.conv_engine(x, prep.wpb, y, spec.K, spec.stride, spec.pad, 1, bias, res, lens, act, wp_pad=prep.pad)


def conv_bwd_data(spec, prep, dy, dx, accumulate=False, gate=None, lens=None):
    """``gate``: the saved output of the LeakyReLU that produced this layer's input - its derivative (and the length mask
    ``lens`` of that output) is applied in the epilogue, so ``dx`` comes out as the gradient of the PRE-activation below
    (one pass instead of this launch plus a separate activation-backward pass over the same tensor)"""
    act = ACT_LEAKY_GATE if gate is not None else ACT_NONE
    if prep.w is not None and _o1(spec):
        assert gate is None
        K.conv_o1_bwd_data(dy, prep.w, dx, spec.K, spec.pad, accumulate)
    elif spec.kind == 'conv':
        K.conv_engine(dy, prep.wpb, dx, spec.K, spec.stride, spec.pad, 1, res=gate, lens=lens, act=act,
                      accumulate=accumulate, wp_pad=prep.pad)
    else:
        K.conv_engine(dy, prep.wpa, dx, spec.K, spec.stride, spec.pad, 0, res=gate, lens=lens, act=act, accumulate=accumulate)


def conv_wgrad(spec, x, dy, dw, db):
    """dw, db must be zero-filled."""
    if _o1(spec):
        K.conv_o1_wgrad(dy, x, dw, spec.K, spec.pad)
    elif spec.kind == 'conv':
        K.conv_wgrad(dy, x, dw, spec.K, spec.stride, spec.pad)
    else:
        K.conv_wgrad(x, dy, dw, spec.K, spec.stride, spec.pad)
    if db is not None:
        K.channel_sum(dy, db)


# --------------------------------------------------------------------------------------
# Generator conv trunk
# --------------------------------------------------------------------------------------
class GTrunk(object):
    """Static description of dense_res_gen: per bottleneck (conv spec, deconv spec), then the
    final conv.  WN items are ordered [conv.w, conv.b, deconv.w, deconv.b] * n + [final.w, final.b]."""

    def __init__(self, bottlenecks, final):
        self.bottlenecks = bottlenecks   # list of (ConvSpec conv, ConvSpec deconv)
        self.final = final               # ConvSpec
        self.group = WNGroup()
        self.ctot = 1 + sum(d.cout for _, d in bottlenecks)


class GTrunkFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x0, trunk, *params):
        B, L = x0.shape
        prep = trunk.group.prepare()
        ctot = trunk.ctot
        if (x0.stride() == (ctot * L, 1) and x0.storage_offset() == 0 and x0.dtype == torch.float32
                and x0.untyped_storage().nbytes() == 4 * B * ctot * L):
            # the recurrent front wrote its frames straight into channel 0 of a fresh [B, ctot, L] slab
            # (recurrent._frames_buffer): the trunk's input is in place
            slab = x0.as_strided((B, ctot, L), (ctot * L, L, 1), 0)
        else:
            slab = torch.empty(B, ctot, L, device=x0.device)
            slab[:, 0, :].copy_(x0)
        hids = []
        cin = 1
        for i, (cs, ds) in enumerate(trunk.bottlenecks):
            pw, pb, qw, qb = prep[4 * i:4 * i + 4]
            lh = cs.out_len(L)
            hid = torch.empty(B, cs.cout, lh, device=x0.device)
            conv_fwd(cs, pw, slab[:, :cin], hid, bias=pb.w, act=ACT_LEAKY)
            assert ds.out_len(lh) == L, 'bottleneck must preserve the clip length'
            res = slab[:, cin - ds.cout:cin] if cin >= ds.cout else None
            conv_fwd(ds, qw, hid, slab[:, cin:cin + ds.cout], bias=qb.w, res=res, act=ACT_LEAKY)
            hids.append(hid)
            cin += ds.cout
        fw, fb = prep[-2:]
        y = torch.empty(B, 1, L, device=x0.device)
        conv_fwd(trunk.final, fw, slab, y, bias=fb.w, act=ACT_NONE)
        ctx.trunk = trunk
        ctx.key = trunk.group._key[1:]
        ctx.save_for_backward(slab, *hids)
        return y.view(B, L)

    @staticmethod
    def backward(ctx, dy):
        trunk = ctx.trunk
        slab, hids = ctx.saved_tensors[0], ctx.saved_tensors[1:]
        prep = trunk.group.prepare()
        assert trunk.group._key[1:] == ctx.key, 'parameters changed between forward and backward'
        B, ctot, L = slab.shape
        dy = dy.contiguous().view(B, 1, L)
        wg = any(ctx.needs_input_grad[2:])
        dws = trunk.group.zero_dws() if wg else None
        dslab = torch.empty_like(slab)
        with K.deferred_reduces():      # every weight-gradient / bias-sum second stage of this backward in ONE launch
            GTrunkFn._backward_layers(trunk, prep, slab, hids, dy, dslab, dws, wg, ctot)
        grads = trunk.group.backward(dws) if wg else [None] * (2 * len(trunk.group.items))
        dx0 = dslab[:, 0, :] if ctx.needs_input_grad[0] else None      # (a view: the front's backward reads it in place)
        return (dx0, None) + tuple(grads)

    @staticmethod
    def _backward_layers(trunk, prep, slab, hids, dy, dslab, dws, wg, ctot):
        # final conv (no activation)
        if wg:
            conv_wgrad(trunk.final, slab, dy, dws[-2], dws[-1])
        conv_bwd_data(trunk.final, prep[-2], dy, dslab, accumulate=False)
        cin = ctot
        for i in reversed(range(len(trunk.bottlenecks))):
            cs, ds = trunk.bottlenecks[i]
            pw, qw = prep[4 * i], prep[4 * i + 2]
            cin -= ds.cout
            out_v = slab[:, cin:cin + ds.cout]
            dout = dslab[:, cin:cin + ds.cout]
            add = dslab[:, cin - ds.cout:cin] if cin >= ds.cout else None
            # dout now holds d(pre-activation); the bias gradient is summed in the same pass
            K.leaky_bwd(dout, out_v, dout, add_into=add, bias_grad=dws[4 * i + 3] if wg else None)
            hid = hids[i]
            if wg:
                conv_wgrad(ds, hid, dout, dws[4 * i + 2], None)
            # the hidden LeakyReLU's backward rides in the epilogue of this backward-data pass (dhid = d(pre-activation))
            dhid = torch.empty_like(hid)
            conv_bwd_data(ds, qw, dout, dhid, gate=hid)
            if wg:
                K.channel_sum(dhid, dws[4 * i + 1])
            if wg:
                conv_wgrad(cs, slab[:, :cin], dhid, dws[4 * i], None)
            conv_bwd_data(cs, pw, dhid, dslab[:, :cin], accumulate=True)


# --------------------------------------------------------------------------------------
# Discriminator conv stack: 6 x [conv -> leaky -> * length mask]
# --------------------------------------------------------------------------------------
class DConvStack(object):
    def __init__(self, specs):
        self.specs = specs           # list of ConvSpec('conv', ...)
        self.group = WNGroup()       # items [w0, b0, w1, b1, ...]


class DConvStackFn(torch.autograd.Function):
    """returns (act_1, ..., act_n); lens_list[i] (int64, device) masks layer i's output."""

    @staticmethod
    def forward(ctx, x, stack, lens_list, *params):
        B, L = x.shape
        ctx.set_materialize_grads(False)
        prep = stack.group.prepare()
        a = x.contiguous().view(B, 1, L)
        acts = []
        for i, sp in enumerate(stack.specs):
            lo = sp.out_len(a.size(2))
            y = torch.empty(B, sp.cout, lo, device=x.device)
            conv_fwd(sp, prep[2 * i], a, y, bias=prep[2 * i + 1].w, lens=lens_list[i], act=ACT_LEAKY)
            acts.append(y)
            a = y
        ctx.stack, ctx.lens_list, ctx.key = stack, lens_list, stack.group._key[1:]
        ctx.save_for_backward(x, *acts)
        return tuple(acts)

    @staticmethod
    def backward(ctx, *dacts):
        stack = ctx.stack
        x, acts = ctx.saved_tensors[0], ctx.saved_tensors[1:]
        prep = stack.group.prepare()
        assert stack.group._key[1:] == ctx.key, 'parameters changed between forward and backward'
        B, L = x.shape
        wg = any(ctx.needs_input_grad[3:])
        dws = stack.group.zero_dws() if wg else None
        n = len(stack.specs)
        d, gated = None, False
        with K.deferred_reduces():      # every weight-gradient / bias-sum second stage of this backward in ONE launch
            for i in reversed(range(n)):
                sp = stack.specs[i]
                g = dacts[i]
                if d is None:
                    if g is None:
                        continue
                    d = torch.empty_like(acts[i])
                    K.leaky_bwd(g.contiguous(), acts[i], d, lens=ctx.lens_list[i],     # out of place: no clone
                                bias_grad=dws[2 * i + 1] if wg else None)
                elif not gated:
                    if g is not None:
                        K.axpby(g.contiguous(), d, 1.0, 1.0)
                    K.leaky_bwd(d, acts[i], d, lens=ctx.lens_list[i], bias_grad=dws[2 * i + 1] if wg else None)
                elif wg:
                    K.channel_sum(d, dws[2 * i + 1])          # (d already is d(pre-activation): see below)
                xin = acts[i - 1] if i > 0 else x.contiguous().view(B, 1, L)
                if wg:
                    conv_wgrad(sp, xin, d, dws[2 * i], None)
                if i > 0 or ctx.needs_input_grad[0]:
                    dx = torch.empty_like(xin)
                    # when the layer below receives no other gradient its LeakyReLU + length-mask backward rides in this
                    # launch's epilogue (dx = d(pre-activation) of layer i - 1) instead of a pass of its own
                    gated = i > 0 and dacts[i - 1] is None
                    conv_bwd_data(sp, prep[2 * i], d, dx, gate=acts[i - 1] if gated else None,
                                  lens=ctx.lens_list[i - 1] if gated else None)
                    d = dx
                else:
                    d = None
        grads = stack.group.backward(dws) if wg else [None] * (2 * len(stack.group.items))
        dx0 = d.view(B, L) if (ctx.needs_input_grad[0] and d is not None) else None
        return (dx0, None, None) + tuple(grads)


# --------------------------------------------------------------------------------------
# Discriminator heads: Residual x n  ->  Linear -> LeakyReLU -> Linear
# --------------------------------------------------------------------------------------
class DHead(object):
    """WN items: [res0.w, res0.b, res1.w, res1.b, ..., cls0.w, cls0.b, cls1.w, cls1.b]"""

    def __init__(self, n_res):
        self.n_res = n_res
        self.group = WNGroup()


class TimeMajorFn(torch.autograd.Function):
    """[B,C,T] -> contiguous [T,B,C] (what the biLSTM reads, audiogan.py:542) and the gradient's way back, each ONE tiled
    transpose launch instead of a strided elementwise copy"""

    @staticmethod
    def forward(ctx, a, store16=False):
        """``store16`` (bf16 storage, kernels.bf16_storage): the time-major features are WRITTEN as bfloat16 - the operand
        type of the products that read them - and their gradient comes back as bfloat16"""
        if a.stride(2) != 1:
            a = a.contiguous()
        ctx.in_dtype = a.dtype
        return K.bct_to_tbc(a, out_dtype=torch.bfloat16 if store16 else None)

    @staticmethod
    def backward(ctx, g):
        return K.tbc_to_bct(g.contiguous(), out_dtype=ctx.in_dtype), None


_EYE = {}


def _eye(w):
    """cached identity of w's (square) shape on w's device"""
    key = (w.device, w.size(0))
    if key not in _EYE:
        _EYE[key] = torch.eye(w.size(0), device=w.device, dtype=w.dtype)
    return _EYE[key]


def _gh(A, B, C=None, C16=None, ta=False, tb=False, bias=None, res=None, gate=None, act=ACT_NONE, defer=False, beta=0.0):
    """a product on bf16-stored operands: ag_gemm_h when the shape fits it, else the fp32-operand kernel on widened copies
    (odd test shapes only: every BASELINE shape takes the first branch)"""
    if K.gemm_h_ok(A, B, ta, tb):
        K.gemm_h(A, B, C=C, C16=C16, ta=ta, tb=tb, bias=bias, res=res, gate=gate, act=act, defer=defer, beta=beta)
        return
    out = C if C is not None else torch.empty(C16.shape, device=C16.device)
    r = res.float() if (res is not None and res.dtype != torch.float32) else res
    K.gemm(A.float(), B.float(), out, ta=ta, tb=tb, bias=bias, res=r, act=act, defer=defer, beta=beta)
    if gate is not None:
        K.act_bwd(out.view(-1), gate.float().view(-1), out.view(-1), ACT_LEAKY)
    if C16 is not None:
        C16.copy_(out)


class DHeadFn(torch.autograd.Function):
    """``rows`` fp32: the round-3 form.  ``rows`` bfloat16 (bf16 storage): every activation of the heads, its gradient and
    both operands of every product are bfloat16 in HBM (ag_gemm_h); logits, parameter gradients and biases stay fp32."""

    @staticmethod
    def _forward16(ctx, rows, head, prep):
        g = head.group
        M, nr = rows.size(0), head.n_res
        a = rows.contiguous()
        saved = [a]
        for i in range(nr):
            y = torch.empty(M, prep[2 * i].w.size(0), device=a.device, dtype=torch.bfloat16)
            _gh(a, g.w16(2 * i), C16=y, tb=True, bias=prep[2 * i + 1].w, res=a, act=ACT_LEAKY)
            saved.append(y)
            a = y
        w1, b0, b1 = prep[2 * nr + 2].w, prep[2 * nr + 1].w, prep[2 * nr + 3].w
        hmid = torch.empty(M, prep[2 * nr].w.size(0), device=a.device, dtype=torch.bfloat16)
        _gh(a, g.w16(2 * nr), C16=hmid, tb=True, bias=b0, act=ACT_LEAKY)
        out = torch.empty(M, w1.size(0), device=a.device)
        if w1.size(0) == 1 and K.rowdot_ok(hmid, w1):
            K.rowdot_fwd(hmid, w1, b1, out)
        else:
            _gh(hmid, g.w16(2 * nr + 2), C=out, tb=True, bias=b1)
        saved.append(hmid)
        ctx.head, ctx.key, ctx.store16 = head, g._key[1:], True
        ctx.save_for_backward(*saved)
        return out

    @staticmethod
    def _backward16(ctx, dout, head, prep):
        g = head.group
        saved = ctx.saved_tensors
        hmid, acts, nr = saved[-1], saved[:-1], head.n_res
        wg = any(ctx.needs_input_grad[2:])
        dws = g.zero_dws() if wg else None
        dout = dout.contiguous()
        with K.deferred_reduces():
            w1 = prep[2 * nr + 2].w
            dh = torch.empty_like(hmid)
            dw1, db1 = (dws[2 * nr + 2], dws[2 * nr + 3]) if wg else (None, None)
            if w1.size(0) == 1 and K.rowdot_ok(hmid, w1) and (not wg or db1.data_ptr() == dw1.data_ptr() + 4 * dw1.numel()):
                K.rowdot_bwd(dout, hmid, w1, dx=dh, dw=dw1, db=db1, gate=True)
            else:
                d16 = dout.to(torch.bfloat16)
                if wg:
                    _gh(d16, hmid, C=dws[2 * nr + 2], ta=True, defer=True)
                    K.col_sum(dout, dws[2 * nr + 3])
                _gh(d16, g.w16(2 * nr + 2), C16=dh, res=hmid, act=ACT_LEAKY_GATE)
            if wg:
                _gh(dh, acts[nr], C=dws[2 * nr], ta=True, defer=True)
                K.col_sum(dh, dws[2 * nr + 1])
            da = torch.empty_like(acts[nr])
            if nr > 0:
                _gh(dh, g.w16(2 * nr), C16=da, res=acts[nr], act=ACT_LEAKY_GATE)
            else:
                _gh(dh, g.w16(2 * nr), C16=da)
            for i in reversed(range(nr)):
                if wg:
                    _gh(da, acts[i], C=dws[2 * i], ta=True, defer=True)
                    K.col_sum(da, dws[2 * i + 1])
                dprev = torch.empty_like(acts[i])
                # d(input of residual i) = W^T da + da, times the LeakyReLU derivative of the residual below (gate)
                _gh(da, g.w16(2 * i), C16=dprev, res=da, gate=acts[i] if i > 0 else None)
                da = dprev
        grads = g.backward(dws) if wg else [None] * (2 * len(g.items))
        return (da if ctx.needs_input_grad[0] else None, None) + tuple(grads)

    @staticmethod
    def forward(ctx, rows, head, *params):
        prep = head.group.prepare()
        if rows.dtype == torch.bfloat16:
            return DHeadFn._forward16(ctx, rows, head, prep)
        ctx.store16 = False
        M = rows.size(0)
        a = rows.contiguous()
        saved = [a]
        for i in range(head.n_res):
            w, b = prep[2 * i].w, prep[2 * i + 1].w
            y = torch.empty(M, w.size(0), device=a.device)
            K.gemm(a, w, y, tb=True, bias=b, res=a, act=ACT_LEAKY)
            saved.append(y)
            a = y
        w0, b0, w1, b1 = [p.w for p in prep[2 * head.n_res:2 * head.n_res + 4]]
        hmid = torch.empty(M, w0.size(0), device=a.device)
        K.gemm(a, w0, hmid, tb=True, bias=b0, act=ACT_LEAKY)
        out = torch.empty(M, w1.size(0), device=a.device)
        if w1.size(0) == 1 and K.rowdot_ok(hmid, w1):
            K.rowdot_fwd(hmid, w1, b1, out)          # a Linear with one output: one wave per row, no padded MFMA tile
        else:
            K.gemm(hmid, w1, out, tb=True, bias=b1)
        saved.append(hmid)
        ctx.head, ctx.key = head, head.group._key[1:]
        ctx.save_for_backward(*saved)
        return out

    @staticmethod
    def backward(ctx, dout):
        head = ctx.head
        prep = head.group.prepare()
        assert head.group._key[1:] == ctx.key, 'parameters changed between forward and backward'
        if ctx.store16:
            return DHeadFn._backward16(ctx, dout, head, prep)
        saved = ctx.saved_tensors
        hmid = saved[-1]
        acts = saved[:-1]          # acts[0] = input rows, acts[i] = output of residual i
        nr = head.n_res
        wg = any(ctx.needs_input_grad[2:])
        dws = head.group.zero_dws() if wg else None
        dout = dout.contiguous()
        with K.deferred_reduces():      # the bias sums' second stages in ONE launch (nothing below reads them)
            w0, w1 = prep[2 * nr].w, prep[2 * nr + 2].w
            dh = torch.empty_like(hmid)
            dw1, db1 = (dws[2 * nr + 2], dws[2 * nr + 3]) if wg else (None, None)
            if w1.size(0) == 1 and K.rowdot_ok(hmid, w1) and (not wg or db1.data_ptr() == dw1.data_ptr() + 4 * dw1.numel()):
                # classifier[2] has ONE output: its whole backward - the input gradient dout (x) w1 gated by hmid's
                # LeakyReLU, dW1 = dout^T hmid and db1 = sum dout - is one pass over hmid (ag_rowdot_bwd) instead of an outer
                # product and a one-row "matrix" on 64-wide MFMA tiles plus a column sum
                K.rowdot_bwd(dout, hmid, w1, dx=dh, dw=dw1, db=db1, gate=True)
            else:
                # classifier[2]: out = hmid @ w1^T + b1
                if wg:
                    K.gemm(dout, hmid, dws[2 * nr + 2], ta=True, defer=True)
                    K.col_sum(dout, dws[2 * nr + 3])
                # every LeakyReLU backward rides in the epilogue of the product that feeds it (ACT_LEAKY_GATE: res = the
                # saved activation), so dh / da below are d(pre-activation) as they leave the GEMM
                K.gemm(dout, w1, dh, res=hmid, act=ACT_LEAKY_GATE)
            # classifier[0]
            if wg:
                K.gemm(dh, acts[nr], dws[2 * nr], ta=True, defer=True)
                K.col_sum(dh, dws[2 * nr + 1])
            da = torch.empty_like(acts[nr])
            if nr > 0:
                K.gemm(dh, w0, da, res=acts[nr], act=ACT_LEAKY_GATE)
            else:
                K.gemm(dh, w0, da)
            fold = K.get_precision() != 'bf16'
            for i in reversed(range(nr)):
                if wg:
                    K.gemm(da, acts[i], dws[2 * i], ta=True, defer=True)
                    K.col_sum(da, dws[2 * i + 1])
                dprev = torch.empty_like(acts[i])
                # d(input of residual i) = W^T da + da (skip connection) = (W + I)^T da: with the identity folded into the
                # weight the skip needs no second epilogue tensor and `res` is free for the gate of the residual below
                # (not in bf16 mode: rounding 1 + w_ii to bfloat16 would lose the diagonal weights; there the skip stays an
                # exact fp32 add in the epilogue and the gate a pass of its own)
                wi = prep[2 * i].w
                if i > 0 and fold:
                    K.gemm(da, wi + _eye(wi), dprev, res=acts[i], act=ACT_LEAKY_GATE)
                else:
                    K.gemm(da, wi, dprev, res=da)
                    if i > 0:
                        K.act_bwd(dprev, acts[i], dprev, ACT_LEAKY)
                da = dprev
        grads = head.group.backward(dws) if wg else [None] * (2 * len(head.group.items))
        return (da if ctx.needs_input_grad[0] else None, None) + tuple(grads)


from .recurrent import LSTMSeqFn, GFront, GFrontFn, GRUFront, GRUFrontFn  # noqa: E402,F401


# --------------------------------------------------------------------------------------
# input assembly of the two networks (one launch each; audiogan.py:433-439, :724-728, :749-751, :844)
# --------------------------------------------------------------------------------------
class BuildZCFn(torch.autograd.Function):
    """z [B,T,ns], c [B,es] -> zc [T,B,ns+es] = the non-recurrent part of every frame's LSTM input (the reference's
    T.cat([z_t, c]) per frame, :433-439) in ONE launch instead of expand + cat + transpose + contiguous"""

    @staticmethod
    def forward(ctx, z, c):
        ctx.ns = z.size(2)
        return K.build_zc(z.contiguous(), c.contiguous())

    @staticmethod
    def backward(ctx, dzc):
        ns = ctx.ns
        dz = dzc[:, :, :ns].transpose(0, 1) if ctx.needs_input_grad[0] else None
        dc = dzc[:, :, ns:].sum(0) if ctx.needs_input_grad[1] else None
        return dz, dc


class CriticInputFn(torch.autograd.Function):
    """x + noise -> the critic's input rows and the rows' lengths after every conv layer ([n_layers, B] int64), ONE launch
    (audiogan.py:844 ``fake + noise``, :533 the per-layer ``div_roundup``); the gradient passes through to x"""

    @staticmethod
    def forward(ctx, x, noise, length, prods):
        xo, lens, _ = K.critic_batch(x, noise, None, None, length, None, prods)
        ctx.mark_non_differentiable(lens)
        return xo, lens

    @staticmethod
    def backward(ctx, dx, _dlens):
        return dx, None, None, None


# --------------------------------------------------------------------------------------
# masked BCE-with-logits, / nframes, mean over batch
# --------------------------------------------------------------------------------------
class BCEFn(torch.autograd.Function):
    """loss = scale * sum_b( sum_{t<n_b} bce(x[b,t], target_b) / n_b ), scale = 1 / B unless given; also returns the
    per-sample sums.  x: [B,T] of ANY strides (the heads hand over a transposed view: no copy); target: a float or a [B]
    tensor of per-row targets.  One launch forward, one backward (ag_bce_logits_*_strided)."""

    @staticmethod
    def forward(ctx, x, target, nframes, scale=None):
        B, T = x.shape
        ctx.set_materialize_grads(False)
        rows = target if torch.is_tensor(target) else None
        tgt = 0.0 if rows is not None else float(target)
        if rows is not None:
            rows = rows.to(device=x.device, dtype=torch.float32).contiguous()
        if nframes is not None:
            nframes = nframes.contiguous()
        sc = 1.0 / B if scale is None else float(scale)
        per = torch.empty(B, device=x.device)
        loss = torch.empty(1, device=x.device)
        K.bce_logits_fwd_strided(x, tgt, nframes, per, loss, sc, target_rows=rows)
        ctx.target, ctx.scale = tgt, sc
        ctx.save_for_backward(x, nframes if nframes is not None else x.new_empty(0),
                              rows if rows is not None else x.new_empty(0))
        ctx.has_n, ctx.has_rows = nframes is not None, rows is not None
        ctx.mark_non_differentiable(per)
        return loss.view(()), per

    @staticmethod
    def backward(ctx, dloss, _dper):
        x, nfr, rows = ctx.saved_tensors
        if dloss is None:
            return None, None, None, None
        dx = torch.empty_like(x)          # (keeps x's memory layout: the gradient flows back through the same view)
        g = dloss.contiguous().view(1).float()
        K.bce_logits_bwd_strided(x, ctx.target, nfr if ctx.has_n else None, g, ctx.scale, dx,
                                 target_rows=rows if ctx.has_rows else None)
        return dx, None, None, None
