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
from .kernels import ACT_NONE, ACT_LEAKY, ACT_TANH

# bumped whenever parameters are rewritten through raw pointers (fused optimiser)
PARAM_EPOCH = [0]


class ConvSpec(object):
    """kind 'conv': weight [Cout,Cin,K] (NN.Conv1d); 'convT': weight [Cin,Cout,K] (NN.ConvTranspose1d)."""
    __slots__ = ('kind', 'cin', 'cout', 'K', 'stride', 'pad')

    def __init__(self, kind, cin, cout, K_, stride, pad):
        self.kind, self.cin, self.cout, self.K, self.stride, self.pad = kind, cin, cout, K_, stride, pad

    def out_len(self, lin):
        if self.kind == 'conv':
            return (lin + 2 * self.pad - self.K) // self.stride + 1
        return (lin - 1) * self.stride - 2 * self.pad + self.K


class Prepared(object):
    __slots__ = ('w', 'wpa', 'wpb')

    def __init__(self, w=None, wpa=None, wpb=None):
        self.w, self.wpa, self.wpb = w, wpa, wpb


class WNGroup(object):
    """All weight-normed tensors of one block.  ``prepare`` materialises w = g*v/||v|| for
    every tensor (and the conv-engine layouts of 3-D weights) with ONE launch into
    persistent buffers; ``backward`` turns dW into (dv, dg) with one launch."""

    def __init__(self):
        self.items = []   # dict(v=Parameter, g=Parameter, stride=int, engine=bool)
        self._bufs = None
        self._key = None

    def add(self, v, g, stride=1, engine=False):
        self.items.append(dict(v=v, g=g, stride=stride, engine=engine))
        return len(self.items) - 1

    def params(self):
        out = []
        for it in self.items:
            out += [it['v'], it['g']]
        return out

    def _alloc(self, dev):
        bufs = []
        for it in self.items:
            v = it['v']
            p = Prepared(w=torch.empty_like(v.data))
            if it['engine']:
                d0, d1, kk = v.shape
                p.wpa = torch.zeros(K.wpa_numel(d0, d1, kk), device=dev)
                p.wpb = torch.zeros(K.wpb_numel(d0, d1, kk, it['stride']), device=dev)
            bufs.append(p)
        self._bufs = bufs
        self._key = None

    def prepare(self):
        dev = self.items[0]['v'].device
        if self._bufs is None or self._bufs[0].w.device != dev:
            self._alloc(dev)
        key = (PARAM_EPOCH[0],) + tuple((it['v'].data_ptr(), it['v']._version, it['g'].data_ptr(),
                                         it['g']._version) for it in self.items)
        if key != self._key:
            ents = []
            for it, p in zip(self.items, self._bufs):
                ents.append(dict(v=it['v'].data, g=it['g'].data.view(-1), w=p.w, wpa=p.wpa, wpb=p.wpb,
                                 stride=it['stride']))
            K.weight_norm_fwd(ents)
            self._key = key
        return self._bufs

    def backward(self, dws):
        """dws[i]: gradient wrt the materialised w of item i (same shape as v), or None.
        Returns the flat list [dv0, dg0, dv1, dg1, ...]."""
        ents, outs = [], []
        for it, dw in zip(self.items, dws):
            if dw is None:
                outs += [None, None]
                continue
            dv = torch.empty_like(it['v'].data)
            dg = torch.empty_like(it['g'].data)
            ents.append(dict(v=it['v'].data, g=it['g'].data.view(-1), dw=dw, dv=dv, dg=dg.view(-1)))
            outs += [dv, dg]
        if ents:
            K.weight_norm_bwd(ents)
        return outs


# --------------------------------------------------------------------------------------
# conv helpers (layout bookkeeping between NN.Conv1d / NN.ConvTranspose1d and the engine)
# --------------------------------------------------------------------------------------
def conv_fwd(spec, prep, x, y, bias=None, res=None, lens=None, act=ACT_NONE):
    if spec.kind == 'conv':
        K.conv_engine(x, prep.wpa, y, spec.K, spec.stride, spec.pad, 0, bias, res, lens, act)
    else:
        K.conv_engine(x, prep.wpb, y, spec.K, spec.stride, spec.pad, 1, bias, res, lens, act)


def conv_bwd_data(spec, prep, dy, dx, accumulate=False):
    if spec.kind == 'conv':
        K.conv_engine(dy, prep.wpb, dx, spec.K, spec.stride, spec.pad, 1, accumulate=accumulate)
    else:
        K.conv_engine(dy, prep.wpa, dx, spec.K, spec.stride, spec.pad, 0, accumulate=accumulate)


def conv_wgrad(spec, x, dy, dw, db):
    """dw, db must be zero-filled."""
    if spec.kind == 'conv':
        K.conv_wgrad(dy, x, dw, spec.K, spec.stride, spec.pad)
    else:
        K.conv_wgrad(x, dy, dw, spec.K, spec.stride, spec.pad)
    if db is not None:
        K.channel_sum(dy, db)


def _zeros_like_list(tensors):
    """one flat zero buffer, viewed as the given shapes (one memset instead of many)"""
    n = sum(t.numel() for t in tensors)
    flat = torch.zeros(n, device=tensors[0].device, dtype=torch.float32)
    out, o = [], 0
    for t in tensors:
        out.append(flat[o:o + t.numel()].view(t.shape))
        o += t.numel()
    return out


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
        slab = torch.empty(B, trunk.ctot, L, device=x0.device)
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
        ctx.key = trunk.group._key
        ctx.save_for_backward(slab, *hids)
        return y.view(B, L)

    @staticmethod
    def backward(ctx, dy):
        trunk = ctx.trunk
        slab, hids = ctx.saved_tensors[0], ctx.saved_tensors[1:]
        prep = trunk.group.prepare()
        assert trunk.group._key == ctx.key, 'parameters changed between forward and backward'
        B, ctot, L = slab.shape
        dy = dy.contiguous().view(B, 1, L)
        dws = _zeros_like_list([it['v'] for it in trunk.group.items])
        dslab = torch.empty_like(slab)
        # final conv (no activation)
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
            K.leaky_bwd(dout, out_v, dout, add_into=add)          # dout now holds d(pre-activation)
            hid = hids[i]
            conv_wgrad(ds, hid, dout, dws[4 * i + 2], dws[4 * i + 3])
            dhid = torch.empty_like(hid)
            conv_bwd_data(ds, qw, dout, dhid)
            K.leaky_bwd(dhid, hid, dhid)
            conv_wgrad(cs, slab[:, :cin], dhid, dws[4 * i], dws[4 * i + 1])
            conv_bwd_data(cs, pw, dhid, dslab[:, :cin], accumulate=True)
        grads = trunk.group.backward(dws)
        dx0 = dslab[:, 0, :].contiguous() if ctx.needs_input_grad[0] else None
        return (dx0, None) + tuple(grads)


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
        ctx.stack, ctx.lens_list, ctx.key = stack, lens_list, stack.group._key
        ctx.save_for_backward(x, *acts)
        return tuple(acts)

    @staticmethod
    def backward(ctx, *dacts):
        stack = ctx.stack
        x, acts = ctx.saved_tensors[0], ctx.saved_tensors[1:]
        prep = stack.group.prepare()
        assert stack.group._key == ctx.key, 'parameters changed between forward and backward'
        B, L = x.shape
        dws = _zeros_like_list([it['v'] for it in stack.group.items])
        n = len(stack.specs)
        d = None
        for i in reversed(range(n)):
            sp = stack.specs[i]
            g = dacts[i]
            if d is None:
                if g is None:
                    continue
                d = g.contiguous().clone()
            elif g is not None:
                K.axpby(g.contiguous(), d, 1.0, 1.0)
            K.leaky_bwd(d, acts[i], d, lens=ctx.lens_list[i])
            xin = acts[i - 1] if i > 0 else x.contiguous().view(B, 1, L)
            conv_wgrad(sp, xin, d, dws[2 * i], dws[2 * i + 1])
            if i > 0 or ctx.needs_input_grad[0]:
                dx = torch.empty_like(xin)
                conv_bwd_data(sp, prep[2 * i], d, dx)
                d = dx
            else:
                d = None
        grads = stack.group.backward(dws)
        dx0 = d.view(B, L) if (ctx.needs_input_grad[0] and d is not None) else None
        return (dx0, None, None) + tuple(grads)


# --------------------------------------------------------------------------------------
# LSTM layer over a padded sequence (uni- or bidirectional), NN.LSTM parameter layout
# --------------------------------------------------------------------------------------
class LSTMSeqFn(torch.autograd.Function):
    """x: [T,B,F] contiguous; lengths: int64 [B] on device or None; weights per direction:
    (w_ih [4H,F], w_hh [4H,H], b_ih [4H], b_hh [4H]).  Returns y [T,B,D*H] with zeros at
    padded steps (pad_packed_sequence semantics, audiogan.py:214-229)."""

    @staticmethod
    def forward(ctx, x, lengths, ndir, *w):
        T, B, F = x.shape
        H = w[1].size(1)
        dev = x.device
        x2 = x.contiguous().view(T * B, F)
        y = torch.empty(T, B, ndir * H, device=dev)
        gates_all, c_all = [], []
        for d in range(ndir):
            w_ih, w_hh, b_ih, b_hh = w[4 * d:4 * d + 4]
            bsum = b_ih.data + b_hh.data
            g = torch.empty(T, B, 4 * H, device=dev)
            K.gemm(x2, w_ih.data, g.view(T * B, 4 * H), tb=True, bias=bsum)
            c = torch.empty(T + 1, B, H, device=dev)   # c[k+1] = cell after the k-th processed step
            c[0].zero_()
            h0 = torch.zeros(B, H, device=dev)
            hstate = [h0, torch.empty(B, H, device=dev)]
            order = range(T) if d == 0 else range(T - 1, -1, -1)
            for k, t in enumerate(order):
                hp, hn = hstate[k & 1], hstate[(k + 1) & 1]
                if k > 0:
                    K.gemm(hp, w_hh.data, g[t], tb=True, beta=1.0)
                K.lstm_cell_fwd(g[t], c[k], c[k + 1], h_out=hn, y_out=y[t, :, d * H:(d + 1) * H],
                                h_prev=hp, valid=lengths, t=t)
            gates_all.append(g)
            c_all.append(c)
        ctx.ndir, ctx.has_len = ndir, lengths is not None
        ctx.save_for_backward(x2, y, lengths if lengths is not None else x2.new_empty(0),
                              *(gates_all + c_all + [t_.data for t_ in w]))
        ctx.shape = (T, B, F, H)
        return y

    @staticmethod
    def backward(ctx, dy):
        T, B, F, H = ctx.shape
        ndir = ctx.ndir
        sv = ctx.saved_tensors
        x2, y = sv[0], sv[1]
        lengths = sv[2] if ctx.has_len else None
        gates_all, c_all, w = sv[3:3 + ndir], sv[3 + ndir:3 + 2 * ndir], sv[3 + 2 * ndir:]
        dev = x2.device
        dy = dy.contiguous()
        dx2 = torch.empty(T * B, F, device=dev)
        outs = []
        for d in range(ndir):
            w_ih, w_hh = w[4 * d], w[4 * d + 1]
            g, c = gates_all[d], c_all[d]
            dg = torch.empty(T, B, 4 * H, device=dev)
            dh = [torch.empty(B, H, device=dev), torch.empty(B, H, device=dev)]
            dc = [torch.zeros(B, H, device=dev), torch.empty(B, H, device=dev)]
            order = list(range(T)) if d == 0 else list(range(T - 1, -1, -1))
            for k in reversed(range(T)):
                t = order[k]
                dyt = dy[t, :, d * H:(d + 1) * H]
                cur, nxt = dh[k & 1], dh[(k + 1) & 1]
                if k == T - 1:
                    dh_in = None
                else:
                    # cur holds dh_pass written by step k+1; add that step's recurrent term
                    K.gemm(dg[order[k + 1]], w_hh, cur, beta=1.0)
                    dh_in = cur
                K.lstm_cell_bwd(g[t], c[k], c[k + 1], dh_in, dyt,
                                dc[(k + 1) & 1] if k < T - 1 else None, dg[t], dc[k & 1],
                                dh_pass=nxt, valid=lengths, t=t)
                # step k-1 reads its future term from dh[(k-1)&1] == nxt
            dg2 = dg.view(T * B, 4 * H)
            dw_ih = torch.empty_like(w_ih)
            K.gemm(dg2, x2, dw_ih, ta=True)
            # h_{prev}(step k) is the output of step k-1 in processing order (zero at padded steps)
            dw_hh = torch.zeros_like(w_hh)
            if T > 1:
                if d == 0:
                    K.gemm(dg[1:].view((T - 1) * B, 4 * H), y[:-1].view((T - 1) * B, ndir * H)[:, :H],
                           dw_hh, ta=True)
                else:
                    K.gemm(dg[:-1].view((T - 1) * B, 4 * H),
                           y[1:].view((T - 1) * B, ndir * H)[:, H:2 * H], dw_hh, ta=True)
            db = torch.zeros(4 * H, device=dev)
            K.col_sum(dg2, db)
            K.gemm(dg2, w_ih, dx2, beta=0.0 if d == 0 else 1.0)
            outs += [dw_ih, dw_hh, db, db.clone()]
        dx = dx2.view(T, B, F) if ctx.needs_input_grad[0] else None
        return (dx, None, None) + tuple(outs)


# --------------------------------------------------------------------------------------
# Discriminator heads: Residual x n  ->  Linear -> LeakyReLU -> Linear
# --------------------------------------------------------------------------------------
class DHead(object):
    """WN items: [res0.w, res0.b, res1.w, res1.b, ..., cls0.w, cls0.b, cls1.w, cls1.b]"""

    def __init__(self, n_res):
        self.n_res = n_res
        self.group = WNGroup()


class DHeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rows, head, *params):
        prep = head.group.prepare()
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
        K.gemm(hmid, w1, out, tb=True, bias=b1)
        saved.append(hmid)
        ctx.head, ctx.key = head, head.group._key
        ctx.save_for_backward(*saved)
        return out

    @staticmethod
    def backward(ctx, dout):
        head = ctx.head
        prep = head.group.prepare()
        assert head.group._key == ctx.key, 'parameters changed between forward and backward'
        saved = ctx.saved_tensors
        hmid = saved[-1]
        acts = saved[:-1]          # acts[0] = input rows, acts[i] = output of residual i
        nr = head.n_res
        dws = _zeros_like_list([it['v'] for it in head.group.items])
        dout = dout.contiguous()
        w0, w1 = prep[2 * nr].w, prep[2 * nr + 2].w
        # classifier[2]: out = hmid @ w1^T + b1
        K.gemm(dout, hmid, dws[2 * nr + 2], ta=True)
        K.col_sum(dout, dws[2 * nr + 3])
        dh = torch.empty_like(hmid)
        K.gemm(dout, w1, dh)
        K.act_bwd(dh, hmid, dh, ACT_LEAKY)
        # classifier[0]
        K.gemm(dh, acts[nr], dws[2 * nr], ta=True)
        K.col_sum(dh, dws[2 * nr + 1])
        da = torch.empty_like(acts[nr])
        K.gemm(dh, w0, da)
        for i in reversed(range(nr)):
            K.act_bwd(da, acts[i + 1], da, ACT_LEAKY)      # da = d(pre-activation)
            K.gemm(da, acts[i], dws[2 * i], ta=True)
            K.col_sum(da, dws[2 * i + 1])
            dprev = torch.empty_like(acts[i])
            K.gemm(da, prep[2 * i].w, dprev, res=da)       # W^T da + da (skip connection)
            da = dprev
        grads = head.group.backward(dws)
        return (da if ctx.needs_input_grad[0] else None, None) + tuple(grads)


# --------------------------------------------------------------------------------------
# Generator recurrent front: T x [LSTMCell stack -> tanh(proj) fed back, stopper logit]
# --------------------------------------------------------------------------------------
class GFront(object):
    """WN items: per layer [w_ih, w_hh, b_ih, b_hh] * num_layers, then [proj.w, proj.b, stop.w, stop.b]"""

    def __init__(self, frame_size, num_layers, state_size):
        self.fs, self.nl, self.ss = frame_size, num_layers, state_size
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
        x = torch.empty(B, T * fs, device=dev)
        gates = [torch.empty(T, B, 4 * S, device=dev) for _ in range(nl)]
        hs = [torch.empty(T, B, S, device=dev) for _ in range(nl)]
        cs = [torch.empty(T + 1, B, S, device=dev) for _ in range(nl)]
        for l in range(nl):
            cs[l][0].zero_()
        bsum = [lw[l][2] + lw[l][3] for l in range(nl)]
        w_ih0 = lw[0][0]
        # all frames at once: zc_t @ W_ih[:, fs:]^T + b_ih + b_hh
        K.gemm(zc.contiguous().view(T * B, Fz), w_ih0[:, fs:], gates[0].view(T * B, 4 * S), tb=True,
               bias=bsum[0])
        for t in range(T):
            if t > 0:
                K.gemm(x[:, (t - 1) * fs:t * fs], w_ih0[:, :fs], gates[0][t], tb=True, beta=1.0)
                K.gemm(hs[0][t - 1], lw[0][1], gates[0][t], tb=True, beta=1.0)
            K.lstm_cell_fwd(gates[0][t], cs[0][t], cs[0][t + 1], h_out=hs[0][t])
            for l in range(1, nl):
                K.gemm(hs[l - 1][t], lw[l][0], gates[l][t], tb=True, bias=bsum[l])
                if t > 0:
                    K.gemm(hs[l][t - 1], lw[l][1], gates[l][t], tb=True, beta=1.0)
                K.lstm_cell_fwd(gates[l][t], cs[l][t], cs[l][t + 1], h_out=hs[l][t])
            K.gemm(hs[-1][t], pw, x[:, t * fs:(t + 1) * fs], tb=True, bias=pb, act=ACT_TANH)
        s = torch.empty(T * B, 1, device=dev)
        K.gemm(hs[-1].view(T * B, S), sw, s, tb=True, bias=sb)
        ctx.front, ctx.key = front, front.group._key
        ctx.dims = (T, B, Fz)
        ctx.save_for_backward(zc, x, *(gates + hs + cs))
        return x, s.view(T, B).t()

    @staticmethod
    def backward(ctx, dx, ds):
        front = ctx.front
        fs, nl, S = front.fs, front.nl, front.ss
        T, B, Fz = ctx.dims
        prep = front.group.prepare()
        assert front.group._key == ctx.key, 'parameters changed between forward and backward'
        sv = ctx.saved_tensors
        zc, x = sv[0], sv[1]
        gates, hs, cs = sv[2:2 + nl], sv[2 + nl:2 + 2 * nl], sv[2 + 2 * nl:2 + 3 * nl]
        dev = zc.device
        lw = [[p.w for p in prep[4 * l:4 * l + 4]] for l in range(nl)]
        pw, sw = prep[4 * nl].w, prep[4 * nl + 2].w
        dws = _zeros_like_list([it['v'] for it in front.group.items])
        dx = dx.contiguous() if dx is not None else torch.zeros(B, T * fs, device=dev)
        # stopper: s[t,b] = h_last[t,b] . sw + sb
        dh_stop = None
        if ds is not None:
            ds_tb = ds.t().contiguous().view(T * B, 1)
            K.gemm(ds_tb, hs[-1].view(T * B, S), dws[4 * nl + 2], ta=True)
            K.col_sum(ds_tb, dws[4 * nl + 3])
            dh_stop = torch.empty(T, B, S, device=dev)
            K.gemm(ds_tb, sw, dh_stop.view(T * B, S))
        dgs = [torch.empty(T, B, 4 * S, device=dev) for _ in range(nl)]
        dxt = torch.empty(T, B, fs, device=dev)      # d(pre-tanh) of the projection, per frame
        dh_rec = [torch.zeros(B, S, device=dev) for _ in range(nl)]   # dL/dh_l[t] from frame t+1
        dcs = [[torch.zeros(B, S, device=dev), torch.empty(B, S, device=dev)] for _ in range(nl)]
        dxfeed = torch.empty(B, fs, device=dev)      # dL/dx_t through the feedback into frame t+1
        dh_cur = torch.empty(B, S, device=dev)
        for t in reversed(range(T)):
            # x_t = tanh(proj(h_last[t])) receives dx (output) + feedback from frame t+1
            xt = x[:, t * fs:(t + 1) * fs]
            gx = dxt[t]
            gx.copy_(dx[:, t * fs:(t + 1) * fs])
            if t < T - 1:
                K.axpby(dxfeed, gx, 1.0, 1.0)
            xt_c = xt.contiguous()
            K.act_bwd(gx, xt_c, gx, ACT_TANH)
            # dh_last = gx @ pw + recurrent + stopper
            K.gemm(gx, pw, dh_cur, res=dh_rec[-1])
            if dh_stop is not None:
                K.axpby(dh_stop[t], dh_cur, 1.0, 1.0)
            dh_l = dh_cur
            for l in reversed(range(nl)):
                k2 = t & 1
                K.lstm_cell_bwd(gates[l][t], cs[l][t], cs[l][t + 1], dh_l, None,
                                dcs[l][(t + 1) & 1] if t < T - 1 else None, dgs[l][t], dcs[l][k2])
                if t > 0:
                    K.gemm(dgs[l][t], lw[l][1], dh_rec[l])          # into h_l[t-1]
                if l > 0:
                    # input of layer l at frame t is h_{l-1}[t]
                    dh_l = torch.empty(B, S, device=dev)
                    K.gemm(dgs[l][t], lw[l][0], dh_l, res=dh_rec[l - 1])
                elif t > 0:
                    K.gemm(dgs[0][t], lw[0][0][:, :fs], dxfeed)     # into x_{t-1}
        # parameter gradients, one GEMM per tensor over all frames
        dxt2 = dxt.view(T * B, fs)
        K.gemm(dxt2, hs[-1].view(T * B, S), dws[4 * nl], ta=True)
        K.col_sum(dxt2, dws[4 * nl + 1])
        for l in range(nl):
            dg2 = dgs[l].view(T * B, 4 * S)
            dwih = dws[4 * l]
            if l == 0:
                K.gemm(dg2, zc.contiguous().view(T * B, Fz), dwih[:, fs:], ta=True)
                if T > 1:
                    # x_{t-1} for frames 1..T-1: gather the [B, (T-1)*fs] prefix as rows (t,b)
                    xprev = x.view(B, T, fs)[:, :T - 1].transpose(0, 1).contiguous().view((T - 1) * B, fs)
                    K.gemm(dgs[0][1:].view((T - 1) * B, 4 * S), xprev, dwih[:, :fs], ta=True)
            else:
                K.gemm(dg2, hs[l - 1].view(T * B, S), dwih, ta=True)
            if T > 1:
                K.gemm(dgs[l][1:].view((T - 1) * B, 4 * S), hs[l][:T - 1].view((T - 1) * B, S),
                       dws[4 * l + 1], ta=True)
            K.col_sum(dg2, dws[4 * l + 2])
            dws[4 * l + 3].copy_(dws[4 * l + 2])
        dzc = None
        if ctx.needs_input_grad[0]:
            dzc = torch.empty(T * B, Fz, device=dev)
            K.gemm(dgs[0].view(T * B, 4 * S), lw[0][0][:, fs:], dzc)
            dzc = dzc.view(T, B, Fz)
        grads = front.group.backward(dws)
        return (dzc, None) + tuple(grads)


# --------------------------------------------------------------------------------------
# masked BCE-with-logits, / nframes, mean over batch
# --------------------------------------------------------------------------------------
class BCEFn(torch.autograd.Function):
    """loss = mean_b( sum_{t<n_b} bce(x[b,t], target) / n_b ); also returns the per-sample sums."""

    @staticmethod
    def forward(ctx, x, target, nframes):
        B, T = x.shape
        x = x.contiguous()
        per = torch.empty(B, device=x.device)
        loss = torch.zeros(1, device=x.device)
        K.bce_logits_fwd(x, target, nframes, per, loss, 1.0 / B)
        ctx.target = target
        ctx.save_for_backward(x, nframes if nframes is not None else x.new_empty(0))
        ctx.has_n = nframes is not None
        ctx.mark_non_differentiable(per)
        return loss.view(()), per

    @staticmethod
    def backward(ctx, dloss, _dper):
        x, nfr = ctx.saved_tensors
        B, T = x.shape
        dx = torch.empty_like(x)
        g = dloss.contiguous().view(1).float()
        K.bce_logits_bwd(x, ctx.target, nfr if ctx.has_n else None, g, 1.0 / B, dx)
        return dx, None, None
