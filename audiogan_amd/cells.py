"""The recurrent cells of the reference's ``cells.py`` on the HIP kernels.

``ProjectedLSTMCell`` (cells.py:105-125): TensorFlow's ``LSTMCell`` with ``num_proj`` - gates from
``[inputs, h_prev] @ kernel + bias`` in TF's gate order i, j, f, o with ``forget_bias`` added to f; cell
``c = sigmoid(f) * c_prev + sigmoid(i) * tanh(j)``; ``m = sigmoid(o) * tanh(c)``; linear projection
``m @ projection_kernel`` - followed by the projection activation, and that ACTIVATED value is the recurrent state.
``FeedbackMultiLSTMCell`` (cells.py:127-178): concat(inputs, last projected output) -> stacked cells (the last one
projected) -> the new output is also carried as the last state entry; ``random_state`` as :169-178.

Parameters keep TF's variable layout and names (``kernel`` [in + hidden, 4 * units], ``bias`` [4 * units],
``projection_kernel`` [units, num_proj]); the kernels' own gate order (i, f, g, o) is reached by a column permutation
of the kernel, which autograd differentiates back into the TF layout.  Forward and backward of the three pieces
(gate GEMM, cell pointwise step, projection GEMM + tanh) run on ag_gemm / ag_lstm_cell_fwd,bwd / ag_act_*.
TensorFlow is not present in this image and the reference has no fixture for these cells: the oracle
(oracle/cells_oracle.py) restates the published TF algorithm - parity for this row is unpinned."""
import torch
from torch import nn

from . import kernels as K
from .convnets import LinearFn
from .kernels import ACT_TANH


class _CellPointFn(torch.autograd.Function):
    """(gate pre-activations [B,4H] in i|f|g|o order, c_prev [B,H]) -> (m = o * tanh(c), c)"""

    @staticmethod
    def forward(ctx, gates_pre, c_prev):
        gates = gates_pre.contiguous().clone()          # overwritten with the activated gates
        c_prev = c_prev.contiguous()
        B, H = c_prev.shape
        c_new, m = torch.empty(B, H, device=gates.device), torch.empty(B, H, device=gates.device)
        K.lstm_cell_fwd(gates, c_prev, c_new, h_out=m)
        ctx.save_for_backward(gates, c_prev, c_new)
        return m, c_new

    @staticmethod
    def backward(ctx, dm, dc):
        gates, c_prev, c_new = ctx.saved_tensors
        dgates, dc_prev = torch.empty_like(gates), torch.empty_like(c_prev)
        K.lstm_cell_bwd(gates, c_prev, c_new, dm.contiguous() if dm is not None else None, None,
                        dc.contiguous() if dc is not None else None, dgates, dc_prev)
        return dgates, dc_prev


class _TanhFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y = torch.empty_like(x.contiguous())
        K.act_fwd(x.contiguous(), y, ACT_TANH)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, = ctx.saved_tensors
        dx = torch.empty_like(y)
        K.act_bwd(dy.contiguous(), y, dx, ACT_TANH)
        return dx


def _tf_to_kernel_order(units):
    """column permutation TF (i, j, f, o) -> kernels (i, f, g=j, o)"""
    idx = torch.arange(4 * units).view(4, units)
    return torch.cat([idx[0], idx[2], idx[1], idx[3]])


class LSTMCell(nn.Module):
    """TensorFlow ``LSTMCell`` without projection: state (c, h), output h = m."""

    def __init__(self, input_size, num_units, forget_bias=1.0):
        super().__init__()
        self.input_size, self.num_units, self.forget_bias = input_size, num_units, float(forget_bias)
        self.hidden_size = num_units
        self._init_gate_params(input_size + self.hidden_size)

    def _init_gate_params(self, fan_in):
        bound = (6.0 / (fan_in + 4 * self.num_units)) ** 0.5        # glorot_uniform, TF's default kernel initializer
        self.kernel = nn.Parameter(torch.empty(fan_in, 4 * self.num_units).uniform_(-bound, bound))
        self.bias = nn.Parameter(torch.zeros(4 * self.num_units))
        self.register_buffer('_perm', _tf_to_kernel_order(self.num_units), persistent=False)

    @property
    def state_size(self):
        return (self.num_units, self.hidden_size)

    @property
    def output_size(self):
        return self.hidden_size

    def zero_state(self, batch_size, device=None):
        dev = device if device is not None else self.kernel.device
        return (torch.zeros(batch_size, self.num_units, device=dev), torch.zeros(batch_size, self.hidden_size, device=dev))

    def _cell(self, inputs, state):
        c_prev, h_prev = state
        w = self.kernel[:, self._perm].t()                        # [4u, in+hidden], kernels' gate order
        b = self.bias[self._perm]
        fb = torch.zeros_like(b)
        fb[self.num_units:2 * self.num_units] = self.forget_bias   # forget gate block in (i, f, g, o)
        z = LinearFn.apply(torch.cat([inputs, h_prev], 1), w, b + fb)
        return _CellPointFn.apply(z, c_prev)

    def forward(self, inputs, state):
        m, c = self._cell(inputs, state)
        return m, (c, m)


class ProjectedLSTMCell(LSTMCell):
    """cells.py:105-125"""

    def __init__(self, input_size, num_units, num_proj, forget_bias=1.0):
        nn.Module.__init__(self)
        self.input_size, self.num_units, self.forget_bias = input_size, num_units, float(forget_bias)
        self.hidden_size = self.num_proj = num_proj
        self._init_gate_params(input_size + num_proj)
        bound = (6.0 / (num_units + num_proj)) ** 0.5
        self.projection_kernel = nn.Parameter(torch.empty(num_units, num_proj).uniform_(-bound, bound))

    def forward(self, inputs, state):
        m, c = self._cell(inputs, state)
        h = _TanhFn.apply(LinearFn.apply(m, self.projection_kernel.t(), None))
        return h, (c, h)


class FeedbackMultiLSTMCell(nn.Module):
    """cells.py:127-178: state = [per-cell (c, h)] * num_layers + [last projected output]"""

    def __init__(self, input_size, num_units, num_proj, num_layers=1, forget_bias=1.0):
        super().__init__()
        self.num_units, self.num_proj, self.num_layers = num_units, num_proj, num_layers
        cells, fan = [], input_size + num_proj
        for i in range(num_layers):
            if i == num_layers - 1:
                cells.append(ProjectedLSTMCell(fan, num_units, num_proj, forget_bias))
            else:
                cells.append(LSTMCell(fan, num_units, forget_bias))
            fan = num_units
        self.cells = nn.ModuleList(cells)

    @property
    def output_size(self):
        return self.num_proj

    @property
    def state_size(self):
        return [c.state_size for c in self.cells] + [self.num_proj]

    def zero_state(self, batch_size, device=None):
        dev = device if device is not None else self.cells[0].kernel.device
        return [c.zero_state(batch_size, dev) for c in self.cells] + [torch.zeros(batch_size, self.num_proj, device=dev)]

    def random_state(self, batch_size, device=None, generator=None):
        """cells.py:169-178: every state tensor ~ N(0, 1)"""
        out = []
        for s in self.zero_state(batch_size, device):
            if isinstance(s, tuple):
                out.append(tuple(torch.randn(t.shape, generator=generator).to(t.device) for t in s))
            else:
                out.append(torch.randn(s.shape, generator=generator).to(s.device))
        return out

    def forward(self, inputs, state):
        x = torch.cat([inputs, state[-1]], 1)
        new_state = []
        for i, cell in enumerate(self.cells):
            x, s = cell(x, state[i])
            new_state.append(s)
        new_state.append(x)
        return x, new_state


# ----------------------------------------------------------------------------------------------------------------------
# Conv2DLSTMCell (cells.py:4-103): convolutional LSTM with peepholes and TF layer normalisation
# ----------------------------------------------------------------------------------------------------------------------
class _Conv2dSameFn(torch.autograd.Function):
    """y[h,b,:,w] = sum_{dh,dw} W[dh,dw]^T x[h+dh-ph, b, :, w+dw-pw] ('SAME', stride 1, odd kernel) on maps laid out
    [H, B, C, W]: one launch of the 1-D conv engine per kernel row dh over row-shifted views ((rows x batch) is the
    engine's batch axis, W its time axis), accumulating into y.  w: TF layout [kh, kw, Cin, Cout]; bias [Cout] or None."""

    @staticmethod
    def _prep(w):
        from .common import Prepared
        from .ops import ConvSpec
        kh, kw, n, m = w.shape
        spec = ConvSpec('conv', n, m, kw, 1, (kw - 1) // 2)
        preps = []
        for dh in range(kh):
            w1 = w[dh].permute(2, 1, 0).contiguous()                 # [Cout, Cin, kw]: NN.Conv1d layout
            p = Prepared(w=None, wpa=torch.zeros(K.wpa_numel(m, n, kw), device=w.device),
                         wpb=torch.zeros(K.wpb_numel(m, n, kw, 1), device=w.device))
            K.prep_conv_weight(w1, p.wpa, p.wpb, 1)
            preps.append(p)
        return spec, preps

    @staticmethod
    def _rows(H, kh, dh):
        off = dh - (kh - 1) // 2
        return max(0, -off), min(H, H - off), off

    @staticmethod
    def forward(ctx, x, w, bias):
        from .ops import conv_fwd
        H, B, n, W = x.shape
        kh, kw, n2, m = w.shape
        assert n2 == n and kh % 2 == 1 and kw % 2 == 1, "Conv2DLSTMCell: 'SAME' with an even kernel pads asymmetrically - not built"
        x = x.contiguous()
        spec, preps = _Conv2dSameFn._prep(w.data)
        y = torch.zeros(H, B, m, W, device=x.device)
        for dh in range(kh):
            h0, h1, off = _Conv2dSameFn._rows(H, kh, dh)
            if h1 > h0:
                K.conv_engine(x[h0 + off:h1 + off].reshape(-1, n, W), preps[dh].wpa, y[h0:h1].reshape(-1, m, W), kw, 1, spec.pad, 0,
                              bias=bias.data if (bias is not None and off == 0) else None, accumulate=True)
        ctx.save_for_backward(x, w.data)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        from .ops import conv_bwd_data, conv_wgrad
        x, w = ctx.saved_tensors
        H, B, n, W = x.shape
        kh, kw, _, m = w.shape
        dy = dy.contiguous()
        spec, preps = _Conv2dSameFn._prep(w)
        dx = torch.zeros_like(x) if ctx.needs_input_grad[0] else None
        dw = torch.zeros(kh, m, n, kw, device=x.device) if ctx.needs_input_grad[1] else None
        for dh in range(kh):
            h0, h1, off = _Conv2dSameFn._rows(H, kh, dh)
            if h1 <= h0:
                continue
            dyv, xv = dy[h0:h1].reshape(-1, m, W), x[h0 + off:h1 + off].reshape(-1, n, W)
            if dx is not None:
                conv_bwd_data(spec, preps[dh], dyv, dx[h0 + off:h1 + off].reshape(-1, n, W), accumulate=True)
            if dw is not None:
                conv_wgrad(spec, xv, dyv, dw[dh], None)
        db = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = torch.zeros(m, device=x.device)
            K.channel_sum(dy.reshape(-1, m, W), db)
        return dx, (dw.permute(0, 3, 2, 1).contiguous() if dw is not None else None), db


class _LayerNormFn(torch.autograd.Function):
    """tf.contrib.layers.layer_norm (begin_norm_axis 1, begin_params_axis -1, eps 1e-12) on a [H,B,F,W] map"""

    @staticmethod
    def forward(ctx, x, gamma, beta):
        x = x.contiguous()
        H, B, F, W = x.shape
        y, mean, rstd = torch.empty_like(x), torch.empty(B, device=x.device), torch.empty(B, device=x.device)
        K.layer_norm_hbfw_fwd(x, gamma.data.contiguous(), beta.data.contiguous(), 1e-12, y, mean, rstd)
        ctx.save_for_backward(x, gamma.data, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, rstd = ctx.saved_tensors
        H, B, F, W = x.shape
        dx = torch.empty_like(x)
        dgp, dbp = torch.empty(B, F, device=x.device), torch.empty(B, F, device=x.device)
        K.layer_norm_hbfw_bwd(dy.contiguous(), x, gamma.contiguous(), mean, rstd, dx, dgp, dbp)
        dg, db = torch.zeros(F, device=x.device), torch.zeros(F, device=x.device)
        K.col_sum(dgp, dg)
        K.col_sum(dbp, db)
        return dx, dg, db


class _PeepholeFn(torch.autograd.Function):
    """conv output y [H,B,4F,W] (blocks j|i|f|o), c -> j, i + W_ci c, f + W_cf c, o   (cells.py:66-70)"""

    @staticmethod
    def forward(ctx, y, c, wci, wcf):
        y, c = y.contiguous(), c.contiguous()
        j, ip, fp, o = (torch.empty_like(c) for _ in range(4))
        K.convlstm_peephole_fwd(y, c, wci.data.contiguous() if wci is not None else None,
                                wcf.data.contiguous() if wcf is not None else None, j, ip, fp, o)
        ctx.peep = wci is not None
        ctx.save_for_backward(c, *([wci.data, wcf.data] if wci is not None else []))
        return j, ip, fp, o

    @staticmethod
    def backward(ctx, dj, di, df, do):
        c = ctx.saved_tensors[0]
        wci, wcf = (ctx.saved_tensors[1].contiguous(), ctx.saved_tensors[2].contiguous()) if ctx.peep else (None, None)
        z = lambda g: g.contiguous() if g is not None else torch.zeros_like(c)  # noqa: E731
        dy = torch.empty(c.size(0), c.size(1), 4 * c.size(2), c.size(3), device=c.device)
        dc = torch.zeros_like(c)
        dwci, dwcf = (torch.empty_like(wci), torch.empty_like(wcf)) if ctx.peep else (None, None)
        K.convlstm_peephole_bwd(z(dj), z(di), z(df), z(do), c, wci, wcf, dy, dc, dwci, dwcf)
        return dy, dc, dwci, dwcf


class _ConvCellFn(torch.autograd.Function):
    """(j, i, f, c, o_raw) -> c' = c sigmoid(f + fb) + sigmoid(i) tanh(j),  o_pre = o_raw + W_co c'   (cells.py:77-82)"""

    @staticmethod
    def forward(ctx, j, i_, f_, c, o_raw, wco, fb):
        j, i_, f_, c, o_raw = (t.contiguous() for t in (j, i_, f_, c, o_raw))
        cn, op = torch.empty_like(c), torch.empty_like(c)
        K.convlstm_cell_fwd(j, i_, f_, c, o_raw, wco.data.contiguous() if wco is not None else None, fb, cn, op)
        ctx.fb, ctx.peep = fb, wco is not None
        ctx.save_for_backward(j, i_, f_, c, cn, *([wco.data] if wco is not None else []))
        return cn, op

    @staticmethod
    def backward(ctx, dcn, dop):
        j, i_, f_, c, cn = ctx.saved_tensors[:5]
        wco = ctx.saved_tensors[5].contiguous() if ctx.peep else None
        dcn = dcn.contiguous().clone() if dcn is not None else torch.zeros_like(c)
        dop = dop.contiguous() if dop is not None else torch.zeros_like(c)
        dj, di, df, dc = (torch.empty_like(c) for _ in range(4))
        dwco = torch.empty_like(wco) if ctx.peep else None
        K.convlstm_cell_bwd(j, i_, f_, c, cn, wco, ctx.fb, dcn, dop, dj, di, df, dc, dwco)
        return dj, di, df, dc, dop, dwco, None


class _ConvOutFn(torch.autograd.Function):
    """h = sigmoid(o) tanh(c)   (cells.py:88-89)"""

    @staticmethod
    def forward(ctx, o, c):
        o, c = o.contiguous(), c.contiguous()
        h = torch.empty_like(o)
        K.convlstm_out_fwd(o, c, h)
        ctx.save_for_backward(o, c)
        return h

    @staticmethod
    def backward(ctx, dh):
        o, c = ctx.saved_tensors
        do, dc = torch.empty_like(o), torch.empty_like(c)
        K.convlstm_out_bwd(o, c, dh.contiguous(), do, dc)
        return do, dc


class Conv2DLSTMCell(nn.Module):
    """cells.py:4-103, data_format 'channels_last' (TF's default; what modeltf.py:318-322 uses with shape (1, frame) and a
    (1, k) kernel).  ``forward(x, state)``: x [B, H, W, Cin]; state = (c [B, H, W, filters], h [B, *output_shape]) ->
    (h, (c, h)).  Parameters keep TF's variable names and layouts: ``kernel`` [kh, kw, Cin + Ch, 4 * filters] (gate blocks
    j | i | f | o), ``bias`` [4 * filters] only when ``normalize`` is off (:64-65), peepholes ``W_ci`` / ``W_cf`` / ``W_co``
    [H, W, filters] (:68-70, :81), and one (gamma, beta) pair per ``layer_norm`` call in call order j, i, f, o, c (TF names
    them LayerNorm, LayerNorm_1 ... LayerNorm_4; here ``ln_gamma[k]`` / ``ln_beta[k]``).  The activation is tanh (the
    reference's default, the only one the kernels implement).  ``pre_rnn_callback`` / ``post_rnn_callback`` are user
    callables on channels_last tensors exactly as in the reference (:53-54, :98-99).  Compute: the 'SAME' convolution on the
    1-D conv engine (one launch per kernel row), everything else on csrc/convlstm.hip."""

    def __init__(self, shape, filters, kernel, in_channels, forget_bias=1.0, normalize=True, peephole=True,
                 pre_rnn_callback=None, post_rnn_callback=None, output_shape=None, state_channels=None):
        super().__init__()
        self.shape, self.filters, self.ksize = tuple(shape), int(filters), tuple(kernel)
        self.forget_bias, self.normalize, self.peephole = float(forget_bias), bool(normalize), bool(peephole)
        self.pre_rnn_callback, self.post_rnn_callback = pre_rnn_callback, post_rnn_callback
        self._output_shape = tuple(output_shape) if output_shape is not None else None
        H, W = self.shape
        ch = state_channels if state_channels is not None else (self._output_shape[-1] if self._output_shape else self.filters)
        n, m = in_channels + ch, 4 * self.filters
        bound = (6.0 / (self.ksize[0] * self.ksize[1] * (n + m))) ** 0.5          # glorot_uniform (TF get_variable default)
        self.kernel = nn.Parameter(torch.empty(self.ksize[0], self.ksize[1], n, m).uniform_(-bound, bound))
        if not self.normalize:
            self.bias = nn.Parameter(torch.zeros(m))
        if self.peephole:
            pb = (6.0 / (H * W + self.filters)) ** 0.5
            for name in ('W_ci', 'W_cf', 'W_co'):
                setattr(self, name, nn.Parameter(torch.empty(H, W, self.filters).uniform_(-pb, pb)))
        if self.normalize:
            self.ln_gamma = nn.ParameterList([nn.Parameter(torch.ones(self.filters)) for _ in range(5)])
            self.ln_beta = nn.ParameterList([nn.Parameter(torch.zeros(self.filters)) for _ in range(5)])

    @property
    def state_size(self):
        return (self.shape + (self.filters,), self.output_size)

    @property
    def output_size(self):
        return self._output_shape or (self.shape + (self.filters,))

    def zero_state(self, batch_size, device=None):
        dev = device if device is not None else self.kernel.device
        return (torch.zeros((batch_size,) + self.shape + (self.filters,), device=dev),
                torch.zeros((batch_size,) + tuple(self.output_size), device=dev))

    @staticmethod
    def _to_hbcw(t):           # [B,H,W,C] -> [H,B,C,W]
        return t.permute(1, 0, 3, 2).contiguous()

    @staticmethod
    def _from_hbcw(t):         # [H,B,C,W] -> [B,H,W,C]
        return t.permute(1, 0, 3, 2).contiguous()

    def _ln(self, k, t):
        return _LayerNormFn.apply(t, self.ln_gamma[k], self.ln_beta[k]) if self.normalize else t

    def forward(self, x, state):
        c, h = state
        if self.pre_rnn_callback is not None:
            x, h = self.pre_rnn_callback(x, h)
        xh = self._to_hbcw(torch.cat([x, h], -1))
        cm = self._to_hbcw(c)
        y = _Conv2dSameFn.apply(xh, self.kernel, None if self.normalize else self.bias)
        pw = (lambda p: p.permute(0, 2, 1).contiguous()) if self.peephole else None      # [H,W,F] -> [H,F,W]
        j, i_, f_, o = _PeepholeFn.apply(y, cm, pw(self.W_ci) if self.peephole else None, pw(self.W_cf) if self.peephole else None)
        j, i_, f_ = self._ln(0, j), self._ln(1, i_), self._ln(2, f_)
        cn, op = _ConvCellFn.apply(j, i_, f_, cm, o, pw(self.W_co) if self.peephole else None, self.forget_bias)
        op, cn = self._ln(3, op), self._ln(4, cn)
        hm = _ConvOutFn.apply(op, cn)
        h_new, c_new = self._from_hbcw(hm), self._from_hbcw(cn)
        if self.post_rnn_callback is not None:
            h_new = self.post_rnn_callback(h_new)
        return h_new, (c_new, h_new)
