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
