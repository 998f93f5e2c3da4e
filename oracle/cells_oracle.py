"""TEST INFRASTRUCTURE ONLY (never imported by audiogan_amd).  CPU restatement of the reference's recurrent cells,
cells.py:105-178, which subclass TensorFlow's ``tf.nn.rnn_cell.LSTMCell`` (tensorflow is a dependency the reference
does not pin - no requirements file - and it is absent from this image).  The published TF 1.x LSTMCell algorithm is
restated here: ``lstm_matrix = [inputs, m_prev] @ kernel + bias``; ``i, j, f, o = split(lstm_matrix, 4)``;
``c = sigmoid(f + forget_bias) * c_prev + sigmoid(i) * tanh(j)``; ``m = sigmoid(o) * tanh(c)``; with ``num_proj``:
``m = m @ projection_kernel``.  PARITY UNPINNED: the reference holds no fixture or test for these cells; the gate
algebra is cross-checked against torch.nn.LSTMCell (tests/test_cells.py)."""
import torch


def lstm_cell(inputs, c_prev, m_prev, kernel, bias, forget_bias=1.0, projection_kernel=None):
    z = torch.cat([inputs, m_prev], 1) @ kernel + bias
    i, j, f, o = z.chunk(4, 1)
    c = torch.sigmoid(f + forget_bias) * c_prev + torch.sigmoid(i) * torch.tanh(j)
    m = torch.sigmoid(o) * torch.tanh(c)
    if projection_kernel is not None:
        m = m @ projection_kernel
    return m, c


def projected_lstm_cell(inputs, state, kernel, bias, projection_kernel, forget_bias=1.0):
    """cells.py:120-125: the projection output goes through the activation and THAT is the recurrent state"""
    c_prev, h_prev = state
    m, c = lstm_cell(inputs, c_prev, h_prev, kernel, bias, forget_bias, projection_kernel)
    h = torch.tanh(m)
    return h, (c, h)


def feedback_multi_lstm_cell(inputs, state, params, forget_bias=1.0):
    """cells.py:158-167.  params: list of dicts(kernel, bias[, projection_kernel]) per layer"""
    x = torch.cat([inputs, state[-1]], 1)
    new_state = []
    for i, p in enumerate(params):
        if 'projection_kernel' in p:
            x, s = projected_lstm_cell(x, state[i], p['kernel'], p['bias'], p['projection_kernel'], forget_bias)
        else:
            m, c = lstm_cell(x, state[i][0], state[i][1], p['kernel'], p['bias'], forget_bias)
            x, s = m, (c, m)
        new_state.append(s)
    new_state.append(x)
    return x, new_state


def _tf_layer_norm(x, gamma, beta):
    """tf.contrib.layers.layer_norm defaults: moments over every axis but the batch (begin_norm_axis = 1), gamma / beta over
    the last axis (begin_params_axis = -1), tf.nn.batch_normalization with variance_epsilon = 1e-12"""
    dims = tuple(range(1, x.dim()))
    m = x.mean(dim=dims, keepdim=True)
    v = ((x - m) ** 2).mean(dim=dims, keepdim=True)
    return (x - m) * torch.rsqrt(v + 1e-12) * gamma + beta


def conv2d_lstm_cell(x, state, kernel, bias=None, peep=None, ln=None, forget_bias=1.0, pre=None, post=None):
    """cells.py:50-103, channels_last.  x [B,H,W,Cin]; state (c [B,H,W,F], h); kernel [kh,kw,Cin+Ch,4F] (TF HWIO, gate blocks
    j | i | f | o); bias [4F] or None (only without layer norm, :64-65); peep = (W_ci, W_cf, W_co) each [H,W,F] or None; ln =
    list of five (gamma, beta) in call order j, i, f, o, c or None.  TF 'SAME' (odd kernels: symmetric zero padding).
    PARITY UNPINNED (TensorFlow is absent; the reference holds no fixture): the published TF op definitions restated."""
    import torch.nn.functional as F_
    c, h = state
    if pre is not None:
        x, h = pre(x, h)
    xh = torch.cat([x, h], -1)
    kh, kw = kernel.shape[:2]
    y = F_.conv2d(xh.permute(0, 3, 1, 2), kernel.permute(3, 2, 0, 1), padding=((kh - 1) // 2, (kw - 1) // 2)).permute(0, 2, 3, 1)
    if ln is None and bias is not None:
        y = y + bias
    j, i, f, o = y.chunk(4, -1)
    if peep is not None:
        i = i + peep[0] * c
        f = f + peep[1] * c
    if ln is not None:
        j, i, f = _tf_layer_norm(j, *ln[0]), _tf_layer_norm(i, *ln[1]), _tf_layer_norm(f, *ln[2])
    f = torch.sigmoid(f + forget_bias)
    i = torch.sigmoid(i)
    c = c * f + i * torch.tanh(j)
    if peep is not None:
        o = o + peep[2] * c
    if ln is not None:
        o, c = _tf_layer_norm(o, *ln[3]), _tf_layer_norm(c, *ln[4])
    o = torch.sigmoid(o)
    h = o * torch.tanh(c)
    if post is not None:
        h = post(h)
    return h, (c, h)
