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
