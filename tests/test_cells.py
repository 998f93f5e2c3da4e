"""cells.py row: oracle gate algebra vs torch.nn.LSTMCell on CPU; HIP cells vs the oracle on GPU."""
import numpy as np
import pytest
import torch

from oracle import cells_oracle as CO


def test_oracle_gate_algebra_matches_torch_lstmcell():
    """TF order (i, j, f, o) + forget_bias restated correctly: with forget_bias 0 and permuted weights the oracle
    must equal torch.nn.LSTMCell (order i, f, g, o; separate ih / hh weights)"""
    torch.manual_seed(0)
    B, I, H = 5, 7, 6
    ref = torch.nn.LSTMCell(I, H)
    x, h, c = torch.randn(B, I), torch.randn(B, H), torch.randn(B, H)
    h1, c1 = ref(x, (h, c))
    w = torch.cat([ref.weight_ih, ref.weight_hh], 1)            # [4H, I+H], rows i, f, g, o
    b = ref.bias_ih + ref.bias_hh
    blocks = lambda t: t.view(4, H, -1)                          # noqa: E731
    wi, wf, wg, wo = blocks(w)
    bi, bf, bg, bo = b.view(4, H)
    kernel = torch.cat([wi, wg, wf, wo], 0).t()                  # TF: columns i, j, f, o
    bias = torch.cat([bi, bg, bf, bo])
    m, c2 = CO.lstm_cell(x, c, h, kernel, bias, forget_bias=0.0)
    np.testing.assert_allclose(m.detach().numpy(), h1.detach().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(c2.detach().numpy(), c1.detach().numpy(), rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize('layers', [1, 2])
def test_feedback_multi_lstm_cell_matches_oracle(layers):
    from audiogan_amd import cells
    torch.manual_seed(1)
    B, I, U, P, T = 6, 12, 32, 16, 3
    cell = cells.FeedbackMultiLSTMCell(I, U, P, num_layers=layers).cuda()
    gen = torch.Generator().manual_seed(2)
    state = cell.random_state(B, generator=gen)
    xs = [torch.randn(B, I, generator=gen) for _ in range(T)]
    # oracle on CPU copies of the same parameters
    params = []
    for c_ in cell.cells:
        p = dict(kernel=c_.kernel.detach().cpu().clone().requires_grad_(True),
                 bias=c_.bias.detach().cpu().clone().requires_grad_(True))
        if hasattr(c_, 'projection_kernel'):
            p['projection_kernel'] = c_.projection_kernel.detach().cpu().clone().requires_grad_(True)
        params.append(p)
    so = [tuple(t.cpu() for t in s) if isinstance(s, tuple) else s.cpu() for s in state]
    s_hip, outs, outs_o = state, [], []
    for x in xs:
        y, s_hip = cell(x.cuda(), s_hip)
        yo, so = CO.feedback_multi_lstm_cell(x, so, params)
        outs.append(y); outs_o.append(yo)
    for y, yo in zip(outs, outs_o):
        np.testing.assert_allclose(y.detach().cpu().numpy(), yo.detach().numpy(), rtol=1e-3, atol=1e-5)
    w = torch.randn(B, P, generator=gen)
    sum((y * w.cuda()).sum() for y in outs).backward()
    sum((yo * w).sum() for yo in outs_o).backward()
    for c_, p in zip(cell.cells, params):
        for name, ref in p.items():
            got = getattr(c_, name).grad.cpu().numpy()
            np.testing.assert_allclose(got, ref.grad.numpy(), rtol=2e-3, atol=2e-5, err_msg=name)
