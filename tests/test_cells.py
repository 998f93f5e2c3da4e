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


def test_conv2d_lstm_oracle_reduces_to_lstm_cell_for_1x1():
    """without peepholes / layer norm and with a 1x1 kernel on a 1x1 map the convolutional cell IS TF's LSTMCell with
    the gate blocks in j | i | f | o order (cells.py:66) instead of i | j | f | o: pins the block order of the restatement"""
    torch.manual_seed(3)
    B, I, F_ = 4, 5, 6
    x, c, h = torch.randn(B, 1, 1, I), torch.randn(B, 1, 1, F_), torch.randn(B, 1, 1, F_)
    kernel, bias = torch.randn(1, 1, I + F_, 4 * F_) * 0.3, torch.randn(4 * F_) * 0.1
    h1, (c1, _) = CO.conv2d_lstm_cell(x, (c, h), kernel, bias=bias, forget_bias=0.7)
    kj, ki, kf, ko = kernel[0, 0].chunk(4, 1)
    bj, bi, bf, bo = bias.chunk(4)
    m, c2 = CO.lstm_cell(x.view(B, I), c.view(B, F_), h.view(B, F_), torch.cat([ki, kj, kf, ko], 1), torch.cat([bi, bj, bf, bo]),
                         forget_bias=0.7)
    np.testing.assert_allclose(h1.view(B, F_).numpy(), m.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(c1.view(B, F_).numpy(), c2.numpy(), rtol=1e-5, atol=1e-6)


def _conv2d_lstm_case(dev, cells, shape, ksize, normalize, peephole, post):
    torch.manual_seed(7)
    B, Cin, F_, T = 3, 2, 4, 2
    H, W = shape
    post_cb = (lambda t: torch.tanh(t.mean(-1, keepdim=True))) if post else None       # modeltf.py:315-316
    out_shape = (H, W, 1) if post else None
    cell = cells.Conv2DLSTMCell(shape, F_, ksize, Cin, forget_bias=1.0, normalize=normalize, peephole=peephole,
                                post_rnn_callback=post_cb, output_shape=out_shape).to(dev)
    gen = torch.Generator().manual_seed(8)
    with torch.no_grad():
        for p in cell.parameters():
            p.add_(torch.randn(p.shape, generator=gen).to(dev) * 0.1)
    ref = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in cell.named_parameters()}
    ln = [(ref['ln_gamma.%d' % k], ref['ln_beta.%d' % k]) for k in range(5)] if normalize else None
    peep = (ref['W_ci'], ref['W_cf'], ref['W_co']) if peephole else None
    xs = [torch.randn(B, H, W, Cin, generator=gen) for _ in range(T)]
    c0 = torch.randn(B, H, W, F_, generator=gen)
    h0 = torch.randn(B, H, W, 1 if post else F_, generator=gen)
    st, sto, outs, outs_o = (c0.to(dev), h0.to(dev)), (c0, h0), [], []
    for x in xs:
        y, st = cell(x.to(dev), st)
        yo, sto = CO.conv2d_lstm_cell(x, sto, ref['kernel'], bias=ref.get('bias'), peep=peep, ln=ln, forget_bias=1.0, post=post_cb)
        outs.append(y); outs_o.append(yo)
    for y, yo in zip(outs, outs_o):
        np.testing.assert_allclose(y.detach().cpu().numpy(), yo.detach().numpy(), rtol=1e-3, atol=2e-5)
    np.testing.assert_allclose(st[0].detach().cpu().numpy(), sto[0].detach().numpy(), rtol=1e-3, atol=2e-5)
    wgt = torch.randn(outs_o[0].shape, generator=gen)
    (sum((y * wgt.to(dev)).sum() for y in outs) + st[0].sum()).backward()
    (sum((yo * wgt).sum() for yo in outs_o) + sto[0].sum()).backward()
    for k, p in cell.named_parameters():
        np.testing.assert_allclose(p.grad.cpu().numpy(), ref[k].grad.numpy(), rtol=2e-3,
                                   atol=2e-5 * max(1.0, float(ref[k].grad.abs().max())), err_msg=k)


CONVLSTM_CASES = [((1, 16), (1, 5), True, True, True),        # the reference's use: shape (1, frame), kernel (1, k), modeltf.py:318-322
                  ((3, 8), (3, 3), True, True, False), ((2, 6), (1, 3), False, True, False), ((3, 5), (3, 1), True, False, False)]


@pytest.mark.parametrize('shape,ksize,normalize,peephole,post', CONVLSTM_CASES)
def test_conv2d_lstm_cell_host_logic(monkeypatch, shape, ksize, normalize, peephole, post):
    from tests import kernel_model
    kernel_model.install(monkeypatch)
    from audiogan_amd import cells
    _conv2d_lstm_case(torch.device('cpu'), cells, shape, ksize, normalize, peephole, post)


@pytest.mark.gpu
@pytest.mark.parametrize('shape,ksize,normalize,peephole,post', CONVLSTM_CASES)
def test_conv2d_lstm_cell_matches_oracle(shape, ksize, normalize, peephole, post):
    from audiogan_amd import cells
    _conv2d_lstm_case(torch.device('cuda'), cells, shape, ksize, normalize, peephole, post)
