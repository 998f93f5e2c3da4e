"""Host logic of audiogan_amd (autograd blocks, slab bookkeeping, weight layouts, module API)
checked on CPU against the oracle and the reference-generated golden vectors, with the HIP
kernels replaced by the torch model in tests/kernel_model.py.  The real kernels are covered by
the -m gpu tests."""
import os

import numpy as np
import pytest
import torch

from oracle import audiogan_oracle as O
from tests import kernel_model

import audiogan_amd
from audiogan_amd import modules as M


@pytest.fixture(autouse=True)
def _model_kernels(monkeypatch):
    kernel_model.install(monkeypatch)


def _load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name)))


def _sd(v, prefix='sd.'):
    return {k[len(prefix):]: torch.from_numpy(a) for k, a in v.items() if k.startswith(prefix)}


def _check_grads(module, v, rtol=2e-4, atol=2e-5):
    for k, p in module.named_parameters():
        ref = v['grad.' + k]
        got = p.grad.numpy() if p.grad is not None else np.zeros_like(ref)
        np.testing.assert_allclose(got, ref, rtol=rtol, atol=atol, err_msg=k)


def test_state_dict_keys_match_oracle():
    g, go = M.Generator(16, 6, 5, 24, 2, struct=[[9, 4, 8, 4], [5, 2, 6, 4]]), \
        O.Generator(16, 6, 5, 24, 2, struct=[[9, 4, 8, 4], [5, 2, 6, 4]])
    assert list(g.state_dict().keys()) == list(go.state_dict().keys())
    d, do = M.Discriminator(16, 6, 1, [[7, 2, 4], [5, 2, 8]]), O.Discriminator(16, 6, 1, [[7, 2, 4], [5, 2, 8]])
    assert list(d.state_dict().keys()) == list(do.state_dict().keys())
    e, eo = M.Embedder(6, 4, 1, 32), O.Embedder(6, 4, 1, 32)
    assert list(e.state_dict().keys()) == list(eo.state_dict().keys())
    for a, b in ((g, go), (d, do)):
        for (k, p), (_, q) in zip(a.state_dict().items(), b.state_dict().items()):
            assert p.shape == q.shape, k


def test_same_seed_same_init():
    torch.manual_seed(7)
    g = M.Generator(16, 6, 5, 24, 1, struct=[[9, 4, 8, 4]])
    torch.manual_seed(7)
    go = O.Generator(16, 6, 5, 24, 1, struct=[[9, 4, 8, 4]])
    for (k, p), (_, q) in zip(g.state_dict().items(), go.state_dict().items()):
        np.testing.assert_allclose(p.numpy(), q.numpy(), rtol=1e-6, atol=1e-7, err_msg=k)


def test_generator_vs_reference_vectors(golden_dir):
    v = _load(golden_dir, 'ref_generator.npz')
    fs, es, ns, ss, nl = [int(a) for a in v['cfg']]
    g = M.Generator(fs, es, ns, ss, nl, struct=v['cfg_struct'].tolist())
    g.load_state_dict(_sd(v), strict=True)
    z, c = torch.from_numpy(v['z']), torch.from_numpy(v['c'])
    x, s, stops, length = g(z=z, c=c, stop='never')
    np.testing.assert_allclose(x.detach().numpy(), v['x'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(s.detach().numpy(), v['s'], rtol=1e-4, atol=1e-5)
    np.testing.assert_array_equal(length.numpy(), v['length'])
    assert len(stops) == z.size(1) and tuple(stops[0].shape) == (z.size(0), 1)
    ((x * torch.from_numpy(v['gy'])).sum() + (s * torch.from_numpy(v['gs'])).sum()).backward()
    _check_grads(g, v)


def test_discriminator_ragged_vs_reference_vectors(golden_dir):
    v = _load(golden_dir, 'ref_discriminator.npz')
    ss, es, nl = [int(a) for a in v['cfg']]
    d = M.Discriminator(ss, es, nl, cnn_struct=v['cfg_struct'].tolist())
    d.load_state_dict(_sd(v), strict=True)
    x = torch.from_numpy(v['x']).requires_grad_(True)
    logits, acts, act_lens, nfr = d(x, torch.from_numpy(v['length']), torch.from_numpy(v['c']))
    np.testing.assert_allclose(logits.detach().numpy(), v['logits'], rtol=1e-4, atol=1e-5)
    np.testing.assert_array_equal(nfr.numpy(), v['nframes'])
    for i, (a, l) in enumerate(zip(acts, act_lens)):
        np.testing.assert_allclose(a.detach().numpy(), v['act%d' % i], rtol=1e-4, atol=1e-5)
        np.testing.assert_array_equal(l.numpy(), v['actlen%d' % i])
    (logits * torch.from_numpy(v['gl'])).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), v['gx'], rtol=2e-4, atol=2e-5)
    _check_grads(d, v)


def test_embedder_vs_reference_vectors(golden_dir):
    v = _load(golden_dir, 'ref_embedder.npz')
    e = M.Embedder(output_size=6, char_embed_size=4, num_chars=32)
    e.load_state_dict(_sd(v), strict=True)
    emb = e(torch.from_numpy(v['chars']), torch.from_numpy(v['clen']))
    np.testing.assert_allclose(emb.detach().numpy(), v['emb'], rtol=1e-4, atol=1e-5)


def test_generator_stop_draws_match_oracle():
    torch.manual_seed(3)
    go = O.Generator(8, 4, 3, 12, 1, struct=[[5, 2, 4, 2]])
    g = M.Generator(8, 4, 3, 12, 1, struct=[[5, 2, 4, 2]])
    g.load_state_dict(go.state_dict(), strict=True)
    z, c = torch.randn(3, 6, 3), torch.randn(3, 4)
    stop = torch.tensor([[0, 0, 1, 0, 0, 0], [0, 1, 0, 0, 1, 0], [0, 0, 0, 1, 0, 0]])
    xo, so, stops_o, len_o = go(z=z, c=c, stop=stop)
    x, s, stops, length = g(z=z, c=c, stop=stop)
    assert x.shape == xo.shape and s.shape == so.shape and len(stops) == len(stops_o) == 4
    np.testing.assert_allclose(x.detach().numpy(), xo.detach().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_array_equal(length.numpy(), len_o.numpy())


def test_stopper_surrogate_trains_only_the_stop_head():
    """REINFORCE of the stop head as a log-prob surrogate: gradients equal the oracle's autograd of the same
    expression, only the stopper's parameters receive any (others frozen as audiogan.py:897-901) - also when the
    gradients are accumulated straight into existing .grad buffers"""
    from audiogan_amd import losses
    torch.manual_seed(4)
    go = O.Generator(8, 4, 3, 12, 1, struct=[[5, 2, 4, 2]])
    g = M.Generator(8, 4, 3, 12, 1, struct=[[5, 2, 4, 2]])
    g.load_state_dict(go.state_dict(), strict=True)
    z, c = torch.randn(3, 6, 3), torch.randn(3, 4)
    stop = torch.tensor([[0, 0, 0, 1, 0, 0], [0, 0, 0, 0, 1, 0], [0, 0, 0, 0, 0, 0]])
    reward = torch.tensor([0.3, -1.2, 0.7])
    for p in g.parameters():                 # existing .grad buffers: the direct-accumulation path
        p.grad = torch.zeros_like(p)
    _, so, stops_o, _ = go(z=z, c=c, stop=stop)
    _, s, stops, _ = g(z=z, c=c, stop=stop)
    st = torch.cat(stops_o, 1).float()
    ref = -(reward.view(-1, 1) * (st * torch.nn.functional.logsigmoid(so) +
                                  (1 - st) * torch.nn.functional.logsigmoid(-so))).sum()
    for p in go.parameters():
        p.requires_grad_(False)
    for p in go.stopper.parameters():
        p.requires_grad_(True)
    ref.backward()
    with losses.only_stopper_trains(g):
        loss = losses.stopper_surrogate_loss(s, stops, reward)
        loss.backward()
    np.testing.assert_allclose(float(loss), float(ref), rtol=1e-5)
    stop_ids = set(id(p) for p in g.stopper.parameters())
    ref_grads = dict(go.named_parameters())
    for name, p in g.named_parameters():
        if id(p) in stop_ids:
            if name.endswith('bias_v'):
                continue
            np.testing.assert_allclose(p.grad.numpy(), ref_grads[name].grad.numpy(), rtol=1e-4, atol=1e-6, err_msg=name)
        else:
            assert float(p.grad.abs().max()) == 0.0, name
        assert p.requires_grad


def test_lstm_layer_with_time_invariant_input_matches_torch_lstm():
    """LSTMSeqFn's static input (projected once per clip, added inside the step kernel) must equal torch.nn.LSTM on
    cat([x_t, c]) for the output AND for every gradient: x, c (the conditioning embedding trains through it), weights
    (both column blocks of weight_ih), biases - ragged lengths included"""
    from audiogan_amd import ops
    torch.manual_seed(6)
    T, B, Fx, Fc, H = 5, 4, 8, 3, 16
    ref = torch.nn.LSTM(Fx + Fc, H, bidirectional=True)
    x = torch.randn(T, B, Fx, requires_grad=True)
    c = torch.randn(B, Fc, requires_grad=True)
    lens = torch.tensor([5, 3, 5, 2])
    w = []
    for sfx in ('', '_reverse'):
        for n in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh'):
            w.append(getattr(ref, '%s_l0%s' % (n, sfx)).detach().clone().requires_grad_(True))
    y = ops.LSTMSeqFn.apply(x, lens, 2, c, *w)
    xin = torch.cat([x.detach(), c.detach().unsqueeze(0).expand(T, B, Fc)], 2).requires_grad_(True)
    packed = torch.nn.utils.rnn.pack_padded_sequence(xin, lens, enforce_sorted=False)
    yo, _ = torch.nn.utils.rnn.pad_packed_sequence(ref(packed)[0], total_length=T)
    np.testing.assert_allclose(y.detach().numpy(), yo.detach().numpy(), rtol=1e-4, atol=1e-5)
    gy = torch.randn(T, B, 2 * H)
    (y * gy).sum().backward()
    (yo * gy).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), xin.grad[:, :, :Fx].numpy(), rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(c.grad.numpy(), xin.grad[:, :, Fx:].sum(0).numpy(), rtol=1e-3, atol=1e-5)
    i = 0
    for sfx in ('', '_reverse'):
        for n in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh'):
            np.testing.assert_allclose(w[i].grad.numpy(), getattr(ref, '%s_l0%s' % (n, sfx)).grad.numpy(),
                                       rtol=1e-3, atol=1e-5, err_msg=n + sfx)
            i += 1


def _tiny_pair():
    torch.manual_seed(11)
    go = O.Generator(16, 6, 5, 24, 1, struct=[[9, 4, 8, 4], [9, 4, 8, 4]])
    do = O.Discriminator(16, 6, 1, cnn_struct=[[7, 2, 4], [7, 2, 8], [5, 2, 8]])
    g = M.Generator(16, 6, 5, 24, 1, struct=[[9, 4, 8, 4], [9, 4, 8, 4]])
    d = M.Discriminator(16, 6, 1, cnn_struct=[[7, 2, 4], [7, 2, 8], [5, 2, 8]])
    g.load_state_dict(go.state_dict())
    d.load_state_dict(do.state_dict())
    return go, do, g, d


@pytest.mark.parametrize('kind', ['rmsprop', 'adam'])
def test_train_steps_match_oracle(kind):
    from audiogan_amd import optim, train
    go, do, g, d = _tiny_pair()
    B, T, fs = 3, 4, 16
    gen = torch.Generator().manual_seed(5)
    real = torch.from_numpy(O.synthetic_clips(B, T * fs, 'sine')).float()
    real_len = torch.full((B,), T * fs, dtype=torch.long)
    c = torch.randn(B, 6, generator=gen)
    opt_do, opt_go = O.make_optimizer(list(do.parameters()), kind, 1e-3), \
        O.make_optimizer(list(go.parameters()), kind, 1e-3)
    opt_d, opt_g = optim.make_optimizer(list(d.parameters()), kind, 1e-3), \
        optim.make_optimizer(list(g.parameters()), kind, 1e-3)
    stop = torch.zeros(B, T, dtype=torch.long)
    for it in range(2):
        z = torch.randn(B, T, 5, generator=gen)
        nr = torch.randn(B, T * fs, generator=gen) * 0.01
        nf = torch.randn(B, T * fs, generator=gen) * 0.01
        lo, _, _ = O.d_step(go, do, opt_do, real, real_len, c, z, nr, nf, 1.0, stop=stop)
        l, _, _ = train.d_step(g, d, opt_d, real, real_len, c, z, nr, nf, 1.0, check=True)
        np.testing.assert_allclose(float(l), float(lo), rtol=1e-4)
        lo, fo, _ = O.g_step(go, do, opt_go, c, z, nf, 0.1, stop=stop)
        l, f, _ = train.g_step(g, d, opt_g, c, z, nf, 0.1, check=True)
        np.testing.assert_allclose(float(l), float(lo), rtol=1e-4)
        np.testing.assert_allclose(f.numpy(), fo.numpy(), rtol=1e-3, atol=1e-5)
    for (k, p), (_, q) in zip(list(d.state_dict().items()) + list(g.state_dict().items()),
                              list(do.state_dict().items()) + list(go.state_dict().items())):
        if k.split('.')[-1].startswith('bias') and k.endswith('_v'):
            # d(loss)/d(bias_v) is analytically 0 (bias = g*sign(v)); what reaches the optimiser is
            # rounding noise that RMSprop/Adam normalise to O(lr) steps of arbitrary sign, in the
            # reference as much as here.  Only the sign of v enters the model.
            np.testing.assert_array_equal(np.sign(p.numpy()), np.sign(q.numpy()), err_msg=k)
            continue
        np.testing.assert_allclose(p.numpy(), q.numpy(), rtol=2e-3, atol=2e-5, err_msg=k)


def test_per_sample_bce_api():
    import audiogan_amd as A
    x = torch.randn(4, 7, requires_grad=True)
    n = torch.tensor([7, 3, 1, 5])
    w = A.length_mask((4, 7), n)
    per = A.binary_cross_entropy_with_logits_per_sample(x, torch.full((4, 7), 0.9), weight=w)
    xo = x.detach().clone().requires_grad_(True)
    ref = O.binary_cross_entropy_with_logits_per_sample(xo, torch.full((4, 7), 0.9), weight=w)
    np.testing.assert_allclose(per.detach().numpy(), ref.detach().numpy(), rtol=1e-5, atol=1e-6)
    (per / n.float()).mean().backward()
    (ref / n.float()).mean().backward()
    np.testing.assert_allclose(x.grad.numpy(), xo.grad.numpy(), rtol=1e-5, atol=1e-7)
    with pytest.raises(ValueError):
        A.binary_cross_entropy_with_logits_per_sample(x, torch.zeros(4, 3))


def test_aligned_shapes_take_fused_paths_and_match_oracle():
    """sizes that satisfy the fused-step / skinny-product requirements (multiples of 8)"""
    torch.manual_seed(21)
    gcfg = dict(frame_size=32, embed_size=8, noise_size=8, state_size=64, num_layers=2,
                struct=[[17, 8, 16, 8], [9, 4, 16, 8]])
    dcfg = dict(state_size=64, embed_size=8, num_layers=2, cnn_struct=[[7, 2, 8], [7, 2, 16]])
    go, do = O.Generator(**gcfg), O.Discriminator(**dcfg)
    g, d = M.Generator(**gcfg), M.Discriminator(**dcfg)
    g.load_state_dict(go.state_dict()); d.load_state_dict(do.state_dict())
    B, T = 5, 4
    z, c = torch.randn(B, T, 8), torch.randn(B, 8)
    lens = torch.tensor([128, 128, 77, 30, 128])
    xo, so, _, _ = go(z=z, c=c, stop=torch.zeros(B, T, dtype=torch.long))
    x, s, _, _ = g(z=z, c=c, stop='never')
    np.testing.assert_allclose(x.detach().numpy(), xo.detach().numpy(), rtol=1e-4, atol=1e-5)
    lo = do(xo, lens, c)[0]
    l = d(x, lens, c)[0]
    np.testing.assert_allclose(l.detach().numpy(), lo.detach().numpy(), rtol=1e-4, atol=1e-5)
    w = torch.randn(lo.shape)
    ((lo * w).sum() + (so * 0.3).sum()).backward()
    ((l * w).sum() + (s * 0.3).sum()).backward()
    for a, b in ((g, go), (d, do)):
        for (k, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
            if k.split('.')[-1].startswith('bias') and k.endswith('_v'):
                continue
            np.testing.assert_allclose(p.grad.numpy(), q.grad.numpy(), rtol=5e-4, atol=2e-5, err_msg=k)


def test_front_backward_persistence_follows_the_gradient_bucket(monkeypatch):
    """a generator whose optimiser carries a gradient bucket (multi-GPU: a collective may run beside its backward) runs the
    recurrent front's backward in the per-frame form; without a bucket the persistent launch stays allowed.  The switch is
    restored afterwards, also when the backward raises (kernels.front_bwd_persist / train._single_gpu)"""
    from audiogan_amd import kernels as K, train
    seen = []

    class _Loss(object):
        def __init__(self, fail=False):
            self.fail = fail

        def backward(self):
            seen.append(K.PERSIST_FRONT_BWD[0])
            if self.fail:
                raise RuntimeError('boom')

    class _Opt(object):
        bucket = None

    assert K.PERSIST_FRONT_BWD[0] is True
    o = _Opt()
    with K.front_bwd_persist(train._single_gpu(o)):
        _Loss().backward()
    o.bucket = object()
    with K.front_bwd_persist(train._single_gpu(o)):
        _Loss().backward()
        with K.front_bwd_persist(True):                 # an inner "on" cannot override an outer "off"
            _Loss().backward()
    with pytest.raises(RuntimeError):
        with K.front_bwd_persist(False):
            _Loss(fail=True).backward()
    assert seen == [True, False, False, False] and K.PERSIST_FRONT_BWD[0] is True
    # and the shape query honours it (no GPU here: the device check comes first, so use the flag path only)
    monkeypatch.setattr(K, 'PERSIST', [True])
    with K.front_bwd_persist(False):
        assert K.gfront_bwd_persist_ok(64, 1024, 256, torch.device('cpu')) is False


def test_g_backward_late_is_per_frame_unless_the_caller_opts_in():
    """the eager phase-split path (bucket.all_reduce(async_op=True, part='early'); g_backward_late(keep))
    must not start a persistent front backward beside the collective - the guard lives in g_backward_late itself"""
    from audiogan_amd import kernels as K, train
    seen = []

    class _X(object):
        def backward(self, grad):
            seen.append(K.PERSIST_FRONT_BWD[0])

    class _Cut(object):
        grad = torch.zeros(1)

    keep = dict(x=_X(), x_cut=_Cut())
    train.g_backward_late(keep)
    train.g_backward_late(keep, persist=True)
    assert seen == [False, True] and K.PERSIST_FRONT_BWD[0] is True


def test_gru_generator_reports_its_persistent_front(monkeypatch):
    """GRUGenerator's frame loop is a persistent launch too, so gd_step / GraphedStep must not fork its
    forward onto a second stream beside the critic's persistent biLSTM (front_is_persistent used to test the exact type)"""
    import audiogan_amd as A
    from audiogan_amd import kernels as K
    g = A.GRUGenerator(frame_size=32, embed_size=8, noise_size=8, state_size=64, struct=[[17, 8, 16, 8]])
    g2 = A.Generator(frame_size=32, embed_size=8, noise_size=8, state_size=64, num_layers=2, struct=[[17, 8, 16, 8]])
    monkeypatch.setattr(K, 'gfront_persist_ok', lambda B, S, fs, dev: True)
    assert g.front_is_persistent(4, torch.device('cpu')) is True
    assert g2.front_is_persistent(4, torch.device('cpu')) is False       # stacked LSTM cells take the per-frame path
    monkeypatch.setattr(K, 'gfront_persist_ok', lambda B, S, fs, dev: False)
    assert g.front_is_persistent(4, torch.device('cpu')) is False
