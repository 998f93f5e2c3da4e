"""train.Feeder: the reference loader's minibatches into the static tensors a GraphedStep replays on
(audiogan.py:94-97, :714-716, :823-828)."""
import types

import numpy as np
import pytest
import torch

from audiogan_amd import dataset as D


def _loader(B, frame):
    words = ['alpha', 'beta', 'gamma', 'delta', 'epsil', 'zetaa', 'etaaa', 'theta', 'iotaa', 'kappa', 'lambd']
    ds = D.SyntheticWordDataset(words, n_per_word=3, min_len=frame + 1, max_len=4 * frame, kind='noise', seed=3)
    args = types.SimpleNamespace(conditional=True, dataset=ds, minwordlen=1, subset=None, amplitudes=0)
    np.random.seed(5)
    _, maxlen, gen_train, _, keys_train, _ = D.dataloader(B, args, maxlen=4 * frame, frame_size=frame)
    return gen_train, maxlen


def test_feeder_order_and_checks_cpu():
    from audiogan_amd import train
    B, frame = 3, 16
    gen, maxlen = _loader(B, frame)
    items = [next(gen) for _ in range(4)]
    host = [train.batch_from_loader(it) for it in items]
    assert host[0]['real'].dtype == np.float32 and host[0]['real_len'].dtype == np.int64
    static = dict(real=torch.zeros(B, maxlen), real_len=torch.zeros(B, dtype=torch.long),
                  noise=torch.zeros(B, maxlen))
    calls = []
    f = train.Feeder(static, keys=['real', 'real_len'], device_fill={'noise': lambda t: calls.append(1) or t.fill_(len(calls))})
    seen = []

    class _G(object):
        def step(self):
            seen.append((static['real'].clone(), static['real_len'].clone(), float(static['noise'][0, 0])))
            return len(seen)
    n = f.run(_G(), host)
    assert n == 4 and len(seen) == 4
    for i, (r, l, nz) in enumerate(seen):
        np.testing.assert_array_equal(r.numpy(), host[i]['real'])
        np.testing.assert_array_equal(l.numpy(), host[i]['real_len'])
        assert nz == i + 1                                   # the device-side refresh ran once per step, before it
    with pytest.raises(ValueError):
        f.stage(dict(real=np.zeros((B, maxlen + 1), np.float32), real_len=host[0]['real_len']))
    with pytest.raises(TypeError):
        f.stage(dict(real=np.zeros((B, maxlen), np.float64), real_len=host[0]['real_len']))
    f.stage(host[0]); f.stage(host[1])
    with pytest.raises(RuntimeError):
        f.stage(host[2])                                     # only one batch may be staged ahead
    with pytest.raises(KeyError):
        train.Feeder(static, keys=['nope'])


@pytest.mark.gpu
def test_fed_graph_steps_equal_eager_steps_on_the_same_data():
    """two minibatches from the loader fed through a captured GraphedStep == two eager gd_steps on the same data"""
    import audiogan_amd as A
    from audiogan_amd import optim, train
    B, frame, T = 4, 32, 4
    gcfg = dict(frame_size=frame, embed_size=8, noise_size=8, state_size=64, num_layers=1,
                struct=[[17, 8, 16, 8], [9, 4, 16, 8]])
    dcfg = dict(state_size=64, embed_size=8, num_layers=1, cnn_struct=[[7, 2, 8], [7, 2, 16]])
    gen, maxlen = _loader(B, frame)
    assert maxlen == T * frame
    host = [train.batch_from_loader(next(gen)) for _ in range(3)]
    gen_t = torch.Generator().manual_seed(9)
    fixed = dict(c=torch.randn(B, 8, generator=gen_t), z=torch.randn(B, T, 8, generator=gen_t),
                 noise_real=torch.randn(B, T * frame, generator=gen_t) * 0.01,
                 noise_fake=torch.randn(B, T * frame, generator=gen_t) * 0.01)

    def models():
        torch.manual_seed(4)
        g, d = A.Generator(**gcfg).cuda(), A.Discriminator(**dcfg).cuda()
        return g, d, optim.make_optimizer(list(g.parameters()), 'adam', 1e-4), optim.make_optimizer(list(d.parameters()), 'adam', 1e-4)

    def dev_batch(h):
        b = {k: v.cuda() for k, v in fixed.items()}
        b['real'], b['real_len'] = torch.from_numpy(h['real']).cuda(), torch.from_numpy(h['real_len']).cuda()
        return b
    # eager: warm-up step on batch 0, then steps on batches 1 and 2
    g, d, og, od = models()
    eager = []
    for i, h in enumerate(host):
        b = dev_batch(h)
        l = train.gd_step(g, d, og, od, b['real'], b['real_len'], b['c'], b['z'], b['noise_real'], b['noise_fake'], overlap=False)
        if i > 0:
            eager.append((float(l[0]), float(l[1])))
    want = [p.detach().clone() for p in list(g.parameters()) + list(d.parameters())]
    # graph: the same warm-up step eagerly, capture, then batches 1 and 2 through the Feeder
    g, d, og, od = models()
    b = dev_batch(host[0])
    train.gd_step(g, d, og, od, b['real'], b['real_len'], b['c'], b['z'], b['noise_real'], b['noise_fake'], overlap=False)
    gs = train.GraphedStep(g, d, og, od, b, overlap=False)
    f = train.Feeder(gs.b, keys=['real', 'real_len'])
    got = []
    n = f.run(gs, host[1:], on_step=lambda i, out: got.append((float(out[0]), float(out[1]))))
    torch.cuda.synchronize()
    assert n == 2 and got == eager, (got, eager)
    for p, q in zip(list(g.parameters()) + list(d.parameters()), want):
        assert torch.equal(p.detach(), q)
    gs.check()
