import os

import numpy as np
import torch

from oracle import audiogan_oracle as O
from tests import kernel_model


def test_checkpoint_roundtrip_and_reference_key_compat(tmp_path, monkeypatch):
    kernel_model.install(monkeypatch)
    import audiogan_amd as A
    from audiogan_amd import checkpoint, optim
    torch.manual_seed(0)
    g = A.Generator(16, 6, 5, 24, 1, struct=[[9, 4, 8, 4]])
    d = A.Discriminator(16, 6, 1, cnn_struct=[[7, 2, 4], [5, 2, 8]])
    e = A.Embedder(6, 4, 1, 32)
    og = optim.Adam(list(g.parameters()), lr=1e-3)
    for p in g.parameters():
        p.grad = torch.randn_like(p) * 1e-3
    og.step(clip_norm=0.1)
    prefix = os.path.join(tmp_path, 'model')
    files = checkpoint.save(prefix, 500, d=d, g=g, e_g=e, opt_g=og, extra={'gen_iter': 500})
    assert sorted(os.path.basename(f) for f in files) == ['model-dis-00500', 'model-eg-00500', 'model-gen-00500',
                                                          'model-opt-00500']
    g2 = A.Generator(16, 6, 5, 24, 1, struct=[[9, 4, 8, 4]])
    d2 = A.Discriminator(16, 6, 1, cnn_struct=[[7, 2, 4], [5, 2, 8]])
    og2 = optim.Adam(list(g2.parameters()), lr=1e-3)
    extra = checkpoint.load(prefix, 500, d=d2, g=g2, opt_g=og2)
    assert extra == {'gen_iter': 500} and og2.step_count == 1
    for (k, a), (_, b) in zip(g.state_dict().items(), g2.state_dict().items()):
        np.testing.assert_array_equal(a.numpy(), b.numpy(), err_msg=k)
    for a, b in zip(og._state['s1'], og2._state['s1']):
        np.testing.assert_array_equal(a.numpy(), b.numpy())
    # the same files load into the reference-shaped (oracle) modules: identical key names
    go = O.Generator(16, 6, 5, 24, 1, struct=[[9, 4, 8, 4]])
    go.load_state_dict(torch.load(prefix + '-gen-00500'), strict=True)
    # a reference-style checkpoint (whole-module pickle, audiogan.py:936-939) loads too
    torch.save(go, prefix + '-gen-00777')
    g3 = A.Generator(16, 6, 5, 24, 1, struct=[[9, 4, 8, 4]])
    import pytest
    with pytest.raises(RuntimeError):
        checkpoint.load(prefix, 777, g=g3)            # whole-module pickles execute code: refused unless asked for
    checkpoint.load(prefix, 777, g=g3, allow_pickle=True)
    for (k, a), (_, b) in zip(g.state_dict().items(), g3.state_dict().items()):
        np.testing.assert_array_equal(a.numpy(), b.numpy(), err_msg=k)
    # the RNG stream is restored: the draw after load() repeats the draw after save()
    checkpoint.save(prefix, 900, g=g, opt_g=og)
    a = torch.randn(5)
    torch.randn(100)
    checkpoint.load(prefix, 900, g=g2, opt_g=og2)
    np.testing.assert_array_equal(torch.randn(5).numpy(), a.numpy())
    assert checkpoint.load.last_rng == 'restored'
    # inference-only load (no optimiser): the RNG streams are left alone
    torch.manual_seed(3)
    a = torch.randn(3)
    torch.manual_seed(3)
    checkpoint.load(prefix, 900, g=g2)
    assert checkpoint.load.last_rng == 'not restored'
    np.testing.assert_array_equal(torch.randn(3).numpy(), a.numpy())


def test_checkpoint_io_errors_are_not_pickle_advice_and_extra_is_validated(tmp_path, monkeypatch):
    """a missing / truncated file must surface as what it is (not as 'pass allow_pickle=True'), and an
    ``extra`` that would make the package's own opt file unreadable by the default load is converted or refused at save"""
    import pytest
    kernel_model.install(monkeypatch)
    import audiogan_amd as A
    from audiogan_amd import checkpoint, optim
    g = A.Generator(16, 6, 5, 24, 1, struct=[[9, 4, 8, 4]])
    prefix = os.path.join(tmp_path, 'm')
    with pytest.raises(FileNotFoundError):
        checkpoint.load(prefix, 1, g=g)
    og = optim.Adam(list(g.parameters()), lr=1e-3)
    checkpoint.save(prefix, 2, g=g, opt_g=og, extra={'it': np.int64(7), 'acc': np.float32(0.5), 'hist': np.arange(3)})
    extra = checkpoint.load(prefix, 2, g=g, opt_g=og)
    assert extra['it'] == 7 and isinstance(extra['it'], int) and abs(extra['acc'] - 0.5) < 1e-7
    np.testing.assert_array_equal(extra['hist'].numpy(), np.arange(3))
    with pytest.raises(TypeError):
        checkpoint.save(prefix, 3, g=g, extra={'f': open})


import pytest  # noqa: E402


@pytest.mark.gpu
def test_resume_from_checkpoint_equals_uninterrupted_run_gpu(tmp_path):
    """audiogan.py:696-701, :936-939 on the device: 2 graph-replayed steps, ``checkpoint.save`` mid-run, FRESH modules and
    optimisers (different init, own captured graph), ``checkpoint.load``, 2 more replayed steps == 4 uninterrupted steps bit
    for bit - parameters, both optimisers' state incl. Adam's device-side step counter, the weight-norm materialisation
    caches after ``load_state_dict`` (the graph is captured BEFORE the load) and the CUDA RNG stream (z and the instance
    noise are drawn on the device before every step)"""
    import audiogan_amd as A
    from audiogan_amd import checkpoint, optim, train
    B, frame, T = 4, 32, 4
    gcfg = dict(frame_size=frame, embed_size=8, noise_size=8, state_size=64, num_layers=1,
                struct=[[17, 8, 16, 8], [9, 4, 16, 8]])
    dcfg = dict(state_size=64, embed_size=8, num_layers=1, cnn_struct=[[7, 2, 8], [7, 2, 16]])
    L = T * frame
    gen = torch.Generator().manual_seed(3)
    real = torch.rand(B, L, generator=gen) * 2 - 1
    c = torch.randn(B, 8, generator=gen)

    def fresh(seed):
        torch.manual_seed(seed)
        g, d = A.Generator(**gcfg).cuda(), A.Discriminator(**dcfg).cuda()
        og, od = optim.make_optimizer(list(g.parameters()), 'adam', 1e-4), optim.make_optimizer(list(d.parameters()), 'adam', 1e-4)
        b = dict(real=real.cuda(), real_len=torch.tensor([L, L - 40, L, L // 2]).cuda(), c=c.cuda(),
                 z=torch.zeros(B, T, 8).cuda(), noise_real=torch.zeros(B, L).cuda(), noise_fake=torch.zeros(B, L).cuda())
        # lazily created resources (optimiser state, workspaces) must exist before capture: one eager step, then put the
        # initial parameters / optimiser state back so that the captured run starts from the seeded init
        p0 = [p.detach().clone() for p in list(g.parameters()) + list(d.parameters())]
        train.gd_step(g, d, og, od, b['real'], b['real_len'], b['c'], b['z'], b['noise_real'], b['noise_fake'], overlap=False)
        with torch.no_grad():
            for p, q in zip(list(g.parameters()) + list(d.parameters()), p0):
                p.copy_(q)
        for o in (og, od):
            sd = o.state_dict()
            sd['step'] = 0
            sd['s1'] = [torch.zeros_like(t) for t in sd['s1']]
            sd['s2'] = [torch.zeros_like(t) for t in sd['s2']]
            o.load_state_dict(sd)
        gs = train.GraphedStep(g, d, og, od, b, overlap=False)
        return g, d, og, od, b, gs

    def draw(b):
        b['z'].normal_()
        b['noise_real'].normal_().mul_(0.01)
        b['noise_fake'].normal_().mul_(0.01)

    def run(gs, b, n):
        out = []
        for _ in range(n):
            draw(b)
            l = gs.step()
            out.append((float(l[0]), float(l[1])))
        return out

    # ---- uninterrupted: 4 steps
    g, d, og, od, b, gs = fresh(4)
    torch.cuda.manual_seed(11)
    want_losses = run(gs, b, 4)
    want = [p.detach().clone() for p in list(g.parameters()) + list(d.parameters())]
    want_state = [t.clone() for o in (og, od) for t in o.state_dict()['s1'] + o.state_dict()['s2']]
    # ---- interrupted: 2 steps, save, fresh everything, load, 2 steps
    g, d, og, od, b, gs = fresh(4)
    torch.cuda.manual_seed(11)
    got = run(gs, b, 2)
    prefix = os.path.join(tmp_path, 'run')
    checkpoint.save(prefix, 2, d=d, g=g, opt_d=od, opt_g=og, extra={'dis_iter': 2})
    del g, d, og, od, gs
    g2, d2, og2, od2, b2, gs2 = fresh(99)                  # other weights, its own graph, captured BEFORE the load
    assert not torch.equal(next(g2.parameters()).detach(), want[0])
    extra = checkpoint.load(prefix, 2, d=d2, g=g2, opt_d=od2, opt_g=og2)
    assert extra == {'dis_iter': 2} and checkpoint.load.last_rng == 'restored'
    assert og2.step_count == 2 and od2.step_count == 2
    got += run(gs2, b2, 2)
    torch.cuda.synchronize()
    assert got == want_losses, (got, want_losses)
    for p, q in zip(list(g2.parameters()) + list(d2.parameters()), want):
        assert torch.equal(p.detach(), q)
    for t, q in zip([t for o in (og2, od2) for t in o.state_dict()['s1'] + o.state_dict()['s2']], want_state):
        assert torch.equal(t, q)
    gs2.check()
