"""C1 tiny conv G/D and the WGAN-GP critic (hand-written double backward) vs the oracle.
Runs on CPU with the kernel model (host logic) and, marked gpu, on the real kernels."""
import numpy as np
import pytest
import torch

from oracle import audiogan_oracle as O
from tests import kernel_model


def _run(dev, A):
    torch.manual_seed(31)
    # ---- C1: tiny conv generator / discriminator, B=4, L=1024 sine clips
    go, do = O.Conv1DGenerator(), O.Conv1DDiscriminator()
    g, d = A.Conv1DGenerator(), A.Conv1DDiscriminator()
    g.load_state_dict(go.state_dict()); d.load_state_dict(do.state_dict())
    g.to(dev); d.to(dev)
    z = torch.randn(4, 1024 // go.multiplier)
    real = torch.from_numpy(O.synthetic_clips(4, 1024, 'sine')).float()
    xo = go(z=z)
    x = g(z=z.to(dev))
    assert x.shape == (4, 1024)
    np.testing.assert_allclose(x.detach().cpu().numpy(), xo.detach().numpy(), rtol=1e-3, atol=1e-5)
    so, s = do(torch.cat([xo, real])), d(torch.cat([x, real.to(dev)]))
    np.testing.assert_allclose(s.detach().cpu().numpy(), so.detach().numpy(), rtol=1e-3, atol=1e-5)
    w = torch.randn(8)
    (so * w).sum().backward(); (s * w.to(dev)).sum().backward()
    for a, b in ((g, go), (d, do)):
        for (k, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
            np.testing.assert_allclose(p.grad.cpu().numpy(), q.grad.numpy(), rtol=1e-3,
                                       atol=1e-5 * max(1.0, float(q.grad.abs().max())), err_msg=k)
    # ---- WGAN-GP on the conv critic (C5), small stack
    struct = [[7, 2, 4], [7, 2, 8], [5, 2, 8]]
    co = O.Conv1DDiscriminator(config=[(c, k, s) for k, s, c in struct])
    cr = A.ConvPoolCritic(cnn_struct=struct)
    cr.load_state_dict(co.state_dict()); cr.to(dev)
    xr, xf, eps = torch.randn(5, 256), torch.randn(5, 256), torch.rand(5, 1)
    lo = O.wgan_gp_d_loss(lambda t: co(t), xr, xf, eps, lam=10.0)
    l = A.wgan_gp_d_loss(cr, xr.to(dev), xf.to(dev), eps.to(dev), lam=10.0)
    np.testing.assert_allclose(float(l), float(lo), rtol=1e-3)
    lo.backward(); l.backward()
    for (k, p), (_, q) in zip(cr.named_parameters(), co.named_parameters()):
        np.testing.assert_allclose(p.grad.cpu().numpy(), q.grad.numpy(), rtol=2e-3,
                                   atol=2e-5 * max(1.0, float(q.grad.abs().max())), err_msg=k)
    np.testing.assert_allclose(float(A.wgan_g_loss(cr, xf.to(dev))), float(O.wgan_g_loss(lambda t: co(t), xf)),
                               rtol=1e-3, atol=1e-6)


def test_convnets_host_logic(monkeypatch):
    kernel_model.install(monkeypatch)
    import audiogan_amd as A
    _run(torch.device('cpu'), A)


@pytest.mark.gpu
def test_convnets_gpu():
    import audiogan_amd as A
    _run(torch.device('cuda'), A)


def _run_gru(dev, A):
    """config C4: GRU generator (vs torch.nn.GRUCell via the oracle) + conv critic"""
    torch.manual_seed(41)
    cfg = dict(frame_size=32, embed_size=8, noise_size=8, state_size=64, struct=[[17, 8, 16, 8], [9, 4, 16, 8]])
    go, g = O.GRUGenerator(**cfg), A.GRUGenerator(**cfg)
    assert list(g.state_dict().keys()) == list(go.state_dict().keys())
    g.load_state_dict(go.state_dict()); g.to(dev)
    z, c = torch.randn(5, 4, 8), torch.randn(5, 8)
    xo, so, _, _ = go(z=z, c=c)
    x, s, _, ln = g(z=z.to(dev), c=c.to(dev), stop='never')
    np.testing.assert_allclose(x.detach().cpu().numpy(), xo.detach().numpy(), rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(s.detach().cpu().numpy(), so.detach().numpy(), rtol=1e-3, atol=1e-5)
    w, w2 = torch.randn(xo.shape), torch.randn(so.shape)
    ((xo * w).sum() + (so * w2).sum()).backward()
    ((x * w.to(dev)).sum() + (s * w2.to(dev)).sum()).backward()
    for (k, p), (_, q) in zip(g.named_parameters(), go.named_parameters()):
        if k.split('.')[-1].startswith('bias') and k.endswith('_v'):
            continue
        np.testing.assert_allclose(p.grad.cpu().numpy(), q.grad.numpy(), rtol=2e-3,
                                   atol=2e-5 * max(1.0, float(q.grad.abs().max())), err_msg=k)


def test_gru_generator_host_logic(monkeypatch):
    kernel_model.install(monkeypatch)
    import audiogan_amd as A
    _run_gru(torch.device('cpu'), A)


@pytest.mark.gpu
def test_gru_generator_gpu():
    import audiogan_amd as A
    _run_gru(torch.device('cuda'), A)


@pytest.mark.gpu
@pytest.mark.parametrize('B', [4, 64])
def test_c4_c5_at_c2_widths_gpu(B):
    """BASELINE configs[3] / [4] at the C2 widths (state 1024, frame 256, default structs, 8192-sample clips), at a small
    batch and at their BASELINE per-GPU batch of 64: one GRU-front + conv-critic BCE step (train.c4_step) and one WGAN-GP
    step (train.wgan_gp_step) vs the oracle's statements; losses, every per-parameter gradient norm of both networks (as
    the fused optimiser measured them), and the generated waveforms after the step"""
    import os
    import audiogan_amd as A
    from audiogan_amd import optim, train
    dev = torch.device('cuda')
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    cfg = [(16, 7, 2), (32, 7, 2), (64, 7, 2), (128, 7, 2), (256, 7, 2), (512, 7, 2)]
    T, fs = 32, 256
    gen = torch.Generator().manual_seed(52)
    real = torch.rand(B, T * fs, generator=gen) * 2 - 1
    c, z = torch.randn(B, 100, generator=gen), torch.randn(B, T, 100, generator=gen)
    nr, nf = torch.randn(B, T * fs, generator=gen) * 0.01, torch.randn(B, T * fs, generator=gen) * 0.01
    eps = torch.rand(B, 1, generator=gen)
    stop = torch.zeros(B, T, dtype=torch.long)
    to = lambda t: t.to(dev)  # noqa: E731
    for wl in ('c4', 'c5'):
        torch.manual_seed(53)
        go = (O.GRUGenerator if wl == 'c4' else O.Generator)(frame_size=fs, embed_size=100, noise_size=100, state_size=1024)
        co = O.Conv1DDiscriminator(config=cfg)
        g = (A.GRUGenerator if wl == 'c4' else A.Generator)(frame_size=fs, embed_size=100, noise_size=100, state_size=1024)
        cr = A.ConvPoolCritic()
        g.load_state_dict(go.state_dict()); cr.load_state_dict(co.state_dict())
        g.to(dev); cr.to(dev)
        ogo, odo = O.make_optimizer(list(go.parameters()), 'adam', 1e-4), O.make_optimizer(list(co.parameters()), 'adam', 1e-4)
        og, od = optim.make_optimizer(list(g.parameters()), 'adam', 1e-4), optim.make_optimizer(list(cr.parameters()), 'adam', 1e-4)
        # the oracle's per-parameter gradient norms: where it clips (c4: before clip_grad rescales them in place), else
        # (c5: no clipping) as its optimisers see them when they step
        rec = {}
        orig = O.clip_grad

        def record(params, clip_norm, rec=rec, orig=orig):
            rec[id(params[0])] = [float(p.grad.norm()) if p.grad is not None else None for p in params]
            return orig(params, clip_norm)

        def wrap_step(opt, rec=rec):
            step = opt.step
            params = [p for gr in opt.param_groups for p in gr['params']]

            def stepped(*a, **k):
                rec.setdefault(id(params[0]), [float(p.grad.norm()) if p.grad is not None else None for p in params])
                return step(*a, **k)
            opt.step = stepped
        wrap_step(ogo); wrap_step(odo)
        O.clip_grad = record
        try:
            if wl == 'c4':
                lo = O.c4_step(go, co, ogo, odo, real, c, z, nr, nf, 1.0, 0.1, stop=stop)
            else:
                lo = O.wgan_gp_step(go, co, ogo, odo, real, c, z, eps, 10.0, stop=stop)
        finally:
            O.clip_grad = orig
        rec = [rec[id(next(co.parameters()))], rec[id(next(go.parameters()))]]
        if wl == 'c4':
            l = train.c4_step(g, cr, og, od, to(real), to(c), to(z), to(nr), to(nf), 1.0, 0.1, check=True)
        else:
            l = train.wgan_gp_step(g, cr, og, od, to(real), to(c), to(z), to(eps), 10.0, check=True)
        np.testing.assert_allclose([float(l[0]), float(l[1])], [float(lo[0]), float(lo[1])], rtol=1e-3, err_msg=wl)
        # per-parameter gradient norms of the critic and the generator iteration (oracle: recorded where it clips)
        for mod, opt, ref_norms in ((cr, od, rec[0]), (g, og, rec[1])):
            live = [k for k, q in mod.named_parameters() if q.grad is not None]
            names = [k for k, _ in mod.named_parameters()]
            got = opt._state['norms'].cpu()
            for i, k in enumerate(live):
                r_ = ref_norms[names.index(k)]
                if (k.split('.')[-1].startswith('bias') and k.endswith('_v')) or r_ is None:
                    continue
                np.testing.assert_allclose(float(got[i]), r_, rtol=2e-3, atol=1e-7, err_msg='%s B=%d grad norm %s' % (wl, B, k))
        xo = go(z=z, c=c, stop=stop)[0]
        x = g(z=to(z), c=to(c), stop='never')[0]
        np.testing.assert_allclose(x.detach().cpu().numpy(), xo.detach().numpy(), rtol=2e-3,
                                   atol=2e-3 * float(xo.abs().max()), err_msg=wl + ' waveform after the step')
