"""feature-matching penalty and FGSM-style perturbations on the HIP path
vs the oracle (CPU with the kernel model; -m gpu on the real kernels)."""
import numpy as np
import pytest
import torch

from oracle import audiogan_oracle as O
from tests import kernel_model


def _run(dev, A):
    from audiogan_amd import extras as X
    torch.manual_seed(51)
    gcfg = dict(frame_size=32, embed_size=8, noise_size=8, state_size=64, num_layers=1,
                struct=[[17, 8, 16, 8], [9, 4, 16, 8]])
    dcfg = dict(state_size=64, embed_size=8, num_layers=1, cnn_struct=[[7, 2, 8], [7, 2, 16]])
    go, do = O.Generator(**gcfg), O.Discriminator(**dcfg)
    g, d = A.Generator(**gcfg), A.Discriminator(**dcfg)
    g.load_state_dict(go.state_dict()); d.load_state_dict(do.state_dict())
    g.to(dev); d.to(dev)
    B, T = 4, 4
    real, rl = torch.randn(B, 128), torch.tensor([128, 96, 128, 64])
    z, c = torch.randn(B, T, 8), torch.randn(B, 8)
    stop = torch.zeros(B, T, dtype=torch.long)
    # ---- feature penalty through D's activations into G (audiogan.py:845-855)
    fo = go(z=z, c=c, stop=stop)[0]
    f = g(z=z.to(dev), c=c.to(dev), stop='never')[0]
    _, hso, hlo, _ = do(fo, torch.full((B,), 128), c)
    _, hs, hl, _ = d(f, torch.full((B,), 128).to(dev), c.to(dev))
    _, hsro, hlro, _ = do(real, rl, c)
    _, hsr, hlr, _ = d(real.to(dev), rl.to(dev), c.to(dev))
    po = O.feature_penalty(O.calc_dists(hsro, hlro), O.calc_dists(hso, hlo), B)
    p = X.feature_penalty(X.calc_dists(hsr, hlr), X.calc_dists(hs, hl), B)
    np.testing.assert_allclose(float(p), float(po), rtol=1e-3)
    # the one-reduction form train.g_step_full uses: the same penalty and the same gradients into the activations
    gr = torch.autograd.grad(p, list(hs), retain_graph=True)
    pf = X.feature_penalty_fused(hsr, hlr, hs, hl, B)
    np.testing.assert_allclose(float(pf), float(p), rtol=1e-5)
    for a_, b_ in zip(torch.autograd.grad(pf, list(hs), retain_graph=True), gr):
        np.testing.assert_allclose(a_.cpu().numpy(), b_.cpu().numpy(), rtol=1e-4, atol=1e-6 * float(b_.abs().max()))
    po.backward(); p.backward()
    for (k, q), (_, qo) in zip(g.named_parameters(), go.named_parameters()):
        if k.split('.')[-1].startswith('bias') and k.endswith('_v'):
            continue
        ref = qo.grad.numpy() if qo.grad is not None else np.zeros(tuple(qo.shape), np.float32)
        got = q.grad.cpu().numpy() if q.grad is not None else np.zeros_like(ref)
        np.testing.assert_allclose(got, ref, rtol=2e-3, atol=2e-5 * max(1.0, float(np.abs(ref).max())), err_msg=k)
    # ---- FGSM movement of D's input (audiogan.py:139-150): signs must agree wherever |grad| is not noise
    xr = real.clone().requires_grad_(True)
    clso, _, _, nfo = do(xr, rl, c)
    lo = O.binary_cross_entropy_with_logits_per_sample(clso, torch.full_like(clso, 0.9),
                                                       weight=O.length_mask(clso.size(), nfo)) / nfo.float()
    go_, = torch.autograd.grad(lo.sum(), xr)
    adv = X.adversarial_movement_d(real.to(dev), rl.to(dev), c.to(dev), 0.9, None, d, scale=1e-3).cpu()
    strong = go_.abs() > 1e-7
    assert strong.float().mean() > 0.5
    np.testing.assert_array_equal(torch.sign(adv)[strong].numpy(), torch.sign(go_)[strong].numpy())
    assert float(adv.abs().max()) == pytest.approx(1e-3)
    assert not adv[1, 112:].any()         # past the clip's length (+ receptive field) the input is masked out
    # ---- adversarial z (audiogan.py:99-137); input-gradient passes must leave every parameter's .grad untouched
    before = [(q, None if q.grad is None else q.grad.clone()) for q in list(g.parameters()) + list(d.parameters())]
    z2 = X.adversarially_sample_z(g, d, B, T, 8, 128, c.to(dev), 0.01, c.to(dev), z=z.to(dev),
                                  noise=torch.zeros(B, 128).to(dev), stop='never')
    assert z2.shape == z.shape and float((z2.cpu() - z).abs().max()) <= 1e-2 + 1e-7
    for q, b in before:
        assert q.requires_grad and ((q.grad is None) if b is None else torch.equal(q.grad, b))


def test_extras_host_logic(monkeypatch):
    kernel_model.install(monkeypatch)
    import audiogan_amd as A
    _run(torch.device('cpu'), A)


@pytest.mark.gpu
def test_extras_gpu():
    import audiogan_amd as A
    _run(torch.device('cuda'), A)
