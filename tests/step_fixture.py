"""Shared driver for ``tests/golden/ref_step.npz`` (written by oracle/pin_reference.py::pin_step from the reference's own
training-loop bodies, audiogan.py:711-788 and :822-921): feeds the stored inputs / draws to a (d_step_full, g_step_full)
pair - the oracle's on CPU, audiogan_amd.train's on the kernel model or on the GPU - and compares every stored result."""
import os

import numpy as np
import torch


def load(golden_dir):
    return dict(np.load(os.path.join(golden_dir, 'ref_step.npz')))


GCFG = dict(frame_size=32, embed_size=8, noise_size=8, state_size=64, num_layers=1, struct=[[17, 8, 16, 8], [9, 4, 16, 8]])
DCFG = dict(state_size=64, embed_size=8, num_layers=1, cnn_struct=[[7, 2, 8], [7, 2, 16]])


def build(M, v, dev):
    """models of module namespace ``M`` (oracle or audiogan_amd) holding the fixture's initial weights"""
    g, d = M.Generator(**GCFG), M.Discriminator(**DCFG)
    e_g, e_d = M.Embedder(8, 6, num_chars=128), M.Embedder(8, 6, num_chars=128)
    for tag, m in (('g', g), ('d', d), ('eg', e_g), ('ed', e_d)):
        pre = 'init.%s.' % tag
        m.load_state_dict({k[len(pre):]: torch.from_numpy(a) for k, a in v.items() if k.startswith(pre)}, strict=True)
        m.to(dev)
    return g, d, e_g, e_d


def _close(got, ref, rtol, atol_scale, msg):
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
    ref = np.asarray(ref)
    np.testing.assert_allclose(got, ref, rtol=rtol, atol=atol_scale * max(1e-3, float(np.abs(ref).max())), err_msg=msg)


def _post(mods, v, pre, rtol, atol_scale):
    for tag, m in mods:
        for k, p in m.state_dict().items():
            if k.split('.')[-1].startswith('bias') and k.endswith('_v'):
                # analytically zero gradient (bias = g*sign(v)): both sides feed rounding noise to RMSprop, which
                # normalises it to full-size steps; only the sign of v matters for the network
                np.testing.assert_array_equal(np.sign(p.cpu().numpy()), np.sign(v[pre + 'post.%s.' % tag + k]), err_msg=k)
                continue
            _close(p, v[pre + 'post.%s.' % tag + k], rtol, atol_scale, pre + tag + '.' + k)


def run(v, mods, opt_d, opt_g, d_step_full, g_step_full, dev, rtol=1e-4, atol_scale=1e-5, post_atol=2e-4):
    """two critic iterations + one generator iteration, compared with the reference's results after each"""
    g, d, e_g, e_d = mods
    scale = float(v['cfg'][3]) * 1e-6
    to = lambda a, dt=None: (torch.from_numpy(np.asarray(a)) if dt is None else torch.from_numpy(np.asarray(a)).to(dt)).to(dev)  # noqa: E731
    f32 = lambda a: torch.from_numpy(np.asarray(a).astype('float32')).to(dev)  # noqa: E731
    for it in (1, 2):
        pre = 'd%d.' % it
        real = f32(v[pre + 'real'])                                   # tovar(real_data): float64 -> float32
        B, L = real.shape
        if it % 2 == 0:
            nr = f32(v[pre + 'noise_real_raw'] * scale)               # tovar(RNG.randn(...) * noisescale)
            nf = to(v[pre + 'noise_fake_raw']) * scale                # T.randn(...) * noisescale
        else:
            nr, nf = torch.zeros(B, L, device=dev), torch.zeros(B, L, device=dev)
        r = d_step_full(g, d, e_g, e_d, opt_d, it, real, to(v[pre + 'real_len'], torch.long), to(v[pre + 'cs'], torch.long),
                        to(v[pre + 'cl'], torch.long), to(v[pre + 'cs2'], torch.long), to(v[pre + 'cl2'], torch.long),
                        to(v[pre + 'z']), nr, nf, 1.0, stop=to(v[pre + 'stop']))
        for k in ('loss', 'loss_d', 'loss_g', 'cls_d', 'cls_g'):
            _close(r[k], v[pre + k], rtol, atol_scale, pre + k)
        np.testing.assert_allclose([r['acc_d'], r['acc_g']], v[pre + 'acc'], atol=1e-6)
        np.testing.assert_allclose(float(r['grad_norm']), float(v[pre + 'grad_norm']), rtol=max(rtol, 1e-4))
        _post((('d', d), ('ed', e_d)), v, pre, rtol, post_atol)
    pre = 'g1.'
    r = g_step_full(g, d, e_g, e_d, opt_g, f32(v[pre + 'real']), to(v[pre + 'real_len'], torch.long),
                    to(v[pre + 'cs'], torch.long), to(v[pre + 'cl'], torch.long), to(v[pre + 'z0']),
                    f32(v[pre + 'noise_real_raw'] * scale), to(v[pre + 'noise_adv_raw']) * scale,
                    to(v[pre + 'noise_fake_raw']) * scale, to(v[pre + 'stop_adv']), to(v[pre + 'stop']), None)
    # the adversarial z: z0 +- 1e-2 along the sign of d(loss)/dz (zero where |grad| <= 1e-9)
    dz, dzo = (r['z'].cpu() - torch.from_numpy(v[pre + 'z0'])), torch.from_numpy(v[pre + 'z'] - v[pre + 'z0'])
    agree = float((torch.sign(dz) == torch.sign(dzo)).float().mean())
    assert agree > 0.99, agree
    np.testing.assert_array_equal(r['fake_len'].cpu().numpy(), v[pre + 'fake_len'])
    loose = max(rtol, 2e-3) if agree < 1.0 else rtol                  # a flipped sign of z moves everything downstream
    for k in ('loss', 'bce', 'feature_penalty', 's', 'fake'):
        _close(r[k], v[pre + k], loose, max(atol_scale, 1e-4 if agree < 1.0 else 0), pre + k)
    assert abs(r['baseline'] - float(v[pre + 'baseline'])) <= loose * abs(float(v[pre + 'baseline'])) + 1e-7
    np.testing.assert_allclose(float(r['grad_norm']), float(v[pre + 'grad_norm']), rtol=max(loose, 1e-4))
    _post((('g', g), ('eg', e_g)), v, pre, loose, post_atol)
    return agree
