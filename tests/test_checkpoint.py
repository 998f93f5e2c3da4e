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
