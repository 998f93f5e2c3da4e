"""audiogan_amd.loop.TrainLoop: the reference's outer loop (audiogan.py:703-940: critic_iter with the accuracy early break,
gencatchup, checkpoints every N generator iterations) over the loader interface.  CPU: host logic on the kernel model;
-m gpu: at the C2 widths on the HIP kernels."""
import os
import types

import numpy as np
import pytest
import torch

from audiogan_amd import dataset as D
from tests import kernel_model


def _setup(A, dev, wide, tmp_path, B):
    from audiogan_amd import loop, optim
    torch.manual_seed(81)
    if wide:
        frame, maxlen = 256, 8192
        gcfg = dict(frame_size=frame, embed_size=100, noise_size=100, state_size=1024, num_layers=1)
        dcfg = dict(state_size=1024, embed_size=100, num_layers=1)
        ecfg = dict(output_size=100, char_embed_size=50, num_layers=1, num_chars=256)
    else:
        frame, maxlen = 32, 128
        gcfg = dict(frame_size=frame, embed_size=8, noise_size=8, state_size=64, num_layers=1, struct=[[17, 8, 16, 8], [9, 4, 16, 8]])
        dcfg = dict(state_size=64, embed_size=8, num_layers=1, cnn_struct=[[7, 2, 8], [7, 2, 16]])
        ecfg = dict(output_size=8, char_embed_size=6, num_chars=256)
    g, d = A.Generator(**gcfg).to(dev), A.Discriminator(**dcfg).to(dev)
    e_g, e_d = A.Embedder(**ecfg).to(dev), A.Embedder(**ecfg).to(dev)
    opt_g = optim.make_optimizer(list(g.parameters()) + list(e_g.parameters()), 'rmsprop', 1e-4)
    opt_d = optim.make_optimizer(list(d.parameters()) + list(e_d.parameters()), 'rmsprop', 1e-4)
    words = ['alpha', 'beta', 'gamma', 'delta', 'epsil', 'zetaa', 'etaaa', 'theta', 'iotaa', 'kappa', 'lambd']
    ds = D.SyntheticWordDataset(words, n_per_word=3, min_len=maxlen // 3, max_len=maxlen, kind='noise', seed=3)
    args = types.SimpleNamespace(conditional=True, dataset=ds, minwordlen=1, subset=None, amplitudes=0)
    np.random.seed(5)
    h5, ml, gen_train, _, keys_train, _ = D.dataloader(B, args, maxlen=maxlen, frame_size=frame)
    pick = loop.words_picker(D, B, ml, h5, keys_train, args, frame_size=frame)
    prefix = os.path.join(tmp_path, 'run')
    mk = lambda **kw: loop.TrainLoop(g, d, e_g, e_d, opt_g, opt_d, gen_train, pick, B, ml, dev, checkpoint_prefix=prefix, **kw)  # noqa: E731
    return mk, (g, d, e_g, e_d), prefix


def _run(A, dev, wide, tmp_path, B):
    mk, mods, prefix = _setup(A, dev, wide, tmp_path, B)
    # the reference's loop: up to critic_iter critic iterations, stopping early once both accuracies pass require_acc
    lp = mk(critic_iter=3, require_acc=0.5, gencatchup=1, checkpoint_every=2)
    ran, rd, rg = lp.outer()
    assert 1 <= ran <= 3 and lp.dis_iter == ran and lp.gen_iter == 1
    if ran < 3:
        assert rd['acc_d'] > 0.5 and rd['acc_g'] > 0.5
    # a fixed count (what bench.py --workload full declares): an odd (FGSM) and an even (instance noise) critic iteration
    lp.fixed_critic_iter = 2
    ran2, rd, rg = lp.outer()
    assert ran2 == 2 and lp.gen_iter == 2
    for tag, *vals in lp.log:
        assert all(np.isfinite(v) for v in vals[1:]), (tag, vals)
    assert 0.0 <= rd['acc_d'] <= 1.0 and 0.0 <= rd['acc_g'] <= 1.0
    # the checkpoint of generator iteration 2 exists under the reference's names and resumes into the loop
    for role in ('dis', 'gen', 'eg', 'ed', 'opt'):
        assert os.path.exists('%s-%s-%05d' % (prefix, role, 2))
    want = [p.detach().clone() for m in mods for p in m.parameters()]
    lp.outer()
    extra = lp.resume(2)
    assert extra['gen_iter'] == 2 and lp.gen_iter == 2 and lp.dis_iter == extra['dis_iter']
    for p, q in zip([p for m in mods for p in m.parameters()], want):
        assert torch.equal(p.detach(), q)
    lp.outer()
    assert lp.gen_iter == 3
    return lp


def test_train_loop_host_logic(monkeypatch, tmp_path):
    kernel_model.install(monkeypatch)
    import audiogan_amd as A
    _run(A, torch.device('cpu'), False, tmp_path, 4)


def test_train_loop_device_scalars_match_host_scalars(monkeypatch, tmp_path):
    """host=False (what the captured iterations run: accuracies, the reward mean and the REINFORCE baseline stay tensors, no
    host read) against host=True on the same loader batches and random draws: same parameters up to the baseline's fp32 vs
    float64 arithmetic"""
    kernel_model.install(monkeypatch)
    import audiogan_amd as A
    res = []
    for host in (True, False):
        mk, mods, _ = _setup(A, torch.device('cpu'), False, tmp_path, 4)
        torch.manual_seed(7)
        lp = mk(fixed_critic_iter=2, gencatchup=1, stop='never', checkpoint_every=0, check=False, host=host)
        for _ in range(2):
            ran, rd, rg = lp.outer()
        assert ran == 2 and lp.dis_iter == 4 and lp.gen_iter == 2
        assert torch.is_tensor(rg['baseline']) == (not host)
        res.append(([p.detach().clone() for m in mods for p in m.parameters()], float(rg['baseline']), float(rd['acc_d'])))
    assert abs(res[0][1] - res[1][1]) <= 1e-6 * max(1.0, abs(res[0][1])) and res[0][2] == res[1][2]
    for a, b in zip(res[0][0], res[1][0]):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7)


@pytest.mark.gpu
def test_train_loop_at_c2_widths_gpu(tmp_path):
    """the outer loop at the C2 widths (default structs, state 1024, frame 256, 8192-sample ragged loader clips, both
    Embedders, Bernoulli stop draws): critic iterations with the accuracy early break, generator iterations, a checkpoint and
    a resume - no persistent launch gave up"""
    import audiogan_amd as A
    from audiogan_amd import kernels as K
    K.lstm_persist_status(reset=True)
    _run(A, torch.device('cuda'), True, tmp_path, 8)
    assert K.lstm_persist_status() == 0


@pytest.mark.gpu
def test_captured_iterations_replay_the_eager_loop_bit_for_bit(tmp_path):
    """TrainLoop(graphed=True): the odd / even critic iteration and the generator iteration captured into three hipGraphs and
    replayed over static inputs must leave the SAME bits in every parameter as the Python-issued loop with the same device
    arithmetic (host=False), on the same loader batches and the same device random stream - the kernels are deterministic and
    a capture may neither drop nor reorder work.  C2 widths, ragged loader clips, 4 passes after the 2 warm-up passes."""
    import audiogan_amd as A
    from audiogan_amd import kernels as K
    dev = torch.device('cuda')
    K.lstm_persist_status(reset=True)
    got = []
    for graphed in (False, True):
        mk, mods, _ = _setup(A, dev, True, tmp_path, 8)
        torch.cuda.manual_seed(17)
        lp = mk(fixed_critic_iter=2, gencatchup=1, stop='never', checkpoint_every=0, check=False, graphed=graphed, host=False)
        # (the first captured call runs two eager passes as its warm-up - real training iterations, counted)
        for _ in range(4 if graphed else 6):
            ran, rd, rg = lp.outer()
        assert lp.dis_iter == 12 and lp.gen_iter == 6
        if graphed:
            assert lp._graphs is not None and set(lp._graphs) == {'d0', 'd1', 'g'}
        got.append(([p.detach().clone() for m in mods for p in m.parameters()],
                    [float(rd['loss']), float(rd['acc_d']), float(rd['acc_g']), float(rg['loss']), float(rg['baseline'])]))
    assert K.lstm_persist_status() == 0
    assert got[0][1] == got[1][1], (got[0][1], got[1][1])
    assert all(np.isfinite(v) for v in got[0][1])
    moved = 0
    for p, q in zip(got[0][0], got[1][0]):
        assert torch.equal(p, q)
    mk, mods, _ = _setup(A, dev, True, tmp_path, 8)
    for p, q in zip(got[0][0], [p for m in mods for p in m.parameters()]):
        moved += int(not torch.equal(p, q.detach()))
    assert moved > 0.9 * len(got[0][0])          # (and the passes did train: nearly every tensor left its initial value)
