"""dataset.py / timer.py interface of the reference (dataset.py:1-118, timer.py:4-32)."""
import types

import numpy as np
import numpy.random as RNG

from audiogan_amd import dataset as D
from audiogan_amd.timer import Timer


def _args(ds, **kw):
    a = types.SimpleNamespace(conditional=True, dataset=ds, minwordlen=1, subset=None, amplitudes=8000)
    a.__dict__.update(kw)
    return a


def test_conditional_loader_shapes_and_semantics():
    RNG.seed(0)
    words = ['hello', 'a', 'to-', '(laugh)', 'x'] + ['word%02d' % i for i in range(21)]
    ds = D.SyntheticWordDataset(words, n_per_word=3, min_len=100, max_len=900, kind='sine')
    ret = D.dataloader(4, _args(ds, minwordlen=2), maxlen=1000, frame_size=256)
    dataset, maxlen, gen_train, gen_val, keys_train, keys_val = ret
    assert dataset is ds and maxlen == 1000
    allk = keys_train + keys_val
    assert 'to-' not in allk and '(laugh)' not in allk and 'a' not in allk and 'x' not in allk
    assert len(keys_train) == len(allk) // 10 * 9
    epoch, batch, samples, lengths, picked, cseq, clen = next(gen_train)
    assert (epoch, batch) == (0, 1)
    assert samples.shape == (4, 1024) and samples.dtype == np.float64     # maxlen rounded up to frames
    assert cseq.shape == (4, max(len(k) for k in keys_train)) and cseq.dtype == np.int32
    for i in range(4):
        assert lengths[i] % 256 == 0 and 0 < lengths[i] <= 1024
        assert abs(np.abs(samples[i]).max() - 1.0) < 1e-12               # peak normalised
        n = np.nonzero(samples[i])[0][-1] + 1
        assert D.roundup(n, 256) == lengths[i]
        assert ''.join(chr(ch) for ch in cseq[i][:clen[i]]) == picked[i]
    assert next(gen_train)[1] == 2
    keys, cs, cl, smp, ln = D.pick_words(3, 1000, ds, keys_train, 6, _args(ds), skip_samples=True)
    assert smp.shape == (3, 1000) and not smp.any() and not ln.any()


def test_too_long_clips_are_redrawn():
    RNG.seed(1)
    ds = {'long': np.ones((2, 500), np.float32), 'short': np.ones((2, 500), np.float32)}
    ds['short'][:, 100:] = 0
    for _ in range(10):
        key, seq, n, out, length = D.pick_word(200, ds, ['long', 'short'], 5, _args(ds))
        assert key == 'short' and length == 100


def test_unconditional_loader():
    RNG.seed(2)
    ds = {'data': np.arange(80 * 10, dtype=np.float32).reshape(80, 10)}
    none, gen_train, gen_val = D.dataloader(8, _args(ds, conditional=False, amplitudes=6))
    assert none is None
    out = next(gen_train)
    assert out[0] == 1 and out[1] == 0 and out[2].shape == (8, 6) and out[3:] == [None] * 6
    assert (out[2][:, 0] < 72 * 10).all()            # first 90% of the rows
    assert (next(gen_val)[2][:, 0] >= 72 * 10).all()
    for _ in range(12):
        e = next(gen_train)
    assert e[0] >= 2                                  # epoch counter advanced


def test_helpers_and_timer():
    assert [D.div_roundup(a, 7) for a in range(15)] == [(a + 6) // 7 for a in range(15)]
    assert D.roundup(8192, 200) == 8200 and D.roundup(8192, 256) == 8192
    Timer.reset()
    with Timer.new('blk'):
        sum(range(1000))
    assert Timer.get('blk') >= 0 and Timer.get('missing') == 0
