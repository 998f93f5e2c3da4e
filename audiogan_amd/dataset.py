"""The reference's minibatch-generator interface (dataset.py:1-118) with a pluggable backing
store.  Same function names, arguments and return tuples:

  dataloader(batch_size, args, maxlen=None, frame_size=None)
      conditional   -> (dataset, maxlen, gen_train, gen_val, keys_train, keys_val)      (:110)
          next(gen) -> [epoch, batch, samples (B, maxlen^frame) f64, lengths (B,), keys,
                        cseq (B, maxchar) i32, clen (B,)]                               (:91)
      unconditional -> (None, gen_train, gen_val)                                       (:41)
          next(gen) -> [epoch, batch, (B, amplitudes)] + [None] * 6                     (:25)
  pick_words(batch_size, maxlen, dataset, keys, maxcharlen, args, frame_size=None, skip_samples=False)
      -> [keys, cseq, clen, samples, lengths] as numpy arrays                          (:76-77)

``args.dataset`` may be (a) a path to an HDF5 file in the reference's layout -- word-keyed
datasets of shape (n, maxlen_word), zero padded (needs h5py, which this image lacks), or (b) an
in-memory mapping with the same layout, e.g. ``SyntheticWordDataset`` / ``{'data': array}``,
which is what bench/tests use (SURVEY.md 8(d) synthetic clips).  Randomness comes from the
global numpy RNG exactly like the reference (RNG.choice / RNG.permutation).
"""
import numpy as NP
import numpy.random as RNG


def div_roundup(x, d):
    """utiltf.py:102-103"""
    return (x + d - 1) // d


def roundup(x, d):
    """utiltf.py:105-106"""
    return div_roundup(x, d) * d


class SyntheticWordDataset(dict):
    """word -> float32 array (n_utterances, maxlen_word), zero padded at the end, the on-disk layout
    written by preprocess-fisher.py:240-250.  kind 'noise' = U(-1,1), 'sine' = 100..1000 Hz tones."""

    def __init__(self, words, n_per_word=4, min_len=2000, max_len=8192, kind='noise', seed=0):
        super().__init__()
        rs = NP.random.RandomState(seed)
        for w in words:
            arr = NP.zeros((n_per_word, max_len), dtype=NP.float32)
            for i in range(n_per_word):
                n = int(rs.randint(min_len, max_len + 1))
                if kind == 'sine':
                    f = rs.uniform(100, 1000)
                    arr[i, :n] = NP.sin(2 * NP.pi * f * NP.arange(n) / 8000.0)
                else:
                    arr[i, :n] = rs.uniform(-1, 1, size=n)
                arr[i, n - 1] = arr[i, n - 1] if arr[i, n - 1] != 0 else 0.5   # length = last non-zero
            self[w] = arr


def _open(spec):
    if isinstance(spec, str):
        try:
            import h5py
        except ImportError as e:
            raise ImportError('args.dataset is a path but h5py is not installed; pass an in-memory '
                              'mapping (e.g. audiogan_amd.dataset.SyntheticWordDataset)') from e
        return h5py.File(spec, 'r')
    return spec


def _unconditional_dataloader(batch_size, data, lower, upper, args):
    """dataset.py:6-25"""
    epoch, batch = 1, 0
    idx = RNG.permutation(range(lower, upper))
    cur = 0
    while True:
        indices = []
        for _ in range(batch_size):
            if cur == len(idx):
                cur = 0
                idx = RNG.permutation(list(set(range(lower, upper)) - set(indices)))
                epoch += 1
                batch = 0
            indices.append(idx[cur])
            cur += 1
        sample = data[sorted(indices)]
        yield [epoch, batch, NP.array(sample)[:, :args.amplitudes]] + [None] * 6
        batch += 1


def unconditional_dataloader(batch_size, args):
    """dataset.py:27-41"""
    dataset = _open(args.dataset)
    data = dataset['data']
    nsamples = data.shape[0]
    if getattr(args, 'subset', None):
        keep = RNG.permutation(range(nsamples))[:args.subset]
        data = data[sorted(keep)]
        nsamples = args.subset
    n_train = nsamples // 10 * 9
    return (None, _unconditional_dataloader(batch_size, data, 0, n_train, args),
            _unconditional_dataloader(batch_size, data, n_train, nsamples, args))


def word_to_seq(word, maxcharlen):
    """dataset.py:43-46"""
    seq = NP.zeros(maxcharlen, dtype=NP.int32)
    seq[:len(word)] = [ord(ch) for ch in word]
    return seq


def _pick_sample_from_word(key, maxlen, dataset, frame_size=None, skip_samples=False):
    """dataset.py:48-61: random utterance of `key`; length = index after the last non-zero sample,
    rounded up to a whole number of frames; (None, None) when it does not fit in maxlen."""
    sample_idx = RNG.choice(dataset[key].shape[0])
    out = NP.zeros(maxlen)
    length = 0
    if not skip_samples:
        src = NP.asarray(dataset[key][sample_idx])
        nz = NP.nonzero(src)[0]
        n = int(nz[-1]) + 1 if len(nz) else 0
        if n > maxlen:
            return None, None
        length = n if frame_size is None else roundup(n, frame_size)
        out[:n] = src[:n]
    return out, length


def pick_word(maxlen, dataset, keys, maxcharlen, args, frame_size=None, skip_samples=False):
    """dataset.py:63-74: redraw until a clip fits and is not silent; peak-normalise (:68-71)"""
    while True:
        key = RNG.choice(keys)
        out, length = _pick_sample_from_word(key, maxlen, dataset, frame_size, skip_samples)
        if out is None:
            continue
        if not skip_samples:
            peak = NP.abs(out).max()
            if peak == 0:
                continue
            out /= peak
        break
    return key, word_to_seq(key, maxcharlen), len(key), out, length


def pick_words(batch_size, maxlen, dataset, keys, maxcharlen, args, frame_size=None, skip_samples=False):
    """dataset.py:76-77"""
    picks = [pick_word(maxlen, dataset, keys, maxcharlen, args, frame_size, skip_samples)
             for _ in range(batch_size)]
    return [NP.array(col) for col in zip(*picks)]


def _conditional_dataloader(batch_size, dataset, maxlen, keys, args, frame_size=None):
    """dataset.py:79-91"""
    epoch, batch = 0, 0
    maxcharlen = max(len(k) for k in keys)
    if frame_size is not None:
        maxlen = roundup(maxlen, frame_size)
    while True:
        batch += 1
        picked, cseq, clen, samples, lengths = pick_words(batch_size, maxlen, dataset, keys, maxcharlen,
                                                          args, frame_size)
        yield [epoch, batch, samples, lengths, picked, cseq, clen]


def _valid_keys(keys, args):
    """dataset.py:93-96"""
    keys = [k for k in keys if not (k[-1] == '-' or k[0] == '(')]
    return [k for k in keys if len(k) >= args.minwordlen]


def conditional_dataloader(batch_size, args, maxlen=None, frame_size=None):
    """dataset.py:98-110"""
    dataset = _open(args.dataset)
    keys = _valid_keys(list(dataset.keys()), args)
    keys = [str(k) for k in RNG.permutation(keys)]
    n_train = len(keys) // 10 * 9
    maxlen = maxlen or max(dataset[k].shape[1] for k in keys)
    train_keys, val_keys = keys[:n_train], keys[n_train:]
    return (dataset, maxlen,
            _conditional_dataloader(batch_size, dataset, maxlen, train_keys, args, frame_size),
            _conditional_dataloader(batch_size, dataset, maxlen, val_keys, args, frame_size),
            train_keys, val_keys)


def dataloader(batch_size, args, maxlen=None, frame_size=None):
    """dataset.py:112-118"""
    if not args.conditional:
        return unconditional_dataloader(batch_size, args)
    return conditional_dataloader(batch_size, args, maxlen=maxlen, frame_size=frame_size)
