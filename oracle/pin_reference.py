"""Pin the oracle against the reference's own code  --  TEST INFRASTRUCTURE.

Runs HERE only (needs /root/reference, which never travels to the GPU box) and
writes ``tests/golden/ref_*.npz``: inputs, parameters, outputs and gradients
produced by the *reference's own* ``Generator`` / ``Discriminator`` /
``dense_res_bottleneck`` / ``Residual`` / loss / mask / clip definitions.

How the reference is executed without copying it: ``audiogan.py`` cannot be
imported (Python-2 ``print`` statements at :594/:811/:940, import-time argparse,
TensorFlow/librosa/h5py imports, ``.cuda()`` everywhere).  The definitions on the
hot path are, however, valid Python-3 syntax on their own.  This script reads the
file as text at run time, cuts out the top-level ``def``/``class`` blocks named in
``WANTED``, and ``exec``s them in a namespace that provides

  * ``tovar``/``tonumpy`` without ``.cuda()``              (audiogan.py:94-97 F6)
  * ``NN.DataParallel`` -> a pass-through holder ``.module`` (numerically identity)
  * Python-2 integer ``/``: ``ast.Div`` -> ``ast.FloorDiv`` inside ``div_roundup``,
    ``roundup`` and ``Discriminator.forward`` only (all three divide ints /
    LongTensors; py2 and torch<0.4 floor them)
  * ``Tensor.multinomial()`` (arg-less form, removed) -> injected all-zero stop
    draw, i.e. "keep generating" (the bench's fixed-length clips)

Nothing else is changed; no reference text is written to the repo.  The npz
files hold data only.

Round 3 adds two fixtures:

``ref_dataset.npz``  the reference's own ``dataset.py`` (valid Python 3) IMPORTED from its path with two stub modules
  in ``sys.modules``: ``h5py`` (``File(name)`` hands back an in-memory mapping with the HDF5 layout) and ``utiltf``
  (only ``roundup`` / ``div_roundup``, exec'ed from the reference's utiltf.py text with the py2 integer division
  floored; the real module imports TensorFlow).  Global numpy RNG seeded; conditional ``next()`` x3 / validation x1,
  ``pick_words(skip_samples=True)``, a too-long and a silent word, unconditional ``next()`` across an epoch boundary.

``ref_step.npz``  the reference's training-loop BODIES executed as they stand: the statements of one critic iteration
  (audiogan.py:711-788, from ``dis_iter += 1`` to ``opt_d.step()``) and of one generator iteration (:822-921, from
  ``gen_iter += 1`` to ``opt_g.step()``) are cut out of the script by those marker lines, dedented and ``exec``ed
  against the reference's own classes / helpers at toy widths, fed by the reference's own ``dataset.py`` loader.
  Shims (mechanical, listed in ``_prepare_body``): ``.cuda()`` dropped; ``fake_len / framesize`` floored (LongTensor
  division); ``d_train_writer.add_summary(...)`` statements dropped, Timer / TF objects replaced by no-ops; ``T.randn`` / ``RNG.randn`` wrapped so that every
  draw is LOGGED (the fixture stores them; the oracle gets them as arguments); the arg-less ``multinomial()`` returns
  INJECTED stop draws.  ``.reinforce(reward)`` + ``T.autograd.backward(stop_list, [None..])`` belong to torch <= 0.3,
  a third-party dependency that is not vendored and not pinned; ``_reinforce_backward`` restates its
  published ``Multinomial.backward`` (torch/autograd/_functions/stochastic.py of 0.2/0.3): grad_probs[sample] =
  -reward / (p[sample] + 1e-6).  Stored: two critic iterations (odd = FGSM branch, even = instance noise) and one
  generator iteration on evolving weights / RMSprop state: losses, logits, accuracies, gradient-norm sums, the
  adversarial z, the generated clips and every post-step parameter.
"""
import ast
import os
import re
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence
from torch.nn.utils import weight_norm as torch_weight_norm

REF = os.environ.get('AUDIOGAN_REFERENCE', '/root/reference/audiogan.py')
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden')

WANTED = ['weight_norm', 'div_roundup', 'roundup', 'log_sigmoid', 'log_one_minus_sigmoid',
          'binary_cross_entropy_with_logits_per_sample', 'advanced_index', 'length_mask',
          'dynamic_rnn', 'check_grad', 'clip_grad', 'Residual', 'dense_res_bottleneck',
          'Embedder', 'Generator', 'Discriminator', 'fourth_moment', 'calc_dists']
FLOORDIV_IN = {'div_roundup', 'roundup', 'Discriminator', 'fourth_moment'}   # integer '/' only


def _blocks(text):
    """Yield (name, source) for top-level def/class blocks."""
    lines = text.split('\n')
    starts = [(i, m.group(2)) for i, l in enumerate(lines)
              for m in [re.match(r'^(def|class)\s+(\w+)', l)] if m]
    for i, name in starts:
        j = i + 1
        while j < len(lines) and (lines[j].strip() == '' or lines[j][0] in ' \t#'):
            j += 1
        yield name, '\n'.join(lines[i:j])


class _Py2Div(ast.NodeTransformer):
    def visit_BinOp(self, node):
        self.generic_visit(node)
        if isinstance(node.op, ast.Div):
            node.op = ast.FloorDiv()
        return node


class _PassThrough(nn.Module):
    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, *a, **k):
        return self.module(*a, **k)


def load_reference(strip_cuda=()):
    text = open(REF).read()
    NN = types.ModuleType('NN')
    NN.__dict__.update(nn.__dict__)
    NN.DataParallel = _PassThrough

    def tovar(*arrs):
        ts = [torch.tensor(a.astype('float32')) if isinstance(a, np.ndarray) else a for a in arrs]
        return ts[0] if len(ts) == 1 else ts

    def tonumpy(*vs):
        arrs = [v.detach().cpu().numpy() for v in vs]
        return arrs[0] if len(arrs) == 1 else arrs

    ns = dict(T=torch, NN=NN, F=F, NP=np, torch_weight_norm=torch_weight_norm,
              pack_padded_sequence=pack_padded_sequence, pad_packed_sequence=pad_packed_sequence,
              tovar=tovar, tonumpy=tonumpy, Parameter=nn.Parameter)
    found = dict(_blocks(text))
    for name in WANTED:
        tree = ast.parse(found[name])
        if name in FLOORDIV_IN:
            tree = ast.fix_missing_locations(_Py2Div().visit(tree))
        if name in strip_cuda:
            tree = ast.fix_missing_locations(_StripCudaFloorLen().visit(tree))
        exec(compile(tree, 'reference:' + name, 'exec'), ns)
    return types.SimpleNamespace(**ns)


class no_stop_multinomial:
    """Patch the removed arg-less Tensor.multinomial(): stop draw := 0 (continue)."""

    def __enter__(self):
        self._orig = torch.Tensor.multinomial

        def mn(t, *a, **k):
            if a or k:
                return self._orig(t, *a, **k)
            return torch.zeros(t.size(0), 1, dtype=torch.long)
        torch.Tensor.multinomial = mn

    def __exit__(self, *e):
        torch.Tensor.multinomial = self._orig


def _sd(module):
    return {k: v.detach().numpy().copy() for k, v in module.state_dict().items()}


def _grads(module):
    return {'grad.' + k: (p.grad.detach().numpy().copy() if p.grad is not None
                          else np.zeros(tuple(p.shape), np.float32))
            for k, p in module.named_parameters()}


def _pack(prefix, d):
    return {prefix + k: v for k, v in d.items()}


def randomize_(module, gen, scale=0.5):
    """Move every parameter off its init so weight-norm g != ||v|| and the bias
    sign structure is exercised."""
    with torch.no_grad():
        for p in module.parameters():
            p.add_(torch.randn(p.shape, generator=gen) * scale * p.abs().mean().clamp(min=1e-3))


# ======================================================================================================================
# round 3: dataset.py and the training-loop bodies
# ======================================================================================================================
REF_DIR = os.path.dirname(REF)


def _utiltf_stub():
    """``utiltf`` with roundup / div_roundup only (utiltf.py:102-106, py2 integer division floored)"""
    text = open(os.path.join(REF_DIR, 'utiltf.py')).read()
    found = dict(_blocks(text))
    mod = types.ModuleType('utiltf')
    for name in ('div_roundup', 'roundup'):
        tree = ast.fix_missing_locations(_Py2Div().visit(ast.parse(found[name])))
        exec(compile(tree, 'reference:utiltf.' + name, 'exec'), mod.__dict__)
    return mod


class _H5Stub(types.ModuleType):
    """``h5py.File(name)`` -> the in-memory mapping registered under that name"""

    def __init__(self):
        super().__init__('h5py')
        self.files = {}

    def File(self, name, *a, **k):
        return self.files[name]


def import_reference_dataset():
    import importlib.util
    h5 = _H5Stub()
    saved = {k: sys.modules.get(k) for k in ('h5py', 'utiltf')}
    sys.modules['h5py'], sys.modules['utiltf'] = h5, _utiltf_stub()
    try:
        spec = importlib.util.spec_from_file_location('reference_dataset', os.path.join(REF_DIR, 'dataset.py'))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return mod, h5


def make_word_dataset(seed=7):
    """word -> (n_utterances, maxlen_word) float32, zero padded: the layout preprocess-fisher.py:240-250 writes.
    Includes words the loader must filter ('to-', '(laugh)', short ones), one word whose clips never fit ``maxlen`` and
    one silent word (all zeros: redrawn by pick_word, dataset.py:69-70)."""
    rs = np.random.RandomState(seed)
    words = ['hello', 'a', 'to-', '(laugh)', 'xy'] + ['word%02d' % i for i in range(18)] + ['toolong', 'silent']
    ds = {}
    for w in words:
        n_utt, width = int(rs.randint(2, 5)), int(rs.randint(90, 130))
        arr = np.zeros((n_utt, width), np.float32)
        for i in range(n_utt):
            n = int(rs.randint(20, width + 1))
            arr[i, :n] = rs.uniform(-0.7, 0.7, size=n).astype(np.float32)
            arr[i, n - 1] = 0.25
            if rs.rand() < 0.3:
                arr[i, rs.randint(0, n - 1)] = 0.0          # zeros INSIDE a clip do not end it
        if w == 'toolong':
            arr = np.full((2, 400), 0.5, np.float32)
        if w == 'silent':
            arr = np.zeros((2, 100), np.float32)
        ds[w] = arr
    return ds


def pin_dataset():
    mod, h5 = import_reference_dataset()
    RNG = np.random
    ds = make_word_dataset()
    h5.files['mem.h5'] = ds
    args = types.SimpleNamespace(conditional=True, dataset='mem.h5', minwordlen=2, subset=None, amplitudes=6)
    out = {'ds.' + w: a for w, a in ds.items()}
    out['ds_order'] = np.array(list(ds.keys()))
    RNG.seed(11)
    dataset_h5, maxlen, gen_train, gen_val, keys_train, keys_val = mod.dataloader(3, args, maxlen=140, frame_size=32)
    assert dataset_h5 is ds
    out['maxlen'] = maxlen
    out['keys_train'], out['keys_val'] = np.array(keys_train), np.array(keys_val)
    for i in range(3):
        e, b, samples, lengths, keys, cseq, clen = next(gen_train)
        out.update({'t%d_epoch_batch' % i: np.array([e, b]), 't%d_samples' % i: samples, 't%d_lengths' % i: lengths,
                    't%d_keys' % i: np.array(keys), 't%d_cseq' % i: cseq, 't%d_clen' % i: clen})
    e, b, samples, lengths, keys, cseq, clen = next(gen_val)
    out.update({'v0_epoch_batch': np.array([e, b]), 'v0_samples': samples, 'v0_lengths': lengths,
                'v0_keys': np.array(keys), 'v0_cseq': cseq, 'v0_clen': clen})
    maxchar = max(len(k) for k in keys_train)
    keys, cs, cl, smp, ln = mod.pick_words(4, maxlen, ds, keys_train, maxchar, args, skip_samples=True)
    out.update(pw_keys=np.array(keys), pw_cseq=cs, pw_clen=cl, pw_samples=smp, pw_lengths=ln)
    # no frame_size, maxlen taken from the data (dataset.py:103), too-long / silent words in play
    RNG.seed(12)
    args2 = types.SimpleNamespace(conditional=True, dataset='mem.h5', minwordlen=1, subset=None, amplitudes=6)
    _, maxlen2, gen2, _, keys2, _ = mod.dataloader(5, args2)
    out['maxlen2'] = maxlen2
    out['keys2'] = np.array(keys2)
    e, b, samples, lengths, keys, cseq, clen = next(gen2)
    out.update(n_samples=samples, n_lengths=lengths, n_keys=np.array(keys), n_cseq=cseq, n_clen=clen)
    RNG.seed(13)
    for i in range(12):           # 'toolong' never fits maxlen 150; 'silent' is redrawn
        k, seq, n, smp, ln = mod.pick_word(150, ds, ['toolong', 'silent', 'hello'], 7, args2)
        assert k == 'hello'
    out['redraw_last'] = smp
    out['redraw_rng_after'] = RNG.randint(0, 1 << 30, size=4)
    # unconditional branch: 'data' (N, sr), 90 % / 10 % split, epoch roll-over (dataset.py:6-41)
    RNG.seed(14)
    h5.files['unc.h5'] = {'data': np.arange(80 * 8, dtype=np.float32).reshape(80, 8)}
    argsu = types.SimpleNamespace(conditional=False, dataset='unc.h5', subset=None, amplitudes=6)
    none, gu, gv = mod.dataloader(8, argsu)
    assert none is None
    rows = []
    for i in range(11):
        r = next(gu)
        assert r[3:] == [None] * 6
        rows.append(np.concatenate([[r[0], r[1]], r[2][:, 0]]))
    out['unc_train'] = np.array(rows)
    r = next(gv)
    out['unc_val'] = np.concatenate([[r[0], r[1]], r[2][:, 0]])
    RNG.seed(15)
    argss = types.SimpleNamespace(conditional=False, dataset='unc.h5', subset=20, amplitudes=8)
    _, gs, _ = mod.dataloader(4, argss)
    out['unc_subset'] = next(gs)[2]
    np.savez_compressed(os.path.join(OUT, 'ref_dataset.npz'), **out)


# ---- training-loop bodies --------------------------------------------------------------------------------------------
class _StripCudaFloorLen(ast.NodeTransformer):
    """``X.cuda()`` -> ``X``;  ``fake_len / Y`` -> ``fake_len // Y`` (LongTensor division of torch <= 0.3);
    ``d_train_writer.add_summary(...)`` statements dropped"""

    def visit_Call(self, node):
        self.generic_visit(node)
        if isinstance(node.func, ast.Attribute) and node.func.attr == 'cuda' and not node.args and not node.keywords:
            return node.func.value
        return node

    def visit_Expr(self, node):
        # ``d_train_writer.add_summary(...)`` statements (TensorBoard logging, out of scope) are dropped whole: their
        # arguments index 1-element tensors the torch <= 0.3 way
        c = node.value
        if isinstance(c, ast.Call) and isinstance(c.func, ast.Attribute) and c.func.attr == 'add_summary' and \
                isinstance(c.func.value, ast.Name) and c.func.value.id == 'd_train_writer':
            return None
        self.generic_visit(node)
        return node

    def visit_BinOp(self, node):
        self.generic_visit(node)
        if isinstance(node.op, ast.Div) and isinstance(node.left, ast.Name) and node.left.id == 'fake_len':
            node.op = ast.FloorDiv()
        return node


def _cut_body(text, first_marker, last_marker):
    """the statements from the line holding ``first_marker`` to the first later line holding ``last_marker``
    (inclusive), dedented; returns (source, first_line_no, last_line_no)"""
    lines = text.split('\n')
    i = next(k for k, l in enumerate(lines) if l.strip() == first_marker)
    j = next(k for k in range(i, len(lines)) if lines[k].strip() == last_marker)
    ind = len(lines[i]) - len(lines[i].lstrip())
    body = [l[ind:] if l.strip() else '' for l in lines[i:j + 1]]
    return '\n'.join(body), i + 1, j + 1


def _prepare_body(src, name):
    tree = _StripCudaFloorLen().visit(ast.parse(src))
    return compile(ast.fix_missing_locations(tree), 'reference:' + name, 'exec')


class _Sink(object):
    """TensorBoard writer / TF.Summary / Timer stand-in: accepts everything, does nothing"""

    def __call__(self, *a, **k):
        return self

    def __getattr__(self, n):
        return self

    def __enter__(self):
        return self

    def __exit__(self, *e):
        return False


class _LogRandn(object):
    """``T`` / ``RNG`` proxy: everything passes through, every ``randn`` draw is logged"""

    def __init__(self, real, log, tag):
        self.__dict__.update(_real=real, _log=log, _tag=tag)

    def __getattr__(self, n):
        return getattr(self._real, n)

    def randn(self, *a, **k):
        v = self._real.randn(*a, **k)
        self._log.append((self._tag, v.clone() if torch.is_tensor(v) else np.array(v)))
        return v


def _reinforce_backward(stop_list):
    """torch <= 0.3 ``Multinomial.backward`` (stochastic.py): grad_probs[i, sample_i] = -reward_i / (p_i[sample_i] + 1e-6),
    back-propagated from the probabilities each draw was taken from"""
    ps, gs = [], []
    for st in stop_list:
        p, r = st._probs, st._reward
        gp = torch.zeros_like(p)
        outp = p.detach().gather(1, st).add(1e-6).reciprocal().neg().mul(r)
        gp.scatter_add_(1, st, outp)
        ps.append(p)
        gs.append(gp)
    torch.autograd.backward(ps, gs)


class injected_stops(object):
    """arg-less ``Tensor.multinomial()`` -> the next column of the queued [B,T] stop arrays; the returned draw remembers
    the probabilities it was 'drawn' from and accepts ``.reinforce(reward)``"""

    def __init__(self, queue):
        self.queue = queue          # list of [B,T] long tensors, one per Generator.forward call, consumed column-wise
        self.col = 0

    def __enter__(self):
        self._orig = torch.Tensor.multinomial
        outer = self

        def mn(t, *a, **k):
            if a or k:
                return outer._orig(t, *a, **k)
            cur = outer.queue[0]
            st = cur[:, outer.col:outer.col + 1].clone()
            outer.col += 1
            st._probs = t
            st.reinforce = lambda r, st=st: setattr(st, '_reward', r.detach().clone())
            return st
        torch.Tensor.multinomial = mn
        return self

    def next_forward(self):
        self.queue.pop(0)
        self.col = 0

    def __exit__(self, *e):
        torch.Tensor.multinomial = self._orig


def pin_step():
    text = open(REF).read()
    d_src, d0, d1 = _cut_body(text, 'dis_iter += 1', 'opt_d.step()')
    g_src, g0, g1 = _cut_body(text, 'gen_iter += 1', 'opt_g.step()')
    assert (d0, d1, g0, g1) == (711, 788, 822, 921), (d0, d1, g0, g1)
    d_code, g_code = _prepare_body(d_src, 'critic iteration'), _prepare_body(g_src, 'generator iteration')

    global WANTED
    saved = list(WANTED)
    WANTED = saved + ['adversarially_sample_z', 'adversarial_movement_d']
    try:
        R = load_reference(strip_cuda=('adversarially_sample_z', 'adversarial_movement_d'))
    finally:
        WANTED = saved
    ns = R.length_mask.__globals__        # the dict the reference's definitions were exec'ed in (their module globals)
    mod, h5 = import_reference_dataset()
    ds = make_word_dataset(21)
    h5.files['mem.h5'] = ds
    B, fs, maxlen_arg = 4, 32, 128
    args = types.SimpleNamespace(conditional=True, dataset='mem.h5', minwordlen=2, subset=None, amplitudes=0,
                                 noisescale=0.01, dgradclip=1.0, ggradclip=0.1, g_optim='boundary_seeking', framesize=fs,
                                 critic_iter=2, gencatchup=1)
    np.random.seed(31)
    dataset_h5, maxlen, loader, _, keys_train, _ = mod.dataloader(B, args, maxlen=maxlen_arg, frame_size=fs)
    maxcharlen_train = max(len(k) for k in keys_train)

    gcfg = dict(frame_size=fs, embed_size=8, noise_size=8, state_size=64, num_layers=1,
                struct=[[17, 8, 16, 8], [9, 4, 16, 8]])
    dcfg = dict(state_size=64, embed_size=8, num_layers=1, cnn_struct=[[7, 2, 8], [7, 2, 16]])
    torch.manual_seed(32)
    g, d = R.Generator(**gcfg), R.Discriminator(**dcfg)
    e_g, e_d = R.Embedder(8, 6, num_chars=128), R.Embedder(8, 6, num_chars=128)
    gen = torch.Generator().manual_seed(33)
    for m in (g, d, e_g, e_d):
        randomize_(m, gen, 0.2)
    param_g = list(g.parameters()) + list(e_g.parameters())
    param_d = list(d.parameters()) + list(e_d.parameters())
    out = {'cfg': np.array([B, fs, maxlen, args.noisescale * 1e6]), 'lines': np.array([d0, d1, g0, g1])}
    for tag, m in (('g', g), ('d', d), ('eg', e_g), ('ed', e_d)):
        out.update(_pack('init.%s.' % tag, _sd(m)))

    log = []

    class _Loader(object):          # py2 generator protocol (dataloader.next()), logging what it hands out
        def next(self):
            r = next(loader)
            log.append(('batch', r))
            return r

    class _DatasetProxy(object):
        def pick_words(self, *a, **k):
            r = mod.pick_words(*a, **k)
            log.append(('pick', r))
            return r

    sink = _Sink()
    ns.update(g=g, d=d, e_g=e_g, e_d=e_d, param_g=param_g, param_d=param_d, args=args, batch_size=B, maxlen=maxlen,
              opt_g=torch.optim.RMSprop(param_g, lr=1e-4), opt_d=torch.optim.RMSprop(param_d, lr=1e-4),
              dataloader=_Loader(), dataset=_DatasetProxy(), dataset_h5=dataset_h5, keys_train=keys_train,
              maxcharlen_train=maxcharlen_train, Timer=sink, d_train_writer=sink, TF=sink, gc=__import__('gc'),
              dis_iter=0, gen_iter=0, baseline=None, lambda_fp=1, epoch=1,
              T=_LogRandn(torch, log, 'T'), RNG=_LogRandn(np.random, log, 'RNG'))
    ns['T'].autograd = types.SimpleNamespace(
        grad=torch.autograd.grad, Variable=getattr(torch.autograd, 'Variable', None),
        backward=lambda tensors, grads=None: _reinforce_backward(tensors))
    T_frames = maxlen // fs
    sgen = torch.Generator().manual_seed(34)

    def stops(ragged):
        s_ = torch.zeros(B, T_frames, dtype=torch.long)
        if ragged:
            s_[1, 2] = 1; s_[3, 1] = 1; s_[0, 3] = 1
        return s_

    def draws(kind):
        """the logged draws of one body, in order"""
        return [v for t, v in log if t == kind]

    # ---- two critic iterations (audiogan.py:706-712: generator frozen)
    for p in param_g:
        p.requires_grad = False
    for p in param_d:
        p.requires_grad = True
    for it in (1, 2):
        del log[:]
        st = stops(it == 1)
        with injected_stops([st]) as inj:
            exec(d_code, ns)
        assert ns['dis_iter'] == it and inj.col >= 1
        batch = [v for t, v in log if t == 'batch'][0]
        pick = [v for t, v in log if t == 'pick'][0]
        tdraws, rdraws = draws('T'), draws('RNG')
        # T.randn draws in order: z of the Generator.forward (:421), then on even iterations the fake-clip noise (:750)
        pre = 'd%d.' % it
        out.update({pre + 'real': batch[2], pre + 'real_len': np.asarray(batch[3]), pre + 'cs': batch[5], pre + 'cl': np.asarray(batch[6]),
                    pre + 'cs2': pick[1], pre + 'cl2': np.asarray(pick[2]), pre + 'z': tdraws[0].numpy(), pre + 'stop': st.numpy(),
                    pre + 'loss': ns['loss'].detach().numpy(), pre + 'loss_d': ns['loss_d'].detach().numpy(),
                    pre + 'loss_g': ns['loss_g'].detach().numpy(), pre + 'cls_d': ns['cls_d'].detach().numpy(),
                    pre + 'cls_g': ns['cls_g'].detach().numpy(),
                    pre + 'acc': np.array([float(ns['correct_d'] / ns['num_d']), float(ns['correct_g'] / ns['num_g'])]),
                    pre + 'grad_norm': float(ns['d_grad_norm']), pre + 'x_grad_norm': float(ns['x_grad_norm'])})
        if it % 2 == 0:
            assert len(tdraws) == 2 and len(rdraws) == 1
            out[pre + 'noise_fake_raw'] = tdraws[1].numpy()
            out[pre + 'noise_real_raw'] = rdraws[0]
        else:
            assert len(tdraws) == 1 and len(rdraws) == 0
        out.update(_pack(pre + 'post.d.', _sd(d)))
        out.update(_pack(pre + 'post.ed.', _sd(e_d)))
    # ---- one generator iteration (:813-816: critic frozen)
    for p in param_g:
        p.requires_grad = True
    for p in param_d:
        p.requires_grad = False
    del log[:]
    st_adv, st = stops(False), stops(True)
    st_adv[2, 1] = 1

    class _TwoForwards(injected_stops):
        pass

    inj = injected_stops([st_adv, st])
    # the body runs Generator.forward twice (inside adversarially_sample_z :103, then :841): switch the queue between them
    orig_g_forward = g.forward
    calls = []

    def g_forward(*a, **k):
        if calls:
            inj.next_forward()
        calls.append(1)
        return orig_g_forward(*a, **k)
    g.forward = g_forward
    with inj:
        exec(g_code, ns)
    g.forward = orig_g_forward
    assert len(calls) == 2 and ns['gen_iter'] == 1
    batch = [v for t, v in log if t == 'batch'][0]
    pick = [v for t, v in log if t == 'pick'][0]
    tdraws, rdraws = draws('T'), draws('RNG')
    # RNG.randn: real-clip noise (:823).  T.randn: z0 (:101), noise inside adversarially_sample_z (:104), noise (:842)
    assert len(tdraws) == 3 and len(rdraws) == 1
    out.update({'g1.real': batch[2], 'g1.real_len': np.asarray(batch[3]), 'g1.cs': pick[1], 'g1.cl': np.asarray(pick[2]),
                'g1.noise_real_raw': rdraws[0], 'g1.z0': tdraws[0].numpy(), 'g1.noise_adv_raw': tdraws[1].numpy(),
                'g1.noise_fake_raw': tdraws[2].numpy(), 'g1.stop_adv': st_adv.numpy(), 'g1.stop': st.numpy(),
                'g1.z': ns['z'].detach().numpy(), 'g1.loss': ns['loss'].detach().numpy(), 'g1.bce': ns['_loss'].detach().numpy(),
                'g1.feature_penalty': ns['feature_penalty'].detach().numpy(), 'g1.baseline': float(ns['baseline']),
                'g1.fake': ns['fake_data'].detach().numpy(), 'g1.fake_len': ns['fake_len'].numpy(),
                'g1.s': ns['fake_s'].detach().numpy(), 'g1.grad_norm': float(ns['g_grad_norm'])})
    out.update(_pack('g1.post.g.', _sd(g)))
    out.update(_pack('g1.post.eg.', _sd(e_g)))
    np.savez_compressed(os.path.join(OUT, 'ref_step.npz'), **out)


def main():
    os.makedirs(OUT, exist_ok=True)
    R = load_reference()
    gen = torch.Generator().manual_seed(0)

    # ---- helpers -----------------------------------------------------------
    x = torch.randn(5, 9, generator=gen) * 4
    tgt = torch.rand(5, 9, generator=gen)
    lens = torch.tensor([9, 3, 0, 7, 1])
    w = R.length_mask((5, 9), lens)
    np.savez(os.path.join(OUT, 'ref_helpers.npz'),
             x=x.numpy(), target=tgt.numpy(), lengths=lens.numpy(), mask=w.numpy(),
             bce=R.binary_cross_entropy_with_logits_per_sample(x, tgt).numpy(),
             bce_w=R.binary_cross_entropy_with_logits_per_sample(x, tgt, weight=w).numpy(),
             log_sigmoid=R.log_sigmoid(x).numpy(),
             log_one_minus_sigmoid=R.log_one_minus_sigmoid(x).numpy(),
             div_roundup=np.array([R.div_roundup(a, 7) for a in range(0, 30)]),
             roundup=np.array([R.roundup(a, 7) for a in range(0, 30)]))

    # ---- clip_grad -----------------------------------------------------------
    ps = [nn.Parameter(torch.randn(s, generator=gen)) for s in [(4, 3), (7,), (2, 2, 2)]]
    gs = [torch.randn(p.shape, generator=gen) * sc for p, sc in zip(ps, [3.0, 0.01, 1.0])]
    for p, g_ in zip(ps, gs):
        p.grad = g_.clone()
    tot = R.clip_grad(ps, 1.0)
    np.savez(os.path.join(OUT, 'ref_clip_grad.npz'), total=float(tot),
             **{'g%d' % i: g_.numpy() for i, g_ in enumerate(gs)},
             **{'c%d' % i: p.grad.numpy() for i, p in enumerate(ps)})

    # ---- dense_res_bottleneck (both residual cases) and Residual ---------------
    for tag, (k, s, cin, hid, cout) in {'bneck_nores': (9, 4, 3, 6, 5), 'bneck_res': (9, 4, 7, 6, 4),
                                        'bneck_s8': (17, 8, 1, 8, 4)}.items():
        torch.manual_seed(1)
        m = R.dense_res_bottleneck(k, s, cin, hid, cout)
        randomize_(m, gen)
        xin = torch.randn(2, cin, 64, generator=gen, requires_grad=True)
        y = m(xin)
        gy = torch.randn(y.shape, generator=gen)
        y.backward(gy)
        np.savez(os.path.join(OUT, 'ref_%s.npz' % tag), cfg=np.array([k, s, cin, hid, cout]),
                 x=xin.detach().numpy(), y=y.detach().numpy(), gy=gy.numpy(), gx=xin.grad.numpy(),
                 **_pack('sd.', _sd(m)), **_grads(m))
    torch.manual_seed(2)
    m = R.Residual(12)
    randomize_(m, gen)
    xin = torch.randn(6, 12, generator=gen, requires_grad=True)
    y = m(xin)
    gy = torch.randn(y.shape, generator=gen)
    y.backward(gy)
    np.savez(os.path.join(OUT, 'ref_residual.npz'), x=xin.detach().numpy(), y=y.detach().numpy(),
             gy=gy.numpy(), gx=xin.grad.numpy(), **_pack('sd.', _sd(m)), **_grads(m))

    # ---- tiny Generator --------------------------------------------------------
    gcfg = dict(frame_size=16, embed_size=6, noise_size=5, state_size=24, num_layers=2,
                struct=[[9, 4, 8, 4], [9, 4, 8, 4], [5, 2, 6, 4]])
    torch.manual_seed(3)
    g = R.Generator(**gcfg)
    randomize_(g, gen, 0.3)
    z = torch.randn(3, 4, 5, generator=gen)
    c = torch.randn(3, 6, generator=gen)
    with no_stop_multinomial():
        xg, s, stop_list, glen = g(z=z, c=c)
    gy = torch.randn(xg.shape, generator=gen)
    gs_ = torch.randn(s.shape, generator=gen)
    (xg * gy).sum().add((s * gs_).sum()).backward()
    np.savez(os.path.join(OUT, 'ref_generator.npz'),
             cfg_struct=np.array(gcfg['struct']),
             cfg=np.array([gcfg[k] for k in ['frame_size', 'embed_size', 'noise_size',
                                              'state_size', 'num_layers']]),
             z=z.numpy(), c=c.numpy(), x=xg.detach().numpy(), s=s.detach().numpy(),
             length=glen.numpy(), gy=gy.numpy(), gs=gs_.numpy(),
             **_pack('sd.', _sd(g)), **_grads(g))

    # ---- tiny Discriminator (ragged lengths) ------------------------------------
    dcfg = dict(state_size=16, embed_size=6, num_layers=1,
                cnn_struct=[[7, 2, 4], [7, 2, 8], [5, 2, 8]])
    torch.manual_seed(4)
    d = R.Discriminator(**dcfg)
    randomize_(d, gen, 0.3)
    xd = torch.randn(3, 64, generator=gen, requires_grad=True)
    ld = torch.tensor([64, 40, 17])
    logits, acts, act_lens, nfr = d(xd, ld, c)
    gl = torch.randn(logits.shape, generator=gen)
    (logits * gl).sum().backward()
    np.savez(os.path.join(OUT, 'ref_discriminator.npz'),
             cfg_struct=np.array(dcfg['cnn_struct']),
             cfg=np.array([dcfg[k] for k in ['state_size', 'embed_size', 'num_layers']]),
             x=xd.detach().numpy(), length=ld.numpy(), c=c.numpy(),
             logits=logits.detach().numpy(), nframes=nfr.numpy(), gl=gl.numpy(),
             gx=xd.grad.numpy(),
             **{'act%d' % i: a.detach().numpy() for i, a in enumerate(acts)},
             **{'actlen%d' % i: a.numpy() for i, a in enumerate(act_lens)},
             **_pack('sd.', _sd(d)), **_grads(d))

    # ---- feature statistics (calc_dists / fourth_moment) on D's activations ---------------
    dists = R.calc_dists([a.detach() for a in acts], act_lens)
    np.savez(os.path.join(OUT, 'ref_calc_dists.npz'), n=len(dists),
             **{'s%d' % i: t[0].numpy() for i, t in enumerate(dists)},
             **{'d%d' % i: t[1].numpy() for i, t in enumerate(dists)})

    # ---- Embedder -------------------------------------------------------------
    torch.manual_seed(5)
    e = R.Embedder(output_size=6, char_embed_size=4, num_chars=32)
    chars = torch.randint(0, 32, (3, 7), generator=gen)
    clen = torch.tensor([7, 2, 5])
    emb = e(chars, clen)
    np.savez(os.path.join(OUT, 'ref_embedder.npz'), chars=chars.numpy(), clen=clen.numpy(),
             emb=emb.detach().numpy(), **_pack('sd.', _sd(e)))
    # ---- C2 widths: the reference's OWN default-struct Generator / Discriminator (audiogan.py:368, :476) ----
    # frame_size 256 (T = 32, L = 8192), embed 100, noise 100, state 1024, exactly bench.py's models.  The weights
    # are NOT stored (60 MB): they are the torch default init under manual_seed(SEED), which the test regenerates
    # (tests/test_oracle_pinned.py::test_c2_width_fixture checks two of them against the checksums kept here).
    SEED = 1234
    B2, Tn, fsz = 2, 32, 256
    torch.manual_seed(SEED)
    g = R.Generator(frame_size=fsz, embed_size=100, noise_size=100, state_size=1024)
    d = R.Discriminator(state_size=1024, embed_size=100)
    gin = torch.Generator().manual_seed(SEED + 1)
    z = torch.randn(B2, Tn, 100, generator=gin)
    c = torch.randn(B2, 100, generator=gin)
    real = torch.rand(B2, Tn * fsz, generator=gin) * 2 - 1
    rlen = torch.tensor([Tn * fsz, 5000])
    gy = torch.randn(B2, Tn * fsz, generator=gin)
    gl = torch.randn(B2, 128, generator=gin)
    gs_ = torch.randn(B2, Tn, generator=gin)
    with no_stop_multinomial():
        xg, s, _, glen = g(z=z, c=c)
    (xg * gy).sum().add((s * gs_).sum()).backward()
    logits_f, _, _, nf_f = d(xg.detach(), glen, c)            # D on the generated clips (full length)
    logits_r, acts_r, _, nf_r = d(real, rlen, c)               # D on ragged "real" clips
    (logits_r * gl).sum().backward()
    gn = {k: float(p.grad.norm()) for k, p in g.named_parameters()}
    dn = {k: float(p.grad.norm()) if p.grad is not None else 0.0 for k, p in d.named_parameters()}
    np.savez_compressed(
        os.path.join(OUT, 'ref_c2_width.npz'), seed=SEED,
        z=z.numpy(), c=c.numpy(), rlen=rlen.numpy(),        # real / gy / gl / gs: re-drawn by the test (same generator)
        wave=xg.detach().numpy(), s=s.detach().numpy(), length=glen.numpy(),
        logits_fake=logits_f.detach().numpy(), logits_real=logits_r.detach().numpy(),
        nframes_real=nf_r.numpy(), act5_real_sum=np.array([float(acts_r[-1].double().sum()),
                                                           float(acts_r[-1].double().abs().sum())]),
        g_names=np.array(list(gn.keys())), g_gradnorm=np.array(list(gn.values())),
        d_names=np.array(list(dn.keys())), d_gradnorm=np.array(list(dn.values())),
        w_check=np.array([float(g.state_dict()['rnn.0.module.weight_hh_v'].double().sum()),
                          float(d.state_dict()['cnn.5.module.weight_v'].double().sum())]))
    pin_dataset()
    pin_step()
    print('wrote', sorted(f for f in os.listdir(OUT) if f.startswith('ref_')))


if __name__ == '__main__':
    main()
