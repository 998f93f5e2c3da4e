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


def load_reference():
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
    print('wrote', sorted(f for f in os.listdir(OUT) if f.startswith('ref_')))


if __name__ == '__main__':
    main()
