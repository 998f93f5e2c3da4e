"""CPU oracle for the audiogan hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

This file is a torch-CPU (fp32) restatement of the reference's G+D training
path.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it; the product package ``audiogan_amd`` never
does (its ops raise if the HIP library is missing).

Pinning status: the reference holds no tests and no golden vectors (SURVEY.md
F2), and its files cannot be imported as they stand (Python-2 syntax, import-time
``.cuda()``/TensorFlow/h5py).  ``oracle/pin_reference.py`` therefore executes the
reference's *own* class/function definitions (read from /root/reference at run
time, never copied) on torch CPU and stores their inputs/outputs under
``tests/golden/ref_*.npz``; ``tests/test_oracle_pinned.py`` checks this
restatement against those vectors.  Pieces with no executable reference
(Adam, WGAN-GP, GRU cell, tiny conv G/D: TensorFlow-only or absent, SURVEY.md
F5/F7/F8) are marked "unpinned" where they are defined.

Every function cites the reference file:line it follows (paths are relative to
/root/reference).
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.utils import weight_norm as _torch_weight_norm
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence

LEAKY_SLOPE = 0.01  # torch default of NN.LeakyReLU() / F.leaky_relu, audiogan.py:261,277,532


# --------------------------------------------------------------------------
# integer helpers
# --------------------------------------------------------------------------
def div_roundup(x, d):
    """audiogan.py:172-173 (Python-2 ``/`` on ints/LongTensors = floor division)."""
    return (x + d - 1) // d


def roundup(x, d):
    """audiogan.py:174-175."""
    return (x + d - 1) // d * d


# --------------------------------------------------------------------------
# weight norm
# --------------------------------------------------------------------------
def weight_norm(module, names):
    """audiogan.py:77-80: stock weight_norm (dim=0) on every listed name.

    For a 1-D bias the per-row norm is |v_i| so bias == g * sign(v)."""
    for n in names:
        module = _torch_weight_norm(module, n)
    return module


class Replicated(nn.Module):
    """Numerically-identity stand-in for the single-process ``NN.DataParallel``
    wrappers of audiogan.py:314,379,384,392,405,409,410,492,504,508.  Kept so that
    ``state_dict`` keys carry the reference's ``.module.`` nesting."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, *a, **k):
        return self.module(*a, **k)


# --------------------------------------------------------------------------
# losses / masks
# --------------------------------------------------------------------------
def log_sigmoid(x):
    """audiogan.py:178-179."""
    return -F.softplus(-x)


def log_one_minus_sigmoid(x):
    """audiogan.py:180-184 (piecewise form; both branches are evaluated and
    blended with a 0/1 sign mask exactly like the reference)."""
    neg_branch = torch.log(1 - torch.sigmoid(x))
    pos_branch = -x - torch.log(1 + torch.exp(-x))
    s = (x > 0).float()
    return s * pos_branch + (1 - s) * neg_branch


def binary_cross_entropy_with_logits_per_sample(input, target, weight=None):
    """audiogan.py:187-197: stable BCE-with-logits, optional weight, sum over dim 1."""
    if target.size() != input.size():
        raise ValueError("Target size ({}) must be the same as input size ({})".format(
            target.size(), input.size()))
    m = (-input).clamp(min=0)
    loss = input - input * target + m + ((-m).exp() + (-input - m).exp()).log()
    if weight is not None:
        loss = loss * weight
    return loss.sum(1)


def length_mask(size, length):
    """audiogan.py:204-211: (B, n) float matrix, row i has ones in [0, length[i])."""
    b, n = int(size[0]), int(size[1])
    ar = torch.arange(n).unsqueeze(0).expand(b, n)
    return (ar < length.view(b, 1).long()).float()


# --------------------------------------------------------------------------------------
# bf16 mode (BASELINE configs[2]; SURVEY 7 "bf16 ... stated against a bf16-rounded oracle").  The reference is
# fp32 only; this is the CPU statement of audiogan_amd's AG_PREC_BF16 contract: EVERY contraction (conv,
# transposed conv, linear, the LSTM / LSTMCell products; forward, backward-data and backward-weight forms)
# rounds BOTH operands to bfloat16 (round-to-nearest-even) and accumulates in fp32; everything else is fp32.
#   forward   y = op(rnd(x), rnd(w)) + bias
#   backward  dx = op_bwd_data(rnd(dy), rnd(w)),  dw = op_bwd_weight(rnd(dy), rnd(x)),  dbias = sum(dy)
# --------------------------------------------------------------------------------------
BF16 = [False]


def _rnd(x):
    return x.bfloat16().float()


class _RoundFwd(torch.autograd.Function):
    """rounds the value; the gradient passes through unchanged"""

    @staticmethod
    def forward(ctx, x):
        return _rnd(x)

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundBwd(torch.autograd.Function):
    """identity; rounds the gradient that comes back"""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return _rnd(g)


# bf16 STORAGE (round 4): on top of the rounded contractions, the critic's sequence path - time-major conv features,
# biLSTM output, the activations of the heads - and the gradients flowing back through those tensors are STORED as
# bfloat16.  A stored tensor is rounded where it is written, so every later use sees the rounded value: the residual add
# of `Residual`, the column sums behind the bias gradients and the dy entering the recurrent backward, besides the
# products, whose operands were rounded anyway.  ``_st`` marks such a tensor: value rounded forward, gradient rounded back.
BF16_STORE = [False]


def _st(x):
    if BF16[0] and BF16_STORE[0]:
        return _RoundBwd.apply(_RoundFwd.apply(x))
    return x


def _st_act(pre, act):
    """a stored activation act(pre): the VALUE is rounded where it is written (after the activation), the GRADIENT where it
    is written (the gradient of the pre-activation, i.e. after the activation's derivative was applied)"""
    if BF16[0] and BF16_STORE[0]:
        return _RoundFwd.apply(act(_RoundBwd.apply(pre)))
    return act(pre)


class bf16_mode(object):
    """``with O.bf16_mode(): ...`` - F.conv1d / F.conv_transpose1d / F.linear round their operands (and the gradient
    entering their backward); nn.LSTMCell and the LSTM layers of ``dynamic_rnn`` are computed from F.linear products
    so that their operands are rounded the same way.  ``store=True``: additionally the tensors marked ``_st`` (the
    critic's sequence path) are rounded where they are produced, value and gradient - the bf16-storage contract."""

    def __init__(self, store=False):
        self.store = bool(store)

    def __enter__(self):
        self._saved_store = BF16_STORE[0]
        BF16_STORE[0] = self.store
        self._saved = (F.conv1d, F.conv_transpose1d, F.linear, nn.LSTMCell.forward, BF16[0])
        o_conv, o_convt, o_lin = F.conv1d, F.conv_transpose1d, F.linear

        def _wrap(orig):
            def op(x, w, bias=None, *a, **k):
                y = _RoundBwd.apply(orig(_RoundFwd.apply(x), _RoundFwd.apply(w), None, *a, **k))
                if bias is not None:
                    y = y + bias.view([1, -1] + [1] * (y.dim() - 2)) if y.dim() > 2 else y + bias
                return y
            return op

        F.conv1d, F.conv_transpose1d, F.linear = _wrap(o_conv), _wrap(o_convt), _wrap(o_lin)
        torch.nn.functional.conv1d, torch.nn.functional.conv_transpose1d = F.conv1d, F.conv_transpose1d
        torch.nn.functional.linear = F.linear

        def cell_forward(self, x, state):
            h, c = state
            gates = F.linear(x, self.weight_ih, self.bias_ih) + F.linear(h, self.weight_hh, self.bias_hh)
            i, f, g, o = gates.chunk(4, 1)
            c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
            return torch.sigmoid(o) * torch.tanh(c2), c2

        nn.LSTMCell.forward = cell_forward
        BF16[0] = True
        return self

    def __exit__(self, *exc):
        BF16_STORE[0] = self._saved_store
        F.conv1d, F.conv_transpose1d, F.linear, nn.LSTMCell.forward, BF16[0] = self._saved
        torch.nn.functional.conv1d, torch.nn.functional.conv_transpose1d = F.conv1d, F.conv_transpose1d
        torch.nn.functional.linear = F.linear


def _lstm_explicit(rnn, seq, length):
    """nn.LSTM (any layers / directions) over a padded batch with pack/unpack semantics, written with F.linear:
    steps past a clip's length keep its state and output zeros"""
    T, B, _ = seq.shape
    ndir = 2 if rnn.bidirectional else 1
    H = rnn.hidden_size
    inp = seq
    last_h, last_c = [], []
    lens = length.view(B, 1)
    for layer in range(rnn.num_layers):
        outs = []
        for d in range(ndir):
            sfx = '_l%d%s' % (layer, '_reverse' if d else '')
            w_ih, w_hh = getattr(rnn, 'weight_ih' + sfx), getattr(rnn, 'weight_hh' + sfx)
            b_ih, b_hh = getattr(rnn, 'bias_ih' + sfx), getattr(rnn, 'bias_hh' + sfx)
            h, c = seq.new_zeros(B, H), seq.new_zeros(B, H)
            ys = [None] * T
            for k in range(T):
                t = k if d == 0 else T - 1 - k
                gates = F.linear(inp[t], w_ih, b_ih) + F.linear(h, w_hh, b_hh)
                i, f, g, o = gates.chunk(4, 1)
                c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
                h2 = torch.sigmoid(o) * torch.tanh(c2)
                live = (t < lens).float()
                h, c = live * h2 + (1 - live) * h, live * c2 + (1 - live) * c
                ys[t] = live * h2
            outs.append(torch.stack(ys, 0))
            last_h.append(h)
            last_c.append(c)
        inp = torch.cat(outs, 2)
    return inp, (torch.stack(last_h, 0), torch.stack(last_c, 0))


def dynamic_rnn(rnn, seq, length, initial_state):
    """audiogan.py:214-229: sort by length, pack, run, unpack, unsort."""
    if BF16[0] and isinstance(rnn, nn.LSTM):
        return _lstm_explicit(rnn, seq, length)
    len_sorted, order = torch.sort(length, descending=True)
    _, inverse = torch.sort(order)
    packed = pack_padded_sequence(seq[:, order], len_sorted.cpu())
    out, last = rnn(packed, initial_state)
    out = pad_packed_sequence(out)[0][:, inverse]
    if isinstance(last, tuple):
        last = tuple(s[:, inverse] for s in last)
    else:
        last = last[:, inverse]
    return out, last


def check_grad(params):
    """audiogan.py:232-240."""
    for p in params:
        if p.grad is None:
            continue
        g = p.grad.data
        assert int((g != g).long().sum()) == 0
        assert int((g.abs() > 1e5).long().sum()) == 0


def clip_grad(params, clip_norm):
    """audiogan.py:243-253: PER-PARAMETER L2 clip; returns the sum of the norms."""
    if clip_norm == 0:
        return None
    total = 0
    for p in params:
        if p.grad is not None:
            n = torch.norm(p.grad.data)
            total = total + n
            if n > clip_norm:
                p.grad.data /= (n / clip_norm)
    return total


# --------------------------------------------------------------------------
# model pieces
# --------------------------------------------------------------------------
class Residual(nn.Module):
    """audiogan.py:256-264: LeakyReLU(W x + b + x)."""

    def __init__(self, size):
        super().__init__()
        self.size = size
        self.linear = weight_norm(nn.Linear(size, size), ['weight', 'bias'])
        self.relu = nn.LeakyReLU()

    def forward(self, x):
        return _st_act(self.linear(x) + x, self.relu)


class dense_res_bottleneck(nn.Module):
    """audiogan.py:266-283: strided conv -> leaky -> transposed conv ->
    (+ last ``outfilters`` input channels when infilters >= outfilters) -> leaky."""

    def __init__(self, kernel, stride, infilters, hidden_filters, outfilters):
        super().__init__()
        self.infilters = infilters
        self.outfilters = outfilters
        self.conv = weight_norm(
            nn.Conv1d(infilters, hidden_filters, kernel_size=kernel, stride=stride,
                      padding=(kernel - 1) // 2), ['weight', 'bias'])
        self.deconv = weight_norm(
            nn.ConvTranspose1d(hidden_filters, outfilters, kernel - 1, stride,
                               padding=stride // 2), ['weight', 'bias'])
        self.relu = nn.LeakyReLU()

    def forward(self, x):
        a = self.deconv(self.relu(self.conv(x)))
        if self.infilters >= self.outfilters:
            a = a + x[:, -self.outfilters:, :]
        return self.relu(a)


class Embedder(nn.Module):
    """audiogan.py:302-334: char embedding -> biLSTM -> last hidden of both directions."""

    def __init__(self, output_size=100, char_embed_size=50, num_layers=1, num_chars=256):
        super().__init__()
        self._output_size = output_size
        self._char_embed_size = char_embed_size
        self._num_layers = num_layers
        self.embed = Replicated(nn.Embedding(num_chars, char_embed_size))
        self.rnn = nn.LSTM(char_embed_size, output_size // 2, num_layers, bidirectional=True)

    def forward(self, chars, length):
        nl, b, o = self._num_layers, chars.size(0), self._output_size
        seq = self.embed(chars).permute(1, 0, 2)
        init = (torch.zeros(nl * 2, b, o // 2), torch.zeros(nl * 2, b, o // 2))
        _, (h, _) = dynamic_rnn(self.rnn, seq, length, init)
        h = h.permute(1, 0, 2)
        return h[:, -2:].reshape(b, o)


class Generator(nn.Module):
    """audiogan.py:361-468.

    Differences from the file, all forced by F3/F6 of SURVEY.md and none numeric:
    no ``.cuda()``, no ``Variable``; the Bernoulli stop draw
    (``p_t.multinomial()`` :450, arg-less form removed from torch) is taken from
    the optional ``stop`` argument ((B,T) long, 1 = stop) or, when absent, drawn
    with ``torch.multinomial(p_t, 1)``."""

    def __init__(self, frame_size=200, embed_size=200, noise_size=100, state_size=1024,
                 num_layers=1,
                 struct=((17, 8, 128, 16), (9, 4, 64, 32), (9, 4, 64, 32), (9, 4, 32, 32))):
        super().__init__()
        self._frame_size = frame_size
        self._noise_size = noise_size
        self._state_size = state_size
        self._embed_size = embed_size
        self._num_layers = num_layers
        names = ['weight_ih', 'weight_hh', 'bias_hh', 'bias_ih']
        self.rnn = nn.ModuleList()
        self.rnn.append(Replicated(weight_norm(
            nn.LSTMCell(frame_size + embed_size + noise_size, state_size), names)))
        for _ in range(1, num_layers):
            self.rnn.append(Replicated(weight_norm(nn.LSTMCell(state_size, state_size), names)))
        self.dense_res_gen = nn.ModuleList()
        cin = 1
        for kernel, stride, hidden, cout in struct:
            self.dense_res_gen.append(Replicated(
                dense_res_bottleneck(kernel, stride, cin, hidden, cout)))
            cin += cout
        self.dense_res_gen.append(Replicated(weight_norm(
            nn.Conv1d(cin, 1, kernel_size=3, stride=1, padding=1), ['weight', 'bias'])))
        self.proj = Replicated(weight_norm(nn.Linear(state_size, frame_size), ['weight', 'bias']))
        self.stopper = Replicated(weight_norm(nn.Linear(state_size, 1), ['weight', 'bias']))

    def forward(self, batch_size=None, length=None, z=None, c=None, stop=None):
        fs, ns, ss, es, nl = (self._frame_size, self._noise_size, self._state_size,
                              self._embed_size, self._num_layers)
        if z is None:
            nframes = div_roundup(length, fs)
            z = torch.randn(batch_size, nframes, ns)
        else:
            batch_size, nframes, _ = z.size()
        zc = torch.cat([z, c.unsqueeze(1).expand(batch_size, nframes, es)], 2)
        hs = [torch.zeros(batch_size, ss) for _ in range(nl)]
        cs = [torch.zeros(batch_size, ss) for _ in range(nl)]
        x_t = torch.zeros(batch_size, fs)
        generating = torch.ones(batch_size).long()
        out_len = torch.zeros(batch_size).long()
        xs, ss_list, stops = [], [], []
        for t in range(nframes):
            inp = torch.cat([x_t, zc[:, t]], 1)
            hs[0], cs[0] = self.rnn[0](inp, (hs[0], cs[0]))
            for i in range(1, nl):
                hs[i], cs[i] = self.rnn[i](hs[i - 1], (hs[i], cs[i]))
            x_t = torch.tanh(self.proj(hs[-1]))
            logit = self.stopper(hs[-1])
            if stop is None:
                p = torch.cat([log_one_minus_sigmoid(logit), log_sigmoid(logit)], 1).exp()
                stop_t = torch.multinomial(p, 1)
            else:
                stop_t = stop[:, t:t + 1].long()
            out_len += generating
            xs.append(x_t)
            ss_list.append(logit.squeeze(1))
            stops.append(stop_t)
            generating = generating * (stop_t.squeeze(1) == 0).long()
            if int(generating.sum()) == 0:
                break
        x = torch.cat(xs, 1).unsqueeze(1)
        s = torch.stack(ss_list, 1)
        for layer in self.dense_res_gen:
            x_next = layer(x)
            x = torch.cat([x, x_next], 1)
        return x_next.squeeze(1), s, stops, out_len * fs


class Discriminator(nn.Module):
    """audiogan.py:471-551."""

    def __init__(self, state_size=1024, embed_size=200, num_layers=1,
                 cnn_struct=((7, 2, 16), (7, 2, 32), (7, 2, 64), (7, 2, 128), (7, 2, 256),
                             (7, 2, 512))):
        super().__init__()
        self._state_size = state_size
        self._embed_size = embed_size
        self._num_layers = num_layers
        self.cnn_struct = [list(l) for l in cnn_struct]
        self._cnn_struct = self.cnn_struct
        self.cnn = nn.ModuleList()
        cin = 1
        for kernel, stride, cout in self.cnn_struct:
            self.cnn.append(Replicated(weight_norm(
                nn.Conv1d(cin, cout, kernel, stride=stride, padding=(kernel - 1) // 2),
                ['weight', 'bias'])))
            cin = cout
        self.frame_size = self._frame_size = cin
        self.rnn = nn.LSTM(cin + embed_size, state_size // 2, num_layers, bidirectional=True)
        self.residual_net = Replicated(nn.Sequential(Residual(state_size), Residual(state_size)))
        self.classifier = Replicated(nn.Sequential(
            weight_norm(nn.Linear(state_size, state_size // 2), ['weight', 'bias']),
            nn.LeakyReLU(),
            weight_norm(nn.Linear(state_size // 2, 1), ['weight', 'bias'])))

    def forward(self, x, length, c, percent_used=0.1):
        ss, nl, es = self._state_size, self._num_layers, self._embed_size
        b = x.size(0)
        init = (torch.zeros(nl * 2, b, ss // 2), torch.zeros(nl * 2, b, ss // 2))
        acts, act_lens = [], []
        a = x.unsqueeze(1)
        nframes = length
        for conv, (_, stride, _) in zip(self.cnn, self.cnn_struct):
            a = F.leaky_relu(conv(a))
            nframes = (nframes + stride - 1) // stride
            a = a * length_mask((b, a.size(2)), nframes).unsqueeze(1)
            acts.append(a)
            act_lens.append(nframes)
        a = _st(a.permute(0, 2, 1))                  # (bf16 storage: the time-major features / their gradient)
        n = a.size(1)
        seq = torch.cat([a, c.unsqueeze(1).expand(b, n, es)], 2).permute(1, 0, 2)
        out, _ = dynamic_rnn(self.rnn, seq, nframes, init)
        out = _st(out.permute(1, 0, 2))              # (the biLSTM output / the dy entering its backward)
        n = out.size(1)
        # the reference uses .view on the permuted tensor (old torch returned a
        # contiguous tensor from pad_packed_sequence+index); reshape is the same data
        rows = out.reshape(b * n, ss)
        cl = self.classifier.module
        hmid = _st_act(cl[0](self.residual_net(rows)), cl[1])        # (= self.classifier(...), the hidden layer marked as stored)
        logits = cl[2](hmid).view(b, n)
        return logits, acts, act_lens, nframes


class GRUGenerator(Generator):
    """Config C4: Generator (audiogan.py:361-468) with the LSTMCell of the frame loop replaced by
    torch.nn.GRUCell.  UNPINNED: the reference has no GRU anywhere."""

    def __init__(self, frame_size=200, embed_size=200, noise_size=100, state_size=1024,
                 struct=((17, 8, 128, 16), (9, 4, 64, 32), (9, 4, 64, 32), (9, 4, 32, 32))):
        super().__init__(frame_size, embed_size, noise_size, state_size, 1, struct)
        self.rnn = nn.ModuleList([Replicated(weight_norm(
            nn.GRUCell(frame_size + embed_size + noise_size, state_size),
            ['weight_ih', 'weight_hh', 'bias_hh', 'bias_ih']))])

    def forward(self, batch_size=None, length=None, z=None, c=None, stop=None):
        fs, ns, ss, es = self._frame_size, self._noise_size, self._state_size, self._embed_size
        batch_size, nframes, _ = z.size()
        zc = torch.cat([z, c.unsqueeze(1).expand(batch_size, nframes, es)], 2)
        h = torch.zeros(batch_size, ss)
        x_t = torch.zeros(batch_size, fs)
        xs, logits = [], []
        for t in range(nframes):
            h = self.rnn[0](torch.cat([x_t, zc[:, t]], 1), h)
            x_t = torch.tanh(self.proj(h))
            xs.append(x_t)
            logits.append(self.stopper(h).squeeze(1))
        x = torch.cat(xs, 1).unsqueeze(1)
        for layer in self.dense_res_gen:
            x_next = layer(x)
            x = torch.cat([x, x_next], 1)
        return (x_next.squeeze(1), torch.stack(logits, 1), None,
                torch.full((batch_size,), nframes * fs, dtype=torch.long))


# --------------------------------------------------------------------------
# C1 "tiny conv" G/D.  Spec: modeltf.py:256-286 and :578-595 restated in PyTorch
# conventions.  UNPINNED: TensorFlow-only in the
# reference, not executable here.
# --------------------------------------------------------------------------
class Conv1DGenerator(nn.Module):
    """z (B, L/prod(stride)) -> N x [ConvTranspose1d 'same' -> LeakyReLU] -> 1x1 conv -> tanh."""

    def __init__(self, config=((16, 5, 2), (16, 5, 2), (8, 5, 2))):
        super().__init__()
        self.config = [tuple(c) for c in config]
        self.multiplier = int(np.prod([s for _, _, s in self.config]))
        self.deconvs = nn.ModuleList()
        cin = 1
        for nf, k, s in self.config:
            # 'same' transposed conv: Lout = Lin*s  <=>  k - 2p + output_padding = s
            p = (k - s + 1) // 2
            op = s - (k - 2 * p)
            self.deconvs.append(nn.ConvTranspose1d(cin, nf, k, s, padding=p, output_padding=op))
            cin = nf
        self.out = nn.Conv1d(cin, 1, 1)

    def forward(self, batch_size=None, length=None, z=None):
        if z is None:
            z = torch.randn(batch_size, length // self.multiplier)
        x = z.unsqueeze(1)
        for d in self.deconvs:
            x = F.leaky_relu(d(x))
        return torch.tanh(self.out(x)).squeeze(1)


class Conv1DDiscriminator(nn.Module):
    """N x [Conv1d k,s 'same' -> LeakyReLU] -> global average pool -> Linear(1)."""

    def __init__(self, config=((8, 5, 2), (16, 5, 2), (16, 5, 2))):
        super().__init__()
        self.config = [tuple(c) for c in config]
        self.convs = nn.ModuleList()
        cin = 1
        for nf, k, s in self.config:
            self.convs.append(nn.Conv1d(cin, nf, k, s, padding=(k - 1) // 2))
            cin = nf
        self.dense = nn.Linear(cin, 1)

    def forward(self, x, c=None):
        a = x.unsqueeze(1)
        for conv in self.convs:
            a = F.leaky_relu(conv(a))
        return self.dense(a.mean(2))[:, 0]


# --------------------------------------------------------------------------
# feature statistics of the reference's current G step (audiogan.py:336-359, :850-855)
# --------------------------------------------------------------------------
def fourth_moment(v):
    """audiogan.py:336-339; Python-2 ``(1/4)`` is 0, so this is x**0."""
    return (((v - v.mean(0).unsqueeze(0)) ** 4).sum(0)) ** 0


def calc_dists(hidden_states, hidden_state_lengths):
    """audiogan.py:341-359."""
    means_d, stds_d, fourth_d = [], [], []
    for h, l in zip(hidden_states, hidden_state_lengths):
        mask = length_mask((h.size(0), h.size(2)), l)
        lf = l.unsqueeze(1).float()
        m = h.sum(2) / lf
        cen = h - m.unsqueeze(2) * mask.unsqueeze(1)
        s = (cen ** 2).sum(2) ** 0.5 / lf
        f = (cen ** 4).sum(2) ** 0.25 / lf
        means_d += [(m.mean(0), m.std(0)), (s.mean(0), s.std(0)), (f.mean(0), f.std(0))]
        stds_d += [(m.std(0), m.std(0)), (s.std(0), s.std(0)), (f.std(0), f.std(0))]
        fourth_d += [(fourth_moment(m), m.std(0)), (fourth_moment(s), s.std(0)), (fourth_moment(f), f.std(0))]
    return means_d + stds_d + fourth_d


def feature_penalty(dists_d, dists_g, batch_size):
    """audiogan.py:850-855."""
    pen = 0
    for r, f in zip(dists_d, dists_g):
        pen = pen + torch.pow(r[0] - f[0], 2).mean() / batch_size
    return pen


# --------------------------------------------------------------------------
# canonical G+D step: audiogan.py:706-788 and :816-921
# minus the out-of-scope extras (FGSM passes, feature matching, REINFORCE,
# logging).  All stochastic inputs are passed in.
# --------------------------------------------------------------------------
def make_optimizer(params, kind, lr):
    if kind == 'rmsprop':  # audiogan.py:693-694
        return torch.optim.RMSprop(params, lr=lr)
    if kind == 'adam':     # computation_graph.py:58-59 (TF defaults; unpinned)
        return torch.optim.Adam(params, lr=lr, betas=(0.9, 0.999), eps=1e-8)
    raise ValueError(kind)


def d_step(g, d, opt_d, real, real_len, c, z, noise_real, noise_fake, dgradclip=1.0, stop=None):
    """One critic iteration of the classic step: audiogan.py:723-728, 739-740,
    748-751, 761-766, 780-788."""
    params = [p for p in d.parameters()]
    with torch.no_grad():
        fake, _, _, fake_len = g(z=z, c=c, stop=stop)
        fake = fake + noise_fake
    cls_d, _, _, nf_d = d(real + noise_real, real_len, c)
    w = length_mask(cls_d.size(), nf_d)
    loss_d = (binary_cross_entropy_with_logits_per_sample(
        cls_d, torch.full_like(cls_d, 0.9), weight=w) / nf_d.float()).mean()
    cls_g, _, _, nf_g = d(fake, fake_len, c)
    w = length_mask(cls_g.size(), nf_g)
    loss_g = (binary_cross_entropy_with_logits_per_sample(
        cls_g, torch.zeros_like(cls_g), weight=w) / nf_g.float()).mean()
    loss = loss_d + loss_g
    opt_d.zero_grad()
    loss.backward()
    check_grad(params)
    clip_grad(params, dgradclip)
    opt_d.step()
    return loss.detach(), cls_d.detach(), cls_g.detach()


def g_step(g, d, opt_g, c, z, noise_fake, ggradclip=0.1, g_optim='boundary_seeking', stop=None):
    """One generator iteration of the classic step: audiogan.py:841-845, 857-864,
    897, 902-903, 909-921."""
    params = [p for p in g.parameters()]
    req = [p.requires_grad for p in d.parameters()]
    for p in d.parameters():
        p.requires_grad_(False)
    fake, _, _, fake_len = g(z=z, c=c, stop=stop)
    cls_g, _, _, nf_g = d(fake + noise_fake, fake_len, c)
    tgt = torch.full_like(cls_g, 0.5) if g_optim == 'boundary_seeking' else torch.zeros_like(cls_g)
    w = length_mask(cls_g.size(), nf_g)
    loss = (binary_cross_entropy_with_logits_per_sample(cls_g, tgt, weight=w) / nf_g.float()).mean()
    opt_g.zero_grad()
    loss.backward()
    for p, r in zip(d.parameters(), req):
        p.requires_grad_(r)
    check_grad(params)
    clip_grad(params, ggradclip)
    opt_g.step()
    return loss.detach(), fake.detach(), cls_g.detach()


# --------------------------------------------------------------------------
# WGAN-GP (config C5).  Spec: modeltf.py:460-469, utiltf.py:43-44,60-61,
# computation_graph.py:101-104, lambda = 10 (maintf.py:201).  UNPINNED (TF-only).
# --------------------------------------------------------------------------
def grad_penalty(critic, x_real, x_fake, eps):
    """eps (B,1) ~ U(0,1); penalty_b = (||dD/dx_hat||_2 over time - 1)^2."""
    x_hat = (eps * x_real + (1 - eps) * x_fake).detach().requires_grad_(True)
    d_hat = critic(x_hat)
    grads, = torch.autograd.grad(d_hat.sum(), x_hat, create_graph=True)
    return (grads.pow(2).sum(1).sqrt() - 1).pow(2)


def wgan_gp_d_loss(critic, x_real, x_fake, eps, lam=10.0):
    comp = (critic(x_fake) - critic(x_real)).mean()
    return comp + lam * grad_penalty(critic, x_real, x_fake, eps).mean()


def wgan_g_loss(critic, x_fake):
    return (-critic(x_fake)).mean()


# --------------------------------------------------------------------------
# synthetic inputs
# --------------------------------------------------------------------------
def synthetic_clips(batch, length, kind, seed=0):
    """kind 'sine': sin(2 pi f t / 8000), f ~ U(100,1000); 'noise': U(-1,1);
    both peak-normalised like dataset.py:68-71.  float64 like NP.zeros(maxlen), :50."""
    rs = np.random.RandomState(seed)
    if kind == 'sine':
        f = rs.uniform(100, 1000, size=(batch, 1))
        x = np.sin(2 * np.pi * f * np.arange(length)[None, :] / 8000.0)
    else:
        x = rs.uniform(-1, 1, size=(batch, length))
    x = x / np.abs(x).max(1, keepdims=True)
    return x


# --------------------------------------------------------------------------
# The reference's CURRENT iterations (audiogan.py:706-788 critic, :816-921 generator): the classic step plus
# the FGSM-style passes (:99-150), the feature-matching penalty (:847-855) and the REINFORCE update of the stop
# head (:444-460, :866-908).  Everything random (instance noise, z, stop draws) is an argument.  Not restated:
# TensorBoard / print / gc (:713, :767-811), data loading (:714-716, :823-828).
# --------------------------------------------------------------------------
def _masked_bce(cls, target, nframes):
    w = length_mask(cls.size(), nframes)
    return binary_cross_entropy_with_logits_per_sample(cls, torch.full_like(cls, target), weight=w) / nframes.float()


def adversarial_movement_d(data, data_len, embed_d, target, d, scale=1e-3):
    """audiogan.py:139-150: +-scale along the sign of d(per-sample loss)/d(input)"""
    data = data.detach().requires_grad_(True)
    cls, _, _, nframes = d(data, data_len, embed_d)
    loss = _masked_bce(cls, target, nframes)
    grad, = torch.autograd.grad(loss, data, grad_outputs=torch.ones_like(loss))
    return (grad > 0).float() * scale - (grad < 0).float() * scale


def adversarially_sample_z(g, d, z, embed_g, embed_d, noise, g_optim='boundary_seeking', scale=1e-2, stop=None):
    """audiogan.py:99-137 with the draw of z (:101), of the instance noise (:104) and of the stop decisions injected;
    its feature-penalty lines (:109-117) never reach the returned z and are left out"""
    z = z.detach().requires_grad_(True)
    fake, _, _, fake_len = g(z=z, c=embed_g, stop=stop)
    cls_g, _, _, nframes_g = d(fake + noise[:, :fake.size(1)], fake_len, embed_d)
    loss = _masked_bce(cls_g, 0.5 if g_optim == 'boundary_seeking' else 0.0, nframes_g)
    grad, = torch.autograd.grad(loss, z, grad_outputs=torch.ones_like(loss))
    advers = (grad > 1e-9).float() * scale - (grad < -1e-9).float() * scale
    return (z + advers).detach()


def stopper_surrogate_loss(stop_logits, stops, reward):
    """-sum reward * log p(stop draw): its gradient is what ``stop_t.reinforce(reward)`` + backward produced
    (audiogan.py:444-451, :898-903)"""
    stops = stops[:, :stop_logits.size(1)].float()
    logp = stops * F.logsigmoid(stop_logits) + (1.0 - stops) * F.logsigmoid(-stop_logits)
    return -(reward.detach() * logp).sum()


def d_step_full(g, d, e_g, e_d, opt_d, dis_iter, real, real_len, cs, cl, cs2, cl2, z, noise_real, noise_fake,
                dgradclip=1.0, stop=None):
    """critic iteration ``dis_iter`` (audiogan.py:706-788).  Even iterations add instance noise to the real and the
    generated clips (:724-728, :749-751); odd iterations feed the clean real clips (their FGSM perturbation at :735-736
    is computed AFTER ``cls_d`` and never used) and move the generated clips by +-1e-3 along the sign of the critic's
    input gradient (:752-759).  opt_d holds the parameters of d and e_d (:691)."""
    params = list(d.parameters()) + list(e_d.parameters())
    embed_real = e_d(cs, cl)
    even = dis_iter % 2 == 0
    cls_d, _, _, nf_d = d(real + noise_real if even else real, real_len, embed_real)
    loss_d = _masked_bce(cls_d, 0.9, nf_d).mean()
    w_d = length_mask(cls_d.size(), nf_d)
    with torch.no_grad():
        embed_g = e_g(cs2, cl2)
    embed_d = e_d(cs2, cl2)
    with torch.no_grad():
        fake, _, _, fake_len = g(z=z, c=embed_g, stop=stop)
    if even:
        fake = fake + noise_fake[:, :fake.size(1)]
    else:
        fake = fake + adversarial_movement_d(fake, fake_len, embed_d.detach(), 0.0, d)
    cls_g, _, _, nf_g = d(fake, fake_len, embed_d)
    loss_g = _masked_bce(cls_g, 0.0, nf_g).mean()
    w_g = length_mask(cls_g.size(), nf_g)
    loss = loss_d + loss_g
    opt_d.zero_grad()
    loss.backward()
    check_grad(params)
    gnorm = clip_grad(params, dgradclip)
    opt_d.step()
    return dict(loss=loss.detach(), loss_d=loss_d.detach(), loss_g=loss_g.detach(), cls_d=cls_d.detach(),
                cls_g=cls_g.detach(), grad_norm=gnorm,
                acc_d=float((((cls_d > 0).float() * w_d).sum() / w_d.sum())),
                acc_g=float((((cls_g < 0).float() * w_g).sum() / w_g.sum())))


def g_step_full(g, d, e_g, e_d, opt_g, real, real_len, cs, cl, z0, noise_real, noise_adv, noise_fake, stop_adv, stop,
                baseline=None, ggradclip=0.1, g_optim='boundary_seeking', lambda_fp=1.0):
    """generator iteration (audiogan.py:816-921): adversarial z (:836), generator + critic forward (:841-847), feature
    penalty against the critic's statistics on real clips (:847-855), BCE towards 0.5 (:857-864), reward / baseline
    (:873-887), loss + penalty (:897), REINFORCE of the stop draws into the stop head only (:898-908), per-parameter
    clip and step over the parameters of g and e_g (:909-921).  Returns the new baseline."""
    params = list(g.parameters()) + list(e_g.parameters())
    B = real.size(0)
    embed_g = e_g(cs, cl)
    with torch.no_grad():
        embed_d = e_d(cs, cl)
    z = adversarially_sample_z(g, d, z0, embed_g.detach(), embed_d, noise_adv, g_optim, 1e-2, stop=stop_adv)
    fake, s, stop_list, fake_len = g(z=z, c=embed_g, stop=stop)
    fake = fake + noise_fake[:, :fake.size(1)]
    cls_g, hs_g, hl_g, nf_g = d(fake, fake_len, embed_d)
    with torch.no_grad():
        _, hs_d, hl_d, _ = d(real + noise_real, real_len, embed_d)
    pen = feature_penalty(calc_dists(hs_d, hl_d), calc_dists(hs_g, hl_g), B)
    loss_ps = _masked_bce(cls_g, 0.5 if g_optim == 'boundary_seeking' else 0.0, nf_g)
    reward = -loss_ps.detach()
    baseline = float(reward.mean()) if baseline is None else baseline * 0.5 + float(reward.mean()) * 0.5
    frames = fake_len // g._frame_size
    weight_r = length_mask((B, int(frames.max())), frames)
    reward = (reward - baseline).unsqueeze(1) * weight_r
    loss = loss_ps.mean() + pen * lambda_fp
    opt_g.zero_grad()
    loss.backward(retain_graph=True)
    flags = [p.requires_grad for p in params]
    keep = set(id(p) for p in g.stopper.parameters())
    for p in params:
        p.requires_grad_(id(p) in keep)
    stops = torch.cat([t.view(B, 1) for t in stop_list], 1)
    stopper_surrogate_loss(s, stops, reward).backward()
    for p, f in zip(params, flags):
        p.requires_grad_(f)
    check_grad(params)
    gnorm = clip_grad(params, ggradclip)
    opt_g.step()
    return dict(loss=loss.detach(), bce=loss_ps.mean().detach(), feature_penalty=pen.detach(), z=z, fake=fake.detach(),
                fake_len=fake_len, s=s.detach(), baseline=baseline, grad_norm=gnorm)


# --------------------------------------------------------------------------
# BASELINE configs[3] / [4] steps with a whole-clip conv critic (Conv1DDiscriminator with the a4 struct): the CPU
# statements of audiogan_amd.train.c4_step / wgan_gp_step.  Parity unpinned (no GRU / WGAN-GP in the PyTorch reference).
# --------------------------------------------------------------------------
def _bce_mean(logits, target):
    return binary_cross_entropy_with_logits_per_sample(logits, torch.full_like(logits, target)).mean()


def c4_step(g, critic, opt_g, opt_d, real, c, z, noise_real, noise_fake, dgradclip=1.0, ggradclip=0.1, stop=None):
    B = real.size(0)
    with torch.no_grad():
        fake = g(z=z, c=c, stop=stop)[0] + noise_fake
    loss_d = _bce_mean(critic(real + noise_real).view(B, 1), 0.9) + _bce_mean(critic(fake).view(B, 1), 0.0)
    opt_d.zero_grad()
    loss_d.backward()
    clip_grad(list(critic.parameters()), dgradclip)
    opt_d.step()
    fake = g(z=z, c=c, stop=stop)[0]
    loss_g = _bce_mean(critic(fake + noise_fake).view(B, 1), 0.5)
    opt_g.zero_grad()
    loss_g.backward()
    clip_grad(list(g.parameters()), ggradclip)
    opt_g.step()
    return loss_d.detach(), loss_g.detach()


def wgan_gp_step(g, critic, opt_g, opt_d, real, c, z, eps, lam=10.0, stop=None):
    with torch.no_grad():
        fake = g(z=z, c=c, stop=stop)[0]
    loss_d = wgan_gp_d_loss(critic, real, fake, eps, lam)
    opt_d.zero_grad()
    loss_d.backward()
    opt_d.step()
    loss_g = wgan_g_loss(critic, g(z=z, c=c, stop=stop)[0])
    opt_g.zero_grad()
    loss_g.backward()
    opt_g.step()
    return loss_d.detach(), loss_g.detach()
