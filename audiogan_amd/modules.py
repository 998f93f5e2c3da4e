"""Drop-in ``Generator`` / ``Discriminator`` / ``Embedder`` modules.

Same constructor arguments, ``forward`` signatures, return values and ``state_dict`` keys
as the reference classes (audiogan.py:302-334, 361-468, 471-551), so a checkpoint of the
reference's parameters loads with ``strict=True``.  What runs underneath is the fused HIP
blocks of ``audiogan_amd.ops``; no stock torch compute op (conv, linear, LSTM) is called.
"""
import torch
import torch.nn as nn

from . import ops
from .common import prepare_groups
from .ops import ConvSpec


def div_roundup(x, d):
    """audiogan.py:172-173 (integer semantics of Python 2)."""
    return (x + d - 1) // d


def roundup(x, d):
    """audiogan.py:174-175."""
    return (x + d - 1) // d * d


class Replicated(nn.Module):
    """Holder that keeps the ``.module.`` level the reference's per-submodule
    ``NN.DataParallel`` wrappers put into ``state_dict`` keys (audiogan.py:379-410, 492-508).
    Data parallelism itself is process-per-GPU (audiogan_amd.ddp), not this wrapper."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, *a, **k):
        return self.module(*a, **k)


def _register_wn(mod, name, w):
    """weight_norm reparameterisation of tensor ``w`` (dim 0): g = ||w|| per row, v = w.
    Registers ``<name>_g`` then ``<name>_v`` like torch.nn.utils.weight_norm (audiogan.py:77-80)."""
    w = w.detach()
    if w.dim() == 1:
        g = w.abs()
    else:
        g = w.reshape(w.size(0), -1).norm(dim=1).view(w.size(0), *([1] * (w.dim() - 1)))
    mod.register_parameter(name + '_g', nn.Parameter(g.clone()))
    mod.register_parameter(name + '_v', nn.Parameter(w.clone()))


class _WNModule(nn.Module):
    """parameter container for one weight-normed layer; ``names`` in reference order"""

    def __init__(self, proto, names):
        super().__init__()
        for n in names:
            _register_wn(self, n, getattr(proto, n))

    def wn(self, name):
        return getattr(self, name + '_v'), getattr(self, name + '_g')


class WNConv1d(_WNModule):
    def __init__(self, cin, cout, k, stride, padding):
        super().__init__(nn.Conv1d(cin, cout, k, stride=stride, padding=padding), ['weight', 'bias'])
        self.spec = ConvSpec('conv', cin, cout, k, stride, padding)


class WNConvTranspose1d(_WNModule):
    def __init__(self, cin, cout, k, stride, padding):
        super().__init__(nn.ConvTranspose1d(cin, cout, k, stride, padding=padding), ['weight', 'bias'])
        self.spec = ConvSpec('convT', cin, cout, k, stride, padding)


class WNLinear(_WNModule):
    def __init__(self, fin, fout):
        super().__init__(nn.Linear(fin, fout), ['weight', 'bias'])


class WNLSTMCell(_WNModule):
    def __init__(self, fin, hidden):
        super().__init__(nn.LSTMCell(fin, hidden), ['weight_ih', 'weight_hh', 'bias_hh', 'bias_ih'])


class WNGRUCell(_WNModule):
    def __init__(self, fin, hidden):
        super().__init__(nn.GRUCell(fin, hidden), ['weight_ih', 'weight_hh', 'bias_hh', 'bias_ih'])


class Residual(nn.Module):
    """audiogan.py:256-264 (parameters only; computed inside ops.DHeadFn)."""

    def __init__(self, size):
        super().__init__()
        self.size = size
        self.linear = WNLinear(size, size)


class dense_res_bottleneck(nn.Module):
    """audiogan.py:266-283 (parameters only; computed inside ops.GTrunkFn)."""

    def __init__(self, kernel, stride, infilters, hidden_filters, outfilters):
        super().__init__()
        self.infilters = infilters
        self.outfilters = outfilters
        self.conv = WNConv1d(infilters, hidden_filters, kernel, stride, (kernel - 1) // 2)
        self.deconv = WNConvTranspose1d(hidden_filters, outfilters, kernel - 1, stride, stride // 2)


def _cumprod(xs):
    p = 1
    for v in xs:
        p *= v
        yield p


def _add(group, mod, name, **kw):
    v, g = mod.wn(name)
    group.add(v, g, **kw)


class Generator(nn.Module):
    """audiogan.py:361-468.  ``stop``: None -> Bernoulli stop draws like the reference
    (:444-460); a [B,T] integer tensor -> use these draws; 'never' -> fixed-length clips
    (no draw, no host sync; what the bench uses)."""

    def __init__(self, frame_size=200, embed_size=200, noise_size=100, state_size=1024,
                 num_layers=1,
                 struct=((17, 8, 128, 16), (9, 4, 64, 32), (9, 4, 64, 32), (9, 4, 32, 32))):
        super().__init__()
        self._frame_size = frame_size
        self._noise_size = noise_size
        self._state_size = state_size
        self._embed_size = embed_size
        self._num_layers = num_layers
        self.rnn = nn.ModuleList()
        self.rnn.append(Replicated(WNLSTMCell(frame_size + embed_size + noise_size, state_size)))
        for _ in range(1, num_layers):
            self.rnn.append(Replicated(WNLSTMCell(state_size, state_size)))
        self.dense_res_gen = nn.ModuleList()
        cin = 1
        for kernel, stride, hidden, cout in struct:
            self.dense_res_gen.append(Replicated(dense_res_bottleneck(kernel, stride, cin, hidden, cout)))
            cin += cout
        self.dense_res_gen.append(Replicated(WNConv1d(cin, 1, 3, 1, 1)))
        self.proj = Replicated(WNLinear(state_size, frame_size))
        self.stopper = Replicated(WNLinear(state_size, 1))

        # fused-block descriptions (share the Parameters above)
        self._front = ops.GFront(frame_size, num_layers, state_size)
        for cell in self.rnn:
            for n in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh'):
                _add(self._front.group, cell.module, n)
        for lin in (self.proj.module, self.stopper.module):
            _add(self._front.group, lin, 'weight')
            _add(self._front.group, lin, 'bias')
        bn = [(m.module.conv.spec, m.module.deconv.spec) for m in self.dense_res_gen[:-1]]
        self._trunk = ops.GTrunk(bn, self.dense_res_gen[-1].module.spec)
        for m in self.dense_res_gen[:-1]:
            for layer in (m.module.conv, m.module.deconv):
                _add(self._trunk.group, layer, 'weight', stride=layer.spec.stride, engine=True, pad=layer.spec.pad)
                _add(self._trunk.group, layer, 'bias')
        fin = self.dense_res_gen[-1].module
        _add(self._trunk.group, fin, 'weight', stride=1, engine=True)
        _add(self._trunk.group, fin, 'bias')
        # the front writes its frames straight into channel 0 of the trunk's activation slab (no copy between the blocks)
        self._front.slab_channels = self._trunk.ctot

    def _front_apply(self, zc):
        return ops.GFrontFn.apply(zc, self._front, *self._front.group.params())

    def front_is_persistent(self, batch_size, dev):
        """does the frame loop of a forward at this batch size run as ONE persistent launch (all CUs, one such launch at
        a time per device)?  Callers that want to run two forwards side by side on two streams must not when it does."""
        from . import kernels as K
        # (LSTM front: single layer only; the GRU front of GRUGenerator is always one layer - both take the same launch)
        return self._num_layers == 1 and K.gfront_persist_ok(batch_size, self._state_size, self._frame_size, dev)

    def prepare_weights(self):
        """materialise the weight-normed weights of every block NOW, on the current stream (no-op when the
        parameters have not changed since the last materialisation).  A caller that runs two forwards of this
        module on two streams calls it before the fork: otherwise the first forward to be ENQUEUED would rewrite
        the shared weight buffers on its stream while the other stream's forward (which finds the cache key
        up to date and skips the rewrite) reads them unordered."""
        prepare_groups([self._front.group, self._trunk.group])      # ONE launch for both blocks

    def early_params(self):
        """parameters whose gradients are complete BEFORE the recurrent front's backward runs (the conv trunk,
        which the backward pass reaches first): their all-reduce can overlap the front's frame loop"""
        return [p for m in self.dense_res_gen for p in m.parameters()]

    def forward(self, batch_size=None, length=None, z=None, c=None, stop=None, cut=None):
        """``cut``: optional dict.  When given, the autograd graph is cut between the recurrent front and the conv
        trunk: the trunk runs on a detached copy of the front's frames (``cut['x_cut']``, a leaf that collects the
        trunk's input gradient) and ``cut['x']`` keeps the front's output, so that the caller can run the front's
        backward later with ``cut['x'].backward(cut['x_cut'].grad)`` (train.g_backward_early / _late)."""
        fs, ns, es = self._frame_size, self._noise_size, self._embed_size
        dev = c.device
        self.prepare_weights()
        if z is None:
            nframes = div_roundup(length, fs)
            z = torch.randn(batch_size, nframes, ns, device=dev)
        else:
            batch_size, nframes, _ = z.size()
        if z.is_cuda and z.dtype == torch.float32 and c.dtype == torch.float32:
            zc = ops.BuildZCFn.apply(z, c)          # [T,B,noise+embed] in one launch
        else:
            zc = torch.cat([z, c.unsqueeze(1).expand(batch_size, nframes, es)], 2).transpose(0, 1).contiguous()
        x, s = self._front_apply(zc)

        if isinstance(stop, str):
            assert stop == 'never'
            t_eff = nframes
            stops = None
            out_len, never = _never_stop_constants(batch_size, nframes, fs, dev)
            if not (dev.type == 'cuda' and torch.cuda.is_current_stream_capturing()):
                out_len = out_len.clone()       # eager callers get their own copy (an in-place edit must not reach the cache)
        else:
            if stop is None:
                stops = torch.bernoulli(torch.sigmoid(s.detach())).long()
            else:
                stops = stop.to(dev).long()
            # frames generated by sample b: up to and including its first stop draw (:451-458);
            # the loop itself runs until every sample has stopped (:459-460)
            first = torch.where(stops.bool().any(1), stops.argmax(1) + 1,
                                torch.full_like(stops[:, 0], nframes))
            t_eff = int(first.max())          # host sync, as the reference's generating.sum()==0
            out_len = first * fs
        if t_eff < nframes:
            x, s = x[:, :t_eff * fs], s[:, :t_eff]
        stop_list = [stops[:, t:t + 1] for t in range(t_eff)] if stops is not None else list(never)
        if cut is not None:
            cut['x'] = x
            x = cut['x_cut'] = x.detach().requires_grad_(True)
        wave = ops.GTrunkFn.apply(x, self._trunk, *self._trunk.group.params())
        return wave, s, stop_list, out_len


_NEVER = {}


def _never_stop_constants(batch_size, nframes, fs, dev):
    """what ``stop='never'`` returns besides the audio: the full length per clip and the all-zero stop draws.  Constants of
    (batch, frames, frame size, device), built once (two fills per forward otherwise) - treat them as read-only."""
    key = (int(batch_size), int(nframes), int(fs), str(dev))
    if key not in _NEVER:
        _NEVER[key] = (torch.full((batch_size,), nframes * fs, dtype=torch.long, device=dev),
                       tuple(torch.zeros(nframes, batch_size, 1, dtype=torch.long, device=dev).unbind(0)))
    return _NEVER[key]


class Discriminator(nn.Module):
    """audiogan.py:471-551."""

    def __init__(self, state_size=1024, embed_size=200, num_layers=1,
                 cnn_struct=((7, 2, 16), (7, 2, 32), (7, 2, 64), (7, 2, 128), (7, 2, 256), (7, 2, 512))):
        super().__init__()
        self._state_size = state_size
        self._embed_size = embed_size
        self._num_layers = num_layers
        self.cnn_struct = [list(l) for l in cnn_struct]
        self._cnn_struct = self.cnn_struct
        self.cnn = nn.ModuleList()
        cin = 1
        for kernel, stride, cout in self.cnn_struct:
            self.cnn.append(Replicated(WNConv1d(cin, cout, kernel, stride, (kernel - 1) // 2)))
            cin = cout
        self.frame_size = self._frame_size = cin
        # parameter container only (names/initialisation of NN.LSTM, :498-503); never called
        self.rnn = nn.LSTM(cin + embed_size, state_size // 2, num_layers, bidirectional=True)
        self.residual_net = Replicated(nn.Sequential(Residual(state_size), Residual(state_size)))
        self.classifier = Replicated(nn.Sequential(
            WNLinear(state_size, state_size // 2), nn.LeakyReLU(), WNLinear(state_size // 2, 1)))

        self._stack = ops.DConvStack([m.module.spec for m in self.cnn])
        for m in self.cnn:
            _add(self._stack.group, m.module, 'weight', stride=m.module.spec.stride, engine=True,
                 pad=m.module.spec.pad)
            _add(self._stack.group, m.module, 'bias')
        self._head = ops.DHead(2)
        for r in self.residual_net.module:
            _add(self._head.group, r.linear, 'weight')
            _add(self._head.group, r.linear, 'bias')
        for i in (0, 2):
            _add(self._head.group, self.classifier.module[i], 'weight')
            _add(self._head.group, self.classifier.module[i], 'bias')

    def _stride_prods(self, dev):
        # built once per device (a host->device copy inside forward would break hipGraph capture)
        t = getattr(self, '_prods', None)
        if t is None or t[0].device != dev:
            p = torch.tensor(list(_cumprod([s for _, s, _ in self.cnn_struct])), device=dev).view(-1, 1)
            t = self._prods = (p, p - 1)
        return t

    def _rnn_weights(self, layer):
        out = []
        for suffix in ('', '_reverse'):
            for n in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh'):
                out.append(getattr(self.rnn, '%s_l%d%s' % (n, layer, suffix)))
        return out

    def stride_products(self):
        """s_1, s_1 s_2, ...: the length after conv layer i is ceil(length / (s_1 ... s_i)) (audiogan.py:533)"""
        return list(_cumprod([s for _, s, _ in self.cnn_struct]))

    def features(self, x, length, lens_all=None):
        """the conv stack of forward() (audiogan.py:529-536): (activations, their lengths).  ``lens_all``: the
        [n_layers, B] int64 table of the clips' lengths after every layer when the caller already has it (the one-launch
        input assembly of train.d_step / g_step, ops.CriticInputFn); ``length`` is then not read."""
        if lens_all is None:
            length = length.to(x.device).long()
            # nframes after layer i = ceil(... ceil(length / s_1) ... / s_i) = ceil(length / (s_1 ... s_i)):
            # all layers in one broadcast op instead of one tiny kernel pair per layer (:533)
            prods, prods_m1 = self._stride_prods(x.device)
            lens_all = torch.div(length.view(1, -1) + prods_m1, prods, rounding_mode='floor')
        lens_list = [lens_all[i] for i in range(lens_all.size(0))]
        acts = ops.DConvStackFn.apply(x, self._stack, lens_list, *self._stack.group.params())
        return acts, lens_list

    def stores_bf16(self, batch, dev):
        """does classify() at this batch size keep its sequence path as bfloat16 in HBM (kernels.bf16_storage: 'bf16'
        precision mode, AG_BF16_STORE != 0, and the persistent recurrent launches - which write / read those tensors in that
        type - fit)?  The oracle's ``bf16_mode(store=...)`` states the same contract."""
        K = ops.K
        dev = torch.device(dev)
        h = self._state_size // 2
        return bool(dev.type == 'cuda' and K.bf16_storage() and K.lstm_step_ok(batch, h) and
                    K.lstm_persist_ok(batch, h, 2, dev) and K.lstm_persist_bwd_ok(batch, h, 2, dev))

    def classify(self, a, n, c):
        """the rest of forward() (audiogan.py:537-549): biLSTM over [conv features, c] + heads -> logits"""
        ss, es = self._state_size, self._embed_size
        b, tq = a.size(0), a.size(2)
        # bf16 storage (kernels.bf16_storage: 'bf16' precision mode): the sequence path - time-major features, biLSTM
        # output, every activation of the heads and all their gradients - lives in HBM as bfloat16; it needs the persistent
        # recurrent launches, which write / read those tensors in that type
        s16 = self.stores_bf16(b, a.device)
        # layer 0 sees cat([features_t, c]) at every frame (:541): c goes in as the layer's time-invariant input
        seq = ops.TimeMajorFn.apply(a, s16)               # [T',B,C], contiguous
        for layer in range(self._num_layers):
            seq = ops.LSTMSeqFn.apply(seq, n, 2, c if layer == 0 else None, *self._rnn_weights(layer))
        # the heads are per-row: keep the LSTM's (time, clip) row order and transpose only the logits
        logits = ops.DHeadFn.apply(seq.reshape(tq * b, ss), self._head, *self._head.group.params())
        return logits.view(tq, b).t()       # (fp32 also on bf16 storage)

    def early_params(self):
        """parameters whose gradients are complete BEFORE the conv stack's backward runs (heads + biLSTM):
        their all-reduce can overlap that backward"""
        conv = set(id(p) for p in self.cnn.parameters())
        return [p for p in self.parameters() if id(p) not in conv]

    def prepare_weights(self):
        """materialise the weight-normed weights of the conv stack and the heads with ONE launch (no-op when they are up to
        date); features() / classify() called on their own prepare their block at its first use"""
        prepare_groups([self._stack.group, self._head.group])

    def forward(self, x, length, c, percent_used=0.1, lens_all=None):
        self.prepare_weights()
        acts, lens_list = self.features(x, length, lens_all)
        n = lens_list[-1]
        return self.classify(acts[-1], n, c), list(acts), lens_list, n


class Embedder(nn.Module):
    """audiogan.py:302-334: char embedding -> biLSTM -> last hidden state of both directions."""

    def __init__(self, output_size=100, char_embed_size=50, num_layers=1, num_chars=256):
        super().__init__()
        self._output_size = output_size
        self._char_embed_size = char_embed_size
        self._num_layers = num_layers
        self.embed = Replicated(nn.Embedding(num_chars, char_embed_size))   # lookup = indexing
        self.rnn = nn.LSTM(char_embed_size, output_size // 2, num_layers, bidirectional=True)

    def forward(self, chars, length):
        nl, b, o = self._num_layers, chars.size(0), self._output_size
        dev = self.embed.module.weight.device
        length = length.to(dev).long()
        seq = self.embed.module.weight[chars.to(dev).long()].permute(1, 0, 2).contiguous()
        for layer in range(nl):
            w = []
            for suffix in ('', '_reverse'):
                for n in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh'):
                    w.append(getattr(self.rnn, '%s_l%d%s' % (n, layer, suffix)))
            seq = ops.LSTMSeqFn.apply(seq, length, 2, None, *w)
        h = o // 2
        # last hidden state: forward direction at t = len-1, reverse direction at t = 0
        idx = (length - 1).clamp(min=0)
        fwd = seq[idx, torch.arange(b, device=dev), :h]
        bwd = seq[0, :, h:]
        return torch.cat([fwd, bwd], 1)


class GRUGenerator(Generator):
    """BASELINE config C4: the Generator of audiogan.py:361-468 with the LSTMCell of its frame loop
    (:377-386, :437-443) replaced by a GRU cell (one layer).  The reference contains no GRU
    (SURVEY.md F5): parity is against torch.nn.GRUCell via oracle.GRUGenerator, not the reference."""

    def __init__(self, frame_size=200, embed_size=200, noise_size=100, state_size=1024,
                 struct=((17, 8, 128, 16), (9, 4, 64, 32), (9, 4, 64, 32), (9, 4, 32, 32))):
        super().__init__(frame_size, embed_size, noise_size, state_size, 1, struct)
        self.rnn = nn.ModuleList([Replicated(WNGRUCell(frame_size + embed_size + noise_size, state_size))])
        self._front = ops.GRUFront(frame_size, state_size)
        for n in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh'):
            _add(self._front.group, self.rnn[0].module, n)
        for lin in (self.proj.module, self.stopper.module):
            _add(self._front.group, lin, 'weight')
            _add(self._front.group, lin, 'bias')
        self._front.slab_channels = self._trunk.ctot

    def _front_apply(self, zc):
        return ops.GRUFrontFn.apply(zc, self._front, *self._front.group.params())
