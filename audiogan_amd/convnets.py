"""Conv-only generator / critic variants on the same HIP conv engine.

  Conv1DGenerator / Conv1DDiscriminator   BASELINE config C1 ("tiny 3-layer conv G + D"): the shape of
        modeltf.py:256-286 / :578-595 restated in PyTorch conventions
  ConvPoolCritic                          a4 conv stack + masked average pool + Linear(1): the critic of
        configs C4/C5 ("D reduced to a4 + pooling head")
  wgan_gp_d_loss / wgan_g_loss            WGAN-GP (modeltf.py:460-469, utiltf.py:43-44,60-61,
        computation_graph.py:101-104, lambda 10): the gradient penalty's DOUBLE backward is written
        out by hand in CriticGPFn -- no autograd-of-autograd, only conv-engine / wgrad launches.
"""
import numpy as np
import torch
import torch.nn as nn

from . import kernels as K
from . import ops
from .common import PARAM_EPOCH, Prepared, _zeros_like_list, param_epoch  # noqa: F401
from .kernels import ACT_NONE, ACT_LEAKY, ACT_TANH
from .ops import ConvSpec, conv_fwd, conv_bwd_data, conv_wgrad


class PlainGroup(object):
    """engine layouts of plain (not weight-normed) conv weights, cached per parameter version"""

    def __init__(self):
        self.items = []      # dict(w=Parameter, b=Parameter, stride=int)
        self._bufs, self._key = None, None

    def add(self, w, b, stride):
        self.items.append(dict(w=w, b=b, stride=stride))

    def params(self):
        out = []
        for it in self.items:
            out += [it['w'], it['b']]
        return out

    def prepare(self):
        dev = self.items[0]['w'].device
        if self._bufs is None or self._bufs[0].wpa.device != dev:
            self._bufs = []
            for it in self.items:
                d0, d1, kk = it['w'].shape
                self._bufs.append(Prepared(w=None, wpa=torch.zeros(K.wpa_numel(d0, d1, kk), device=dev),
                                           wpb=torch.zeros(K.wpb_numel(d0, d1, kk, it['stride']), device=dev)))
            self._key = None
        from .common import capture_tag
        key = (capture_tag(dev),) + tuple((it['w'].data_ptr(), it['w']._version, param_epoch(it['w'])) for it in self.items)
        if key != self._key:
            for it, p in zip(self.items, self._bufs):
                K.prep_conv_weight(it['w'].data.contiguous(), p.wpa, p.wpb, it['stride'])
            self._key = key
        return self._bufs


class ConvChain(object):
    """static description: list of (ConvSpec, act, out_len_fn)"""

    def __init__(self, layers):
        self.layers = layers          # list of (ConvSpec, act)
        self.group = PlainGroup()


def _act_bwd_bcl(d, y, act):
    """in place: d <- d * act'(.) from the saved output y ([B,C,L] contiguous)"""
    if act == ACT_LEAKY:
        K.leaky_bwd(d, y, d)
    elif act == ACT_TANH:
        K.act_bwd(d.view(-1), y.view(-1), d.view(-1), ACT_TANH)


class ConvChainFn(torch.autograd.Function):
    """x [B,C0,L0] -> act_n(conv_n(... act_1(conv_1(x))));  params = w1, b1, w2, b2, ..."""

    @staticmethod
    def forward(ctx, x, chain, *params):
        prep = chain.group.prepare()
        a = x.contiguous()
        acts = []
        for i, (sp, act) in enumerate(chain.layers):
            lo = sp.out_len(a.size(2))
            y = torch.empty(a.size(0), sp.cout, lo, device=a.device)
            conv_fwd(sp, prep[i], a, y, bias=params[2 * i + 1].data, act=act)
            acts.append(y)
            a = y
        ctx.chain, ctx.key = chain, chain.group._key[1:]
        ctx.save_for_backward(x, *acts)
        return a

    @staticmethod
    def backward(ctx, dy):
        chain = ctx.chain
        x, acts = ctx.saved_tensors[0], ctx.saved_tensors[1:]
        prep = chain.group.prepare()
        assert chain.group._key[1:] == ctx.key, 'parameters changed between forward and backward'
        wg = any(ctx.needs_input_grad[2:])
        d = dy.contiguous().clone()
        grads = []
        for i in reversed(range(len(chain.layers))):
            sp, act = chain.layers[i]
            _act_bwd_bcl(d, acts[i], act)
            xin = acts[i - 1] if i > 0 else x.contiguous()
            if wg:
                dw = torch.zeros_like(chain.group.items[i]['w'].data)
                db = torch.zeros_like(chain.group.items[i]['b'].data)
                conv_wgrad(sp, xin, d, dw, db)
                grads = [dw, db] + grads
            else:
                grads = [None, None] + grads
            if i > 0 or ctx.needs_input_grad[0]:
                dx = torch.empty_like(xin)
                conv_bwd_data(sp, prep[i], d, dx)
                d = dx
        return (d if ctx.needs_input_grad[0] else None, None) + tuple(grads)


class LinearFn(torch.autograd.Function):
    """y = x @ w^T + b on the MFMA GEMM"""

    @staticmethod
    def forward(ctx, x, w, b):
        x = x.contiguous()
        y = torch.empty(x.size(0), w.size(0), device=x.device)
        wc = w.data.contiguous()            # w may be a transposed / permuted view of the parameter
        K.gemm(x, wc, y, tb=True, bias=b.data.contiguous() if b is not None else None)
        ctx.save_for_backward(x, wc)
        ctx.has_b = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            K.gemm(dy, w.contiguous(), dx)
        if ctx.needs_input_grad[1]:
            dw = torch.empty(w.shape, device=w.device)
            K.gemm(dy, x, dw, ta=True)
        if ctx.has_b and ctx.needs_input_grad[2]:
            db = torch.zeros(w.size(0), device=x.device)
            K.col_sum(dy, db)
        return dx, dw, db


def _register_chain(module_list, chain):
    for m in module_list:
        stride = m.stride[0]
        chain.group.add(m.weight, m.bias, stride)


class Conv1DGenerator(nn.Module):
    """z (B, L/prod(stride)) -> N x [ConvTranspose1d 'same' -> LeakyReLU] -> 1x1 conv -> tanh
    (modeltf.py:256-286 in PyTorch conventions; same parameters as oracle.Conv1DGenerator)."""

    def __init__(self, config=((16, 5, 2), (16, 5, 2), (8, 5, 2))):
        super().__init__()
        self.config = [tuple(c) for c in config]
        self.multiplier = int(np.prod([s for _, _, s in self.config]))
        self.deconvs = nn.ModuleList()
        layers, cin = [], 1
        for nf, k, s in self.config:
            p = (k - s + 1) // 2
            op = s - (k - 2 * p)
            self.deconvs.append(nn.ConvTranspose1d(cin, nf, k, s, padding=p, output_padding=op))
            layers.append((ConvSpec('convT', cin, nf, k, s, p, out_pad=op), ACT_LEAKY))
            cin = nf
        self.out = nn.Conv1d(cin, 1, 1)
        layers.append((ConvSpec('conv', cin, 1, 1, 1, 0), ACT_TANH))
        self._chain = ConvChain(layers)
        _register_chain(list(self.deconvs) + [self.out], self._chain)

    def forward(self, batch_size=None, length=None, z=None):
        if z is None:
            z = torch.randn(batch_size, length // self.multiplier, device=self.out.weight.device)
        y = ConvChainFn.apply(z.unsqueeze(1), self._chain, *self._chain.group.params())
        return y.squeeze(1)


class Conv1DDiscriminator(nn.Module):
    """N x [Conv1d k,s 'same' -> LeakyReLU] -> global average pool -> Linear(1)  (modeltf.py:578-595)"""

    def __init__(self, config=((8, 5, 2), (16, 5, 2), (16, 5, 2))):
        super().__init__()
        self.config = [tuple(c) for c in config]
        self.convs = nn.ModuleList()
        layers, cin = [], 1
        for nf, k, s in self.config:
            self.convs.append(nn.Conv1d(cin, nf, k, s, padding=(k - 1) // 2))
            layers.append((ConvSpec('conv', cin, nf, k, s, (k - 1) // 2), ACT_LEAKY))
            cin = nf
        self.dense = nn.Linear(cin, 1)
        self._chain = ConvChain(layers)
        _register_chain(list(self.convs), self._chain)

    def forward(self, x, c=None):
        a = ConvChainFn.apply(x.unsqueeze(1), self._chain, *self._chain.group.params())
        return LinearFn.apply(a.mean(2), self.dense.weight, self.dense.bias)[:, 0]


# --------------------------------------------------------------------------------------
# critic with a hand-written double backward (WGAN-GP)
# --------------------------------------------------------------------------------------
class CriticGPFn(torch.autograd.Function):
    """(d, pen) = (D(x), (||dD/dx||_2 - 1)^2) for D = Linear(mean_t(chain(x))).

    forward : conv chain, pooled head, then the INPUT gradient g0 = dD/dx by the ordinary
              backward-data pass (v_n = w/L, v_{i-1} = conv_i^T(M_i v_i), M_i = act_i').
    backward: through d   -- the ordinary backward again, scaled by dL/dd;
              through pen -- u_0 = dL/dpen * 2(||g0||-1)/||g0|| * g0 pushed FORWARD through the chain
              (u_i = M_i conv_i(u_{i-1}), no bias; LeakyReLU has zero second derivative), and
              dW_i += wgrad(input = u_{i-1}, out-grad = M_i v_i),  dw_head += sum_t u_n / L."""

    @staticmethod
    def forward(ctx, x, chain, hw, hb, *params):
        prep = chain.group.prepare()
        B = x.size(0)
        a = x.contiguous().view(B, 1, -1)
        acts = []
        for i, (sp, act) in enumerate(chain.layers):
            assert act == ACT_LEAKY
            y = torch.empty(B, sp.cout, sp.out_len(a.size(2)), device=a.device)
            conv_fwd(sp, prep[i], a, y, bias=params[2 * i + 1].data, act=act)
            acts.append(y)
            a = y
        Ln = a.size(2)
        pooled = a.mean(2)
        d = torch.empty(B, 1, device=x.device)
        K.gemm(pooled, hw.data.contiguous(), d, tb=True, bias=hb.data)
        # input gradient of sum_b d[b]
        v = (hw.data.view(1, -1, 1) / Ln).expand(B, -1, Ln).contiguous()
        ms = []
        for i in reversed(range(len(chain.layers))):
            sp, _ = chain.layers[i]
            K.leaky_bwd(v, acts[i], v)                   # m_i = M_i * v_i
            ms.insert(0, v)
            xin_shape = acts[i - 1].shape if i > 0 else (B, 1, x.size(-1))
            dx = torch.empty(xin_shape, device=x.device)
            conv_bwd_data(sp, prep[i], v, dx)
            v = dx
        g0 = v.view(B, -1)
        nrm = g0.norm(dim=1)
        pen = (nrm - 1) ** 2
        ctx.chain, ctx.key = chain, chain.group._key[1:]
        ctx.save_for_backward(x, hw.data, g0, nrm, *(acts + ms))
        ctx.nl = len(chain.layers)
        return d.view(B), pen

    @staticmethod
    def backward(ctx, dd, dpen):
        chain, nl = ctx.chain, ctx.nl
        prep = chain.group.prepare()
        assert chain.group._key[1:] == ctx.key, 'parameters changed between forward and backward'
        sv = ctx.saved_tensors
        x, hw, g0, nrm = sv[0], sv[1], sv[2], sv[3]
        acts, ms = sv[4:4 + nl], sv[4 + nl:4 + 2 * nl]
        B, dev = x.size(0), x.device
        Ln = acts[-1].size(2)
        dws = [torch.zeros_like(it['w'].data) for it in chain.group.items]
        dbs = [torch.zeros_like(it['b'].data) for it in chain.group.items]
        dhw = torch.zeros_like(hw)
        dhb = torch.zeros(1, device=dev)
        dx = None
        if dd is not None:
            # ordinary backward of d = w . mean_t(a_n) + b
            dd = dd.contiguous().view(B, 1)
            K.gemm(dd, acts[-1].mean(2), dhw, ta=True)
            dhb += dd.sum()
            d = (dd.view(B, 1, 1) * hw.view(1, -1, 1) / Ln).expand(B, -1, Ln).contiguous()
            for i in reversed(range(nl)):
                sp, _ = chain.layers[i]
                K.leaky_bwd(d, acts[i], d)
                xin = acts[i - 1] if i > 0 else x.contiguous().view(B, 1, -1)
                conv_wgrad(sp, xin, d, dws[i], dbs[i])
                if i > 0 or ctx.needs_input_grad[0]:
                    nd = torch.empty_like(xin)
                    conv_bwd_data(sp, prep[i], d, nd)
                    d = nd
            if ctx.needs_input_grad[0]:
                dx = d.view(x.shape)
        if dpen is not None:
            coef = dpen.contiguous().view(B, 1) * 2 * (nrm - 1).view(B, 1) / nrm.view(B, 1)
            u = (coef * g0).view(B, 1, -1).contiguous()
            for i in range(nl):
                sp, _ = chain.layers[i]
                conv_wgrad(sp, u, ms[i], dws[i], None)          # d<u, conv^T(m; W)>/dW
                a = torch.empty_like(acts[i])
                conv_fwd(sp, prep[i], u, a)                     # no bias, no activation
                K.leaky_bwd(a, acts[i], a)                      # u_i = M_i * conv(u_{i-1})
                u = a
            dhw += (u.sum((0, 2)) / Ln).view_as(dhw)
        grads = []
        for i in range(nl):
            grads += [dws[i], dbs[i]]
        return (dx, None, dhw, dhb.view(1)) + tuple(grads)


class ConvPoolCritic(nn.Module):
    """a4-style conv stack (Conv1d k7 s2 + LeakyReLU, audiogan.py:476,483-494) -> average pool ->
    Linear(1).  ``forward`` returns the critic value; ``value_and_penalty`` also returns the
    WGAN-GP penalty of each sample with a double backward built on the conv kernels."""

    def __init__(self, cnn_struct=((7, 2, 16), (7, 2, 32), (7, 2, 64), (7, 2, 128), (7, 2, 256), (7, 2, 512))):
        super().__init__()
        self.cnn_struct = [list(l) for l in cnn_struct]
        self.convs = nn.ModuleList()
        layers, cin = [], 1
        for k, s, cout in self.cnn_struct:
            self.convs.append(nn.Conv1d(cin, cout, k, s, padding=(k - 1) // 2))
            layers.append((ConvSpec('conv', cin, cout, k, s, (k - 1) // 2), ACT_LEAKY))
            cin = cout
        self.dense = nn.Linear(cin, 1)
        self._chain = ConvChain(layers)
        _register_chain(list(self.convs), self._chain)

    def forward(self, x, c=None):
        a = ConvChainFn.apply(x.unsqueeze(1), self._chain, *self._chain.group.params())
        return LinearFn.apply(a.mean(2), self.dense.weight, self.dense.bias)[:, 0]

    def value_and_penalty(self, x):
        return CriticGPFn.apply(x, self._chain, self.dense.weight, self.dense.bias,
                                *self._chain.group.params())


def wgan_gp_d_loss(critic, x_real, x_fake, eps, lam=10.0):
    """mean(D(fake) - D(real)) + lam * mean((||dD/dx_hat|| - 1)^2), x_hat = eps x_real + (1-eps) x_fake
    (modeltf.py:460-469; computation_graph.py:101; utiltf.py:43-44)"""
    x_hat = (eps * x_real + (1 - eps) * x_fake).detach()
    _, pen = critic.value_and_penalty(x_hat)
    return (critic(x_fake) - critic(x_real)).mean() + lam * pen.mean()


def wgan_g_loss(critic, x_fake):
    """utiltf.py:60-61, computation_graph.py:102-104"""
    return (-critic(x_fake)).mean()
