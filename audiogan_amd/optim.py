"""Fused optimisers: check_grad + per-parameter clip + update in two launches per network.

Replaces ``check_grad`` (audiogan.py:232-240), ``clip_grad`` (:243-253) and
``T.optim.RMSprop(...).step()`` (:693-694, :788, :921); Adam follows the TF defaults of the
obsolete graph (computation_graph.py:58-59), which equal torch.optim.Adam's.
"""
import torch

from . import kernels as K
from . import ops  # noqa: F401
from . import common


class _Fused(object):
    kind = None

    def __init__(self, params, lr):
        self.params = [p for p in params]
        self.lr = float(lr)
        self.step_count = 0
        self._state = None
        self.last_norm_sum = None
        self.last_flags = None

    # -- state -----------------------------------------------------------------------
    def _init_state(self):
        dev = self.params[0].device
        offs, n = common.flat_offsets([p.numel() for p in self.params])      # every tensor's state 16-byte aligned
        flat1 = torch.zeros(n, device=dev)
        flat2 = torch.zeros(n, device=dev) if self.kind == K.OPT_ADAM else None
        s1, s2 = [], []
        for p, o in zip(self.params, offs):
            s1.append(flat1[o:o + p.numel()])
            if flat2 is not None:
                s2.append(flat2[o:o + p.numel()])
        self._state = dict(s1=s1, s2=s2 if flat2 is not None else None,
                           norms=torch.zeros(len(self.params), device=dev),
                           norm_sum=torch.zeros(1, device=dev),
                           flags=torch.zeros(1, dtype=torch.int32, device=dev),
                           step=torch.full((1,), self.step_count, dtype=torch.int32, device=dev))

    def zero_grad(self):
        """keeps .grad allocated (stable pointers -> cached descriptor tables)"""
        if getattr(self, 'bucket', None) is not None:
            self.bucket.zero()
            return
        # gradients that exist are moved ONCE into one flat buffer (``.grad`` become views), so that
        # clearing them is one memset instead of one launch per parameter tensor; parameters that have
        # never received a gradient keep ``.grad is None`` (the step skips them, as torch.optim does)
        views = getattr(self, '_gviews', None)
        if views is None or any(p.grad is not v for p, v in views):
            have = [p for p in self.params if p.grad is not None]
            if not have:
                return
            offs, n = common.flat_offsets([p.numel() for p in have])
            flat = torch.zeros(n, device=have[0].device, dtype=torch.float32)
            views = []
            for p, o in zip(have, offs):
                p.grad = flat[o:o + p.numel()].view(p.shape)
                views.append((p, p.grad))
            self._gflat, self._gviews = flat, views
            return
        self._gflat.zero_()
        if len(views) != len(self.params):
            for p in self.params:
                if p.grad is not None and not any(p is q for q, _ in views):
                    self._gviews = None      # a new gradient appeared: re-flatten next time
                    p.grad.detach_()
                    p.grad.zero_()

    def state_dict(self):
        self._ensure()
        self.step_count = int(self._state['step'].item())
        return dict(step=self.step_count, lr=self.lr,
                    s1=[t.clone() for t in self._state['s1']],
                    s2=[t.clone() for t in self._state['s2']] if self._state['s2'] else None)

    def load_state_dict(self, sd):
        self._ensure()
        self.step_count, self.lr = sd['step'], sd['lr']
        self._state['step'].fill_(self.step_count)
        for a, b in zip(self._state['s1'], sd['s1']):
            a.copy_(b)
        if self._state['s2']:
            for a, b in zip(self._state['s2'], sd['s2']):
                a.copy_(b)

    def _ensure(self):
        if self._state is None or self._state['norms'].device != self.params[0].device:
            self._init_state()

    # -- step ------------------------------------------------------------------------
    def step(self, clip_norm=0.0, grad_scale=1.0, check=False):
        """``clip_norm``: per-PARAMETER L2 clip (0 = off) exactly as clip_grad; ``grad_scale``
        multiplies every gradient first (1/world_size after a sum all-reduce); ``check`` reads
        the NaN/|g|>1e5 flags back (one host sync) and asserts like check_grad; it also reads the sticky status word of
        the persistent recurrent launches and raises ``kernels.PersistentLaunchError`` if one of them gave up."""
        self._ensure()
        st = self._state
        live = [(p, p.grad, a, (st['s2'][i] if st['s2'] else None))
                for i, (p, a) in enumerate(zip(self.params, st['s1'])) if p.grad is not None]
        if not live:
            return None
        ps = [l[0].data for l in live]
        gs = [l[1].contiguous() for l in live]
        s1 = [l[2] for l in live]
        s2 = [l[3] for l in live] if st['s2'] else None
        self.step_count += 1
        # check=True reads the flags BEFORE the update (check_grad asserts before clip_grad / step, audiogan.py:786-788):
        # the norms are finished by a launch of their own.  Otherwise the update launch finishes them itself (2 launches)
        part = K.grad_norms(ps, gs, s1, s2, st['norms'], st['norm_sum'], st['flags'], grad_scale, st['step'], finish=bool(check))
        if check:
            part = None
            f = int(st['flags'].item())
            if self.params[0].is_cuda:
                K.check_persist_status(self.params[0].device)     # a persistent launch that gave up (its outputs are NaN)
            assert not (f & 1), 'NaN in gradients (check_grad)'
            assert not (f & 2), '|grad| > 1e5 (check_grad)'
        self._launch(ps, gs, s1, s2, st['norms'], clip_norm, grad_scale, part)
        common.bump_param_epoch(self.params)
        self.last_norm_sum, self.last_flags = st['norm_sum'], st['flags']
        return st['norm_sum']


class RMSprop(_Fused):
    """torch.optim.RMSprop defaults: alpha 0.99, eps 1e-8, no momentum (audiogan.py:693-694)."""
    kind = K.OPT_RMSPROP

    def __init__(self, params, lr=1e-2, alpha=0.99, eps=1e-8):
        super().__init__(params, lr)
        self.alpha, self.eps = alpha, eps

    def _launch(self, ps, gs, s1, s2, norms, clip, gscale, part=None):
        K.opt_step(ps, gs, s1, None, norms, K.OPT_RMSPROP, self.lr, float(clip), float(gscale),
                   self.alpha, 0.0, self.eps, self.step_count, self._state['step'], part=part,
                   norm_sum=self._state['norm_sum'], flags=self._state['flags'])


class Adam(_Fused):
    """lr 1e-3, betas (0.9, 0.999), eps 1e-8 (computation_graph.py:58-59)."""
    kind = K.OPT_ADAM

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, lr)
        self.betas, self.eps = betas, eps

    def _launch(self, ps, gs, s1, s2, norms, clip, gscale, part=None):
        K.opt_step(ps, gs, s1, s2, norms, K.OPT_ADAM, self.lr, float(clip), float(gscale),
                   self.betas[0], self.betas[1], self.eps, self.step_count, self._state['step'], part=part,
                   norm_sum=self._state['norm_sum'], flags=self._state['flags'])


def make_optimizer(params, kind, lr):
    if kind == 'rmsprop':
        return RMSprop(params, lr=lr)
    if kind == 'adam':
        return Adam(params, lr=lr)
    raise ValueError(kind)
