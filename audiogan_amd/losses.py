"""Loss / mask helpers with the reference's names (audiogan.py:178-211), on the HIP kernels."""
import torch

from . import ops


def binary_cross_entropy_with_logits_per_sample(input, target, weight=None, nframes=None):
    """audiogan.py:187-197.  ``target`` is a Python scalar or a tensor filled with one value (the
    reference only ever uses 0.9 / 0 / 0.5, :727,:762,:857-860).  The 0/1 ``weight`` matrix of the
    reference is expressed by ``nframes`` (row b weighs columns [0, nframes[b])); passing a
    ``weight`` tensor is accepted when it is such a prefix mask.  Returns the per-sample sums."""
    if torch.is_tensor(target):
        if target.size() != input.size():
            raise ValueError("Target size ({}) must be the same as input size ({})".format(
                target.size(), input.size()))
        target = float(target.reshape(-1)[0])
    if weight is not None and nframes is None:
        nframes = weight.sum(1).long()
    return _PerSample.apply(input, float(target), nframes)


class _PerSample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, target, nframes):
        from . import kernels as K
        x = x.contiguous()
        per = torch.empty(x.size(0), device=x.device)
        K.bce_logits_fwd(x, target, nframes, per, None, 1.0)
        ctx.target, ctx.has_n = target, nframes is not None
        ctx.save_for_backward(x, nframes if nframes is not None else x.new_empty(0))
        return per

    @staticmethod
    def backward(ctx, dper):
        from . import kernels as K
        x, nfr = ctx.saved_tensors
        B, T = x.shape
        # d per[b] / dx = (sigmoid - target) * mask: reuse the fused kernel with n := 1 scaling undone
        dx = torch.empty_like(x)
        n = nfr if ctx.has_n else torch.full((B,), T, dtype=torch.long, device=x.device)
        K.bce_logits_bwd(x, ctx.target, n, None, 1.0, dx)
        return dx * (dper * n.float()).view(B, 1), None, None


def masked_bce_mean(logits, target, nframes, scale=None):
    """mean_b( sum_{t<n_b} bce(x[b,t], target) / n_b )  --  the loss assembly at
    audiogan.py:739-740, :766+:780, :864+:897 in one kernel.  Returns (loss, per_sample_sums).  ``target``: a float, or
    a [B] tensor of per-row targets; ``scale``: replaces the 1 / B of the mean (e.g. 2 / B when real and fake clips are
    scored in one call and the loss is the SUM of their means)."""
    return ops.BCEFn.apply(logits, target if torch.is_tensor(target) else float(target), nframes, scale)


_TARGET_ROWS = {}


def real_fake_targets(batch, device, real=0.9, fake=0.0):
    """cached [2 * batch] per-row targets: `real` for the first half, `fake` for the second"""
    key = (int(batch), str(device), float(real), float(fake))
    if key not in _TARGET_ROWS:
        _TARGET_ROWS[key] = torch.cat([torch.full((batch,), float(real)), torch.full((batch,), float(fake))]).to(device)
    return _TARGET_ROWS[key]


def length_mask(size, length):
    """audiogan.py:204-211 (device-side, no host loop)."""
    b, n = int(size[0]), int(size[1])
    ar = torch.arange(n, device=length.device).unsqueeze(0)
    return (ar < length.view(b, 1)).float()


def stopper_surrogate_loss(stop_logits, stops, reward, weight=None):
    """Log-prob surrogate of the reference's REINFORCE update of the stop head (audiogan.py:444-451 draws
    ``stop_t ~ Categorical([1 - sigmoid(s_t), sigmoid(s_t)])``; :866-887 builds ``reward = -(loss) - baseline``
    per sample, times the frame weights; :887-902 ``stop_t.reinforce(reward[:, t])`` + backward into the stopper's
    parameters only).  ``.reinforce`` no longer exists in torch; minimising

        L = - sum_{b,t} reward[b,t] * log p(stops[b,t] | s[b,t])

    gives the same gradient.  ``stops``: [B,T] tensor or the forward's ``stop_list`` (list of [B,1]); ``reward``:
    [B] or [B,T] (treated as a constant); ``weight``: optional [B,T] frame mask.  Use it inside
    ``only_stopper_trains(g)`` to reproduce the reference's freezing of every other generator parameter."""
    import torch.nn.functional as F
    if isinstance(stops, (list, tuple)):
        stops = torch.cat([t.view(-1, 1) for t in stops], 1)
    stops = stops.to(stop_logits.device).float()
    T_ = min(stops.size(1), stop_logits.size(1))
    s, stops = stop_logits[:, :T_], stops[:, :T_]
    r = reward.detach()
    if r.dim() == 1:
        r = r.view(-1, 1).expand_as(s)
    r = r[:, :T_]
    if weight is not None:
        r = r * weight[:, :T_].to(r.device)
    logp = stops * F.logsigmoid(s) + (1.0 - stops) * F.logsigmoid(-s)
    return -(r * logp).sum()


class only_stopper_trains(object):
    """context manager: every generator parameter except the stop head is frozen (audiogan.py:897-901)"""

    def __init__(self, g, *also_frozen):
        self.g = g
        self.also = also_frozen          # further modules trained by the same optimiser (the text embedder e_g, :691)

    def __enter__(self):
        keep = set(id(p) for p in self.g.stopper.parameters())
        self.flags = [(p, p.requires_grad) for m in (self.g,) + tuple(self.also) for p in m.parameters()]
        for p, _ in self.flags:
            p.requires_grad_(id(p) in keep)
        from . import recurrent
        self._prev = recurrent.STOPPER_ONLY[0]
        recurrent.STOPPER_ONLY[0] = True
        return self

    def __exit__(self, *exc):
        from . import recurrent
        recurrent.STOPPER_ONLY[0] = self._prev
        for p, r in self.flags:
            p.requires_grad_(r)
