"""The parts of the reference's *current* D/G iterations that sit on top of the classic step
(SURVEY.md section 8(f), rank 2): the feature-matching penalty over D's conv activations and the
FGSM-style input / latent perturbations.  They reuse the hot-path kernels for every D/G pass and
input gradient; the per-(clip, channel) moment statistics over time are one HIP kernel per
activation (ag_time_moments_fwd/bwd), their batch means / stds tiny reductions over [B,C].

  fourth_moment, calc_dists        audiogan.py:336-359
  feature_penalty                  audiogan.py:850-855 (and :112-117)
  adversarial_movement_d           audiogan.py:139-150
  adversarially_sample_z           audiogan.py:99-137
"""
import torch

from . import kernels as K
from .common import frozen
from .losses import binary_cross_entropy_with_logits_per_sample, length_mask  # noqa: F401


class _TimeMomentsFn(torch.autograd.Function):
    """h [B,C,L], lengths [B] -> (m, s, f) [B,C] as in calc_dists (audiogan.py:345-350)"""

    @staticmethod
    def forward(ctx, h, lens):
        B, C, _ = h.shape
        lens = lens.to(h.device).long().contiguous()
        m, s, f = (torch.empty(B, C, device=h.device) for _ in range(3))
        if h.stride(2) != 1:
            h = h.contiguous()
        K.time_moments_fwd(h, lens, m, s, f)
        ctx.save_for_backward(h, lens)
        return m, s, f

    @staticmethod
    def backward(ctx, gm, gs, gf):
        h, lens = ctx.saved_tensors
        dh = torch.empty(h.shape, device=h.device)
        cont = lambda t: t.contiguous() if t is not None else None      # noqa: E731
        K.time_moments_bwd(h, lens, cont(gm), cont(gs), cont(gf), dh)
        return dh, None


def fourth_moment(v):
    """audiogan.py:336-339.  The reference is Python 2, where the exponent ``(1/4)`` is the integer 0:
    the function returns x**0 == 1 for every feature.  Reproduced as written (it makes the nine
    'fourth' terms of the penalty vanish, since both sides are identically 1)."""
    v_mean = v.mean(0)
    return (((v - v_mean.unsqueeze(0)) ** 4).sum(0)) ** 0


def calc_dists(hidden_states, hidden_state_lengths):
    """audiogan.py:341-359: per layer, per-clip mean / (un-normalised) 2nd and 4th central moment roots
    over time, then batch mean and std of each; returns (statistic, std) pairs in the reference order."""
    means_d, stds_d, fourth_d = [], [], []
    for h, l in zip(hidden_states, hidden_state_lengths):
        m, s, f = _TimeMomentsFn.apply(h, l)
        means_d += [(m.mean(0), m.std(0)), (s.mean(0), s.std(0)), (f.mean(0), f.std(0))]
        stds_d += [(m.std(0), m.std(0)), (s.std(0), s.std(0)), (f.std(0), f.std(0))]
        fourth_d += [(fourth_moment(m), m.std(0)), (fourth_moment(s), s.std(0)), (fourth_moment(f), f.std(0))]
    return means_d + stds_d + fourth_d


def feature_penalty(dists_d, dists_g, batch_size):
    """audiogan.py:850-855: sum over statistics of mean((real - fake)^2) / batch_size"""
    pen = 0
    for r, f in zip(dists_d, dists_g):
        pen = pen + torch.pow(r[0] - f[0], 2).mean() / batch_size
    return pen


_PEN_W = {}


def feature_penalty_fused(hs_d, hl_d, hs_g, hl_g, batch_size):
    """``feature_penalty(calc_dists(hs_d, hl_d), calc_dists(hs_g, hl_g), batch_size)`` (audiogan.py:847-855) without its
    ~600 tiny launches: the per-clip time moments of all layers of one side are concatenated along the channel axis into ONE
    [B, 3 * sum(C)] matrix, whose batch mean and batch std (one reduction each) are every 'mean' and 'std' statistic of
    calc_dists at once; each statistic's ``mean()`` over its own channels becomes a per-column weight 1 / C_layer.  The nine
    'fourth' terms per layer are identically 1 on both sides (``fourth_moment`` as the reference wrote it) and contribute
    an exact zero, the unused second element of every pair is not computed.  Same sum, different association (1e-6)."""
    def side(hs, hl):
        cols = []
        for h, l in zip(hs, hl):
            cols += list(_TimeMomentsFn.apply(h, l))
        return torch.cat(cols, 1)
    xr, xg = side(hs_d, hl_d), side(hs_g, hl_g)
    key = (tuple(int(h.size(1)) for h in hs_d), str(xr.device))
    w = _PEN_W.get(key)
    if w is None:
        w = torch.cat([torch.full((3 * c,), 1.0 / c) for c in key[0]]).to(xr.device)
        _PEN_W[key] = w
    dm = xr.mean(0) - xg.mean(0)
    ds = xr.std(0) - xg.std(0)
    return ((dm * dm + ds * ds) * w).sum() / batch_size


def _input_grad_sign(d, data, data_len, embed_d, target, nframes_hint=None):
    data = data.detach().requires_grad_(True)
    # only d(loss)/d(input) is wanted: with D's parameters frozen the blocks compute no weight gradient and do not
    # touch ``.grad`` (which may be the flat all-reduce bucket holding this iteration's gradients)
    with frozen(d):
        cls, _, _, nframes = d(data, data_len, embed_d)
        loss = binary_cross_entropy_with_logits_per_sample(cls, target, nframes=nframes) / nframes.float()
        grad, = torch.autograd.grad(loss.sum(), data)
    return grad


def adversarial_movement_d(data, data_len, embed_d, target, weight, d, scale=1e-3):
    """audiogan.py:139-150: +-scale in the direction that increases D's loss on ``data``
    (``weight`` is accepted for signature compatibility; the mask is rebuilt from D's nframes)."""
    grad = _input_grad_sign(d, data, data_len, embed_d, target)
    return (grad > 0).float() * scale - (grad < 0).float() * scale


def adversarially_sample_z(g, d, batch_size, nframes, noise_size, maxlen, embed_g, noisescale, embed_d,
                           g_optim='boundary_seeking', scale=1e-2, z=None, noise=None, stop=None):
    """audiogan.py:99-137 (minus its feature-penalty lines, whose value never reaches the returned z):
    draw z, push it through G and D, and move it by +-scale along the sign of d(loss)/dz wherever
    |grad| > 1e-9."""
    dev = embed_g.device
    z = torch.randn(batch_size, nframes, noise_size, device=dev) if z is None else z
    z = z.detach().requires_grad_(True)
    with frozen(g, d):          # input gradient only: no weight gradients, ``.grad`` of G and D untouched
        fake, _, _, fake_len = g(batch_size=batch_size, length=maxlen, c=embed_g, z=z, stop=stop)
        if noise is None:
            noise = torch.randn_like(fake) * noisescale
        cls_g, _, _, nframes_g = d(fake + noise[:, :fake.size(1)], fake_len, embed_d)
        target = 0.5 if g_optim == 'boundary_seeking' else 0.0
        loss = binary_cross_entropy_with_logits_per_sample(cls_g, target, nframes=nframes_g) / nframes_g.float()
        grad, = torch.autograd.grad(loss.sum(), z)
    advers = (grad > 1e-9).float() * scale - (grad < -1e-9).float() * scale
    return (z + advers).detach()
