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


def masked_bce_mean(logits, target, nframes):
    """mean_b( sum_{t<n_b} bce(x[b,t], target) / n_b )  --  the loss assembly at
    audiogan.py:739-740, :766+:780, :864+:897 in one kernel.  Returns (loss, per_sample_sums)."""
    return ops.BCEFn.apply(logits, float(target), nframes)


def length_mask(size, length):
    """audiogan.py:204-211 (device-side, no host loop)."""
    b, n = int(size[0]), int(size[1])
    ar = torch.arange(n, device=length.device).unsqueeze(0)
    return (ar < length.view(b, 1)).float()
