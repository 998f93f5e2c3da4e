"""Process-per-GPU data parallelism for the G+D step: gradients of each network live in ONE
flat fp32 buffer (parameters' ``.grad`` are views into it) and are summed across ranks with a
single RCCL all-reduce per network per step over xGMI; the 1/world_size factor is folded into
the fused optimiser kernel (``grad_scale``).  New in this build: the reference has no
distributed path, only per-submodule ``NN.DataParallel``."""
import torch
import torch.distributed as dist


class GradBucket(object):
    """Owns the flat gradient buffer of one network."""

    def __init__(self, params, group=None, early=None, force_collective=False, comm_dtype='f32'):
        """``early``: parameters whose gradients are final before the rest of backward has run (they
        are laid out first, so their all-reduce can be issued while backward continues).
        ``comm_dtype='bf16'`` (BASELINE configs[2]): the gradients cross xGMI as bfloat16 - each rank's fp32 sum is
        rounded once into a bf16 staging buffer, RCCL sums that, and the result is widened back into the fp32 bucket
        the optimiser reads (half the bytes per step; accumulation inside a rank and the optimiser state stay fp32)."""
        params = [p for p in params]
        early = [p for p in (early or [])]
        eid = set(id(p) for p in early)
        self.params = early + [p for p in params if id(p) not in eid]
        from .common import flat_offsets
        # (every tensor starts on a 16-byte boundary of the flat buffer: the optimiser kernels read it 16 bytes at a time)
        self.offsets, n = flat_offsets([p.numel() for p in self.params])
        self.n_early = self.offsets[len(early)] if len(early) < len(self.params) else n
        self.group = group
        dev = self.params[0].device
        self.flat = torch.zeros(n, device=dev, dtype=torch.float32)
        assert comm_dtype in ('f32', 'bf16')
        self.comm = torch.zeros(n, device=dev, dtype=torch.bfloat16) if comm_dtype == 'bf16' else None
        self._pending = []        # (work or None, lo, hi) of every all-reduce not yet waited for / widened
        for p, o in zip(self.params, self.offsets):
            p.grad = self.flat[o:o + p.numel()].view(p.shape)
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        # force_collective: issue the all-reduce even in a 1-rank group (single-GPU rehearsal of the RCCL path)
        self.force = bool(force_collective) and dist.is_available() and dist.is_initialized()

    def zero(self):
        self.flat.zero_()

    def check_views(self):
        """autograd must have accumulated IN PLACE into the flat buffer"""
        base = self.flat.data_ptr()
        for p, o in zip(self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != base + 4 * o:
                return False
        return True

    def all_reduce(self, async_op=False, part='all'):
        """sum over ranks; returns the scale the optimiser must apply (1/world).  part: 'all', 'early'
        (the leading n_early elements) or 'late' (the rest).  async_op=True: the collective runs on
        RCCL's own stream (after everything already enqueued on the current stream); call wait()."""
        if self.world > 1 or self.force:
            lo, hi = {'all': (0, self.flat.numel()), 'early': (0, self.n_early),
                      'late': (self.n_early, self.flat.numel())}[part]
            if hi > lo:
                assert all(h <= lo or hi <= l for _, l, h in self._pending), \
                    'GradBucket.all_reduce: this range already has an all-reduce in flight; call wait() first'
                buf = self.flat[lo:hi]
                if self.comm is not None:
                    buf = self.comm[lo:hi]
                    buf.copy_(self.flat[lo:hi])            # fp32 -> bf16 (RNE), once per rank
                work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
                self._pending.append((work if async_op else None, lo, hi))
                if not async_op:
                    self.wait()
        return 1.0 / self.world

    def wait(self):
        """complete every all-reduce issued so far, in issue order (and widen bf16 sums back into the fp32 bucket the
        optimiser reads)"""
        pend, self._pending = self._pending, []
        for work, lo, hi in pend:
            if work is not None:
                work.wait()
            if self.comm is not None:
                self.flat[lo:hi].copy_(self.comm[lo:hi])


def broadcast_parameters(module, src=0, group=None):
    """make every rank start from rank ``src``'s weights (one flat broadcast)"""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    ps = [p.data for p in module.parameters()]
    flat = torch.cat([p.reshape(-1) for p in ps])
    dist.broadcast(flat, src=src, group=group)
    o = 0
    for p in ps:
        p.copy_(flat[o:o + p.numel()].view(p.shape))
        o += p.numel()
