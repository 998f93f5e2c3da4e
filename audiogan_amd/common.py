"""Pieces shared by the autograd blocks: the weight-norm group and scratch helpers."""
import torch

from . import kernels as K

# bumped whenever parameters are rewritten through raw pointers (fused optimiser): globally, and per parameter
# (attribute ``_ag_epoch``) so that a block only re-materialises its weights when ITS parameters were stepped
PARAM_EPOCH = [0]


# Weight materialisations are cached on the HOST (a key of parameter versions).  Under hipGraph capture that decision is
# frozen into the graph, so a graph captured while the cache happened to be warm would never re-materialise its weights
# on replay.  ``new_capture()`` (call it right before every ``torch.cuda.graph`` capture) makes the first use of each
# weight group inside that capture materialise unconditionally.
CAPTURE_EPOCH = [0]


def new_capture():
    CAPTURE_EPOCH[0] += 1
    return CAPTURE_EPOCH[0]


def capture_tag(dev):
    if torch.device(dev).type == 'cuda' and torch.cuda.is_current_stream_capturing():
        return CAPTURE_EPOCH[0]
    return None


def param_epoch(p):
    return getattr(p, '_ag_epoch', 0)


def bump_param_epoch(params):
    PARAM_EPOCH[0] += 1
    for p in params:
        p._ag_epoch = getattr(p, '_ag_epoch', 0) + 1


# Parameter gradients are accumulated by the kernels straight into an already allocated ``.grad``
# (then the autograd block returns None for that parameter) instead of returning a fresh tensor that
# autograd adds to ``.grad`` with one extra elementwise launch per parameter tensor (~100 per step).
DIRECT_GRADS = [True]


def grad_target(p):
    """the buffer to accumulate dL/dp into, or None (no .grad yet / feature off -> return the gradient)"""
    if not DIRECT_GRADS[0] or not isinstance(p, torch.nn.Parameter) or not p.requires_grad:
        return None
    g = p.grad
    if g is None or not g.is_contiguous() or g.dtype != torch.float32 or g.requires_grad:
        return None
    return g


class frozen(object):
    """context manager: parameters of the given modules (or iterables of parameters) get requires_grad False inside
    the block and their old flag back afterwards.  A block's ``needs_input_grad`` is fixed when its forward runs, so
    a forward inside this context computes NO weight gradient in its backward and leaves ``.grad`` untouched
    (audiogan.py:897-901 freezes the same way around the stopper's REINFORCE backward)."""

    def __init__(self, *modules):
        self.params = []
        for m in modules:
            self.params += list(m.parameters()) if hasattr(m, 'parameters') else list(m)

    def __enter__(self):
        self.flags = [p.requires_grad for p in self.params]
        for p in self.params:
            p.requires_grad_(False)
        return self

    def __exit__(self, *exc):
        for p, r in zip(self.params, self.flags):
            p.requires_grad_(r)


# Network-level backward scope.  ``with network_backward(): loss.backward()`` - every second stage the backward's blocks
# defer (kernels.deferred_reduces) is summed by ONE launch when the whole backward is over, and the weight-norm backward of
# EVERY block of the network (dW -> dv, dg, accumulated straight into ``.grad``) runs as ONE launch behind it, instead of
# one pair per autograd Function.  Blocks whose parameters have no ``.grad`` buffer yet (they return fresh gradient tensors
# to autograd) finish on the spot, as they do outside a scope.
_WN_PENDING = dict(ents=None, keep=None, groups=None)


def finish_pending():
    """inside a network scope: make everything deferred so far final NOW (one flush + one weight-norm backward launch);
    the scope stays open.  A block whose backward runs a SECOND time inside one scope (the critic applied to real and to
    generated clips in two passes) calls this before it reuses its dW scratch."""
    ents = _WN_PENDING['ents']
    if ents is None:
        return
    K.flush_reduces()
    if ents:
        K.weight_norm_bwd(ents)
    _WN_PENDING['ents'], _WN_PENDING['keep'], _WN_PENDING['groups'] = [], [], set()


class network_backward(object):
    def __enter__(self):
        self.dr = K.deferred_reduces(outer=True)
        self.dr.__enter__()
        self.owner = self.dr.role == 'owner'
        if self.owner:
            _WN_PENDING['ents'], _WN_PENDING['keep'], _WN_PENDING['groups'] = [], [], set()
        return self

    def __exit__(self, et, ev, tb):
        if not self.owner:
            return self.dr.__exit__(et, ev, tb)
        ents = _WN_PENDING['ents']
        _WN_PENDING['ents'] = None
        try:
            self.dr.__exit__(et, ev, tb)           # the one flush: every deferred sum is final behind it
            if et is None and ents:
                K.weight_norm_bwd(ents)
        finally:
            _WN_PENDING['keep'] = _WN_PENDING['groups'] = None
        return False


class Bf16Images(object):
    """bfloat16 images of fp32 tensors (weights, or column blocks of them), made by ONE small launch each and cached until
    the source changes: what ag_gemm_h reads on the bf16-storage path.  The key follows WNGroup's: parameter pointer /
    version / epoch, and the capture tag - inside a hipGraph capture the first use re-converts, so a replayed graph never
    reads an image that was filled outside it."""

    def __init__(self):
        self._img = {}

    def get(self, name, src, owner=None):
        """``src``: a 2-D fp32 tensor (any row pitch) or contiguous; ``owner``: the Parameter whose version identifies its
        content (default: src itself)"""
        o = owner if owner is not None else src
        key = (capture_tag(src.device), o.data_ptr(), o._version, param_epoch(o), src.data_ptr(), tuple(src.shape))
        hit = self._img.get(name)
        if hit is not None and hit[0] == key:
            return hit[1]
        buf = hit[1] if (hit is not None and hit[1].shape == src.shape and hit[1].device == src.device) else None
        img = K.to_bf16(src, out=buf)
        self._img[name] = (key, img)
        return img


class Prepared(object):
    __slots__ = ('w', 'wpa', 'wpb', 'pad')

    def __init__(self, w=None, wpa=None, wpb=None, pad=0):
        self.w, self.wpa, self.wpb, self.pad = w, wpa, wpb, pad      # pad: what the scatter layout is prepared for


class WNGroup(object):
    """All weight-normed tensors of one block.  ``prepare`` materialises w = g*v/||v|| for
    every tensor (and the conv-engine layouts of 3-D weights) with ONE launch into
    persistent buffers; ``backward`` turns dW into (dv, dg) with one launch."""

    def __init__(self):
        self.items = []   # dict(v=Parameter, g=Parameter, stride=int, engine=bool)
        self._bufs = None
        self._key = None

    def add(self, v, g, stride=1, engine=False, pad=0):
        self.items.append(dict(v=v, g=g, stride=stride, engine=engine, pad=pad))
        return len(self.items) - 1

    def params(self):
        out = []
        for it in self.items:
            out += [it['v'], it['g']]
        return out

    def _alloc(self, dev):
        bufs = []
        for it in self.items:
            v = it['v']
            p = Prepared(w=torch.empty_like(v.data), pad=it.get('pad', 0))
            if it['engine']:
                d0, d1, kk = v.shape
                p.wpa = torch.zeros(K.wpa_numel(d0, d1, kk), device=dev)
                p.wpb = torch.zeros(K.wpb_numel(d0, d1, kk, it['stride']), device=dev)
            bufs.append(p)
        self._bufs = bufs
        self._key = None

    def _current_key(self):
        """(key of the parameters as they are now); allocates the buffers on first use"""
        dev = self.items[0]['v'].device
        if self._bufs is None or self._bufs[0].w.device != dev:
            self._alloc(dev)
        return (capture_tag(dev),) + tuple((it['v'].data_ptr(), it['v']._version, param_epoch(it['v']), it['g'].data_ptr(),
                                            it['g']._version, param_epoch(it['g'])) for it in self.items)

    def _entries(self):
        return [dict(v=it['v'].data, g=it['g'].data.view(-1), w=p.w, wpa=p.wpa, wpb=p.wpb, stride=it['stride'],
                     pad=it.get('pad', 0)) for it, p in zip(self.items, self._bufs)]

    def prepare(self):
        key = self._current_key()
        if key != self._key:
            K.weight_norm_fwd(self._entries())
            self._mark_fresh(key)
        return self._bufs

    def _mark_fresh(self, key):
        """the materialised weights were just rewritten: their bf16 images are stale"""
        self._key = key
        self._w16 = {}

    def w16(self, i):
        """bfloat16 image of the materialised weight of item i (made on first use after every materialisation; inside a
        hipGraph capture that first use lies inside the graph, like the materialisation itself)"""
        d = getattr(self, '_w16', None)
        if d is None:
            d = self._w16 = {}
        img = d.get(i)
        if img is None:
            keep = getattr(self, '_w16_buf', None)
            if keep is None:
                keep = self._w16_buf = {}
            img = d[i] = K.to_bf16(self._bufs[i].w, out=keep.get(i))       # (stable address: the buffer is reused)
            keep[i] = img
        return img

    def zero_dws(self):
        """zero-filled scratch for the gradients wrt the materialised weights, one tensor per item (views of ONE
        persistent flat buffer: a single memset, and - unlike a fresh allocation per backward - stable addresses, so the
        descriptor table of the weight-norm backward is built once and a captured graph needs no upload node for it:
        under replay such an upload runs as ~24 serial 64-byte blits per table, 0.35 ms per step)"""
        dev = self.items[0]['v'].device
        if _WN_PENDING['groups'] and id(self) in _WN_PENDING['groups']:
            finish_pending()        # second backward of this block inside one scope: the first one's dW must be consumed first
        if getattr(self, '_dws', None) is None or self._dws[0].device != dev:
            n = sum(it['v'].numel() for it in self.items)
            flat = torch.empty(n, device=dev, dtype=torch.float32)
            views, o = [], 0
            for it in self.items:
                views.append(flat[o:o + it['v'].numel()].view(it['v'].shape))
                o += it['v'].numel()
            self._dws, self._dws_flat = views, flat
        self._dws_flat.zero_()
        return list(self._dws)

    def backward(self, dws):
        """dws[i]: gradient wrt the materialised w of item i (same shape as v), or None.
        Returns the flat list [dv0, dg0, dv1, dg1, ...]."""
        ents, outs = [], []
        for it, dw in zip(self.items, dws):
            if dw is None or not (it['v'].requires_grad or it['g'].requires_grad):
                outs += [None, None]        # no gradient arrived / tensor frozen (e.g. stopper-only backward)
                continue
            tv, tg = grad_target(it['v']), grad_target(it['g'])
            if tv is not None and tg is not None:
                ents.append(dict(v=it['v'].data, g=it['g'].data.view(-1), dw=dw, dv=tv, dg=tg.view(-1),
                                 accumulate=True))
                outs += [None, None]
                continue
            dv = torch.empty_like(it['v'].data)
            dg = torch.empty_like(it['g'].data)
            ents.append(dict(v=it['v'].data, g=it['g'].data.view(-1), dw=dw, dv=dv, dg=dg.view(-1)))
            outs += [dv, dg]
        if not ents:
            return outs
        if _WN_PENDING['ents'] is not None and K.reduces_outer() and all(o is None for o in outs):
            # inside a network-level scope and every gradient goes straight into ``.grad``: dW is final only when that
            # scope flushes, and nothing reads dv / dg before the optimiser - join the network's one launch
            _WN_PENDING['ents'] += ents
            _WN_PENDING['keep'].append(dws)
            _WN_PENDING['groups'].add(id(self))
            return outs
        if K.reduces_outer():
            K.flush_reduces()       # fresh gradient tensors go back to autograd now: they must be final now
        K.weight_norm_bwd(ents)
        return outs


def prepare_groups(groups):
    """materialise every STALE group of `groups` with ONE weight-norm launch (a network's blocks are otherwise prepared one
    launch each at their first use; a launch of this size is ~25 us of latency whatever it carries).  Groups on different
    devices, or with nothing stale, fall back to / cost nothing."""
    stale, ents = [], []
    for gr in groups:
        if not gr.items:
            continue
        key = gr._current_key()
        if key != gr._key:
            stale.append((gr, key))
            ents += gr._entries()
    if ents:
        if len(set(e['v'].device for e in ents)) > 1:
            for gr, _ in stale:
                gr.prepare()
            return
        K.weight_norm_fwd(ents)
        for gr, key in stale:
            gr._mark_fresh(key)


def flat_offsets(numels, align=4):
    """offsets of tensors packed into one flat fp32 buffer, each starting on a 16-byte boundary (``align`` elements): the
    optimiser and gradient-norm kernels then move every tensor with 16-byte accesses.  Returns (offsets, total)."""
    offs, o = [], 0
    for n in numels:
        offs.append(o)
        o += (int(n) + align - 1) // align * align
    return offs, o


def _zeros_like_list(tensors):
    """one flat zero buffer, viewed as the given shapes (one memset instead of many)"""
    n = sum(t.numel() for t in tensors)
    flat = torch.zeros(n, device=tensors[0].device, dtype=torch.float32)
    out, o = [], 0
    for t in tensors:
        out.append(flat[o:o + t.numel()].view(t.shape))
        o += t.numel()
    return out
