"""Tensor-level wrappers over the C ABI (include/audiogan_hip.h).

Every function takes torch CUDA fp32 tensors (views allowed where noted), checks what
the kernels assume (device, dtype, unit stride on the last axis, shapes) and enqueues
the kernel on torch's current HIP stream.  Nothing here computes on the host and there
is no fallback: a non-CUDA tensor raises."""
import ctypes as C

import torch

from . import _lib
from ._lib import lib, check, ACT_NONE, ACT_LEAKY, ACT_TANH, ACT_LEAKY_GATE, OPT_RMSPROP, OPT_ADAM  # noqa: F401

LEAKY_SLOPE = 0.01


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _chk(t, name, dtype=torch.float32):
    if t is None:
        return
    if not t.is_cuda:
        raise RuntimeError('audiogan_amd: %s must be a CUDA (HIP) tensor; there is no CPU path' % name)
    if t.dtype != dtype:
        raise TypeError('audiogan_amd: %s must be %s, got %s' % (name, dtype, t.dtype))


BF16 = torch.bfloat16


def _is16(t):
    return t is not None and t.dtype == torch.bfloat16


def _chk_any(t, name):
    """fp32 or bfloat16 device tensor"""
    if t is None:
        return
    if not t.is_cuda:
        raise RuntimeError('audiogan_amd: %s must be a CUDA (HIP) tensor; there is no CPU path' % name)
    if t.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError('audiogan_amd: %s must be float32 or bfloat16, got %s' % (name, t.dtype))


def _mat_any(t, name):
    _chk_any(t, name)
    if t.dim() != 2 or (t.size(1) > 1 and t.stride(1) != 1):
        raise ValueError('audiogan_amd: %s must be 2-D with unit stride along dim 1' % name)
    return t.stride(0) if t.size(0) > 1 else max(t.stride(0), t.size(1))


# bf16 STORAGE (BASELINE configs[2], round 4): in 'bf16' precision mode the critic's sequence path keeps its activations,
# their gradients and the GEMM operands as bfloat16 in HBM (ag_gemm_h reads them straight into LDS); fp32 accumulators,
# master weights, gate pre-activations / cell states of the recurrent kernels and optimiser state are unchanged.
# AG_BF16_STORE=0 keeps the round-3 form (fp32 in memory, rounded per use) for A/B runs.
import os as _os1
BF16_STORE = [_os1.environ.get('AG_BF16_STORE', '1') != '0']


def bf16_storage():
    return BF16_STORE[0] and lib.ag_get_precision() == 1


def _bcl(t, name):
    """(batch stride, channel stride) of a [B,C,L] view with unit time stride."""
    _chk(t, name)
    if t.dim() != 3 or (t.size(2) > 1 and t.stride(2) != 1):
        raise ValueError('audiogan_amd: %s must be [B,C,L] with unit stride along L' % name)
    return t.stride(0), t.stride(1)


def _mat(t, name):
    """leading dimension of a 2-D row-major view"""
    _chk(t, name)
    if t.dim() != 2 or (t.size(1) > 1 and t.stride(1) != 1):
        raise ValueError('audiogan_amd: %s must be 2-D with unit stride along dim 1' % name)
    return t.stride(0) if t.size(0) > 1 else max(t.stride(0), t.size(1))


# ------------------------------------------------------------------------------------
# two-stage (deterministic) reductions: a per-call workspace bound right before the C call
# ------------------------------------------------------------------------------------
import os as _os0

if _os0.environ.get('AG_DETERMINISTIC', '1') == '0':
    raise ImportError('audiogan_amd: AG_DETERMINISTIC=0 asked for the float-atomic reductions of round 1; they were removed '
                      '(round 4) - every cross-workgroup sum is two-stage and bitwise reproducible.  Unset the variable.')
_SMALL_WS = 1 << 17        # floats; enough for the bias / channel / column sums at every BASELINE size


# process-wide, like the C side (api.hip): a scope is opened by the thread that calls loss.backward() and the recording
# calls come from torch's autograd thread (never concurrently: the opener blocks inside backward())
#   keep   workspaces kept alive until the flush (None: no scope)
#   outer  the scope was opened by a network-level caller (common.network_backward): it RECORDS only inside the inner
#          ``deferred_reduces()`` blocks of the autograd Functions (each block vouches that nothing inside it reads the
#          sums it defers) and flushes once, when the whole backward is over
#   depth  inner blocks currently open
_DEFER = dict(keep=None, outer=False, depth=0)


def reduces_recording():
    return _DEFER['keep'] is not None and (_DEFER['depth'] > 0 or not _DEFER['outer'])


def reduces_outer():
    """is a network-level scope open?  (its sums are final only when IT closes: see common.WNGroup.backward)"""
    return _DEFER['keep'] is not None and _DEFER['outer']


class deferred_reduces(object):
    """``with K.deferred_reduces(): ...`` - the second stages of every two-stage reduction issued inside (conv weight
    gradients, bias / channel / column sums, split-K weight-gradient GEMMs called with ``defer=True``) run as ONE launch at
    the end instead of one each (ag_defer_reduces / ag_flush_reduces; same results bit for bit).  Nothing inside the block
    may read those outputs.  The partial-sum workspaces are kept alive until the flush.
    ``outer=True`` (common.network_backward): open a scope around a whole backward pass; it records only inside the
    Functions' own ``deferred_reduces()`` blocks, and those blocks then do NOT flush when they end - everything recorded
    by the whole backward is summed by one launch when the outer scope closes."""

    def __init__(self, outer=False):
        self.outer = bool(outer)
        self.role = None

    def __enter__(self):
        st = _DEFER
        if st['keep'] is None:
            check(lib.ag_defer_reduces(1), 'ag_defer_reduces')
            st['keep'], st['outer'], st['depth'] = [], self.outer, 0
            self.role = 'owner'
            if self.outer:
                lib.ag_defer_reduces(2)                # paused until an inner block opens
        elif self.outer:
            self.role = 'nested'                       # a network scope inside another scope: the outermost one rules
        else:
            if st['outer'] and st['depth'] == 0:
                check(lib.ag_defer_reduces(3), 'ag_defer_reduces')
            st['depth'] += 1
            self.role = 'inner'
        return self

    def __exit__(self, et, ev, tb):
        st = _DEFER
        if self.role == 'inner':
            st['depth'] -= 1
            if st['outer'] and st['depth'] == 0:
                lib.ag_defer_reduces(2)
        elif self.role == 'owner':
            try:
                if et is None:
                    check(lib.ag_flush_reduces(_stream()), 'ag_flush_reduces')
            finally:
                st['keep'], st['outer'], st['depth'] = None, False, 0
                lib.ag_defer_reduces(1)        # drops whatever is still recorded (error path) ...
                lib.ag_defer_reduces(0)        # ... and turns deferral off
        return False


def flush_reduces():
    """sum everything recorded so far NOW (a reader inside a scope needs final values); the scope stays open"""
    if _DEFER['keep'] is not None:
        check(lib.ag_flush_reduces(_stream()), 'ag_flush_reduces')


class _immediate(object):
    """the reducing call inside runs its second stage at once even inside a recording scope"""

    def __enter__(self):
        self.on = reduces_recording()
        if self.on:
            lib.ag_defer_reduces(2)

    def __exit__(self, *exc):
        if self.on:
            lib.ag_defer_reduces(3)


def _bind_ws(numel, dev):
    """bind `numel` floats of scratch for the NEXT reducing call (ag_bind_workspace); the tensor comes from torch's
    stream-ordered caching allocator, so it is safe to drop it as soon as the call has been enqueued"""
    if numel <= 0:
        lib.ag_bind_workspace(None, 0)       # no workspace for this call: drop whatever an earlier, failed call left bound
        return None
    ws = torch.empty(int(numel), dtype=torch.float32, device=dev)
    check(lib.ag_bind_workspace(_p(ws), int(numel)), 'ag_bind_workspace')
    if _DEFER['keep'] is not None:
        _DEFER['keep'].append(ws)
    return ws


# ------------------------------------------------------------------------------------
# precision mode of the contractions (ag_set_precision): 'f32' (default) or 'bf16'
# ------------------------------------------------------------------------------------
def set_precision(mode):
    """'f32': exact fp32 contractions.  'bf16': every contraction rounds both operands to bfloat16 (RNE) and
    accumulates in fp32 (GEMMs and the persistent recurrent kernels on v_mfma_f32_32x32x16_bf16 / rounded operands);
    memory, epilogues, losses and the optimiser stay fp32.  Returns the previous mode."""
    old = get_precision()
    check(lib.ag_set_precision({'f32': 0, 'bf16': 1, 'f32x3': 2}[mode]), 'ag_set_precision')
    return old


def get_precision():
    """'f32x3' is an experiment (bench.py --dtype f32x3): the large GEMMs on three bf16 MFMAs per product of bf16 hi + lo
    operand parts (~2^-16 relative per product), everything else exact fp32"""
    return {0: 'f32', 1: 'bf16', 2: 'f32x3'}[lib.ag_get_precision()]


class precision(object):
    """``with K.precision('bf16'): ...``"""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.old = set_precision(self.mode)
        return self

    def __exit__(self, *exc):
        set_precision(self.old)


def wpa_numel(d0, d1, K):
    return int(lib.ag_wpa_numel(d0, d1, K))


def wpb_numel(d0, d1, K, stride):
    return int(lib.ag_wpb_numel(d0, d1, K, stride))


# ------------------------------------------------------------------------------------
# descriptor tables (device copies of small C struct arrays), cached by content
# ------------------------------------------------------------------------------------
from collections import OrderedDict

_table_cache = OrderedDict()     # descriptor tables created eagerly: LRU, bounded
_table_pinned = {}               # tables created while a hipGraph was being captured: the graph's copy node re-reads
                                 # the pinned slot and its kernels read the device table on every replay - never evicted
_TABLE_LRU = 4096


def _table(kind, structs, device):
    raw = b''.join(bytes(s) for s in structs)
    key = (kind, raw, str(device))
    t = _table_pinned.get(key)
    if t is not None:
        return t[0]
    capturing = torch.device(device).type == 'cuda' and torch.cuda.is_current_stream_capturing()
    t = _table_cache.get(key)
    if t is not None:
        if capturing:
            # uploaded eagerly before the capture began (complete: capture starts behind a synchronize): the graph's
            # kernels may read it as it is, without a copy node - but from now on it must never be evicted
            _table_pinned[key] = _table_cache.pop(key)
            _capture_log.append(key)
            return t[0]
        _table_cache.move_to_end(key)
        return t[0]
    src = torch.frombuffer(bytearray(raw), dtype=torch.uint8)
    if torch.device(device).type == 'cuda':
        # staged through a pre-allocated pinned arena + async copy: legal inside hipGraph capture (the copy becomes a
        # graph node that re-reads the arena slot on replay, so slots are never recycled)
        host = _pinned_slot(len(raw))
        host.copy_(src)
        t = torch.empty(len(raw), dtype=torch.uint8, device=device)
        t.copy_(host, non_blocking=True)
    else:
        host, t = src, src.to(device)
    if capturing:
        _table_pinned[key] = (t, host)
        _capture_log.append(key)
        TABLE_STATS['uploads_in_capture'] += 1
        TABLE_STATS['kinds'].append(kind)
    else:
        _table_cache[key] = (t, host)
        while len(_table_cache) > _TABLE_LRU:
            _table_cache.popitem(last=False)
    return t


_capture_log = []
# descriptor tables that had to be UPLOADED by a copy node of a captured graph (their content was not seen by an eager step
# before the capture): each costs a ~5 us node per replay - bench.py reports the count
TABLE_STATS = dict(uploads_in_capture=0, kinds=[])


def capture_mark():
    """position in the log of tables created under capture (see drop_captured_tables)"""
    return len(_capture_log)


def drop_captured_tables(mark):
    """forget the tables created under a capture that FAILED or will never be replayed (their device content was
    never written): call with the capture_mark() taken before the capture started"""
    while len(_capture_log) > mark:
        _table_pinned.pop(_capture_log.pop(), None)


_arena = {'buf': None, 'off': 0}


def _pinned_slot(nbytes):
    n = (nbytes + 63) // 64 * 64
    a = _arena
    if a['buf'] is None or a['off'] + n > a['buf'].numel():
        a['buf'] = torch.empty(max(4 << 20, n), dtype=torch.uint8).pin_memory()
        a['off'] = 0
    s = a['buf'][a['off']:a['off'] + n]
    a['off'] += n
    return s[:nbytes]


def reserve_table_arena():
    """allocate the pinned staging arena now (call before hipGraph capture)"""
    if _arena['buf'] is None:
        _pinned_slot(64)


# ------------------------------------------------------------------------------------
# weight norm
# ------------------------------------------------------------------------------------
def weight_norm_fwd(entries):
    """entries: list of dict(v, g, w=None, wpa=None, wpb=None, stride=1, pad=0).  v is the
    parameter tensor ([d0], [d0,d1] or [d0,d1,K]); outputs are written in place.  ``pad``: the conv padding the
    scatter layout wpb is prepared for (aligned layout, see conv_engine's wp_pad)."""
    descs, max_rows = [], 1
    for e in entries:
        v, g = e['v'], e['g']
        _chk(v, 'v'); _chk(g, 'g')
        assert v.is_contiguous() and g.is_contiguous() and g.numel() == v.size(0)
        rows = v.size(0)
        cols = v.numel() // rows
        if v.dim() == 3:
            d1, K = v.size(1), v.size(2)
        else:
            d1, K = cols, 1
        for k in ('w', 'wpa', 'wpb'):
            _chk(e.get(k), k)
        if e.get('w') is not None:
            assert e['w'].is_contiguous() and e['w'].numel() == v.numel()
        s = int(e.get('stride', 1))
        if e.get('wpa') is not None:
            assert e['wpa'].numel() == wpa_numel(rows, d1, K)
        if e.get('wpb') is not None:
            assert e['wpb'].numel() == wpb_numel(rows, d1, K, s)
        descs.append(_lib.WnDesc(v.data_ptr(), g.data_ptr(), _p(e.get('w')).value or 0,
                                 _p(e.get('wpa')).value or 0, _p(e.get('wpb')).value or 0, 0,
                                 rows, cols, d1, K, s, int(e.get('pad', 0))))
        max_rows = max(max_rows, rows)
    tab = _table('wn', descs, entries[0]['v'].device)
    check(lib.ag_weight_norm_fwd(_p(tab), len(descs), max_rows, _stream()), 'ag_weight_norm_fwd')


def weight_norm_bwd(entries):
    """entries: list of dict(v, g, dw, dv, dg[, accumulate]); dv/dg are written (or added to)."""
    descs, max_rows = [], 1
    for e in entries:
        for k in ('v', 'g', 'dw', 'dv', 'dg'):
            _chk(e[k], k)
            assert e[k].is_contiguous()
        rows = e['v'].size(0)
        cols = e['v'].numel() // rows
        assert e['dw'].numel() == e['v'].numel() == e['dv'].numel() and e['dg'].numel() == rows
        descs.append(_lib.WnBwdDesc(e['v'].data_ptr(), e['g'].data_ptr(), e['dw'].data_ptr(),
                                    e['dv'].data_ptr(), e['dg'].data_ptr(), rows, cols,
                                    1 if e.get('accumulate') else 0, 0))
        max_rows = max(max_rows, rows)
    tab = _table('wnb', descs, entries[0]['v'].device)
    check(lib.ag_weight_norm_bwd(_p(tab), len(descs), max_rows, _stream()), 'ag_weight_norm_bwd')


def prep_conv_weight(w, wpa, wpb, stride, pad=0):
    _chk(w, 'w'); _chk(wpa, 'wpa'); _chk(wpb, 'wpb')
    assert w.dim() == 3 and w.is_contiguous()
    d0, d1, K = w.shape
    if wpa is not None:
        assert wpa.numel() == wpa_numel(d0, d1, K)
    if wpb is not None:
        assert wpb.numel() == wpb_numel(d0, d1, K, stride)
    check(lib.ag_prep_conv_weight(_p(w), _p(wpa), _p(wpb), d0, d1, K, stride, int(pad), _stream()),
          'ag_prep_conv_weight')


# ------------------------------------------------------------------------------------
# conv engine
# ------------------------------------------------------------------------------------
def conv_engine(x, wp, y, K, stride, pad, mode, bias=None, res=None, lens=None, act=ACT_NONE,
                slope=LEAKY_SLOPE, accumulate=False, wp_pad=0):
    """mode 0: y[b,o,t] = sum W x[b,c,s*t+k-p]   (wp = gather layout of the weight)
    mode 1: y[b,o,s*t+k-p] += W x[b,c,t]        (wp = scatter layout; wp_pad = the padding it was prepared for)
    x: [B,C,Lin] view, y: [B,O,Lout] view (written in place).  act=ACT_LEAKY_GATE: ``res`` is the saved output of a
    LeakyReLU and scales the result by that activation's derivative instead of being added."""
    assert act != ACT_LEAKY_GATE or res is not None, 'ACT_LEAKY_GATE needs the saved activation in `res`'
    x_bs, x_cs = _bcl(x, 'x')
    y_bs, y_cs = _bcl(y, 'y')
    _chk(wp, 'wp'); _chk(bias, 'bias'); _chk(lens, 'lens', torch.int64)
    B, Cc, Lin = x.shape
    B2, O, Lout = y.shape
    assert B == B2
    if mode == 0:
        assert wp.numel() == wpa_numel(O, Cc, K), 'gather weight layout does not match shapes'
    else:
        assert wp.numel() == wpb_numel(Cc, O, K, stride), 'scatter weight layout does not match'
    r_bs = r_cs = 0
    if res is not None:
        r_bs, r_cs = _bcl(res, 'res')
        assert tuple(res.shape) == tuple(y.shape)
    if bias is not None:
        assert bias.numel() == O and bias.is_contiguous()
    if lens is not None:
        assert lens.numel() == B and lens.is_contiguous()
    a = _lib.ConvArgs(x.data_ptr(), wp.data_ptr(), _p(bias).value or 0, _p(res).value or 0,
                      y.data_ptr(), _p(lens).value or 0, x_bs, x_cs, y_bs, y_cs, r_bs, r_cs,
                      B, Cc, Lin, O, Lout, K, stride, pad, mode, act, slope, 1 if accumulate else 0, int(wp_pad))
    check(lib.ag_conv1d_engine(C.byref(a), _stream()), 'ag_conv1d_engine')


def conv_wgrad(sh, lg, dw, K, stride, pad):
    """dw[a,c,k] += sum_{b,t} sh[b,a,t] * lg[b,c,s*t+k-p];  dw: [A,C,K] contiguous."""
    sh_bs, sh_cs = _bcl(sh, 'sh')
    lg_bs, lg_cs = _bcl(lg, 'lg')
    _chk(dw, 'dw')
    B, A, Lsh = sh.shape
    B2, Cc, Llg = lg.shape
    assert B == B2 and dw.is_contiguous() and dw.numel() == A * Cc * K
    _ws = _bind_ws(lib.ag_conv1d_wgrad_ws_numel(B, A, Lsh, Cc, K), dw.device)  # noqa: F841
    check(lib.ag_conv1d_wgrad(_p(sh), sh_bs, sh_cs, _p(lg), lg_bs, lg_cs, _p(dw), B, A, Lsh, Cc, Llg,
                              K, stride, pad, _stream()), 'ag_conv1d_wgrad')


def channel_sum(dy, db, accumulate=True):
    """db[c] (+)= sum_{b,t} dy[b,c,t]"""
    bs, cs = _bcl(dy, 'dy')
    _chk(db, 'db')
    B, Cc, L = dy.shape
    assert db.numel() == Cc and db.is_contiguous()
    _ws = _bind_ws(_SMALL_WS, db.device)  # noqa: F841
    check(lib.ag_channel_sum(_p(dy), bs, cs, _p(db), B, Cc, L, int(bool(accumulate)), _stream()), 'ag_channel_sum')


def leaky_bwd(dy, y, dpre, lens=None, slope=LEAKY_SLOPE, add_into=None, bias_grad=None):
    """dpre = dy * (y > 0 ? 1 : slope) * (t < lens[b]); all [B,C,L] views; dpre may alias dy.
    add_into (optional [B,C,L] view, must not overlap dpre): add_into += dpre.
    bias_grad (optional [C], pre-zeroed): bias_grad[c] += sum_{b,t} dpre[b,c,t]."""
    a = _bcl(dy, 'dy'); b = _bcl(y, 'y'); c = _bcl(dpre, 'dpre')
    d = _bcl(add_into, 'add_into') if add_into is not None else (0, 0)
    _chk(lens, 'lens', torch.int64)
    assert tuple(dy.shape) == tuple(y.shape) == tuple(dpre.shape)
    assert add_into is None or tuple(add_into.shape) == tuple(dy.shape)
    B, Cc, L = dy.shape
    if bias_grad is not None:
        _chk(bias_grad, 'bias_grad')
        assert bias_grad.is_contiguous() and bias_grad.numel() == Cc
        _ws = _bind_ws(_SMALL_WS, bias_grad.device)  # noqa: F841
    check(lib.ag_leaky_bwd(_p(dy), a[0], a[1], _p(y), b[0], b[1], _p(dpre), c[0], c[1], _p(add_into),
                           d[0], d[1], _p(lens), _p(bias_grad), B, Cc, L, slope, _stream()), 'ag_leaky_bwd')


# ------------------------------------------------------------------------------------
# GEMM
# ------------------------------------------------------------------------------------
def gemm(A, B, Cm, ta=False, tb=False, alpha=1.0, beta=0.0, bias=None, res=None, act=ACT_NONE,
         slope=LEAKY_SLOPE, defer=False):
    """Cm[M,N] = act(alpha * op(A) @ op(B) + beta*Cm + bias + res).
    ta: A is stored [K,M];  tb: B is stored [N,K] (a Linear weight).
    ``defer``: Cm is a parameter gradient that nothing reads before the enclosing ``deferred_reduces`` scope closes - a
    split-K product may then leave its second stage to the scope's one launch.  Default: the product is complete when the
    call returns (in stream order), also inside a scope."""
    lda, ldb, ldc = _mat(A, 'A'), _mat(B, 'B'), _mat(Cm, 'C')
    M, N = Cm.shape
    K = A.size(0) if ta else A.size(1)
    assert (A.size(1) if ta else A.size(0)) == M, 'gemm: A shape'
    assert (B.size(1) if tb else B.size(0)) == K and (B.size(0) if tb else B.size(1)) == N, 'gemm: B shape'
    ldres = 0
    if res is not None:
        ldres = _mat(res, 'res')
        assert tuple(res.shape) == (M, N)
    if bias is not None:
        _chk(bias, 'bias')
        assert bias.numel() == N and bias.is_contiguous()
    nws = lib.ag_gemm_ws_numel(M, N, K, act)
    hold = (not defer) and nws > 0 and reduces_recording()
    if hold:
        lib.ag_defer_reduces(2)
    try:
        _ws = _bind_ws(nws, Cm.device)  # noqa: F841
        check(lib.ag_gemm(_p(A), lda, int(ta), _p(B), ldb, int(tb), _p(Cm), ldc, M, N, K, alpha, beta,
                          _p(bias), _p(res), ldres, act, slope, _stream()), 'ag_gemm')
    finally:
        if hold:
            lib.ag_defer_reduces(3)


def col_sum(X, out, accumulate=True, defer=True):
    """out[n] (+)= sum_m X[m,n].  ``defer=False``: complete when the call returns (in stream order) also inside a recording
    ``deferred_reduces`` scope - for sums that are read before that scope closes."""
    ldx = _mat_any(X, 'X')
    _chk(out, 'out')
    M, N = X.shape
    assert out.numel() == N and out.is_contiguous()
    hold = (not defer) and reduces_recording()
    if hold:
        lib.ag_defer_reduces(2)
    try:
        _ws = _bind_ws(min(_SMALL_WS, 64 * N) if M > 16 else 0, out.device)  # noqa: F841
        check(lib.ag_col_sum(_p(X), int(_is16(X)), ldx, _p(out), M, N, int(bool(accumulate)), _stream()), 'ag_col_sum')
    finally:
        if hold:
            lib.ag_defer_reduces(3)


# ------------------------------------------------------------------------------------
# LSTM cell pointwise
# ------------------------------------------------------------------------------------
def lstm_cell_fwd(gates, c_prev, c_out, h_out=None, y_out=None, h_prev=None, valid=None, t=0):
    ldg = _mat(gates, 'gates')
    B, H4 = gates.shape
    H = H4 // 4
    _chk(valid, 'valid', torch.int64)
    ld = lambda x, n: (_mat(x, n) if x is not None else 0)  # noqa: E731
    for x in (c_prev, c_out, h_out, y_out, h_prev):
        assert x is None or tuple(x.shape) == (B, H)
    check(lib.ag_lstm_cell_fwd(_p(gates), ldg, _p(c_prev), ld(c_prev, 'c_prev'), _p(h_out),
                               ld(h_out, 'h_out'), _p(c_out), ld(c_out, 'c_out'), _p(y_out),
                               ld(y_out, 'y_out'), _p(h_prev), ld(h_prev, 'h_prev'), _p(valid), t, B, H,
                               _stream()), 'ag_lstm_cell_fwd')


def lstm_cell_bwd(gates_act, c_prev, c_new, dh, dy, dc_next, dgates, dc_prev, dh_pass=None,
                  valid=None, t=0):
    """dh: gradient from later steps (or None), dy: gradient of this step's output (or None)"""
    B, H4 = gates_act.shape
    H = H4 // 4
    _chk(valid, 'valid', torch.int64)
    ld = lambda x, n: (_mat(x, n) if x is not None else 0)  # noqa: E731
    for x in (c_prev, c_new, dh, dy, dc_next, dc_prev, dh_pass):
        assert x is None or tuple(x.shape) == (B, H)
    assert tuple(dgates.shape) == (B, H4)
    check(lib.ag_lstm_cell_bwd(_p(gates_act), _mat(gates_act, 'gates_act'), _p(c_prev),
                               ld(c_prev, 'c_prev'), _p(c_new), ld(c_new, 'c_new'), _p(dh), ld(dh, 'dh'),
                               _p(dy), ld(dy, 'dy'), _p(dc_next), ld(dc_next, 'dc_next'), _p(dgates), _mat(dgates, 'dgates'),
                               _p(dc_prev), ld(dc_prev, 'dc_prev'), _p(dh_pass), ld(dh_pass, 'dh_pass'),
                               _p(valid), t, B, H, _stream()), 'ag_lstm_cell_bwd')


# ------------------------------------------------------------------------------------
# BCE / activations
# ------------------------------------------------------------------------------------
def bce_logits_fwd(x, target, nframes, per_sample, loss, scale):
    """per_sample[b] = masked sum; loss[0] += scale * sum_b per_sample[b]/n[b]"""
    ldx = _mat(x, 'x')
    _chk(nframes, 'nframes', torch.int64); _chk(per_sample, 'per_sample'); _chk(loss, 'loss')
    B, T = x.shape
    check(lib.ag_bce_logits_fwd(_p(x), ldx, float(target), _p(nframes), _p(per_sample), _p(loss),
                                float(scale), B, T, _stream()), 'ag_bce_logits_fwd')


def bce_logits_bwd(x, target, nframes, gscale, scale, dx):
    ldx, lddx = _mat(x, 'x'), _mat(dx, 'dx')
    _chk(nframes, 'nframes', torch.int64); _chk(gscale, 'gscale')
    B, T = x.shape
    check(lib.ag_bce_logits_bwd(_p(x), ldx, float(target), _p(nframes), _p(gscale), float(scale),
                                _p(dx), lddx, B, T, _stream()), 'ag_bce_logits_bwd')


def bce_logits_fwd_strided(x, target, nframes, per_sample, loss, scale, target_rows=None):
    """x [B,T] of any strides; per_sample[b] = masked sum (optional); loss[0] = scale * sum_b per_sample[b]/n[b] (written);
    target_rows: optional [B] targets (else `target` for every row)"""
    _chk(x, 'x'); _chk(nframes, 'nframes', torch.int64); _chk(per_sample, 'per_sample'); _chk(loss, 'loss')
    _chk(target_rows, 'target_rows')
    B, T = x.shape
    assert target_rows is None or (target_rows.is_contiguous() and target_rows.numel() == B)
    assert nframes is None or (nframes.is_contiguous() and nframes.numel() == B)
    assert per_sample is None or (per_sample.is_contiguous() and per_sample.numel() == B)
    check(lib.ag_bce_logits_fwd_strided(_p(x), x.stride(0), x.stride(1), float(target), _p(target_rows), _p(nframes),
                                        _p(per_sample), _p(loss), float(scale), B, T, _stream()), 'ag_bce_logits_fwd_strided')


def bce_logits_bwd_strided(x, target, nframes, gscale, scale, dx, target_rows=None):
    _chk(x, 'x'); _chk(dx, 'dx'); _chk(nframes, 'nframes', torch.int64); _chk(gscale, 'gscale'); _chk(target_rows, 'target_rows')
    B, T = x.shape
    assert tuple(dx.shape) == (B, T)
    check(lib.ag_bce_logits_bwd_strided(_p(x), x.stride(0), x.stride(1), float(target), _p(target_rows), _p(nframes),
                                        _p(gscale), float(scale), _p(dx), dx.stride(0), dx.stride(1), B, T, _stream()),
          'ag_bce_logits_bwd_strided')


def act_fwd(x, y, act, slope=LEAKY_SLOPE):
    _chk(x, 'x'); _chk(y, 'y')
    assert x.is_contiguous() and y.is_contiguous() and x.numel() == y.numel()
    check(lib.ag_act_fwd(_p(x), _p(y), x.numel(), act, slope, _stream()), 'ag_act_fwd')


def act_bwd(dy, y, dx, act, slope=LEAKY_SLOPE):
    for t_, n in ((dy, 'dy'), (y, 'y'), (dx, 'dx')):
        _chk(t_, n)
        assert t_.is_contiguous()
    assert dy.numel() == y.numel() == dx.numel()
    check(lib.ag_act_bwd(_p(dy), _p(y), _p(dx), dy.numel(), act, slope, _stream()), 'ag_act_bwd')


def bct_to_tbc(x, out=None, out_dtype=None):
    """[B,C,T] (any batch / channel pitch, time contiguous) -> contiguous [T,B,C] (ag_transpose_batched); fp32 or bf16 on
    either side (``out_dtype``: default = x's)"""
    _chk_any(x, 'x')
    B, Cc, T = x.shape
    assert x.stride(2) == 1
    if out is None:
        out = torch.empty(T, B, Cc, device=x.device, dtype=out_dtype or x.dtype)
    _chk_any(out, 'out')
    check(lib.ag_transpose_batched(_p(x), int(_is16(x)), x.stride(0), x.stride(1), _p(out), int(_is16(out)), Cc, B * Cc, B, Cc,
                                   T, _stream()), 'ag_transpose_batched')
    return out


def tbc_to_bct(x, out=None, out_dtype=None):
    """contiguous [T,B,C] -> contiguous [B,C,T] (ag_transpose_batched); fp32 or bf16 on either side"""
    _chk_any(x, 'x')
    T, B, Cc = x.shape
    assert x.is_contiguous()
    if out is None:
        out = torch.empty(B, Cc, T, device=x.device, dtype=out_dtype or x.dtype)
    _chk_any(out, 'out')
    check(lib.ag_transpose_batched(_p(x), int(_is16(x)), Cc, B * Cc, _p(out), int(_is16(out)), Cc * T, T, B, T, Cc, _stream()),
          'ag_transpose_batched')
    return out


def rowdot_ok(x, w):
    """can the one-output Linear kernels (ag_rowdot_*) take x [M,K] (unit stride along K) and w [1,K] / [K]?"""
    return x.dim() == 2 and x.stride(1) == 1 and w.numel() == x.size(1) and w.is_contiguous()


def rowdot_fwd(x, w, bias, y):
    """y[m] = x[m,:] . w + bias[0]; y: [M] or [M,1] of any stride; x fp32 or bf16"""
    ldx = _mat_any(x, 'x')
    _chk(w, 'w'); _chk(bias, 'bias'); _chk(y, 'y')
    M, Kd = x.shape
    assert rowdot_ok(x, w) and y.numel() == M
    ldy = y.stride(0) if M > 1 else 1
    check(lib.ag_rowdot_fwd(_p(x), int(_is16(x)), ldx, _p(w), _p(bias), _p(y), ldy, M, Kd, _stream()), 'ag_rowdot_fwd')


def rowdot_bwd(dy, x, w, dx=None, dw=None, db=None, gate=False, slope=LEAKY_SLOPE, accumulate=True):
    """backward of a one-output Linear in ONE pass over x: dx = dy (x) w (times LeakyReLU'(x) when ``gate``: x is the saved
    output of the LeakyReLU below), dw (+)= dy^T x, db (+)= sum dy.  dw [1,K] / [K] and db [1] must be adjacent in memory
    (db right behind dw: one gradient row), as the views of WNGroup.zero_dws are; dy: [M] or [M,1] of any stride."""
    ldx = _mat_any(x, 'x')
    _chk(dy, 'dy'); _chk(w, 'w'); _chk_any(dx, 'dx'); _chk(dw, 'dw'); _chk(db, 'db')
    M, Kd = x.shape
    assert rowdot_ok(x, w) and dy.numel() == M
    lddx = _mat_any(dx, 'dx') if dx is not None else 0
    assert dx is None or (tuple(dx.shape) == (M, Kd) and dx.dtype == x.dtype), 'x and dx share a storage type'
    h16 = int(_is16(x))
    if dw is not None:
        assert dw.is_contiguous() and dw.numel() == Kd and db is not None and db.numel() == 1
        assert db.data_ptr() == dw.data_ptr() + 4 * Kd, 'db must sit right behind dw'
        _ws = _bind_ws(lib.ag_rowdot_bwd_ws_numel(M, Kd), dw.device)  # noqa: F841
    check(lib.ag_rowdot_bwd(_p(dy), dy.stride(0) if M > 1 else 1, _p(x), ldx, _p(w), _p(dx), lddx, h16, _p(dw), _p(db),
                            int(bool(accumulate)), M, Kd, int(bool(gate)), slope, _stream()), 'ag_rowdot_bwd')


def to_bf16(src, out=None):
    """a bfloat16 image of a contiguous or row-pitched 2-D fp32 tensor / of any contiguous fp32 tensor (ag_to_bf16_2d)"""
    _chk(src, 'src')
    if src.dim() == 2 and src.stride(1) == 1 and src.size(0) <= 65535:
        rows, cols, ld = src.size(0), src.size(1), (src.stride(0) if src.size(0) > 1 else src.size(1))
    else:
        assert src.is_contiguous()
        n = src.numel()
        cols = next((c for c in (8192, 4096, 2048, 1024, 512, 256, 64, 8, 1) if n % c == 0 and n // c <= 65535), None)
        assert cols is not None, 'to_bf16: tensor too large for one launch'
        rows, ld = n // cols, cols
    if out is None:
        out = torch.empty(src.shape, device=src.device, dtype=torch.bfloat16)
    _chk(out, 'out', torch.bfloat16)
    assert out.is_contiguous() and out.numel() == src.numel()
    check(lib.ag_to_bf16_2d(_p(src), ld, _p(out), cols, rows, cols, _stream()), 'ag_to_bf16_2d')
    return out


def gemm_h_ok(A, B, ta, tb):
    """can ag_gemm_h take these bf16 operands?  (K % 64 == 0, 16-byte aligned rows, k-strided row counts % 8 == 0)"""
    if not (_is16(A) and _is16(B) and A.dim() == 2 and B.dim() == 2 and A.stride(1) == 1 and B.stride(1) == 1):
        return False
    M, Kd = (A.size(1), A.size(0)) if ta else (A.size(0), A.size(1))
    N = B.size(0) if tb else B.size(1)
    lda = A.stride(0) if A.size(0) > 1 else A.size(1)
    ldb = B.stride(0) if B.size(0) > 1 else B.size(1)
    return bool(lib.ag_gemm_h_ok(M, N, Kd, int(ta), int(tb), lda, ldb)) and A.data_ptr() % 16 == 0 and B.data_ptr() % 16 == 0


def gemm_h(A, B, C=None, C16=None, ta=False, tb=False, alpha=1.0, beta=0.0, bias=None, res=None, gate=None, act=ACT_NONE,
           slope=LEAKY_SLOPE, defer=False):
    """C / C16 [M,N] = act(alpha * op(A) @ op(B) + beta * C + bias + res) on operands STORED as bfloat16 (ag_gemm_h).
    C: fp32 output, C16: bf16 output (either or both); res: fp32 or bf16 residual (with ACT_LEAKY_GATE: the saved activation);
    gate: bf16 saved LeakyReLU output applied after bias / res.  ``defer`` as for gemm()."""
    lda, ldb = _mat_any(A, 'A'), _mat_any(B, 'B')
    assert _is16(A) and _is16(B), 'gemm_h: bf16 operands'
    out = C if C is not None else C16
    M, N = out.shape
    Kd = A.size(0) if ta else A.size(1)
    assert (A.size(1) if ta else A.size(0)) == M, 'gemm_h: A shape'
    assert (B.size(1) if tb else B.size(0)) == Kd and (B.size(0) if tb else B.size(1)) == N, 'gemm_h: B shape'
    ldc = ldc16 = ldres = ldres16 = ldg = 0
    if C is not None:
        ldc = _mat(C, 'C')
        assert tuple(C.shape) == (M, N)
    if C16 is not None:
        _chk(C16, 'C16', torch.bfloat16)
        ldc16 = _mat_any(C16, 'C16')
        assert tuple(C16.shape) == (M, N)
    r32 = res if (res is not None and not _is16(res)) else None
    r16 = res if _is16(res) else None
    if res is not None:
        assert tuple(res.shape) == (M, N)
        ldres, ldres16 = (_mat(r32, 'res') if r32 is not None else 0), (_mat_any(r16, 'res') if r16 is not None else 0)
    if gate is not None:
        _chk(gate, 'gate', torch.bfloat16)
        ldg = _mat_any(gate, 'gate')
        assert tuple(gate.shape) == (M, N)
    if bias is not None:
        _chk(bias, 'bias')
        assert bias.numel() == N and bias.is_contiguous()
    nws = lib.ag_gemm_h_ws_numel(M, N, Kd, act, int(C16 is not None)) if gate is None else 0
    hold = (not defer) and nws > 0 and reduces_recording()
    if hold:
        lib.ag_defer_reduces(2)
    try:
        _ws = _bind_ws(nws, out.device)  # noqa: F841
        check(lib.ag_gemm_h(_p(A), lda, int(ta), _p(B), ldb, int(tb), _p(C), ldc, _p(C16), ldc16, M, N, Kd, alpha, beta,
                            _p(bias), _p(r32), ldres, _p(r16), ldres16, _p(gate), ldg, act, slope, _stream()), 'ag_gemm_h')
    finally:
        if hold:
            lib.ag_defer_reduces(3)


def _work_gemm_h(A, B, C=None, C16=None, ta=False, tb=False, *a_, **kw):
    out = C if C is not None else C16
    M, N = out.shape
    Kd = A.size(0) if ta else A.size(1)
    return 'gemm_bf16s_kernel<%d,%d>' % (int(ta), int(tb)), 2.0 * M * N * Kd, 2.0 * (M * Kd + N * Kd) + (4.0 if C is not None else 2.0) * M * N


def build_zc(z, c, out=None):
    """zc[t,b,:] = [z[b,t,:] | c[b,:]]: z [B,T,ns], c [B,es] -> contiguous [T,B,ns+es] (ag_build_zc; audiogan.py:433-439)"""
    _chk(z, 'z'); _chk(c, 'c')
    B, T, ns = z.shape
    es = c.size(1)
    assert z.is_contiguous() and c.is_contiguous() and c.size(0) == B
    if out is None:
        out = torch.empty(T, B, ns + es, device=z.device)
    check(lib.ag_build_zc(_p(z), _p(c), _p(out), B, T, ns, es, _stream()), 'ag_build_zc')
    return out


def critic_batch(xa, na=None, xb=None, nb=None, len_a=None, len_b=None, prods=(), ca=None, cb=None):
    """the critic's minibatch in ONE launch (ag_critic_batch): rows [xa + na ; xb + nb] -> x [nA+nB, L]; the rows' lengths
    after every conv layer, lens [len(prods), nA+nB] int64 with lens[i] = ceil(len / prods[i]) (len_a / len_b None: L); the
    conditioning rows [ca ; cb] (None: skipped).  Returns (x, lens or None, c or None)."""
    nA, L = xa.shape
    nB = xb.size(0) if xb is not None else 0
    dev = xa.device
    lda = _rows(xa, 'xa', nA, L)
    ldna = _rows(na, 'na', nA, L) if na is not None else 0
    ldb = _rows(xb, 'xb', nB, L) if xb is not None else 0
    ldnb = _rows(nb, 'nb', nB, L) if nb is not None else 0
    for t_, n_, k_ in ((len_a, 'len_a', nA), (len_b, 'len_b', nB)):
        _chk(t_, n_, torch.int64)
        assert t_ is None or (t_.is_contiguous() and t_.numel() == k_)
    x = torch.empty(nA + nB, L, device=dev)
    nl = len(prods)
    lens = torch.empty(nl, nA + nB, dtype=torch.int64, device=dev) if nl else None
    E, c = 0, None
    if ca is not None:
        E = ca.size(1)
        for t_, n_, k_ in ((ca, 'ca', nA), (cb, 'cb', nB)):
            _chk(t_, n_)
            assert t_ is None or (t_.is_contiguous() and tuple(t_.shape) == (k_, E))
        assert cb is not None or nB == 0
        c = torch.empty(nA + nB, E, device=dev)
    arr = (C.c_int32 * max(nl, 1))(*[int(v) for v in prods])
    check(lib.ag_critic_batch(_p(xa), lda, _p(na), ldna, nA, _p(xb), ldb, _p(nb), ldnb, nB, L, _p(x), _p(len_a), _p(len_b),
                              arr, nl, _p(lens), _p(ca), _p(cb), E, _p(c), _stream()), 'ag_critic_batch')
    return x, lens, c


def axpby(x, y, a, b):
    """y = a*x + b*y (contiguous)"""
    _chk(x, 'x'); _chk(y, 'y')
    assert x.is_contiguous() and y.is_contiguous() and x.numel() == y.numel()
    check(lib.ag_axpby(_p(x), _p(y), x.numel(), float(a), float(b), _stream()), 'ag_axpby')


# ------------------------------------------------------------------------------------
# optimiser
# ------------------------------------------------------------------------------------
def _opt_table(params, grads, s1, s2):
    descs = []
    for i, p in enumerate(params):
        _chk(p, 'param'); _chk(grads[i], 'grad'); _chk(s1[i], 's1')
        assert p.is_contiguous() and grads[i].is_contiguous() and grads[i].numel() == p.numel()
        descs.append(_lib.OptDesc(p.data_ptr(), grads[i].data_ptr(), s1[i].data_ptr(),
                                  s2[i].data_ptr() if s2 is not None else 0, p.numel()))
    return _table('opt', descs, params[0].device)


def grad_norms(params, grads, s1, s2, norms, norm_sum, flags, grad_scale=1.0, step_dev=None, finish=True):
    """norms[i] = ||grads[i]*grad_scale||; norm_sum[0] = sum_i norms[i]; flags = NaN/BIG bits (written).  ``finish=False``:
    only the per-chunk partial sums are computed; returns the workspace holding them, to be handed to ``opt_step(part=...)``
    right behind on the same stream (norms / norm_sum / flags are then written by that launch)."""
    tab = _opt_table(params, grads, s1, s2)
    _chk(norms, 'norms'); _chk(norm_sum, 'norm_sum'); _chk(flags, 'flags', torch.int32)
    assert norms.numel() >= len(params)
    _chk(step_dev, 'step_dev', torch.int32)
    ws = torch.empty(512 * len(params), dtype=torch.float32, device=norms.device)
    check(lib.ag_bind_workspace(_p(ws), ws.numel()), 'ag_bind_workspace')
    check(lib.ag_grad_norms(_p(tab), len(params), _p(norms), _p(norm_sum), _p(flags), grad_scale,
                            _p(step_dev), int(bool(finish)), _stream()), 'ag_grad_norms')
    return ws


def opt_step(params, grads, s1, s2, norms, kind, lr, clip, grad_scale, a1, b2, eps, step, step_dev=None, part=None,
             norm_sum=None, flags=None):
    """``part``: the workspace ``grad_norms(finish=False)`` returned - this launch then also finishes the norms"""
    tab = _opt_table(params, grads, s1, s2)
    _chk(step_dev, 'step_dev', torch.int32); _chk(part, 'part'); _chk(norm_sum, 'norm_sum'); _chk(flags, 'flags', torch.int32)
    assert part is None or part.numel() >= 512 * len(params)
    check(lib.ag_opt_step(_p(tab), len(params), _p(norms), kind, lr, clip, grad_scale, a1, b2, eps,
                          step, _p(step_dev), _p(part), _p(norms) if part is not None else None, _p(norm_sum), _p(flags),
                          _stream()), 'ag_opt_step')


# ------------------------------------------------------------------------------------
# per-kernel timing hook for bench.py's roofline figure: HIP events recorded on the launch
# stream around the launches of ONE kernel class (or all of them in the discovery pass).
# ------------------------------------------------------------------------------------
class Profiler(object):
    enabled = False
    only = None          # kernel-class key to time, or None = every class
    records = []         # (key, flops, bytes, ev_start, ev_stop)

    @classmethod
    def start(cls, only=None):
        cls.enabled, cls.only, cls.records = True, only, []

    @classmethod
    def stop(cls):
        cls.enabled = False
        torch.cuda.synchronize()
        out = {}
        for key, fl, by, e0, e1, nl in cls.records:
            r = out.setdefault(key, dict(n=0, ms=0.0, flops=0.0, bytes=0.0))
            r['n'] += nl
            r['ms'] += e0.elapsed_time(e1)
            r['flops'] += fl
            r['bytes'] += by
        cls.records = []
        return out


def _cdiv(a, b):
    return (a + b - 1) // b


def _bf16_tag():
    """bf16 mode: the persistent recurrent kernels run their v_mfma_f32_*_bf16 instantiation"""
    return '<bf16>' if lib.ag_get_precision() == 1 else ''


def _work_gemm(A, B, Cm, ta=False, tb=False, *a_, **kw):
    M, N = Cm.shape
    Kd = A.size(0) if ta else A.size(1)
    big = _cdiv(M, 128) * _cdiv(N, 128)
    use128 = M > 64 and N > 64 and (big >= 192 or Kd >= 2048)
    # mirrors ag_gemm's dispatch: the LDS-DMA kernel takes the 128x128 case when K % 16 == 0 and rows are 16-B aligned
    dma = use128 and Kd % 16 == 0 and (not ta or M % 4 == 0) and (tb or N % 4 == 0) and \
        A.stride(0) % 4 == 0 and B.stride(0) % 4 == 0 and _al16(A) and _al16(B)
    bf_shape_ok = (M > 32 and N > 32 and Kd % 4 == 0 and A.stride(0) % 4 == 0 and
                   B.stride(0) % 4 == 0 and _al16(A) and _al16(B) and (not ta or M % 4 == 0) and (tb or N % 4 == 0))
    bf = lib.ag_get_precision() == 1 and bf_shape_ok
    x3 = lib.ag_get_precision() == 2 and use128 and bf_shape_ok
    if bf or x3:
        key = 'gemm_bf16_kernel<%d,%d%s>' % (int(ta), int(tb), ',x3' if x3 else '')
    elif dma:
        # mirrors gemm_pick_tile (gemm.hip): the largest tile that still gives every CU a workgroup
        act = kw.get('act', ACT_NONE)
        ks = 1
        wsn = int(lib.ag_gemm_ws_numel(M, N, Kd, int(act)))
        if wsn:
            kchunk = _cdiv(_cdiv(Kd, wsn // (M * N)), 64) * 64
            ks = _cdiv(Kd, kchunk)
        best, best_t = None, 0.0
        for (bm, bn, ti, tj), rate in (((256, 256, 2, 4), 137.), ((256, 128, 2, 2), 133.), ((128, 256, 2, 2), 133.),
                                       ((128, 128, 2, 2), 120.)):
            wgs = _cdiv(M, bm) * _cdiv(N, bn) * ks
            t = _cdiv(wgs, 256) * bm * bn / rate + 1e-9 * wgs * bm * bn
            if best is None or t < best_t * 0.98:
                best, best_t = (bm, bn, ti, tj), t
        key = 'gemm_tile_kernel<%d,%d,%d,%d,%d,%d,0>' % ((int(ta), int(tb)) + best)
    else:
        key = 'gemm_kernel<%s,%d,%d>' % ('2,2,2,2' if use128 else '1,1,2,2', int(ta), int(tb))
    return key, 2.0 * M * N * Kd, 4.0 * (M * Kd + N * Kd + M * N)


def _work_conv(x, wp, y, K_, stride, pad, mode, *a_, **kw):
    B, Cc, Lin = x.shape
    _, O, Lout = y.shape
    rows = O if mode == 0 else O * stride
    ncnt = Lout if mode == 0 else _cdiv(Lout, stride)
    if rows <= 32:
        tile = '1,2,1,4'
    elif rows <= 64:
        tile = '2,1,1,4'
    elif ncnt <= 64:
        tile = '2,1,2,2'
    else:
        tile = '2,2,2,2'
    if mode == 1 and _cdiv(K_, stride) == 2 and rows > 64 and lib.ag_get_precision() != 1 and \
            _os0.environ.get('AG_CONV_HALF', '-1') != '0':
        tile = '2,1,2,2'         # (conv_engine.hip: two tap slots take 128 x 64 tiles)
    macs = B * O * Lout * Cc * K_ if mode == 0 else B * Cc * Lin * O * K_
    if (K_, stride) == (7, 2) and _os0.environ.get('AG_CONV_C1', '1') != '0':
        # the critic's single-input-channel first layer and its backward-data: streaming kernels (conv_c1.hip)
        if mode == 0 and Cc == 1 and kw.get('res') is None and not kw.get('accumulate'):
            return 'conv_c1_fwd_kernel<%d,%d>' % (stride, K_), 2.0 * macs, 4.0 * (x.numel() + y.numel() + O * K_)
        if mode == 1 and O == 1 and kw.get('res') is None and kw.get('bias') is None and kw.get('lens') is None:
            return 'conv_c1_bwdx_kernel<%d,%d>' % (stride, K_), 2.0 * macs, 4.0 * (x.numel() + y.numel() + Cc * K_)
    taps = K_ if mode == 0 else _cdiv(K_, stride)
    ts = (taps, stride if mode == 0 else 0)
    if ts not in ((17, 8), (9, 4), (7, 2), (3, 1), (16, 8), (8, 4), (2, 0), (3, 0), (4, 0)):
        ts = (0, 0)
    nbytes = 4.0 * (x.numel() + y.numel() + O * Cc * K_)
    if lib.ag_get_precision() == 1 and Cc >= 16 and Lin % 4 == 0:
        # mirrors launch_bf16 (conv_engine.hip): the bf16-MFMA kernel takes the launch when a chunk's staging fits
        ot = {'1,2,1,4': 32, '2,1,1,4': 64, '2,1,2,2': 128, '2,2,2,2': 128}[tile]
        tt = {'1,2,1,4': 256, '2,1,1,4': 128, '2,1,2,2': 64, '2,2,2,2': 128}[tile]
        sp = stride if mode == 0 else 1
        ncols = tt + ((K_ - 1) // stride if mode == 0 else taps - 1)
        nq = (sp * ncols + 6) // 4 + 1
        deep = ot >= 128 and (-(-Cc // 16)) * taps >= 32
        if 2 * nq <= 512 and taps * 2 * ot <= (16 if deep else 8) * 256:
            return 'conv_engine_bf16_kernel<%s>' % tile, 2.0 * macs, nbytes
    return 'conv_engine_kernel<%s,%d,%d>' % (tile, ts[0], ts[1]), 2.0 * macs, nbytes


def _work_wgrad(sh, lg, dw, K_, stride, pad):
    B, A, Lsh = sh.shape
    Cc = lg.size(1)
    if Cc == 1 and (K_, stride) == (7, 2) and _os0.environ.get('AG_CONV_C1', '1') != '0':
        return 'conv_c1_wgrad4_kernel<%d,%d>' % (stride, K_), 2.0 * B * A * Lsh * K_, 4.0 * (sh.numel() + lg.numel() + dw.numel())
    tile = '1,1,1,4' if A <= 32 else ('1,1,2,2' if (A <= 64 or Cc * K_ <= 64) else '2,2,2,2')
    return 'conv_wgrad_kernel<%s>' % tile, 2.0 * B * A * Lsh * Cc * K_, 4.0 * (sh.numel() + lg.numel() + dw.numel())


def _instrument(name, work):
    fn = globals()[name]

    def wrapped(*a, **k):
        if not Profiler.enabled:
            return fn(*a, **k)
        w_ = work(*a, **k) if work is not None else (name, 0.0, 0.0)
        key, fl, by = w_[:3]
        nl = w_[3] if len(w_) > 3 else 1        # launches behind this call (a whole recurrent layer pass: T)
        if Profiler.only is not None and key != Profiler.only:
            return fn(*a, **k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*a, **k)
        e1.record()
        Profiler.records.append((key, fl, by, e0, e1, nl))
        return r
    wrapped.__name__ = name
    wrapped.__doc__ = fn.__doc__
    wrapped.__module__ = __name__
    globals()[name] = wrapped


for _n, _w in (('gemm', _work_gemm), ('conv_engine', _work_conv), ('conv_wgrad', _work_wgrad),
               ('channel_sum', None), ('leaky_bwd', None), ('col_sum', None), ('lstm_cell_fwd', None),
               ('lstm_cell_bwd', None), ('weight_norm_fwd', None), ('weight_norm_bwd', None),
               ('bce_logits_fwd', None), ('bce_logits_bwd', None), ('act_fwd', None), ('act_bwd', None),
               ('axpby', None), ('grad_norms', None), ('opt_step', None)):
    _instrument(_n, _w)


# ------------------------------------------------------------------------------------
# skinny products / fused recurrent steps (lstm_step.hip)
# ------------------------------------------------------------------------------------
def _al16(t):
    return t.data_ptr() % 16 == 0


def skinny_ok(A, B, tb):
    """can ag_skinny_gemm take this product? (M <= 64, K % 8 == 0, 16-byte aligned rows)"""
    M, Kd = A.shape
    ok = M <= 256 and Kd % 8 == 0 and A.stride(1) == 1 and A.stride(0) % 4 == 0 and _al16(A)
    if tb:
        ok = ok and B.stride(1) == 1 and B.stride(0) % 4 == 0 and _al16(B)
    return ok


def skinny_gemm(A, B, Cm, tb=False, beta=0.0, bias=None, act=ACT_NONE, slope=LEAKY_SLOPE, atomic=False):
    """Cm[M<=64,N] = act(A @ op(B) + beta*Cm + bias), or with atomic=True: Cm += A @ op(B) (+bias)"""
    lda, ldb, ldc = _mat(A, 'A'), _mat(B, 'B'), _mat(Cm, 'C')
    M, N = Cm.shape
    Kd = A.size(1)
    assert A.size(0) == M and (B.size(1) if tb else B.size(0)) == Kd and (B.size(0) if tb else B.size(1)) == N
    if bias is not None:
        _chk(bias, 'bias')
        assert bias.numel() == N and bias.is_contiguous()
    _ws = _bind_ws(lib.ag_skinny_ws_numel(M, N, Kd), Cm.device) if atomic else None  # noqa: F841
    check(lib.ag_skinny_gemm(_p(A), lda, _p(B), ldb, int(tb), _p(Cm), ldc, M, N, Kd, beta, _p(bias), act,
                             slope, int(atomic), _stream()), 'ag_skinny_gemm')


def lstm_front_bwd_ok(B, H, fs, dxa_t, x_t):
    return (H % 16 == 0 and fs % 16 == 0 and B <= 65535 and dxa_t.stride(1) == 1 and x_t.stride(1) == 1
            and dxa_t.stride(0) % 4 == 0 and x_t.stride(0) % 4 == 0 and _al16(dxa_t) and _al16(x_t))


def lstm_front_bwd_step(dxa_t, x_t, gx_out, w_proj, dh_acc, gates, c_prev, c_new, dc_next, dgates, dc_prev):
    """fused backward of one Generator-front frame: gx = dxa_t*(1-x_t^2) -> gx_out; dh = dh_acc + gx @ w_proj;
    LSTMCell backward -> dgates, dc_prev.  dxa_t / x_t: [B,fs] row views; the rest contiguous."""
    B, fs = dxa_t.shape
    H = w_proj.size(1)
    for t_, n, shp in ((gx_out, 'gx_out', (B, fs)), (w_proj, 'w_proj', (fs, H)),
                       (gates, 'gates', (B, 4 * H)), (c_prev, 'c_prev', (B, H)), (c_new, 'c_new', (B, H)),
                       (dgates, 'dgates', (B, 4 * H)), (dc_prev, 'dc_prev', (B, H))):
        _chk(t_, n)
        assert t_.is_contiguous() and tuple(t_.shape) == shp, (n, tuple(t_.shape), shp)
    _chk(dxa_t, 'dxa_t'); _chk(x_t, 'x_t'); _chk(dc_next, 'dc_next'); _chk(dh_acc, 'dh_acc')
    assert tuple(x_t.shape) == (B, fs) and lstm_front_bwd_ok(B, H, fs, dxa_t, x_t)
    assert tuple(dh_acc.shape) == (B, H) and dh_acc.stride(1) == 1
    assert dc_next is None or (dc_next.is_contiguous() and tuple(dc_next.shape) == (B, H))
    check(lib.ag_lstm_front_bwd_step(_p(dxa_t), dxa_t.stride(0), _p(x_t), x_t.stride(0), _p(gx_out), fs, fs,
                                     _p(w_proj), _p(dh_acc), dh_acc.stride(0), _p(gates), _p(c_prev), _p(c_new), _p(dc_next),
                                     _p(dgates), _p(dc_prev), B, H, _stream()), 'ag_lstm_front_bwd_step')


def lstm_step_ok(B, H, x=None, wx=None):
    ok = B <= 256 and H % 8 == 0
    if x is not None:
        ok = ok and x.size(1) % 8 == 0 and x.stride(1) == 1 and x.stride(0) % 4 == 0 and _al16(x) \
            and wx.stride(1) == 1 and wx.stride(0) % 4 == 0 and _al16(wx)
    return ok


def lstm_step_fwd(gates_pre, x, wx, h_prev, whh, c_prev, c_out, h_out, first_step):
    """fused LSTMCell step; gates_pre [B,4H] contiguous is overwritten with the activated gates"""
    for t_, n in ((gates_pre, 'gates_pre'), (h_prev, 'h_prev'), (whh, 'whh'), (c_prev, 'c_prev'),
                  (c_out, 'c_out'), (h_out, 'h_out')):
        _chk(t_, n)
        assert t_.is_contiguous()
    B, H4 = gates_pre.shape
    H = H4 // 4
    ldx = ldwx = Kx = 0
    if x is not None:
        ldx, ldwx, Kx = _mat(x, 'x'), _mat(wx, 'wx'), x.size(1)
        assert wx.size(0) == 4 * H and wx.size(1) == Kx
    check(lib.ag_lstm_step_fwd(_p(gates_pre), _p(x), ldx, _p(wx), ldwx, Kx, _p(h_prev), _p(whh), _p(c_prev),
                               _p(c_out), _p(h_out), B, H, int(first_step), _stream()), 'ag_lstm_step_fwd')


def _ptr_table(tensors):
    arr = (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    return arr


# ---- persistent (weights-resident) layer pass: workspace per device, status word, switch -------------------------
import os as _os

PERSIST = [_os.environ.get('AG_LSTM_PERSIST', '1') != '0']
_persist_ws = {}
_ncu = {}


def _n_cu(dev):
    n = _ncu.get(dev)
    if n is None:
        n = _ncu[dev] = torch.cuda.get_device_properties(dev).multi_processor_count
    return n


def lstm_persist_ok(B, H, ndir, dev):
    return bool(PERSIST[0] and torch.device(dev).type == 'cuda' and lib.ag_lstm_persist_ok(B, H, ndir, _n_cu(dev)))


_PERSIST_WS_MIN = 8 << 20      # covers every BASELINE shape (biLSTM at 256 clips, H 768: 3.2 MB; the generator front: 0.7 MB)


def _persist_workspace(dev, nbytes):
    """the per-device workspace of the persistent launches (sticky status word + header + exchange buffers).  Allocated
    once at a size that covers every shape the kernels accept at BASELINE batch sizes; if a later call still needs more,
    a NEW buffer becomes current and the old one stays alive for good: a captured hipGraph has its pointer baked into
    memset and kernel nodes and keeps using it on every replay (freeing it would hand that memory to another tensor)."""
    lst = _persist_ws.setdefault(dev, [])
    if not lst or lst[0].numel() < nbytes:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError('audiogan_amd: the persistent LSTM workspace must exist before hipGraph capture '
                               '(run one eager step first)')
        lst.insert(0, torch.zeros(max(int(nbytes), _PERSIST_WS_MIN), dtype=torch.uint8, device=dev))
    return lst[0]


def lstm_persist_status(dev=None, reset=False):
    """sticky status of EVERY persistent launch on `dev` since the workspace was allocated (or since the last
    ``reset=True``): 0 = all completed (host sync).  Non-zero: a bounded wait timed out, i.e. a launch's workgroups were
    not all co-resident (another persistent or RCCL kernel holding CUs, a CU mask, a partition mode); everything that
    launch produced afterwards is NaN."""
    dev = torch.device('cuda', torch.cuda.current_device()) if dev is None else torch.device(dev)
    if dev.index is None:
        dev = torch.device('cuda', torch.cuda.current_device())
    st = 0
    for ws in _persist_ws.get(dev, []):
        st |= int(ws[:4].view(torch.int32).item()) & 0xFFFFFFFF
        if reset:
            ws[:4].zero_()
    return st


class PersistentLaunchError(RuntimeError):
    pass


def check_persist_status(dev=None):
    """raise if any persistent launch on `dev` gave up (see lstm_persist_status); resets the sticky word"""
    st = lstm_persist_status(dev)
    if st:
        lstm_persist_status(dev, reset=True)
        raise PersistentLaunchError(
            'audiogan_amd: a persistent recurrent launch timed out waiting for its group (status 0x%08x: step %d); its '
            'workgroups were not all co-resident - is another persistent or collective kernel holding compute units? '
            'Results of that launch are NaN.  AG_LSTM_PERSIST=0 selects the per-step kernels.' % (st, st & 0x7FFFFFFF))


def persist_debug(timeout_ticks=0, mute_block=-1):
    """test hook (ag_persist_debug)"""
    check(lib.ag_persist_debug(int(timeout_ticks), int(mute_block)), 'ag_persist_debug')


def lstm_seq_fwd_persist(pre, whh, c_all, y, valid, static=None):
    """ONE persistent launch for the whole layer pass (ag_lstm_seq_fwd_persist)"""
    ndir = len(pre)
    T, B, H4 = pre[0].shape
    H = H4 // 4
    for d in range(ndir):
        for t_, shp in ((pre[d], (T, B, 4 * H)), (whh[d], (4 * H, H)), (c_all[d], (T + 1, B, H))):
            _chk(t_, 'lstm_seq tensor')
            assert t_.is_contiguous() and tuple(t_.shape) == shp, (tuple(t_.shape), shp)
    _chk_any(y, 'y'); _chk(valid, 'valid', torch.int64)
    assert y.is_contiguous() and tuple(y.shape) == (T, B, ndir * H)
    if static is not None:
        assert len(static) == ndir
        for t_ in static:
            _chk(t_, 'static')
            assert t_.is_contiguous() and tuple(t_.shape) == (B, 4 * H)
    nb = int(lib.ag_lstm_persist_ws_bytes(B, H, ndir))
    ws = _persist_workspace(y.device, nb)
    check(lib.ag_lstm_seq_fwd_persist(_ptr_table(pre), _ptr_table(whh), _ptr_table(c_all), _p(y), int(_is16(y)), _p(valid),
                                      _ptr_table(static) if static is not None else None, _p(ws), ws.numel(),
                                      T, B, H, ndir, _n_cu(y.device), _stream()), 'ag_lstm_seq_fwd_persist')


def lstm_seq_fwd(pre, whh, c_all, hbuf, y, valid, static=None):
    """pre/whh/c_all/hbuf: lists (one per direction) of contiguous tensors; static: optional list of [B,4H]
    tensors added to every step's pre-activations; see ag_lstm_seq_fwd.  Shapes that fit the chip take the
    persistent weights-resident launch (AG_LSTM_PERSIST=0 forces one launch per step)."""
    T = pre[0].size(0)
    if lstm_persist_ok(pre[0].size(1), pre[0].size(2) // 4, len(pre), y.device):
        _lstm_seq_fwd_persist_call(pre, whh, c_all, y, valid, static)
        return
    # (the profiler times the whole chain of T back-to-back launches with one pair of events and divides by T:
    # events around every single launch add ~3 us to a 10 us kernel)
    _lstm_seq_fwd_range(pre, whh, c_all, hbuf, y, valid, 0, T, static)


def _lstm_seq_fwd_persist_call(pre, whh, c_all, y, valid, static=None):
    lstm_seq_fwd_persist(pre, whh, c_all, y, valid, static)


def _lstm_seq_fwd_range(pre, whh, c_all, hbuf, y, valid, k0, k1, static=None):
    ndir = len(pre)
    T, B, H4 = pre[0].shape
    H = H4 // 4
    for d in range(ndir):
        for t_, shp in ((pre[d], (T, B, 4 * H)), (whh[d], (4 * H, H)), (c_all[d], (T + 1, B, H)),
                        (hbuf[d], (2, B, H))):
            _chk(t_, 'lstm_seq tensor')
            assert t_.is_contiguous() and tuple(t_.shape) == shp, (tuple(t_.shape), shp)
    _chk(y, 'y'); _chk(valid, 'valid', torch.int64)
    assert y.is_contiguous() and tuple(y.shape) == (T, B, ndir * H)
    if static is not None:
        assert len(static) == ndir
        for t_ in static:
            _chk(t_, 'static')
            assert t_.is_contiguous() and tuple(t_.shape) == (B, 4 * H)
    check(lib.ag_lstm_seq_fwd(_ptr_table(pre), _ptr_table(whh), _ptr_table(c_all), _ptr_table(hbuf), _p(y),
                              _p(valid), _ptr_table(static) if static is not None else None, T, B, H, ndir, k0, k1,
                              _stream()), 'ag_lstm_seq_fwd')


def gfront_persist_ok(B, S, fs, dev):
    return bool(PERSIST[0] and torch.device(dev).type == 'cuda' and lib.ag_gfront_persist_ok(B, S, fs, _n_cu(dev)))


def _rows(t, name, B, n):
    """row pitch of a [B, n] view with unit stride along dim 1 (a slab channel, a column block)"""
    _chk(t, name)
    assert t.dim() == 2 and tuple(t.shape) == (B, n) and (n == 1 or t.stride(1) == 1), (name, tuple(t.shape), t.stride())
    return t.stride(0) if B > 1 else max(t.stride(0), n)


def gfront_fwd_persist(gates, wx, whh, wp, bp, hs, cs, x, xt=None):
    """the Generator front's frame loop in ONE persistent launch (ag_gfront_fwd_persist).  x: [B, T*fs] rows of any pitch
    (channel 0 of the conv trunk's slab); xt: optional [T,B,fs], receives the same frames time-major"""
    T, B, S4 = gates.shape
    S = S4 // 4
    fs = wp.size(0)
    for t_, n, shp in ((gates, 'gates', (T, B, 4 * S)), (whh, 'whh', (4 * S, S)), (wp, 'wp', (fs, S)), (bp, 'bp', (fs,)),
                       (hs, 'hs', (T, B, S)), (cs, 'cs', (T + 1, B, S))) + (((xt, 'xt', (T, B, fs)),) if xt is not None else ()):
        _chk(t_, n)
        assert t_.is_contiguous() and tuple(t_.shape) == shp, (n, tuple(t_.shape), shp)
    ldx = _rows(x, 'x', B, T * fs)
    _chk(wx, 'wx')
    assert tuple(wx.shape) == (4 * S, fs) and wx.stride(1) == 1
    nb = int(lib.ag_gfront_persist_ws_bytes(B, S, fs))
    ws = _persist_workspace(x.device, nb)
    check(lib.ag_gfront_fwd_persist(_p(gates), _p(wx), wx.stride(0), _p(whh), _p(wp), _p(bp), _p(hs), _p(cs), _p(x), ldx,
                                    _p(xt), _p(ws), ws.numel(), T, B, S, fs, _n_cu(x.device), _stream()),
          'ag_gfront_fwd_persist')


# the front's backward runs beside the conv trunk's gradient all-reduce in the multi-GPU step (train.GraphedStep, phase
# g3b; eager: the bucket's hook): a persistent launch wants its CUs to itself, so a generator whose optimiser carries a
# gradient bucket runs its backward with this switch off (train.g_step / g_backward / GraphedStep)
PERSIST_FRONT_BWD = [True]


class front_bwd_persist(object):
    """``with K.front_bwd_persist(on): loss.backward()`` - `on` False: the front's backward takes the per-frame form"""

    def __init__(self, on):
        self.on = bool(on)

    def __enter__(self):
        self.old = PERSIST_FRONT_BWD[0]
        PERSIST_FRONT_BWD[0] = self.old and self.on

    def __exit__(self, *exc):
        PERSIST_FRONT_BWD[0] = self.old


def gfront_bwd_persist_ok(B, S, fs, dev):
    return bool(PERSIST[0] and PERSIST_FRONT_BWD[0] and torch.device(dev).type == 'cuda'
                and lib.ag_gfront_bwd_persist_ok(B, S, fs, _n_cu(dev)))


def _ext_grads(dh_ext, dx_ext, T, B, S, fs):
    if dh_ext is not None:
        _chk(dh_ext, 'dh_ext')
        assert dh_ext.is_contiguous() and tuple(dh_ext.shape) == (T, B, S), tuple(dh_ext.shape)
    return _rows(dx_ext, 'dx_ext', B, T * fs) if dx_ext is not None else 0


def gfront_bwd_persist(gates, cs, x, dh_ext, dx_ext, whh, wx, wp, dgs, dxt):
    """the backward through time of the Generator front's frame loop in ONE persistent launch (ag_gfront_bwd_persist);
    the external gradients, read only, each may be None: dh_ext [T,B,S] = dL/dh_t (the stop head's), dx_ext [B,T*fs] rows of
    any pitch = dL/dx_t (the conv trunk's: channel 0 of its gradient slab); x likewise [B,T*fs] rows of any pitch"""
    T, B, S4 = gates.shape
    S = S4 // 4
    fs = wp.size(0)
    for t_, n, shp in ((gates, 'gates', (T, B, 4 * S)), (cs, 'cs', (T + 1, B, S)), (whh, 'whh', (4 * S, S)),
                       (wp, 'wp', (fs, S)), (dgs, 'dgs', (T, B, 4 * S)), (dxt, 'dxt', (T, B, fs))):
        _chk(t_, n)
        assert t_.is_contiguous() and tuple(t_.shape) == shp, (n, tuple(t_.shape), shp)
    ldx, lddx = _rows(x, 'x', B, T * fs), _ext_grads(dh_ext, dx_ext, T, B, S, fs)
    _chk(wx, 'wx')
    assert tuple(wx.shape) == (4 * S, fs) and wx.stride(1) == 1
    ws = _persist_workspace(x.device, _PERSIST_WS_MIN)
    check(lib.ag_gfront_bwd_persist(_p(gates), _p(cs), _p(x), ldx, _p(dh_ext), _p(dx_ext), lddx, _p(whh), _p(wx),
                                    wx.stride(0), _p(wp), _p(dgs), _p(dxt), _p(ws), ws.numel(), T, B, S, fs,
                                    _n_cu(x.device), _stream()), 'ag_gfront_bwd_persist')


def grufront_bwd_persist(gates, hs, gh, x, dh_ext, dx_ext, whh, wx, wp, dgi, dgh, dxt):
    """the GRU front's backward through time in ONE persistent launch (ag_grufront_bwd_persist); hs [T+1,B,S] with
    hs[t] = h_{t-1}; dh_ext / dx_ext: the external gradients as for gfront_bwd_persist"""
    T, B, S3 = gates.shape
    S = S3 // 3
    fs = wp.size(0)
    for t_, n, shp in ((gates, 'gates', (T, B, 3 * S)), (hs, 'hs', (T + 1, B, S)), (gh, 'gh', (T, B, 3 * S)),
                       (whh, 'whh', (3 * S, S)), (wp, 'wp', (fs, S)),
                       (dgi, 'dgi', (T, B, 3 * S)), (dgh, 'dgh', (T, B, 3 * S)), (dxt, 'dxt', (T, B, fs))):
        _chk(t_, n)
        assert t_.is_contiguous() and tuple(t_.shape) == shp, (n, tuple(t_.shape), shp)
    ldx, lddx = _rows(x, 'x', B, T * fs), _ext_grads(dh_ext, dx_ext, T, B, S, fs)
    _chk(wx, 'wx')
    assert tuple(wx.shape) == (3 * S, fs) and wx.stride(1) == 1
    ws = _persist_workspace(x.device, _PERSIST_WS_MIN)
    check(lib.ag_grufront_bwd_persist(_p(gates), _p(hs), _p(gh), _p(x), ldx, _p(dh_ext), _p(dx_ext), lddx, _p(whh), _p(wx),
                                      wx.stride(0), _p(wp), _p(dgi), _p(dgh), _p(dxt), _p(ws), ws.numel(), T, B, S, fs,
                                      _n_cu(x.device), _stream()), 'ag_grufront_bwd_persist')


def _work_grufront_bwd(gates, hs, gh, x, dh_ext, dx_ext, whh, wx, wp, *a_, **kw):
    T, B, S3 = gates.shape
    S, fs = S3 // 3, wp.size(0)
    return 'gfront_persist_bwd_kernel<gru>', 2.0 * B * ((T - 1) * S3 * (S + fs) + T * fs * S), \
        4.0 * (S3 * (S + fs) + fs * S + T * B * (4 * S3 + 2 * S + (S + fs) + 2 * fs)), 1


def grufront_fwd_persist(gates, gh, wx, whh, bhn, wp, bp, hs, x, xt=None):
    """the GRU-front generator's frame loop in ONE persistent launch (ag_grufront_fwd_persist); hs: [T,B,S] (h_t); x / xt as
    for gfront_fwd_persist"""
    T, B, S3 = gates.shape
    S = S3 // 3
    fs = wp.size(0)
    for t_, n, shp in ((gates, 'gates', (T, B, 3 * S)), (gh, 'gh', (T, B, 3 * S)), (whh, 'whh', (3 * S, S)), (bhn, 'bhn', (S,)),
                       (wp, 'wp', (fs, S)), (bp, 'bp', (fs,)), (hs, 'hs', (T, B, S))) + (((xt, 'xt', (T, B, fs)),) if xt is not None else ()):
        _chk(t_, n)
        assert t_.is_contiguous() and tuple(t_.shape) == shp, (n, tuple(t_.shape), shp)
    ldx = _rows(x, 'x', B, T * fs)
    _chk(wx, 'wx')
    assert tuple(wx.shape) == (3 * S, fs) and wx.stride(1) == 1
    nb = int(lib.ag_gfront_persist_ws_bytes(B, S, fs))
    ws = _persist_workspace(x.device, nb)
    check(lib.ag_grufront_fwd_persist(_p(gates), _p(gh), _p(wx), wx.stride(0), _p(whh), _p(bhn), _p(wp), _p(bp), _p(hs),
                                      _p(x), ldx, _p(xt), _p(ws), ws.numel(), T, B, S, fs, _n_cu(x.device), _stream()),
          'ag_grufront_fwd_persist')


def _work_grufront(gates, gh, wx, whh, bhn, wp, *a_, **kw):
    T, B, S3 = gates.shape
    S, fs = S3 // 3, wp.size(0)
    return 'gfront_persist_fwd_kernel<gru>', T * 2.0 * B * (S3 * (S + fs) + fs * S), \
        4.0 * (S3 * (S + fs) + fs * S + T * B * (2 * S3 + 2 * S + fs)), 1


def _work_gfront(gates, wx, whh, wp, *a_, **kw):
    T, B, S4 = gates.shape
    S, fs = S4 // 4, wp.size(0)
    return 'gfront_persist_fwd_kernel' + _bf16_tag(), T * 2.0 * B * (S4 * (S + fs) + fs * S), \
        4.0 * (S4 * (S + fs) + fs * S + T * B * (2 * S4 + 2 * S + fs)), 1


def _work_gfront_bwd(gates, cs, x, dh_ext, dx_ext, whh, wx, wp, *a_, **kw):
    T, B, S4 = gates.shape
    S, fs = S4 // 4, wp.size(0)
    return 'gfront_persist_bwd_kernel' + _bf16_tag(), 2.0 * B * ((T - 1) * S4 * (S + fs) + T * fs * S), \
        4.0 * (S4 * (S + fs) + fs * S + T * B * (2 * S4 + 2 * S + (S + fs) + 2 * fs)), 1


def lstm_persist_bwd_ok(B, H, ndir, dev):
    return bool(PERSIST[0] and lib.ag_lstm_persist_bwd_ok(B, H, ndir, _n_cu(dev)))


def _lstm_seq_bwd_persist_call(gates, whh, c_all, dy, dgates, valid, dgsum=None, dg16=None):
    """ONE persistent launch for the whole backward through time (ag_lstm_seq_bwd_persist); dgsum: optional list of
    [B,4H] outputs (one per direction) = the sum over time of dgates; dy: fp32 or bf16; dg16: optional list of [T,B,4H]
    bfloat16 outputs = dgates rounded (the operand of the gradient products on bf16 storage)"""
    ndir = len(gates)
    T, B, H4 = gates[0].shape
    H = H4 // 4
    for d in range(ndir):
        for t_, shp in ((gates[d], (T, B, 4 * H)), (whh[d], (4 * H, H)), (c_all[d], (T + 1, B, H)),
                        (dgates[d], (T, B, 4 * H))):
            _chk(t_, 'lstm_seq tensor')
            assert t_.is_contiguous() and tuple(t_.shape) == shp, (tuple(t_.shape), shp)
    _chk_any(dy, 'dy'); _chk(valid, 'valid', torch.int64)
    assert dy.is_contiguous() and tuple(dy.shape) == (T, B, ndir * H)
    if dgsum is not None:
        assert len(dgsum) == ndir
        for t_ in dgsum:
            _chk(t_, 'dgsum')
            assert t_.is_contiguous() and tuple(t_.shape) == (B, 4 * H)
    ld16 = 0
    if dg16 is not None:
        assert len(dg16) == ndir
        for t_ in dg16:      # views [T,B,4H] of one [T,B,ndir*4H] tensor (or separate contiguous tensors): a common row pitch
            _chk(t_, 'dg16', torch.bfloat16)
            assert tuple(t_.shape) == (T, B, 4 * H) and t_.stride(2) == 1 and t_.stride(0) == B * t_.stride(1)
        ld16 = dg16[0].stride(1)
        assert all(t_.stride(1) == ld16 for t_ in dg16)
    ws = _persist_workspace(dy.device, 8192 + 256)
    check(lib.ag_lstm_seq_bwd_persist(_ptr_table(gates), _ptr_table(whh), _ptr_table(c_all), _p(dy), int(_is16(dy)),
                                      _ptr_table(dgates), _ptr_table(dgsum) if dgsum is not None else None,
                                      _ptr_table(dg16) if dg16 is not None else None, ld16, _p(valid),
                                      _p(ws), ws.numel(), T, B, H, ndir, _n_cu(dy.device), _stream()),
          'ag_lstm_seq_bwd_persist')


def _work_seq_bwd_persist(gates, whh, c_all, dy, dgates, valid, dgsum=None, dg16=None):
    T, B, H4 = gates[0].shape
    nd = len(gates)
    return 'lstm_persist_bwd_kernel' + _bf16_tag(), T * 2.0 * nd * B * H4 * (H4 // 4), \
        4.0 * nd * (H4 * (H4 // 4) + T * 5.5 * B * H4), 1


def lstm_seq_bwd(gates, whh, c_all, dy, dgates, dhbuf, dcbuf, valid, dgsum=None, dg16=None):
    """``dgsum``: optional list of [B,4H] tensors (one per direction) that receive the sum over time of dgates.  Returns True
    when they were filled (the persistent launch sums them in registers); False: the caller sums dgates itself.
    ``dg16`` / a bf16 ``dy``: the persistent launch only (callers check lstm_persist_bwd_ok first)."""
    T = gates[0].size(0)
    if lstm_persist_bwd_ok(gates[0].size(1), gates[0].size(2) // 4, len(gates), dy.device):
        _lstm_seq_bwd_persist_call(gates, whh, c_all, dy, dgates, valid, dgsum, dg16)
        return dgsum is not None
    assert dg16 is None and not _is16(dy), 'bf16 storage needs the persistent launch'

    if Profiler.enabled:
        H = gates[0].size(2) // 4
        if H % 16 == 0:         # the fused step kernel: the whole chain between one pair of events
            _lstm_seq_bwd_step(gates, whh, c_all, dy, dgates, dhbuf, dcbuf, valid, 0, T, 3)
            return
        for k_ in reversed(range(T)):
            _lstm_seq_bwd_cell(gates, whh, c_all, dy, dgates, dhbuf, dcbuf, valid, k_, k_ + 1, 1)
            if k_ > 0:
                _lstm_seq_bwd_prod(gates, whh, c_all, dy, dgates, dhbuf, dcbuf, valid, k_, k_ + 1, 2)
    else:
        _lstm_seq_bwd_range(gates, whh, c_all, dy, dgates, dhbuf, dcbuf, valid, 0, T)
    return False


def _lstm_seq_bwd_cell(*a):
    _lstm_seq_bwd_range(*a)


def _lstm_seq_bwd_prod(*a):
    _lstm_seq_bwd_range(*a)


def _lstm_seq_bwd_step(*a):
    _lstm_seq_bwd_range(*a)


def _lstm_seq_bwd_range(gates, whh, c_all, dy, dgates, dhbuf, dcbuf, valid, k0, k1, phases=3):
    ndir = len(gates)
    T, B, H4 = gates[0].shape
    H = H4 // 4
    for d in range(ndir):
        for t_, shp in ((gates[d], (T, B, 4 * H)), (whh[d], (4 * H, H)), (c_all[d], (T + 1, B, H)),
                        (dgates[d], (T, B, 4 * H)), (dhbuf[d], (2, B, H)), (dcbuf[d], (2, B, H))):
            _chk(t_, 'lstm_seq tensor')
            assert t_.is_contiguous() and tuple(t_.shape) == shp, (tuple(t_.shape), shp)
    _chk(dy, 'dy'); _chk(valid, 'valid', torch.int64)
    assert dy.is_contiguous() and tuple(dy.shape) == (T, B, ndir * H)
    _ws = _bind_ws(ndir * lib.ag_skinny_ws_numel(B, H, 4 * H), dy.device) if H % 16 else None  # noqa: F841
    check(lib.ag_lstm_seq_bwd(_ptr_table(gates), _ptr_table(whh), _ptr_table(c_all), _p(dy),
                              _ptr_table(dgates), _ptr_table(dhbuf), _ptr_table(dcbuf), _p(valid), T, B, H,
                              ndir, k0, k1, phases, _stream()), 'ag_lstm_seq_bwd')


def _work_skinny(A, B, Cm, tb=False, *a_, **kw):
    M, N = Cm.shape
    return 'skinny_gemm_kernel', 2.0 * M * N * A.size(1), \
        4.0 * (A.numel() + N * A.size(1) + M * N)


def _work_step(gates_pre, x, wx, h_prev, whh, *a_, **kw):
    B, H4 = gates_pre.shape
    Kd = H4 // 4 + (x.size(1) if x is not None else 0)
    return 'lstm_step_fwd_kernel', 2.0 * B * H4 * Kd, 4.0 * (H4 * Kd + 3 * B * H4)


def _work_seq_fwd(pre, whh, c_all, hbuf, y, valid, k0, k1, *a_, **kw):
    T, B, H4 = pre[0].shape
    nd, n = len(pre), k1 - k0
    return 'lstm_step_fwd_kernel', n * 2.0 * nd * B * H4 * (H4 // 4), n * 4.0 * nd * (H4 * (H4 // 4) + 3 * B * H4), n


def _work_seq_fwd_persist(pre, whh, c_all, y, valid, static=None):
    T, B, H4 = pre[0].shape
    nd = len(pre)
    # one launch = the whole layer pass: T steps of 2*B*4H*H flops per direction; algorithmic bytes: W_hh ONCE
    # + gates in/out, cells, outputs per step
    return 'lstm_persist_fwd_kernel' + _bf16_tag(), T * 2.0 * nd * B * H4 * (H4 // 4), \
        4.0 * nd * (H4 * (H4 // 4) + T * 3 * B * H4), 1


def _work_seq_bwd_prod(gates, whh, *a_, **kw):
    T, B, H4 = gates[0].shape
    nd = len(gates)
    return 'skinny_gemm_kernel', 2.0 * nd * B * H4 * (H4 // 4), 4.0 * nd * (H4 * (H4 // 4) + 2 * B * H4)


def _work_seq_bwd_step(gates, whh, c_all, dy, dgates, dhbuf, dcbuf, valid, k0, k1, *a_, **kw):
    T, B, H4 = gates[0].shape
    nd, n = len(gates), k1 - k0
    return 'lstm_step_bwd_kernel', n * 2.0 * nd * B * H4 * (H4 // 4), n * 4.0 * nd * (H4 * (H4 // 4) + 5.5 * B * H4), n


def _work_seq_bwd_cell(gates, whh, *a_, **kw):
    T, B, H4 = gates[0].shape
    return 'lstm_cell_bwd2_kernel', 0.0, 4.0 * len(gates) * B * H4 * 3.5


for _n, _w in (('skinny_gemm', _work_skinny), ('lstm_step_fwd', _work_step),
               ('_lstm_seq_fwd_range', _work_seq_fwd), ('_lstm_seq_fwd_persist_call', _work_seq_fwd_persist),
               ('_lstm_seq_bwd_persist_call', _work_seq_bwd_persist), ('gfront_fwd_persist', _work_gfront), ('gfront_bwd_persist', _work_gfront_bwd), ('grufront_bwd_persist', _work_grufront_bwd),
               ('grufront_fwd_persist', _work_grufront), ('_lstm_seq_bwd_prod', _work_seq_bwd_prod),
               ('_lstm_seq_bwd_cell', _work_seq_bwd_cell), ('_lstm_seq_bwd_step', _work_seq_bwd_step)):
    _instrument(_n, _w)


# ------------------------------------------------------------------------------------
# GRU cell pointwise (config C4)
# ------------------------------------------------------------------------------------
def gru_cell_fwd(gi, gh, h_prev, h_out):
    """gi, gh: [B,3H] contiguous (complete products incl. biases); gi <- (r,z,n); h_out <- h'"""
    for t_, n in ((gi, 'gi'), (gh, 'gh')):
        _chk(t_, n)
        assert t_.is_contiguous()
    B, H3 = gi.shape
    H = H3 // 3
    assert tuple(gh.shape) == (B, H3) and tuple(h_prev.shape) == (B, H) and tuple(h_out.shape) == (B, H)
    check(lib.ag_gru_cell_fwd(_p(gi), _p(gh), _p(h_prev), _mat(h_prev, 'h_prev'), _p(h_out), _mat(h_out, 'h_out'),
                              B, H, _stream()), 'ag_gru_cell_fwd')


def gru_cell_bwd(gates_act, gh, h_prev, dh, dgi, dgh, dh_prev):
    for t_, n in ((gates_act, 'gates_act'), (gh, 'gh'), (dgi, 'dgi'), (dgh, 'dgh')):
        _chk(t_, n)
        assert t_.is_contiguous()
    B, H3 = gates_act.shape
    H = H3 // 3
    check(lib.ag_gru_cell_bwd(_p(gates_act), _p(gh), _p(h_prev), _mat(h_prev, 'h_prev'), _p(dh), _mat(dh, 'dh'),
                              _p(dgi), _p(dgh), _p(dh_prev), _mat(dh_prev, 'dh_prev'), B, H, _stream()),
          'ag_gru_cell_bwd')


for _n in ('gru_cell_fwd', 'gru_cell_bwd', 'rowdot_fwd', 'rowdot_bwd', 'build_zc', 'critic_batch', 'to_bf16'):
    _instrument(_n, None)
_instrument('gemm_h', _work_gemm_h)


def act_bwd2d(dy, y, dx, act, slope=LEAKY_SLOPE):
    """dx = dy * act'(.) on 2-D row-strided views (unit stride along dim 1)"""
    a, b, c = _mat(dy, 'dy'), _mat(y, 'y'), _mat(dx, 'dx')
    assert tuple(dy.shape) == tuple(y.shape) == tuple(dx.shape)
    check(lib.ag_act_bwd2d(_p(dy), a, _p(y), b, _p(dx), c, dy.size(0), dy.size(1), act, slope, _stream()),
          'ag_act_bwd2d')


_instrument('act_bwd2d', None)


# ------------------------------------------------------------------------------------
# single-output-channel stride-1 conv (Generator's final conv)
# ------------------------------------------------------------------------------------
def conv_o1_ok(spec_kind, cout, K_, stride, pad):
    return spec_kind == 'conv' and cout == 1 and stride == 1 and K_ <= 9 and 2 * pad == K_ - 1


def conv_o1_fwd(x, w, bias, y, K_, pad, act=ACT_NONE, slope=LEAKY_SLOPE):
    """x [B,C,L] view, w [1,C,K] contiguous, y [B,1,L] view"""
    x_bs, x_cs = _bcl(x, 'x')
    y_bs, _ = _bcl(y, 'y')
    _chk(w, 'w'); _chk(bias, 'bias')
    B, Cc, L = x.shape
    assert w.is_contiguous() and w.numel() == Cc * K_ and tuple(y.shape) == (B, 1, L)
    check(lib.ag_conv1d_o1_fwd(_p(x), x_bs, x_cs, _p(w), _p(bias), _p(y), y_bs, B, Cc, L, K_, pad, act, slope,
                               _stream()), 'ag_conv1d_o1_fwd')


def conv_o1_bwd_data(dy, w, dx, K_, pad, accumulate=False):
    dy_bs, _ = _bcl(dy, 'dy')
    dx_bs, dx_cs = _bcl(dx, 'dx')
    B, Cc, L = dx.shape
    assert w.is_contiguous() and w.numel() == Cc * K_ and tuple(dy.shape) == (B, 1, L)
    check(lib.ag_conv1d_o1_bwd_data(_p(dy), dy_bs, _p(w), _p(dx), dx_bs, dx_cs, B, Cc, L, K_, pad,
                                    int(accumulate), _stream()), 'ag_conv1d_o1_bwd_data')


def conv_o1_wgrad(dy, x, dw, K_, pad):
    dy_bs, _ = _bcl(dy, 'dy')
    x_bs, x_cs = _bcl(x, 'x')
    B, Cc, L = x.shape
    assert dw.is_contiguous() and dw.numel() == Cc * K_ and tuple(dy.shape) == (B, 1, L)
    _ws = _bind_ws(_SMALL_WS, dw.device)  # noqa: F841
    check(lib.ag_conv1d_o1_wgrad(_p(dy), dy_bs, _p(x), x_bs, x_cs, _p(dw), B, Cc, L, K_, pad, _stream()),
          'ag_conv1d_o1_wgrad')


for _n in ('conv_o1_fwd', 'conv_o1_bwd_data', 'conv_o1_wgrad'):
    _instrument(_n, None)


# ------------------------------------------------------------------------------------
# feature-matching statistics over time (calc_dists)
# ------------------------------------------------------------------------------------
def time_moments_fwd(h, lens, m, s, f):
    """h [B,C,L] view (unit stride along L), lens int64 [B]; m, s, f [B,C] contiguous (written)"""
    a = _bcl(h, 'h')
    _chk(lens, 'lens', torch.int64)
    B, Cc, L = h.shape
    for t_, n in ((m, 'm'), (s, 's'), (f, 'f')):
        _chk(t_, n)
        assert t_.is_contiguous() and tuple(t_.shape) == (B, Cc)
    check(lib.ag_time_moments_fwd(_p(h), a[0], a[1], _p(lens), _p(m), _p(s), _p(f), B, Cc, L, _stream()),
          'ag_time_moments_fwd')


def time_moments_bwd(h, lens, gm, gs, gf, dh):
    a, d = _bcl(h, 'h'), _bcl(dh, 'dh')
    _chk(lens, 'lens', torch.int64)
    B, Cc, L = h.shape
    assert tuple(dh.shape) == (B, Cc, L)
    for t_, n in ((gm, 'gm'), (gs, 'gs'), (gf, 'gf')):
        _chk(t_, n)
        assert t_ is None or (t_.is_contiguous() and tuple(t_.shape) == (B, Cc))
    check(lib.ag_time_moments_bwd(_p(h), a[0], a[1], _p(lens), _p(gm), _p(gs), _p(gf), _p(dh), d[0], d[1], B, Cc, L,
                                  _stream()), 'ag_time_moments_bwd')


# ------------------------------------------------------------------------------------
# Conv2DLSTMCell pieces (csrc/convlstm.hip); every map [H, B, C, W] contiguous, peephole weights [H, F, W]
# ------------------------------------------------------------------------------------
def _hbcw(t, name, shape=None):
    _chk(t, name)
    assert t.is_contiguous() and t.dim() == 4, name + ' must be a contiguous [H,B,C,W] map'
    if shape is not None:
        assert tuple(t.shape) == tuple(shape), (name, tuple(t.shape), tuple(shape))
    return t


def _peep(t, name, H, F, W):
    if t is None:
        return None
    _chk(t, name)
    assert t.is_contiguous() and tuple(t.shape) == (H, F, W), (name, tuple(t.shape))
    return t


def convlstm_peephole_fwd(y, c, wci, wcf, j, ip, fp, o):
    H, B, F, W = c.shape
    _hbcw(y, 'y', (H, B, 4 * F, W))
    for t_, n in ((c, 'c'), (j, 'j'), (ip, 'i_pre'), (fp, 'f_pre'), (o, 'o_raw')):
        _hbcw(t_, n, (H, B, F, W))
    check(lib.ag_convlstm_peephole_fwd(_p(y), _p(c), _p(_peep(wci, 'w_ci', H, F, W)), _p(_peep(wcf, 'w_cf', H, F, W)), _p(j),
                                       _p(ip), _p(fp), _p(o), H, B, F, W, _stream()), 'ag_convlstm_peephole_fwd')


def convlstm_peephole_bwd(dj, di, df, do, c, wci, wcf, dy, dc, dwci, dwcf):
    """dy [H,B,4F,W] written; dc accumulated into; dwci / dwcf written when given"""
    H, B, F, W = c.shape
    _hbcw(dy, 'dy', (H, B, 4 * F, W))
    for t_, n in ((dj, 'dj'), (di, 'di'), (df, 'df'), (do, 'do'), (c, 'c'), (dc, 'dc')):
        _hbcw(t_, n, (H, B, F, W))
    check(lib.ag_convlstm_peephole_bwd(_p(dj), _p(di), _p(df), _p(do), _p(c), _p(_peep(wci, 'w_ci', H, F, W)),
                                       _p(_peep(wcf, 'w_cf', H, F, W)), _p(dy), _p(dc), _p(_peep(dwci, 'dw_ci', H, F, W)),
                                       _p(_peep(dwcf, 'dw_cf', H, F, W)), H, B, F, W, _stream()), 'ag_convlstm_peephole_bwd')


def convlstm_cell_fwd(j, i_, f_, c, o_raw, wco, forget_bias, c_new, o_pre):
    H, B, F, W = c.shape
    for t_, n in ((j, 'j'), (i_, 'i'), (f_, 'f'), (c, 'c'), (o_raw, 'o_raw'), (c_new, 'c_new'), (o_pre, 'o_pre')):
        _hbcw(t_, n, (H, B, F, W))
    check(lib.ag_convlstm_cell_fwd(_p(j), _p(i_), _p(f_), _p(c), _p(o_raw), _p(_peep(wco, 'w_co', H, F, W)), float(forget_bias),
                                   _p(c_new), _p(o_pre), H, B, F, W, _stream()), 'ag_convlstm_cell_fwd')


def convlstm_cell_bwd(j, i_, f_, c, c_new, wco, forget_bias, dc_new, do_pre, dj, di, df, dc, dwco):
    """dc_new: in = dL/dc', out = dL/dc' + do_pre * W_co; dj, di, df, dc written; dwco written when given"""
    H, B, F, W = c.shape
    for t_, n in ((j, 'j'), (i_, 'i'), (f_, 'f'), (c, 'c'), (c_new, 'c_new'), (dc_new, 'dc_new'), (do_pre, 'do_pre'),
                  (dj, 'dj'), (di, 'di'), (df, 'df'), (dc, 'dc')):
        _hbcw(t_, n, (H, B, F, W))
    check(lib.ag_convlstm_cell_bwd(_p(j), _p(i_), _p(f_), _p(c), _p(c_new), _p(_peep(wco, 'w_co', H, F, W)), float(forget_bias),
                                   _p(dc_new), _p(do_pre), _p(dj), _p(di), _p(df), _p(dc), _p(_peep(dwco, 'dw_co', H, F, W)),
                                   H, B, F, W, _stream()), 'ag_convlstm_cell_bwd')


def convlstm_out_fwd(o, c, h):
    for t_, n in ((o, 'o'), (c, 'c'), (h, 'h')):
        _chk(t_, n)
        assert t_.is_contiguous() and t_.numel() == o.numel()
    check(lib.ag_convlstm_out_fwd(_p(o), _p(c), _p(h), o.numel(), _stream()), 'ag_convlstm_out_fwd')


def convlstm_out_bwd(o, c, dh, do, dc):
    for t_, n in ((o, 'o'), (c, 'c'), (dh, 'dh'), (do, 'do'), (dc, 'dc')):
        _chk(t_, n)
        assert t_.is_contiguous() and t_.numel() == o.numel()
    check(lib.ag_convlstm_out_bwd(_p(o), _p(c), _p(dh), _p(do), _p(dc), o.numel(), _stream()), 'ag_convlstm_out_bwd')


def layer_norm_hbfw_fwd(x, gamma, beta, eps, y, mean, rstd):
    H, B, F, W = x.shape
    _hbcw(x, 'x'); _hbcw(y, 'y', x.shape)
    for t_, n, k in ((gamma, 'gamma', F), (beta, 'beta', F), (mean, 'mean', B), (rstd, 'rstd', B)):
        _chk(t_, n)
        assert t_.is_contiguous() and t_.numel() == k, n
    check(lib.ag_layer_norm_hbfw_fwd(_p(x), _p(gamma), _p(beta), float(eps), _p(y), _p(mean), _p(rstd), H, B, F, W, _stream()),
          'ag_layer_norm_hbfw_fwd')


def layer_norm_hbfw_bwd(dy, x, gamma, mean, rstd, dx, dgp, dbp):
    H, B, F, W = x.shape
    _hbcw(x, 'x'); _hbcw(dy, 'dy', x.shape); _hbcw(dx, 'dx', x.shape)
    for t_, n, k in ((gamma, 'gamma', F), (mean, 'mean', B), (rstd, 'rstd', B), (dgp, 'dgamma_part', B * F), (dbp, 'dbeta_part', B * F)):
        _chk(t_, n)
        assert t_.is_contiguous() and t_.numel() == k, n
    check(lib.ag_layer_norm_hbfw_bwd(_p(dy), _p(x), _p(gamma), _p(mean), _p(rstd), _p(dx), _p(dgp), _p(dbp), H, B, F, W, _stream()),
          'ag_layer_norm_hbfw_bwd')


for _n in ('convlstm_peephole_fwd', 'convlstm_peephole_bwd', 'convlstm_cell_fwd', 'convlstm_cell_bwd', 'convlstm_out_fwd',
           'convlstm_out_bwd', 'layer_norm_hbfw_fwd', 'layer_norm_hbfw_bwd'):
    _instrument(_n, None)
