"""Torch-CPU model of every entry point of ``audiogan_amd.kernels``  --  TEST INFRASTRUCTURE.

Written from the contracts in include/audiogan_hip.h.  The CPU test-suite monkeypatches
``audiogan_amd.kernels`` with these functions so that the HOST logic of the package
(autograd blocks, slab bookkeeping, weight layouts, module API, DDP, optimiser wiring) can be
checked against the oracle without a GPU.  The product never imports this file; on the GPU
box the real HIP kernels run and are themselves compared with the oracle (tests -m gpu).
"""
import torch
import torch.nn.functional as F

ACT_NONE, ACT_LEAKY, ACT_TANH, ACT_LEAKY_GATE = 0, 1, 2, 3
OPT_RMSPROP, OPT_ADAM = 0, 1
LEAKY_SLOPE = 0.01


def _r(a, b):
    return (a + b - 1) // b * b


def wpa_numel(d0, d1, K):
    return _r(d1, 2) * K * _r(d0, 32)


def wpb_numel(d0, d1, K, stride):
    return _r(d0, 2) * ((K + stride - 1) // stride) * _r(d1 * stride, 32)


def _scatter_shift(s, pad, r):
    qmax = (pad + s - 1) // s
    qr = (pad - r + s - 1) // s if pad > r else 0
    return qmax - qr


def _scatter_aligned(K, s, pad):
    """common.h ag_scatter_aligned: the aligned scatter layout applies (pad % s != 0 and no extra tap slot)"""
    if s <= 1 or pad % s == 0:
        return False
    mt = (K + s - 1) // s
    for r in range(s):
        taps_r = (K - r + s - 1) // s if r < K else 0
        if taps_r + _scatter_shift(s, pad, r) > mt:
            return False
    return True


def _fill_layouts(w, wpa, wpb, stride, pad=0):
    d0, d1, K = w.shape
    if wpa is not None:
        wpa.view(_r(d1, 2), K, _r(d0, 32))[:d1, :, :d0] = w.permute(1, 2, 0)
    if wpb is not None:
        mt, mp = (K + stride - 1) // stride, _r(d1 * stride, 32)
        v = wpb.view(_r(d0, 2), mt, mp)
        v.zero_()
        al = _scatter_aligned(K, stride, pad)
        for k in range(K):
            m = k // stride + (_scatter_shift(stride, pad, k % stride) if al else 0)
            v[:d0, m, torch.arange(d1) * stride + k % stride] = w[:, :, k]


def weight_norm_fwd(entries):
    for e in entries:
        v, g = e['v'], e['g']
        rows = v.size(0)
        v2 = v.reshape(rows, -1)
        w = (v2 * (g.view(rows, 1) / v2.norm(dim=1, keepdim=True))).view(v.shape)
        if e.get('w') is not None:
            e['w'].copy_(w)
        if v.dim() == 3:
            _fill_layouts(w, e.get('wpa'), e.get('wpb'), int(e.get('stride', 1)), int(e.get('pad', 0)))


def weight_norm_bwd(entries):
    for e in entries:
        v, g, dw = e['v'], e['g'], e['dw']
        rows = v.size(0)
        v2, dw2 = v.reshape(rows, -1), dw.reshape(rows, -1)
        inv = 1.0 / v2.norm(dim=1, keepdim=True)
        dot = (v2 * dw2).sum(1, keepdim=True)
        dg_, dv_ = (dot * inv).view(-1), (g.view(rows, 1) * inv * (dw2 - v2 * dot * inv * inv)).view(v.shape)
        if e.get('accumulate'):
            e['dg'].add_(dg_)
            e['dv'].add_(dv_)
        else:
            e['dg'].copy_(dg_)
            e['dv'].copy_(dv_)


def prep_conv_weight(w, wpa, wpb, stride, pad=0):
    _fill_layouts(w, wpa, wpb, stride, pad)


def _act(v, act, slope):
    if act == ACT_LEAKY:
        return torch.where(v > 0, v, v * slope)
    if act == ACT_TANH:
        return torch.tanh(v)
    return v


def _fit(t, n):
    if t.size(2) >= n:
        return t[:, :, :n]
    return F.pad(t, (0, n - t.size(2)))


def conv_engine(x, wp, y, K, stride, pad, mode, bias=None, res=None, lens=None, act=ACT_NONE,
                slope=LEAKY_SLOPE, accumulate=False, wp_pad=0):
    B, C, Lin = x.shape
    _, O, Lout = y.shape
    if mode == 0:
        assert wp.numel() == wpa_numel(O, C, K)
        w = wp.view(_r(C, 2), K, _r(O, 32))[:C, :, :O].permute(2, 0, 1)          # [O,C,K]
        out = _fit(F.conv1d(x, w, None, stride, pad), Lout)
    else:
        assert wp.numel() == wpb_numel(C, O, K, stride)
        mt, mp = (K + stride - 1) // stride, _r(O * stride, 32)
        v = wp.view(_r(C, 2), mt, mp)
        al = wp_pad == pad and _scatter_aligned(K, stride, pad)
        w = torch.stack([v[:C, k // stride + (_scatter_shift(stride, pad, k % stride) if al else 0),
                           torch.arange(O) * stride + k % stride] for k in range(K)], 2)
        full = F.conv_transpose1d(x, w, None, stride, 0)                    # w: [C,O,K]; u + pad
        out = _fit(full[:, :, pad:], Lout)
    if bias is not None:
        out = out + bias.view(1, O, 1)
    if res is not None:
        out = out * torch.where(res > 0, torch.ones_like(res), torch.full_like(res, slope)) if act == ACT_LEAKY_GATE else out + res
    out = _act(out, act if act != ACT_LEAKY_GATE else ACT_NONE, slope)
    if lens is not None:
        out = out * (torch.arange(Lout).view(1, 1, Lout) < lens.view(B, 1, 1)).float()
    if accumulate:
        out = out + y
    y.copy_(out)


def conv_wgrad(sh, lg, dw, K, stride, pad):
    B, A, Lsh = sh.shape
    _, C, Llg = lg.shape
    need = stride * (Lsh - 1) + K
    lp = F.pad(lg, (pad, max(0, need - pad - Llg)))
    acc = torch.zeros(A, C, K)
    for k in range(K):
        acc[:, :, k] = torch.einsum('bat,bct->ac', sh, lp[:, :, k:k + stride * (Lsh - 1) + 1:stride])
    dw.add_(acc.view(dw.shape))


def channel_sum(dy, db, accumulate=True):
    if accumulate:
        db.add_(dy.sum((0, 2)))
    else:
        db.copy_(dy.sum((0, 2)))


def lstm_front_bwd_ok(B, H, fs, dxa_t, x_t):
    return H % 16 == 0 and fs % 16 == 0


def lstm_front_bwd_step(dxa_t, x_t, gx_out, w_proj, dh_acc, gates, c_prev, c_new, dc_next, dgates, dc_prev):
    gx = dxa_t * (1 - x_t * x_t)
    gx_out.copy_(gx)
    dh = dh_acc + gx @ w_proj
    lstm_cell_bwd(gates, c_prev, c_new, dh, None, dc_next, dgates, dc_prev)


def leaky_bwd(dy, y, dpre, lens=None, slope=LEAKY_SLOPE, add_into=None, bias_grad=None):
    B, C, L = dy.shape
    g = torch.where(y > 0, dy, dy * slope)
    if lens is not None:
        g = g * (torch.arange(L).view(1, 1, L) < lens.view(B, 1, 1)).float()
    dpre.copy_(g)
    if add_into is not None:
        add_into.add_(g)
    if bias_grad is not None:
        bias_grad.add_(g.sum((0, 2)))


def gemm(A, B, Cm, ta=False, tb=False, alpha=1.0, beta=0.0, bias=None, res=None, act=ACT_NONE,
         slope=LEAKY_SLOPE, defer=False):
    a = A.t() if ta else A
    b = B.t() if tb else B
    out = alpha * (a @ b)
    if beta != 0.0:
        out = out + beta * Cm
    if bias is not None:
        out = out + bias.view(1, -1)
    if act == ACT_LEAKY_GATE:        # res = the saved LeakyReLU output whose derivative scales the result (not added)
        out = out * torch.where(res > 0, torch.ones_like(res), torch.full_like(res, slope))
    elif res is not None:
        out = out + res
    Cm.copy_(_act(out, act if act != ACT_LEAKY_GATE else ACT_NONE, slope))


def col_sum(X, out, accumulate=True, defer=True):
    if accumulate:
        out.add_(X.sum(0))
    else:
        out.copy_(X.sum(0))


def lstm_cell_fwd(gates, c_prev, c_out, h_out=None, y_out=None, h_prev=None, valid=None, t=0):
    B, H4 = gates.shape
    H = H4 // 4
    i, f, g, o = [gates[:, k * H:(k + 1) * H] for k in range(4)]
    ia, fa, ga, oa = torch.sigmoid(i), torch.sigmoid(f), torch.tanh(g), torch.sigmoid(o)
    cn = fa * c_prev + ia * ga
    hn = oa * torch.tanh(cn)
    if valid is not None:
        ok = (t < valid).view(B, 1)
    else:
        ok = torch.ones(B, 1, dtype=torch.bool)
    hp = h_prev if h_prev is not None else torch.zeros(B, H)
    new_gates = torch.where(ok, torch.cat([ia, fa, ga, oa], 1), gates)
    cn2, hn2, yn = torch.where(ok, cn, c_prev), torch.where(ok, hn, hp), torch.where(ok, hn, torch.zeros(B, H))
    gates.copy_(new_gates)
    c_out.copy_(cn2)
    if h_out is not None:
        h_out.copy_(hn2)
    if y_out is not None:
        y_out.copy_(yn)


def lstm_cell_bwd(gates_act, c_prev, c_new, dh, dy, dc_next, dgates, dc_prev, dh_pass=None,
                  valid=None, t=0):
    B, H4 = gates_act.shape
    H = H4 // 4
    ig, fg, gg, og = [gates_act[:, k * H:(k + 1) * H] for k in range(4)]
    z = torch.zeros(B, H)
    dhf = dh if dh is not None else z
    dcn = dc_next if dc_next is not None else z
    dhv = dhf + (dy if dy is not None else z)
    tc = torch.tanh(c_new)
    dc = dcn + dhv * og * (1 - tc * tc)
    dg = torch.cat([dc * gg * ig * (1 - ig), dc * c_prev * fg * (1 - fg), dc * ig * (1 - gg * gg),
                    dhv * tc * og * (1 - og)], 1)
    ok = (t < valid).view(B, 1) if valid is not None else torch.ones(B, 1, dtype=torch.bool)
    dgates.copy_(torch.where(ok, dg, torch.zeros_like(dg)))
    dc_prev.copy_(torch.where(ok, dc * fg, dcn))
    if dh_pass is not None:
        dh_pass.copy_(torch.where(ok, z, dhf))


def _bce_elem(x, target):
    m = (-x).clamp(min=0)
    return x - x * target + m + ((-m).exp() + (-x - m).exp()).log()


def bce_logits_fwd(x, target, nframes, per_sample, loss, scale):
    B, T = x.shape
    n = nframes if nframes is not None else torch.full((B,), T, dtype=torch.long)
    mask = (torch.arange(T).view(1, T) < n.view(B, 1)).float()
    per = (_bce_elem(x, target) * mask).sum(1)
    if per_sample is not None:
        per_sample.copy_(per)
    if loss is not None:
        loss.add_(scale * (per / n.float()).sum())


def bce_logits_bwd(x, target, nframes, gscale, scale, dx):
    B, T = x.shape
    n = nframes if nframes is not None else torch.full((B,), T, dtype=torch.long)
    mask = (torch.arange(T).view(1, T) < n.view(B, 1)).float()
    gs = gscale.view(()) if gscale is not None else 1.0
    dx.copy_(gs * scale / n.float().view(B, 1) * (torch.sigmoid(x) - target) * mask)


def bce_logits_fwd_strided(x, target, nframes, per_sample, loss, scale, target_rows=None):
    B, T = x.shape
    n = nframes if nframes is not None else torch.full((B,), T, dtype=torch.long)
    mask = (torch.arange(T).view(1, T) < n.view(B, 1)).float()
    tg = target_rows.view(B, 1) if target_rows is not None else target
    per = (_bce_elem(x, tg) * mask).sum(1)
    if per_sample is not None:
        per_sample.copy_(per)
    loss.copy_((scale * (per / n.float()).sum()).view(1))


def bce_logits_bwd_strided(x, target, nframes, gscale, scale, dx, target_rows=None):
    B, T = x.shape
    n = nframes if nframes is not None else torch.full((B,), T, dtype=torch.long)
    mask = (torch.arange(T).view(1, T) < n.view(B, 1)).float()
    gs = gscale.view(()) if gscale is not None else 1.0
    tg = target_rows.view(B, 1) if target_rows is not None else target
    dx.copy_(gs * scale / n.float().view(B, 1) * (torch.sigmoid(x) - tg) * mask)


def act_fwd(x, y, act, slope=LEAKY_SLOPE):
    y.copy_(_act(x, act, slope))


def act_bwd(dy, y, dx, act, slope=LEAKY_SLOPE):
    if act == ACT_LEAKY:
        g = torch.where(y > 0, dy, dy * slope)
    elif act == ACT_TANH:
        g = dy * (1 - y * y)
    else:
        g = dy
    dx.copy_(g)


def axpby(x, y, a, b):
    y.copy_(a * x + (b * y if b != 0.0 else 0.0))


def grad_norms(params, grads, s1, s2, norms, norm_sum, flags, grad_scale=1.0, step_dev=None, finish=True):
    """(the model always finishes the norms; ``opt_step(part=...)`` then finds them in place)"""
    if step_dev is not None:
        step_dev.add_(1)
    f = 0
    tot = 0.0
    for i, g in enumerate(grads):
        gs = g * grad_scale
        n = gs.norm()
        norms[i] = n
        tot = tot + n
        if bool((gs != gs).any()):
            f |= 1
        if bool((gs.abs() > 1e5).any()):
            f |= 2
    if norm_sum is not None:
        norm_sum.fill_(float(tot))
    if flags is not None:
        flags.fill_(f)
    return None if finish else norms


def opt_step(params, grads, s1, s2, norms, kind, lr, clip, grad_scale, a1, b2, eps, step, step_dev=None, part=None,
             norm_sum=None, flags=None):
    if step_dev is not None:
        step = int(step_dev.item())
    for i, p in enumerate(params):
        g = grads[i].reshape(-1) * grad_scale
        n = norms[i]
        pf = p.data.view(-1)
        if clip > 0 and float(n) > clip:
            g = g / (n / clip)
        if kind == OPT_RMSPROP:
            s1[i].view(-1).mul_(a1).add_((1 - a1) * g * g)
            pf.sub_(lr * g / (s1[i].view(-1).sqrt() + eps))
        else:
            s1[i].view(-1).mul_(a1).add_((1 - a1) * g)
            s2[i].view(-1).mul_(b2).add_((1 - b2) * g * g)
            bc1 = 1 - a1 ** step
            bc2s = (1 - b2 ** step) ** 0.5
            pf.sub_((lr / bc1) * s1[i].view(-1) / (s2[i].view(-1).sqrt() / bc2s + eps))


ALL = [n for n, v in list(globals().items()) if callable(v) and not n.startswith('_')
       and n not in ('F',)]


def install(monkeypatch):
    """Replace every function of audiogan_amd.kernels by its CPU model (pytest monkeypatch)."""
    import audiogan_amd.kernels as K
    for n in ALL:
        if n == 'install':
            continue
        assert hasattr(K, n), 'kernel model has %s but audiogan_amd.kernels does not' % n
        monkeypatch.setattr(K, n, globals()[n])


def lstm_persist_ok(B, H, ndir, dev):
    """(the model's lstm_seq_fwd walks the steps: the caller zero-fills c_0)"""
    return False


def gfront_persist_ok(B, S, fs, dev):
    """the persistent Generator-front launch has no CPU model: the host logic takes the frame-by-frame path"""
    return False


def bct_to_tbc(x, out=None, out_dtype=None):
    r = x.permute(2, 0, 1).contiguous()
    if out is not None:
        out.copy_(r)
        return out
    return r


def tbc_to_bct(x, out=None, out_dtype=None):
    r = x.permute(1, 2, 0).contiguous()
    if out is not None:
        out.copy_(r)
        return out
    return r


class deferred_reduces(object):
    """(no second stages in the model)"""

    def __init__(self, outer=False):
        self.role = 'owner'

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def bf16_storage():
    """(the CPU model keeps fp32 storage)"""
    return False


def reduces_outer():
    return False


def reduces_recording():
    return False


def flush_reduces():
    pass


def rowdot_ok(x, w):
    return x.dim() == 2 and x.stride(1) == 1 and w.numel() == x.size(1) and w.is_contiguous()


def rowdot_fwd(x, w, bias, y):
    r = x @ w.reshape(-1)
    if bias is not None:
        r = r + bias.reshape(-1)[0]
    y.copy_(r.view(y.shape))


def rowdot_bwd(dy, x, w, dx=None, dw=None, db=None, gate=False, slope=LEAKY_SLOPE, accumulate=True):
    g = dy.reshape(-1, 1)
    if dx is not None:
        r = g * w.reshape(1, -1)
        if gate:
            r = r * torch.where(x > 0, torch.ones_like(x), torch.full_like(x, slope))
        dx.copy_(r)
    if dw is not None:
        gw, gb = (g * x).sum(0).view(dw.shape), g.sum().view(db.shape)
        if accumulate:
            dw.add_(gw); db.add_(gb)
        else:
            dw.copy_(gw); db.copy_(gb)


def build_zc(z, c, out=None):
    B, T, ns = z.shape
    r = torch.cat([z, c.unsqueeze(1).expand(B, T, c.size(1))], 2).transpose(0, 1).contiguous()
    if out is not None:
        out.copy_(r)
        return out
    return r


def critic_batch(xa, na=None, xb=None, nb=None, len_a=None, len_b=None, prods=(), ca=None, cb=None):
    rows = [xa + na if na is not None else xa.clone()]
    L = xa.size(1)
    lens = [len_a if len_a is not None else torch.full((xa.size(0),), L, dtype=torch.long)]
    if xb is not None:
        rows.append(xb + nb if nb is not None else xb.clone())
        lens.append(len_b if len_b is not None else torch.full((xb.size(0),), L, dtype=torch.long))
    x = torch.cat(rows, 0)
    ln = torch.cat([l.long() for l in lens], 0)
    tab = torch.stack([(ln + d - 1) // d for d in prods], 0) if len(prods) else None
    c = None
    if ca is not None:
        c = torch.cat([ca, cb], 0) if xb is not None else ca.clone()
    return x, tab, c


def gfront_bwd_persist_ok(B, S, fs, dev):
    """(likewise for its backward)"""
    return False


# ---- skinny products / fused recurrent steps (same contracts as lstm_step.hip) -----------------
def skinny_ok(A, B, tb):
    M, Kd = A.shape
    ok = M <= 256 and Kd % 8 == 0 and A.stride(1) == 1 and A.stride(0) % 4 == 0
    if tb:
        ok = ok and B.stride(1) == 1 and B.stride(0) % 4 == 0
    return ok


def skinny_gemm(A, B, Cm, tb=False, beta=0.0, bias=None, act=ACT_NONE, slope=LEAKY_SLOPE, atomic=False):
    assert skinny_ok(A, B, tb)
    out = A @ (B.t() if tb else B)
    if atomic:
        assert act == ACT_NONE
        Cm.add_(out + (bias.view(1, -1) if bias is not None else 0.0))
        return
    if beta != 0.0:
        out = out + beta * Cm
    if bias is not None:
        out = out + bias.view(1, -1)
    Cm.copy_(_act(out, act, slope))


def lstm_step_ok(B, H, x=None, wx=None):
    ok = B <= 256 and H % 8 == 0
    if x is not None:
        ok = ok and x.size(1) % 8 == 0 and x.stride(0) % 4 == 0 and wx.stride(0) % 4 == 0
    return ok


def lstm_step_fwd(gates_pre, x, wx, h_prev, whh, c_prev, c_out, h_out, first_step):
    if not first_step:
        if x is not None:
            gates_pre.add_(x @ wx.t())
        gates_pre.add_(h_prev @ whh.t())
    lstm_cell_fwd(gates_pre, c_prev, c_out, h_out=h_out)


def lstm_seq_fwd(pre, whh, c_all, hbuf, y, valid, static=None):
    ndir = len(pre)
    T, B, H4 = pre[0].shape
    H = H4 // 4
    for d in range(ndir):
        hbuf[d][0].zero_()
        for k in range(T):
            t = k if d == 0 else T - 1 - k
            hp, hn = hbuf[d][k & 1], hbuf[d][(k + 1) & 1]
            if k > 0:
                pre[d][t].add_(hp @ whh[d].t())
            if static is not None:
                pre[d][t].add_(static[d])
            lstm_cell_fwd(pre[d][t], c_all[d][k], c_all[d][k + 1], h_out=hn,
                          y_out=y[t, :, d * H:(d + 1) * H], h_prev=hp, valid=valid, t=t)


def lstm_seq_bwd(gates, whh, c_all, dy, dgates, dhbuf, dcbuf, valid, dgsum=None):
    ndir = len(gates)
    T, B, H4 = gates[0].shape
    H = H4 // 4
    for d in range(ndir):
        for k in reversed(range(T)):
            t = k if d == 0 else T - 1 - k
            dh = None if k == T - 1 else dhbuf[d][k & 1]
            dcn = None if k == T - 1 else dcbuf[d][(k + 1) & 1]
            dpass = dhbuf[d][(k + 1) & 1]
            lstm_cell_bwd(gates[d][t], c_all[d][k], c_all[d][k + 1], dh, dy[t, :, d * H:(d + 1) * H], dcn,
                          dgates[d][t], dcbuf[d][k & 1], dh_pass=dpass, valid=valid, t=t)
            if k > 0:
                dpass.add_(dgates[d][t] @ whh[d])
    return False        # (dgsum is left to the caller, like the per-step kernels do)


def gru_cell_fwd(gi, gh, h_prev, h_out):
    B, H3 = gi.shape
    H = H3 // 3
    r = torch.sigmoid(gi[:, :H] + gh[:, :H])
    z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
    n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
    hn = (1 - z) * n + z * h_prev
    gi.copy_(torch.cat([r, z, n], 1))
    h_out.copy_(hn)


def gru_cell_bwd(gates_act, gh, h_prev, dh, dgi, dgh, dh_prev):
    B, H3 = gates_act.shape
    H = H3 // 3
    r, z, n = gates_act[:, :H], gates_act[:, H:2 * H], gates_act[:, 2 * H:]
    hn = gh[:, 2 * H:]
    dn_pre = dh * (1 - z) * (1 - n * n)
    dz_pre = dh * (h_prev - n) * z * (1 - z)
    dr_pre = dn_pre * hn * r * (1 - r)
    dgi.copy_(torch.cat([dr_pre, dz_pre, dn_pre], 1))
    dgh.copy_(torch.cat([dr_pre, dz_pre, dn_pre * r], 1))
    dh_prev.copy_(dh * z)


def act_bwd2d(dy, y, dx, act, slope=LEAKY_SLOPE):
    act_bwd(dy, y, dx, act, slope)


def conv_o1_ok(spec_kind, cout, K_, stride, pad):
    return spec_kind == 'conv' and cout == 1 and stride == 1 and K_ <= 9 and 2 * pad == K_ - 1


def conv_o1_fwd(x, w, bias, y, K_, pad, act=ACT_NONE, slope=LEAKY_SLOPE):
    out = F.conv1d(x, w.view(1, x.size(1), K_), bias, 1, pad)
    y.copy_(_act(out, act, slope))


def conv_o1_bwd_data(dy, w, dx, K_, pad, accumulate=False):
    out = F.conv_transpose1d(dy, w.view(1, dx.size(1), K_), None, 1, pad)
    dx.copy_(out + dx if accumulate else out)


def conv_o1_wgrad(dy, x, dw, K_, pad):
    xp = F.pad(x, (pad, pad))
    L = x.size(2)
    for k in range(K_):
        dw.view(x.size(1), K_)[:, k] += torch.einsum('bt,bct->c', dy[:, 0], xp[:, :, k:k + L])


ALL = [n for n, v in list(globals().items()) if callable(v) and not n.startswith('_')
       and n not in ('F', 'install')]


def time_moments_fwd(h, lens, m, s, f):
    B, C, L = h.shape
    mask = (torch.arange(L).view(1, L) < lens.view(B, 1)).float()
    lf = lens.view(B, 1).float()
    mm = h.sum(2) / lf
    cen = h - mm.unsqueeze(2) * mask.unsqueeze(1)
    m.copy_(mm)
    s.copy_((cen ** 2).sum(2) ** 0.5 / lf)
    f.copy_((cen ** 4).sum(2) ** 0.25 / lf)


def time_moments_bwd(h, lens, gm, gs, gf, dh):
    with torch.enable_grad():       # called from inside an autograd backward
        hh = h.detach().clone().requires_grad_(True)
        B, C, L = hh.shape
        mask = (torch.arange(L).view(1, L) < lens.view(B, 1)).float()
        lf = lens.view(B, 1).float()
        mm = hh.sum(2) / lf
        cen = hh - mm.unsqueeze(2) * mask.unsqueeze(1)
        z = torch.zeros(B, C)
        tot = (mm * (gm if gm is not None else z)).sum() \
            + (((cen ** 2).sum(2) ** 0.5 / lf) * (gs if gs is not None else z)).sum() \
            + (((cen ** 4).sum(2) ** 0.25 / lf) * (gf if gf is not None else z)).sum()
        g, = torch.autograd.grad(tot, hh)
    dh.copy_(g)


ALL = [n for n, v in list(globals().items()) if callable(v) and not n.startswith('_')
       and n not in ('F', 'install')]


# ---- Conv2DLSTMCell pieces (csrc/convlstm.hip): maps [H,B,C,W], peephole weights [H,F,W] --------------------------------
def _pw(w):
    return w.unsqueeze(1) if w is not None else 0.0          # [H,1,F,W] broadcasts over the batch axis


def convlstm_peephole_fwd(y, c, wci, wcf, j, ip, fp, o):
    F_ = c.size(2)
    yj, yi, yf, yo = (y[:, :, k * F_:(k + 1) * F_] for k in range(4))
    j.copy_(yj); ip.copy_(yi + _pw(wci) * c); fp.copy_(yf + _pw(wcf) * c); o.copy_(yo)


def convlstm_peephole_bwd(dj, di, df, do, c, wci, wcf, dy, dc, dwci, dwcf):
    dy.copy_(torch.cat([dj, di, df, do], 2))
    dc.add_(di * _pw(wci) + df * _pw(wcf))
    if dwci is not None:
        dwci.copy_((di * c).sum(1))
    if dwcf is not None:
        dwcf.copy_((df * c).sum(1))


def convlstm_cell_fwd(j, i_, f_, c, o_raw, wco, forget_bias, c_new, o_pre):
    v = c * torch.sigmoid(f_ + forget_bias) + torch.sigmoid(i_) * torch.tanh(j)
    c_new.copy_(v)
    o_pre.copy_(o_raw + _pw(wco) * v)


def convlstm_cell_bwd(j, i_, f_, c, c_new, wco, forget_bias, dc_new, do_pre, dj, di, df, dc, dwco):
    if dwco is not None:
        dwco.copy_((do_pre * c_new).sum(1))
    g = dc_new + do_pre * _pw(wco)
    sf, si, tj = torch.sigmoid(f_ + forget_bias), torch.sigmoid(i_), torch.tanh(j)
    dc_new.copy_(g)
    dc.copy_(g * sf); df.copy_(g * c * sf * (1 - sf)); di.copy_(g * tj * si * (1 - si)); dj.copy_(g * si * (1 - tj * tj))


def convlstm_out_fwd(o, c, h):
    h.copy_(torch.sigmoid(o) * torch.tanh(c))


def convlstm_out_bwd(o, c, dh, do, dc):
    so, tc = torch.sigmoid(o), torch.tanh(c)
    do.copy_(dh * tc * so * (1 - so)); dc.copy_(dh * so * (1 - tc * tc))


def layer_norm_hbfw_fwd(x, gamma, beta, eps, y, mean, rstd):
    m = x.mean(dim=(0, 2, 3))
    v = ((x - m.view(1, -1, 1, 1)) ** 2).mean(dim=(0, 2, 3))
    r = torch.rsqrt(v + eps)
    mean.copy_(m); rstd.copy_(r)
    y.copy_((x - m.view(1, -1, 1, 1)) * r.view(1, -1, 1, 1) * gamma.view(1, 1, -1, 1) + beta.view(1, 1, -1, 1))


def layer_norm_hbfw_bwd(dy, x, gamma, mean, rstd, dx, dgp, dbp):
    xh = (x - mean.view(1, -1, 1, 1)) * rstd.view(1, -1, 1, 1)
    g = dy * gamma.view(1, 1, -1, 1)
    m1, m2 = g.mean(dim=(0, 2, 3)).view(1, -1, 1, 1), (g * xh).mean(dim=(0, 2, 3)).view(1, -1, 1, 1)
    dx.copy_(rstd.view(1, -1, 1, 1) * (g - m1 - xh * m2))
    dgp.copy_((dy * xh).sum(dim=(0, 3)).view(dgp.shape)); dbp.copy_(dy.sum(dim=(0, 3)).view(dbp.shape))


ALL = [n for n, v in list(globals().items()) if callable(v) and not n.startswith('_')
       and getattr(v, '__module__', None) == __name__ and n not in ('install',)]
