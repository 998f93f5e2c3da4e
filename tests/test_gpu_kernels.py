"""-m gpu: every HIP kernel called through the C ABI (audiogan_amd.kernels -> ctypes ->
libaudiogan_hip.so) and compared with torch-CPU fp32 on the same seeded inputs.
Tolerance: 1e-3 relative (north_star), written per test; most kernels are far tighter because
v_mfma_f32_32x32x2_f32 is an exact fp32 fma chain."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests import kernel_model as KM

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def K():
    assert torch.cuda.is_available(), 'gpu tests need a GPU'
    import audiogan_amd.kernels as K_
    assert K_.conv_engine.__module__ == 'audiogan_amd.kernels', 'real kernels must be in place'
    return K_


def dev(t):
    return t.cuda() if t is not None else None


def close(got, ref, rtol=1e-3, atol=None, msg=''):
    ref = ref.detach().cpu().float().numpy() if torch.is_tensor(ref) else np.asarray(ref)
    got = got.detach().cpu().float().numpy() if torch.is_tensor(got) else np.asarray(got)
    if atol is None:
        atol = 1e-5 * max(1.0, float(np.abs(ref).max()))
    np.testing.assert_allclose(got, ref, rtol=rtol, atol=atol, err_msg=msg)


# C2 layer table: kind, cin, cout, K, stride, pad, Lin
G_LAYERS = [('conv', 1, 128, 17, 8, 8, 8192), ('convT', 128, 16, 16, 8, 4, 1024),
            ('conv', 17, 64, 9, 4, 4, 8192), ('convT', 64, 32, 8, 4, 2, 2048),
            ('conv', 49, 64, 9, 4, 4, 8192), ('conv', 81, 32, 9, 4, 4, 8192),
            ('convT', 32, 32, 8, 4, 2, 2048), ('conv', 113, 1, 3, 1, 1, 8192)]
D_LAYERS = [('conv', 1, 16, 7, 2, 3, 8192), ('conv', 16, 32, 7, 2, 3, 4096), ('conv', 32, 64, 7, 2, 3, 2048),
            ('conv', 64, 128, 7, 2, 3, 1024), ('conv', 128, 256, 7, 2, 3, 512), ('conv', 256, 512, 7, 2, 3, 256)]
EDGE_LAYERS = [('conv', 3, 5, 5, 2, 2, 16), ('convT', 5, 3, 4, 2, 1, 8), ('conv', 2, 3, 1, 1, 0, 7),
               ('conv', 7, 33, 5, 5, 2, 333), ('convT', 6, 7, 5, 5, 0, 41), ('conv', 5, 70, 3, 3, 1, 100),
               ('convT', 3, 20, 5, 2, 2, 129), ('conv', 1, 1, 3, 1, 1, 1), ('conv', 130, 140, 3, 1, 1, 70)]


def _out_len(kind, lin, k, s, p):
    return (lin + 2 * p - k) // s + 1 if kind == 'conv' else (lin - 1) * s - 2 * p + k


def _mk(kind, cin, cout, k, s, p, lin, B, seed):
    gen = torch.Generator().manual_seed(seed)
    w = torch.randn((cout, cin, k) if kind == 'conv' else (cin, cout, k), generator=gen) / (cin * k) ** 0.5
    x = torch.randn(B, cin, lin, generator=gen)
    return w, x


def _check_bf16_image(wp, n32, cp2, taps):
    """behind the fp32 layout [cp2][taps][mpad] sits its bf16 image [ceil(cp2/16)][taps][2][mpad][8] (common.h ag_wq_*)"""
    wp = wp.cpu()
    mpad = n32 // (cp2 * taps)
    G = (cp2 + 15) // 16
    assert wp.numel() == n32 + G * 16 * taps * mpad // 2
    full = torch.zeros(G * 16, taps, mpad)
    full[:cp2] = wp[:n32].view(cp2, taps, mpad)
    want = full.to(torch.bfloat16).view(torch.int16).view(G, 2, 8, taps, mpad).permute(0, 3, 1, 4, 2).contiguous()
    got = wp[n32:].view(torch.int16).view(G, taps, 2, mpad, 8)
    assert torch.equal(got, want)


@pytest.mark.parametrize('layer', G_LAYERS + D_LAYERS + EDGE_LAYERS)
def test_conv_forward_backward_data(K, layer):
    kind, cin, cout, k, s, p, lin = layer
    B = 2
    w, x = _mk(kind, cin, cout, k, s, p, lin, B, 1)
    lout = _out_len(kind, lin, k, s, p)
    gen = torch.Generator().manual_seed(2)
    bias = torch.randn(cout, generator=gen)
    res = torch.randn(B, cout, lout, generator=gen)
    lens = torch.tensor([lout, max(1, lout // 3)])
    d0, d1, _ = w.shape
    wpa, wpb = torch.zeros(K.wpa_numel(d0, d1, k)).cuda(), torch.zeros(K.wpb_numel(d0, d1, k, s)).cuda()
    K.prep_conv_weight(dev(w), wpa, wpb, s)
    na, nb = KM.wpa_numel(d0, d1, k), KM.wpb_numel(d0, d1, k, s)      # the fp32 layouts; their bf16 images follow
    ra, rb = torch.zeros(na), torch.zeros(nb)
    KM.prep_conv_weight(w, ra, rb, s)
    close(wpa[:na], ra, rtol=0, atol=0)
    close(wpb[:nb], rb, rtol=0, atol=0)
    _check_bf16_image(wpa, na, (d1 + 1) // 2 * 2, k)
    _check_bf16_image(wpb, nb, (d0 + 1) // 2 * 2, (k + s - 1) // s)
    fwd_mode = 0 if kind == 'conv' else 1
    fwd_wp, bwd_wp = (wpa, wpb) if kind == 'conv' else (wpb, wpa)
    # forward, full epilogue
    xr = x.clone().requires_grad_(True)
    lin_out = F.conv1d(xr, w, bias, s, p) if kind == 'conv' else F.conv_transpose1d(xr, w, bias, s, p)
    ref = F.leaky_relu(lin_out + res) * (torch.arange(lout).view(1, 1, -1) < lens.view(B, 1, 1)).float()
    y = torch.full((B, cout, lout), float('nan')).cuda()
    K.conv_engine(dev(x), fwd_wp, y, k, s, p, fwd_mode, bias=dev(bias), res=dev(res), lens=dev(lens),
                  act=K.ACT_LEAKY)
    close(y, ref, msg='forward')
    # plain forward + accumulate
    y2 = dev(res.clone())
    K.conv_engine(dev(x), fwd_wp, y2, k, s, p, fwd_mode, accumulate=True)
    plain = F.conv1d(x, w, None, s, p) if kind == 'conv' else F.conv_transpose1d(x, w, None, s, p)
    close(y2, plain + res, msg='accumulate')
    # backward-data (adjoint)
    gy = torch.randn(B, cout, lout, generator=gen)
    lin_out.backward(gy)
    dx = torch.full((B, cin, lin), float('nan')).cuda()
    K.conv_engine(dev(gy), bwd_wp, dx, k, s, p, 1 - fwd_mode)
    close(dx, xr.grad, msg='backward-data')
    # the scatter layout prepared FOR this padding (aligned phases, one column range): same results
    wpb2 = torch.zeros(wpb.numel()).cuda()
    K.prep_conv_weight(dev(w), None, wpb2, s, pad=p)
    rb2 = torch.zeros(nb)
    KM.prep_conv_weight(w, None, rb2, s, pad=p)
    close(wpb2[:nb], rb2, rtol=0, atol=0)
    _check_bf16_image(wpb2, nb, (d0 + 1) // 2 * 2, (k + s - 1) // s)
    if kind == 'conv':
        dx2 = dev(torch.randn(B, cin, lin, generator=gen))
        base = dx2.clone()
        K.conv_engine(dev(gy), wpb2, dx2, k, s, p, 1, accumulate=True, wp_pad=p)
        close(dx2 - base, xr.grad, msg='backward-data, aligned layout', rtol=2e-3)
    else:
        y3 = torch.full((B, cout, lout), float('nan')).cuda()
        K.conv_engine(dev(x), wpb2, y3, k, s, p, 1, bias=dev(bias), res=dev(res), lens=dev(lens), act=K.ACT_LEAKY,
                      wp_pad=p)
        close(y3, ref, msg='forward, aligned layout')


def test_deferred_second_stages_are_bitwise_the_same(K):
    """K.deferred_reduces(): the second stages of many two-stage reductions in ONE launch at the end of the block - every
    output bit for bit what the single launches give (also when two of them write the same tensor: flushed in between),
    nothing written before the flush, and an error path that leaves deferral off"""
    gen = torch.Generator().manual_seed(44)
    layers = (G_LAYERS + D_LAYERS)[:9]
    B = 5
    data = []
    for kind, cin, cout, k, s, p, lin in layers:
        w, x = _mk(kind, cin, cout, k, s, p, lin, B, 3)
        y = F.conv1d(x, w, None, s, p) if kind == 'conv' else F.conv_transpose1d(x, w, None, s, p)
        data.append((kind, k, s, p, dev(x), dev(torch.randn(y.shape, generator=gen)), w.shape, cout))

    def run(deferred):
        outs = [(torch.zeros(shp).cuda(), torch.zeros(cout).cuda()) for *_, shp, cout in data]
        shared = torch.zeros(data[0][7]).cuda()

        def body():
            for (kind, k, s, p, x, gy, shp, cout), (dw, db) in zip(data, outs):
                if kind == 'conv':
                    K.conv_wgrad(gy, x, dw, k, s, p)
                else:
                    K.conv_wgrad(x, gy, dw, k, s, p)
                K.channel_sum(gy, db)
            if deferred:
                torch.cuda.synchronize()
                pending = sum(not bool(t.any()) for pair in outs for t in pair)
                assert pending >= len(outs), 'second stages ran before the flush (%d outputs still zero)' % pending
            K.channel_sum(data[0][5], shared)            # the same output twice: must not share a launch
            K.channel_sum(data[0][5], shared)
        if deferred:
            with K.deferred_reduces():
                body()
        else:
            body()
        torch.cuda.synchronize()
        return [t for pair in outs for t in pair] + [shared]

    plain, deferred = run(False), run(True)
    assert all(bool(t.any()) for t in plain)
    for a, b in zip(plain, deferred):
        assert torch.equal(a, b)
    with pytest.raises(RuntimeError):
        with K.deferred_reduces():
            K.channel_sum(data[0][5], torch.zeros(data[0][7]).cuda())
            raise RuntimeError('boom')
    again = run(False)                                   # deferral is off again, nothing stale is flushed later
    for a, b in zip(plain, again):
        assert torch.equal(a, b)


@pytest.mark.parametrize('layer', G_LAYERS + D_LAYERS + EDGE_LAYERS)
def test_conv_weight_bias_grad(K, layer):
    kind, cin, cout, k, s, p, lin = layer
    B = 3
    w, x = _mk(kind, cin, cout, k, s, p, lin, B, 3)
    wr, xr = w.clone().requires_grad_(True), x
    b = torch.zeros(cout, requires_grad=True)
    y = F.conv1d(xr, wr, b, s, p) if kind == 'conv' else F.conv_transpose1d(xr, wr, b, s, p)
    gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(4))
    y.backward(gy)
    dw = torch.zeros_like(w).cuda()
    db = torch.zeros(cout).cuda()
    if kind == 'conv':
        K.conv_wgrad(dev(gy), dev(x), dw, k, s, p)
    else:
        K.conv_wgrad(dev(x), dev(gy), dw, k, s, p)
    K.channel_sum(dev(gy), db)
    close(dw, wr.grad, msg='dw')
    close(db, b.grad, msg='db')


def test_conv_slab_views_and_residual_grad(K):
    """strided views into a channel slab (what removes the T.cat) + leaky_bwd add_into"""
    B, L = 2, 256
    gen = torch.Generator().manual_seed(5)
    slab = torch.randn(B, 12, L, generator=gen)
    w = torch.randn(4, 8, 8, generator=gen) * 0.2          # convT 8 -> 4, k8 s4 p2 on hidden
    hid = torch.randn(B, 8, L // 4, generator=gen)
    wpb = torch.zeros(K.wpb_numel(8, 4, 8, 4)).cuda()
    K.prep_conv_weight(dev(w.permute(1, 0, 2).contiguous()), None, wpb, 4)
    wt = w.permute(1, 0, 2).contiguous()                   # [Cin=8, Cout=4, K]
    ref = F.leaky_relu(F.conv_transpose1d(hid, wt, None, 4, 2) + slab[:, 4:8])
    s_d = dev(slab.clone())
    K.conv_engine(dev(hid), wpb, s_d[:, 8:12], 8, 4, 2, 1, res=s_d[:, 4:8], act=K.ACT_LEAKY)
    close(s_d[:, 8:12], ref)
    close(s_d[:, :8], slab[:, :8], rtol=0, atol=0)         # neighbours untouched
    dsl = torch.randn(B, 12, L, generator=gen)
    d_d = dev(dsl.clone())
    db = torch.full((4,), 0.5).cuda()                      # accumulates: starts from a non-zero value
    K.leaky_bwd(d_d[:, 8:12], s_d[:, 8:12], d_d[:, 8:12], add_into=d_d[:, 4:8], bias_grad=db)
    g = torch.where(ref > 0, dsl[:, 8:12], dsl[:, 8:12] * 0.01)
    close(d_d[:, 8:12], g)
    close(db, 0.5 + g.sum((0, 2)), rtol=1e-4, atol=1e-4)   # bias gradient in the same pass
    close(d_d[:, 4:8], dsl[:, 4:8] + g)
    close(d_d[:, :4], dsl[:, :4], rtol=0, atol=0)


@pytest.mark.parametrize('shape', [(64, 4096, 1480), (8192, 1024, 1024), (8192, 1, 512), (37, 53, 29),
                                   (64, 256, 1024), (130, 70, 5), (1, 1, 1), (2048, 4096, 200),
                                   (256, 384, 4096), (200, 132, 2048),     # LDS-DMA kernel: split-K, ragged tiles
                                   # the larger tiles of gemm_tile.h as gemm_pick_tile selects them, ragged in both directions:
                                   # 256 x 256, 256 x 128, 128 x 256
                                   (4000, 4072, 64), (8100, 1000, 48), (128, 65500, 32)])
@pytest.mark.parametrize('ta,tb', [(False, True), (False, False), (True, False), (True, True)])
def test_gemm(K, shape, ta, tb):
    M, N, Kd = shape
    if M * N * Kd > 3e9 and (ta or not tb):
        pytest.skip('largest shape only in the Linear orientation')
    gen = torch.Generator().manual_seed(6)
    A = torch.randn((Kd, M) if ta else (M, Kd), generator=gen) / Kd ** 0.5
    Bm = torch.randn((N, Kd) if tb else (Kd, N), generator=gen)
    C0 = torch.randn(M, N, generator=gen)
    bias, res = torch.randn(N, generator=gen), torch.randn(M, N, generator=gen)
    ref = F.leaky_relu(0.5 * ((A.t() if ta else A) @ (Bm.t() if tb else Bm)) + 2.0 * C0 + bias + res)
    Cd = dev(C0.clone())
    K.gemm(dev(A), dev(Bm), Cd, ta=ta, tb=tb, alpha=0.5, beta=2.0, bias=dev(bias), res=dev(res),
           act=K.ACT_LEAKY)
    close(Cd, ref)


@pytest.mark.parametrize('shape,ta,tb', [((8192, 1024, 1024), False, False), ((8192, 512, 1024), False, True), ((130, 70, 24), False, False),
                                         ((37, 53, 32), True, True), ((64, 256, 1024), False, True)])
@pytest.mark.parametrize('prec', ['f32', 'bf16'])
def test_gemm_leaky_gate_epilogue(K, shape, ta, tb, prec):
    """ACT_LEAKY_GATE: the product scaled by the derivative of the LeakyReLU whose saved OUTPUT is passed as `res` (not
    added) - every GEMM kernel (128-tile DMA / register-staged, 64-tile, bf16 MFMA), interior and ragged tiles"""
    M, N, Kd = shape
    gen = torch.Generator().manual_seed(16)
    A = torch.randn((Kd, M) if ta else (M, Kd), generator=gen) / Kd ** 0.5
    Bm = torch.randn((N, Kd) if tb else (Kd, N), generator=gen)
    bias, y = torch.randn(N, generator=gen), torch.randn(M, N, generator=gen)
    rb = (lambda t: t.to(torch.bfloat16).float()) if prec == 'bf16' else (lambda t: t)
    ref = ((rb(A).t() if ta else rb(A)) @ (rb(Bm).t() if tb else rb(Bm)) + bias) * torch.where(y > 0, 1.0, 0.01)
    Cd = torch.full((M, N), float('nan')).cuda()
    with K.precision(prec):
        K.gemm(dev(A), dev(Bm), Cd, ta=ta, tb=tb, bias=dev(bias), res=dev(y), act=K.ACT_LEAKY_GATE)
        with pytest.raises(Exception):
            K.gemm(dev(A), dev(Bm), Cd.clone(), ta=ta, tb=tb, act=K.ACT_LEAKY_GATE)       # the gate needs `res`
    close(Cd, ref, rtol=2e-3 if prec == 'bf16' else 1e-3, atol=1e-4 if prec == 'bf16' else 1e-5)


@pytest.mark.parametrize('B,C,T', [(128, 512, 128), (3, 70, 33), (1, 1, 1), (5, 32, 100)])
def test_time_major_transposes(K, B, C, T):
    """[B,C,T] <-> [T,B,C] through the tiled transpose (ag_transpose_batched), also from a channel slice of a wider slab"""
    gen = torch.Generator().manual_seed(12)
    wide = torch.randn(B, C + 3, T, generator=gen).cuda()
    x = wide[:, 1:C + 1]
    y = K.bct_to_tbc(x)
    assert y.is_contiguous() and torch.equal(y, x.permute(2, 0, 1))
    back = K.tbc_to_bct(y)
    assert back.is_contiguous() and torch.equal(back, x)
    from audiogan_amd import ops
    a = x.clone().requires_grad_(True)
    g = torch.randn(T, B, C, generator=gen).cuda()
    ops.TimeMajorFn.apply(a).backward(g)
    assert torch.equal(a.grad, g.permute(1, 2, 0))


def test_gemm_strided_views(K):
    gen = torch.Generator().manual_seed(7)
    big = torch.randn(64, 456, generator=gen)
    w = torch.randn(128, 456, generator=gen)
    out = torch.zeros(10, 64, 128)
    o_d = dev(out)
    K.gemm(dev(big)[:, 256:], dev(w)[:, 256:], o_d[3], tb=True)
    close(o_d[3], big[:, 256:] @ w[:, 256:].t())
    x = torch.randn(64, 8 * 32, generator=gen)
    x_d = dev(x.clone())
    h = torch.randn(64, 100, generator=gen)
    pw = torch.randn(32, 100, generator=gen)
    K.gemm(dev(h), dev(pw), x_d[:, 64:96], tb=True, act=K.ACT_TANH)
    ref = x.clone()
    ref[:, 64:96] = torch.tanh(h @ pw.t())
    close(x_d, ref)
    cs = torch.zeros(128).cuda()
    K.col_sum(dev(w)[:, 200:328], cs)
    close(cs, w[:, 200:328].sum(0))


def test_lstm_cell(K):
    B, H = 5, 70
    gen = torch.Generator().manual_seed(8)
    gates, cp, hp = torch.randn(B, 4 * H, generator=gen), torch.randn(B, H, generator=gen), torch.randn(B, H, generator=gen)
    valid = torch.tensor([3, 1, 0, 7, 2])
    for v, t in ((None, 0), (valid, 2)):
        rg, rc, rh, ry = gates.clone(), torch.empty(B, H), torch.empty(B, H), torch.empty(B, H)
        KM.lstm_cell_fwd(rg, cp, rc, rh, ry, hp, v, t)
        g_d, c_d, h_d, y_d = dev(gates.clone()), torch.empty(B, H).cuda(), torch.empty(B, H).cuda(), torch.empty(B, H).cuda()
        K.lstm_cell_fwd(g_d, dev(cp), c_d, h_d, y_d, dev(hp), dev(v), t)
        ok = torch.ones(B, dtype=torch.bool) if v is None else (t < v)
        close(g_d[ok], rg[ok]); close(c_d, rc); close(h_d, rh); close(y_d, ry)
        dh, dy, dcn = torch.randn(B, H, generator=gen), torch.randn(B, H, generator=gen), torch.randn(B, H, generator=gen)
        rdg, rdc, rdp = torch.empty(B, 4 * H), torch.empty(B, H), torch.empty(B, H)
        KM.lstm_cell_bwd(rg, cp, rc, dh, dy, dcn, rdg, rdc, rdp, v, t)
        dg_d, dc_d, dp_d = torch.empty(B, 4 * H).cuda(), torch.empty(B, H).cuda(), torch.empty(B, H).cuda()
        K.lstm_cell_bwd(dev(rg), dev(cp), dev(rc), dev(dh), dev(dy), dev(dcn), dg_d, dc_d, dp_d, dev(v), t)
        close(dg_d, rdg); close(dc_d, rdc); close(dp_d, rdp)


def test_weight_norm(K):
    gen = torch.Generator().manual_seed(9)
    shapes = [(128, 1, 17), (128, 16, 16), (64, 49, 9), (4096, 456), (4096,), (1, 113, 3), (512, 256, 7), (3,)]
    strides = [8, 8, 4, 1, 1, 1, 2, 1]
    ents_d, ents_c = [], []
    for sh, s in zip(shapes, strides):
        v = torch.randn(sh, generator=gen)
        g = torch.rand(sh[0], generator=gen) + 0.5
        e = dict(v=v, g=g, w=torch.empty(sh), stride=s)
        d = {k: (dev(t) if torch.is_tensor(t) else t) for k, t in e.items()}
        if len(sh) == 3:
            e['wpa'] = torch.zeros(KM.wpa_numel(sh[0], sh[1], sh[2]))
            e['wpb'] = torch.zeros(KM.wpb_numel(sh[0], sh[1], sh[2], s))
            d['wpa'] = torch.zeros(K.wpa_numel(sh[0], sh[1], sh[2])).cuda()      # fp32 layout + its bf16 image
            d['wpb'] = torch.zeros(K.wpb_numel(sh[0], sh[1], sh[2], s)).cuda()
        ents_c.append(e)
        ents_d.append(d)
    KM.weight_norm_fwd(ents_c)
    K.weight_norm_fwd(ents_d)
    for a, b, sh, s in zip(ents_d, ents_c, shapes, strides):
        for k in ('w', 'wpa', 'wpb'):
            if k in b:
                close(a[k][:b[k].numel()], b[k], rtol=1e-5, msg=k)
        if len(sh) == 3:
            _check_bf16_image(a['wpa'], b['wpa'].numel(), (sh[1] + 1) // 2 * 2, sh[2])
            _check_bf16_image(a['wpb'], b['wpb'].numel(), (sh[0] + 1) // 2 * 2, (sh[2] + s - 1) // s)
    bd, bc = [], []
    for e in ents_c:
        dw = torch.randn(e['v'].shape, generator=gen)
        c = dict(v=e['v'], g=e['g'], dw=dw, dv=torch.empty_like(dw), dg=torch.empty_like(e['g']))
        bc.append(c)
        bd.append({k: dev(t) for k, t in c.items()})
    KM.weight_norm_bwd(bc)
    K.weight_norm_bwd(bd)
    for a, b in zip(bd, bc):
        close(a['dg'], b['dg'], rtol=1e-4)
        if b['v'].dim() == 1:
            # analytically zero (w = g*sign(v)); what is left is cancellation noise of size
            # eps * |g/v| * |dw| on both sides
            bound = 1e-5 * float((b['g'] / b['v'].abs() * b['dw'].abs()).max())
            assert float(a['dv'].abs().max()) <= bound and float(b['dv'].abs().max()) <= bound
        else:
            close(a['dv'], b['dv'], rtol=1e-4, atol=1e-6)


def test_bce(K):
    gen = torch.Generator().manual_seed(10)
    x = torch.randn(64, 128, generator=gen) * 5
    x[0, 0], x[0, 1] = 80.0, -80.0
    n = torch.randint(1, 129, (64,), generator=gen)
    for tgt in (0.9, 0.0, 0.5):
        for nn_ in (n, None):
            per_r, loss_r = torch.empty(64), torch.zeros(1)
            KM.bce_logits_fwd(x, tgt, nn_, per_r, loss_r, 1 / 64)
            per, loss = torch.empty(64).cuda(), torch.zeros(1).cuda()
            K.bce_logits_fwd(dev(x), tgt, dev(nn_), per, loss, 1 / 64)
            close(per, per_r, rtol=1e-5); close(loss, loss_r, rtol=1e-5)
            dx_r, dx = torch.empty_like(x), torch.empty_like(x).cuda()
            gs = torch.tensor([0.7])
            KM.bce_logits_bwd(x, tgt, nn_, gs, 1 / 64, dx_r)
            K.bce_logits_bwd(dev(x), tgt, dev(nn_), dev(gs), 1 / 64, dx)
            close(dx, dx_r, rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize('B,T', [(128, 128), (5, 37), (1, 1), (70, 200)])
def test_bce_strided_rows(K, B, T):
    """the one-launch loss on a TRANSPOSED view (what the heads hand over), with per-row targets and a caller's scale:
    against torch's binary_cross_entropy_with_logits in float64, forward (per-sample sums, loss WRITTEN over garbage) and
    backward (into a gradient of the same transposed layout); and through ops.BCEFn / autograd"""
    gen = torch.Generator().manual_seed(13)
    base = torch.randn(T, B, generator=gen) * 4                    # memory layout [T,B]; the op sees [B,T]
    n = torch.randint(1, T + 1, (B,), generator=gen)
    rows = torch.where(torch.arange(B) < (B + 1) // 2, 0.9, 0.0)
    mask = (torch.arange(T).view(1, T) < n.view(B, 1)).double()
    for tr, tgt, nn_ in ((rows, 0.0, n), (None, 0.5, n), (rows, 0.0, None)):
        x64 = base.t().double().requires_grad_(True)
        tg = (tr.view(B, 1).expand(B, T) if tr is not None else torch.full((B, T), tgt)).double()
        m = mask if nn_ is not None else torch.ones(B, T, dtype=torch.float64)
        nn64 = nn_.double() if nn_ is not None else torch.full((B,), float(T), dtype=torch.float64)
        per64 = (F.binary_cross_entropy_with_logits(x64, tg, reduction='none') * m).sum(1)
        loss64 = (2.0 / B) * (per64 / nn64).sum()
        loss64.backward()
        xd = dev(base).t()                                         # strides (1, B)
        per, loss = torch.empty(B).cuda(), torch.full((1,), float('nan')).cuda()
        K.bce_logits_fwd_strided(xd, tgt, dev(nn_), per, loss, 2.0 / B, target_rows=dev(tr))
        close(per, per64.float(), rtol=1e-5, atol=1e-6); close(loss, loss64.float().view(1), rtol=1e-5)
        dx = torch.empty_like(xd)
        assert dx.stride() == xd.stride()
        K.bce_logits_bwd_strided(xd, tgt, dev(nn_), dev(torch.tensor([0.7])), 2.0 / B, dx, target_rows=dev(tr))
        close(dx, 0.7 * x64.grad.float(), rtol=1e-5, atol=1e-9)
        from audiogan_amd import ops
        xa = dev(base).requires_grad_(True)
        l2, _ = ops.BCEFn.apply(xa.t(), dev(tr) if tr is not None else tgt, dev(nn_), 2.0 / B)
        l2.backward()
        close(l2.view(1), loss64.float().view(1), rtol=1e-5)
        close(xa.grad, x64.grad.float().t(), rtol=1e-5, atol=1e-9)


def test_elementwise(K):
    gen = torch.Generator().manual_seed(11)
    x, dy = torch.randn(100003, generator=gen), torch.randn(100003, generator=gen)
    for act in (K.ACT_LEAKY, K.ACT_TANH, K.ACT_NONE):
        y, yr = torch.empty_like(x).cuda(), torch.empty_like(x)
        K.act_fwd(dev(x), y, act); KM.act_fwd(x, yr, act); close(y, yr, rtol=1e-6)
        dx, dxr = torch.empty_like(x).cuda(), torch.empty_like(x)
        K.act_bwd(dev(dy), y, dx, act); KM.act_bwd(dy, yr, dxr, act); close(dx, dxr, rtol=1e-5)
    y = dev(dy.clone())
    K.axpby(dev(x), y, 2.0, -0.5)
    close(y, 2.0 * x - 0.5 * dy, rtol=1e-6)


@pytest.mark.parametrize('kind', ['rmsprop', 'adam'])
def test_fused_optimizer_vs_torch(K, kind):
    from audiogan_amd import optim
    gen = torch.Generator().manual_seed(12)
    shapes = [(512, 256, 7), (4096, 1024), (4096,), (1,), (33, 5)]
    ps_ref = [torch.nn.Parameter(torch.randn(s, generator=gen)) for s in shapes]
    ps = [torch.nn.Parameter(p.detach().clone().cuda()) for p in ps_ref]
    oref = (torch.optim.RMSprop(ps_ref, lr=1e-3) if kind == 'rmsprop' else torch.optim.Adam(ps_ref, lr=1e-3))
    o = optim.make_optimizer(ps, kind, 1e-3)
    from oracle import audiogan_oracle as O
    for it in range(3):
        scales = [30.0, 1e-3, 1.0, 5.0, 0.1]
        for p, q, sc in zip(ps_ref, ps, scales):
            g = torch.randn(p.shape, generator=gen) * sc
            p.grad, q.grad = g.clone(), g.clone().cuda()
        tot = O.clip_grad(ps_ref, 1.0)
        oref.step()
        # check=True: the norms are finished by a launch of their own (the flags are read before the update); check=False:
        # the update launch finishes them itself (two launches per network) - the same numbers either way
        ns = o.step(clip_norm=1.0, check=(it != 1))
        close(ns, float(tot), rtol=1e-5)
        assert int(o.last_flags.item()) == 0
        for p, q in zip(ps_ref, ps):
            close(q, p, rtol=1e-5, atol=2e-6)   # lr=1e-3 steps: a 1-ulp difference in g/sqrt(v)
    ps[0].grad[0, 0, 0] = float('nan')
    o.step(clip_norm=1.0, check=False)
    assert int(o.last_flags.item()) & 1, 'the fused form must still report NaN gradients'
    ps[0].grad.zero_()
    ps[1].grad[5, 5] = 3e5
    o2 = optim.make_optimizer([torch.nn.Parameter(p.detach().clone()) for p in ps], kind, 1e-3)
    for q, p in zip(o2.params, ps):
        q.grad = p.grad.clone()
    with pytest.raises(AssertionError):
        o2.step(clip_norm=1.0, check=True)


def test_error_convention(K):
    """C status codes surface as Python exceptions; nothing aborts"""
    x = torch.zeros(2, 3, 8).cuda()
    with pytest.raises(ValueError):
        K.conv_engine(x, torch.zeros(K.wpa_numel(4, 3, 3)).cuda(), torch.zeros(2, 4, 9999).cuda(), 3, 1, 1, 0)
    with pytest.raises(RuntimeError):
        K.act_fwd(torch.zeros(4), torch.zeros(4), K.ACT_LEAKY)   # CPU tensor: no fallback


@pytest.mark.parametrize('shape', [(64, 2048, 512), (64, 512, 2048), (64, 256, 1024), (64, 1024, 4096), (128, 512, 2048),
                                   (3, 40, 24), (33, 7, 8), (64, 4096, 256), (40, 100, 512)])
@pytest.mark.parametrize('tb', [True, False])
def test_skinny_gemm(K, shape, tb):
    M, N, Kd = shape
    gen = torch.Generator().manual_seed(13)
    A = torch.randn(M, Kd, generator=gen) / Kd ** 0.5
    Bm = torch.randn((N, Kd) if tb else (Kd, N), generator=gen)
    C0, bias = torch.randn(M, N, generator=gen), torch.randn(N, generator=gen)
    prod = A @ (Bm.t() if tb else Bm)
    Cd = dev(C0.clone())
    K.skinny_gemm(dev(A), dev(Bm), Cd, tb=tb, beta=1.0, bias=dev(bias), act=K.ACT_TANH)
    close(Cd, torch.tanh(prod + C0 + bias))
    Cd = dev(C0.clone())
    K.skinny_gemm(dev(A), dev(Bm), Cd, tb=tb, atomic=True)
    close(Cd, prod + C0)
    Cd = dev(C0.clone())
    K.skinny_gemm(dev(A), dev(Bm), Cd, tb=tb)
    close(Cd, prod)


def test_skinny_rejects_unaligned(K):
    with pytest.raises(ValueError):
        K.skinny_gemm(torch.zeros(4, 12).cuda(), torch.zeros(5, 12).cuda(), torch.zeros(4, 5).cuda(), tb=True)
    assert not K.skinny_ok(torch.zeros(4, 12).cuda(), torch.zeros(5, 12).cuda(), True)
    assert not K.skinny_ok(torch.zeros(300, 16).cuda(), torch.zeros(5, 16).cuda(), True)


@pytest.mark.parametrize('B,H,Kx', [(64, 1024, 256), (64, 512, 0), (5, 24, 16), (33, 64, 32)])
def test_lstm_step_fwd(K, B, H, Kx):
    gen = torch.Generator().manual_seed(14)
    pre = torch.randn(B, 4 * H, generator=gen)
    xfull = torch.randn(B, 3 * max(Kx, 8), generator=gen)
    wfull = torch.randn(4 * H, max(Kx, 8) + 40, generator=gen) / 8
    x, wx = (xfull[:, Kx:2 * Kx], wfull[:, :Kx]) if Kx else (None, None)
    hp, cp = torch.randn(B, H, generator=gen), torch.randn(B, H, generator=gen)
    whh = torch.randn(4 * H, H, generator=gen) / H ** 0.5
    for first in (False, True):
        rg, rc, rh = pre.clone(), torch.empty(B, H), torch.empty(B, H)
        KM.lstm_step_fwd(rg, x, wx, hp, whh, cp, rc, rh, first)
        g_d, c_d, h_d = dev(pre.clone()), torch.empty(B, H).cuda(), torch.empty(B, H).cuda()
        xd, wd = dev(xfull), dev(wfull)
        K.lstm_step_fwd(g_d, xd[:, Kx:2 * Kx] if Kx else None, wd[:, :Kx] if Kx else None, dev(hp), dev(whh),
                        dev(cp), c_d, h_d, first)
        close(g_d, rg, rtol=1e-4); close(c_d, rc, rtol=1e-4); close(h_d, rh, rtol=1e-4)


@pytest.mark.parametrize('persist', [True, False])
@pytest.mark.parametrize('T,B,H,ndir,ragged', [(128, 64, 512, 2, True), (16, 128, 64, 2, True), (5, 3, 8, 2, True), (7, 33, 24, 1, False),
                                               (1, 4, 16, 2, True), (6, 40, 96, 1, True), (9, 70, 32, 2, True), (3, 150, 512, 2, True),
                                               (33, 128, 512, 2, True), (5, 70, 192, 1, True), (4, 33, 64, 2, False),
                                               (3, 256, 256, 2, True), (6, 20, 128, 2, True), (9, 17, 256, 1, True)])
def test_lstm_seq_fwd_bwd(K, T, B, H, ndir, ragged, persist):
    """persist=True: shapes that fit the chip take the persistent weights-resident launch (lstm_persist.hip), the
    others one launch per step; persist=False forces the per-step kernels for every shape"""
    fits = K.lib.ag_lstm_persist_ok(B, H, ndir, 256) or K.lib.ag_lstm_persist_bwd_ok(B, H, ndir, 256)
    if persist and not fits:
        pytest.skip('shape does not take the persistent kernel')
    old = K.PERSIST[0]
    K.PERSIST[0] = persist
    try:
        _lstm_seq_case(K, T, B, H, ndir, ragged)
        assert K.lstm_persist_status() == 0, 'a bounded wait of the persistent launch timed out'
    finally:
        K.PERSIST[0] = old


def _lstm_seq_case(K, T, B, H, ndir, ragged):
    gen = torch.Generator().manual_seed(15)
    pre = [torch.randn(T, B, 4 * H, generator=gen) for _ in range(ndir)]
    whh = [torch.randn(4 * H, H, generator=gen) / H ** 0.5 for _ in range(ndir)]
    valid = torch.randint(1, T + 1, (B,), generator=gen) if ragged else None
    if ragged:
        valid[0] = T
    def run(mod, to):
        p = [to(t.clone()) for t in pre]
        w = [to(t) for t in whh]
        c = [to(torch.zeros(T + 1, B, H)) for _ in range(ndir)]
        hb = [to(torch.zeros(2, B, H)) for _ in range(ndir)]
        y = to(torch.full((T, B, ndir * H), float('nan')))
        v = to(valid) if valid is not None else None
        # odd T: also exercise the time-invariant pre-activation term (static input + biases, added in the step)
        st = [to(torch.randn(B, 4 * H, generator=torch.Generator().manual_seed(17 + d_))) for d_ in range(ndir)] \
            if T % 2 == 1 else None
        mod.lstm_seq_fwd(p, w, c, hb, y, v, static=st)
        dy = to(torch.randn(T, B, ndir * H, generator=torch.Generator().manual_seed(16)))
        dg = [to(torch.full((T, B, 4 * H), float('nan'))) for _ in range(ndir)]
        dh = [to(torch.zeros(2, B, H)) for _ in range(ndir)]
        dc = [to(torch.zeros(2, B, H)) for _ in range(ndir)]
        mod.lstm_seq_bwd(p, w, c, dy, dg, dh, dc, v)
        return y, c, dg, p
    yr, cr, dgr, pr = run(KM, lambda t: t)
    y, c, dg, p = run(K, lambda t: t.cuda())
    close(y, yr, rtol=1e-3, atol=1e-5)
    for d in range(ndir):
        close(c[d], cr[d], rtol=1e-3, atol=1e-5)
        close(dg[d], dgr[d], rtol=2e-3, atol=2e-5)


@pytest.mark.parametrize('C,L,Kk', [(113, 8192, 3), (5, 100, 1), (7, 1030, 9), (1, 4, 3)])
def test_conv_single_output_channel(K, C, L, Kk):
    B, p = 3, (Kk - 1) // 2
    gen = torch.Generator().manual_seed(17)
    x, w, b = torch.randn(B, C + 2, L, generator=gen)[:, 1:C + 1], torch.randn(1, C, Kk, generator=gen) / (C * Kk) ** 0.5, torch.randn(1, generator=gen)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    yref = torch.tanh(F.conv1d(xr, wr, b, 1, p))
    xd = dev(torch.randn(B, C + 2, L))
    xd[:, 1:C + 1] = dev(x)
    y = torch.empty(B, 1, L).cuda()
    K.conv_o1_fwd(xd[:, 1:C + 1], dev(w), dev(b), y, Kk, p, act=K.ACT_TANH)
    close(y, yref, msg='fwd')
    gy = torch.randn(B, 1, L, generator=gen)
    lin = F.conv1d(xr, wr, None, 1, p)
    lin.backward(gy)
    dx0 = torch.randn(B, C, L, generator=gen)
    dx = dev(dx0.clone())
    K.conv_o1_bwd_data(dev(gy), dev(w), dx, Kk, p, accumulate=True)
    close(dx, dx0 + xr.grad, msg='bwd-data')
    dw = torch.zeros(1, C, Kk).cuda()
    K.conv_o1_wgrad(dev(gy), xd[:, 1:C + 1], dw, Kk, p)
    close(dw, wr.grad, msg='wgrad')
    assert K.conv_o1_ok('conv', 1, Kk, 1, p) and not K.conv_o1_ok('conv', 2, Kk, 1, p)


def test_act_bwd2d(K):
    gen = torch.Generator().manual_seed(18)
    dy, y = torch.randn(64, 8192, generator=gen), torch.tanh(torch.randn(64, 8192, generator=gen))
    out = torch.empty(64, 256).cuda()
    K.act_bwd2d(dev(dy)[:, 512:768], dev(y)[:, 512:768], out, K.ACT_TANH)
    close(out, dy[:, 512:768] * (1 - y[:, 512:768] ** 2), rtol=1e-5)


@pytest.mark.parametrize('B,H,fs,T', [(64, 1024, 256, 3), (5, 32, 16, 2), (40, 512, 48, 1), (130, 64, 32, 2)])
def test_lstm_front_bwd_step(K, B, H, fs, T):
    """fused Generator-front backward step (tanh' on load, projection product, cell backward) on row views"""
    gen = torch.Generator().manual_seed(21)
    dacc = torch.randn(T, B, H + fs, generator=gen)            # [dL/dh | dL/dx] joint rows
    x = torch.tanh(torch.randn(B, T * fs, generator=gen))
    wp = torch.randn(fs, H, generator=gen) / fs ** 0.5
    gates = torch.sigmoid(torch.randn(B, 4 * H, generator=gen))
    gates[:, 2 * H:3 * H] = torch.tanh(torch.randn(B, H, generator=gen))
    cp, cn, dcn = (torch.randn(B, H, generator=gen) for _ in range(3))
    t = T - 1
    for dc_next in (None, dcn):
        rg, rdg, rdc = torch.empty(B, fs), torch.empty(B, 4 * H), torch.empty(B, H)
        KM.lstm_front_bwd_step(dacc[t, :, H:], x[:, t * fs:(t + 1) * fs], rg, wp, dacc[t, :, :H], gates, cp, cn,
                               dc_next, rdg, rdc)
        d_acc, d_x = dev(dacc), dev(x)
        g_, dg_, dc_ = torch.empty(B, fs).cuda(), torch.empty(B, 4 * H).cuda(), torch.empty(B, H).cuda()
        assert K.lstm_front_bwd_ok(B, H, fs, d_acc[t, :, H:], d_x[:, t * fs:(t + 1) * fs])
        K.lstm_front_bwd_step(d_acc[t, :, H:], d_x[:, t * fs:(t + 1) * fs], g_, dev(wp), d_acc[t, :, :H], dev(gates),
                              dev(cp), dev(cn), dev(dc_next) if dc_next is not None else None, dg_, dc_)
        close(g_, rg, rtol=1e-5, atol=1e-6)
        close(dg_, rdg, rtol=1e-3, atol=1e-5)
        close(dc_, rdc, rtol=1e-3, atol=1e-5)
        close(d_acc, dacc, rtol=0, atol=0)                     # inputs untouched


@pytest.mark.parametrize('B,C,L', [(4, 16, 4096), (3, 5, 100), (2, 1, 7), (64, 512, 128)])
def test_time_moments(K, B, C, L):
    """calc_dists statistics over time (mean, 2nd / 4th central moment roots) and their backward, ragged lengths,
    rows taken as a channel slice of a wider activation"""
    gen = torch.Generator().manual_seed(31)
    full = torch.randn(B, C + 3, L, generator=gen)
    lens = torch.randint(max(1, L // 2), L + 1, (B,), generator=gen)
    lens[0] = L
    full = full * (torch.arange(L).view(1, 1, L) < lens.view(B, 1, 1)).float()      # the critic zeroes padded steps
    h = full[:, 1:C + 1]
    m, s, f = (torch.empty(B, C) for _ in range(3))
    KM.time_moments_fwd(h, lens, m, s, f)
    hd = dev(full)[:, 1:C + 1]
    md, sd, fd = (torch.empty(B, C).cuda() for _ in range(3))
    K.time_moments_fwd(hd, dev(lens), md, sd, fd)
    close(md, m, rtol=1e-4, atol=1e-6); close(sd, s, rtol=1e-4, atol=1e-6); close(fd, f, rtol=1e-4, atol=1e-6)
    gm, gs, gf = (torch.randn(B, C, generator=gen) for _ in range(3))
    dh = torch.empty(B, C, L)
    KM.time_moments_bwd(h, lens, gm, gs, gf, dh)
    dhd = torch.zeros(B, C + 3, L).cuda()
    K.time_moments_bwd(hd, dev(lens), dev(gm), dev(gs), dev(gf), dhd[:, 1:C + 1])
    close(dhd[:, 1:C + 1], dh, rtol=2e-3, atol=1e-5)
    assert not dhd[:, 0].any() and not dhd[:, C + 1:].any()


@pytest.mark.parametrize('S,fs,B,T', [(128, 64, 7, 1), (128, 64, 5, 3), (128, 64, 40, 6), (128, 64, 64, 2), (1024, 256, 64, 4), (1024, 256, 33, 3)])
def test_gfront_persistent_launch(K, S, fs, B, T):
    """the Generator front's frame loop as ONE persistent launch (lstm_persist.hip) vs one launch per operation: frames,
    stop logits and every gradient (the backward runs on the saved gates / cells / hidden states of either forward)"""
    from audiogan_amd import ops
    from audiogan_amd.common import WNGroup
    assert K.lib.ag_gfront_persist_ok(B, S, fs, 256)
    gen = torch.Generator().manual_seed(23)
    Fz = 24
    front = ops.GFront(fs, 1, S)
    shapes = [(4 * S, fs + Fz), (4 * S, S), (4 * S,), (4 * S,), (fs, S), (fs,), (1, S), (1,)]
    params = []
    for shp in shapes:
        v = torch.nn.Parameter((torch.randn(shp, generator=gen) / (shp[-1] ** 0.5 if len(shp) > 1 else 4.0)).cuda())
        gq = torch.nn.Parameter((v.detach().reshape(shp[0], -1).norm(dim=1) if len(shp) > 1 else v.detach().abs())
                                .view([shp[0]] + [1] * (len(shp) - 1)).clone())
        front.group.add(v, gq)
        params += [v, gq]
    zc = torch.randn(T, B, Fz, generator=gen).cuda()
    gx, gs = torch.randn(B, T * fs, generator=gen).cuda(), torch.randn(B, T, generator=gen).cuda()
    outs = []
    old = K.PERSIST[0]
    gx_wide = torch.zeros(B, 3, T * fs).cuda()
    gx_wide[:, 0] = gx
    try:
        # third pass: the frames written straight into channel 0 of a [B, 3, T*fs] slab (what the Generator does) and the
        # output gradient handed back as a row-pitched view - the persistent launches read / write both in place
        for persist, ctot in ((False, 0), (True, 0), (True, 3)):
            K.PERSIST[0] = persist
            front.slab_channels = ctot
            for q in params:
                q.grad = None
            x, s = ops.GFrontFn.apply(zc, front, *front.group.params())
            assert x.stride(0) == (ctot if ctot else 1) * T * fs
            if ctot:
                x.backward(gx_wide[:, 0], retain_graph=True)
                (s * gs).sum().backward()
            else:
                ((x * gx).sum() + (s * gs).sum()).backward()
            torch.cuda.synchronize()
            assert K.lstm_persist_status() == 0
            outs.append((x.detach().clone(), s.detach().clone(), [q.grad.clone() for q in params]))
    finally:
        K.PERSIST[0] = old
        front.slab_channels = 0
    for k in (1, 2):
        close(outs[k][0], outs[0][0], rtol=1e-4, atol=1e-6)
        close(outs[k][1], outs[0][1], rtol=1e-4, atol=1e-5)
        for a, b in zip(outs[k][2], outs[0][2]):
            close(a, b, rtol=1e-3, atol=1e-5 * max(1.0, float(b.abs().max())))
    assert torch.equal(outs[2][0], outs[1][0])


def test_persistent_launch_timeout_surfaces(K):
    """a persistent launch whose group never completes (one workgroup muted on purpose, 2 ms timeout): the launch
    still ENDS, the sticky status word is set and survives later (healthy) launches, the stalled group's outputs are NaN,
    and the product's checks raise - ADVICE round 2: such a failure used to leave plausible garbage with rc 0"""
    from audiogan_amd import ops, optim
    T, B, H, ndir = 6, 40, 128, 2
    assert K.lstm_persist_ok(B, H, ndir, torch.device('cuda', 0))
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(T, B, 24, generator=gen).cuda()
    w = []
    for _ in range(ndir):
        w += [torch.nn.Parameter(torch.randn(4 * H, 24, generator=gen).cuda() * 0.1),
              torch.nn.Parameter(torch.randn(4 * H, H, generator=gen).cuda() * 0.1),
              torch.nn.Parameter(torch.zeros(4 * H).cuda()), torch.nn.Parameter(torch.zeros(4 * H).cuda())]
    K.lstm_persist_status(reset=True)
    y_ok = ops.LSTMSeqFn.apply(x, None, ndir, None, *w).detach().clone()
    torch.cuda.synchronize()
    assert K.lstm_persist_status() == 0 and bool(torch.isfinite(y_ok).all())
    try:
        K.persist_debug(timeout_ticks=200000, mute_block=0)          # 2 ms; block 0 never raises its flag
        y_bad = ops.LSTMSeqFn.apply(x, None, ndir, None, *w)
        torch.cuda.synchronize()
    finally:
        K.persist_debug(0, -1)
    st = K.lstm_persist_status()
    assert st & 0x80000000, hex(st)
    assert bool(torch.isnan(y_bad).any()), 'a group that gave up must poison its outputs'
    # a healthy launch afterwards does not clear the sticky word ...
    y2 = ops.LSTMSeqFn.apply(x, None, ndir, None, *w)
    torch.cuda.synchronize()
    assert torch.equal(y2.detach(), y_ok) and K.lstm_persist_status() == st
    # ... and the optimiser's check= path raises (then the word is reset)
    opt = optim.make_optimizer(w, 'adam', 1e-4)
    y2.sum().backward()
    with pytest.raises(K.PersistentLaunchError):
        opt.step(check=True)
    assert K.lstm_persist_status() == 0
    opt.step(check=True)


@pytest.mark.parametrize('cell', ['lstm', 'gru'])
def test_front_backward_timeout_surfaces(K, cell):
    """the fronts' persistent backward with one workgroup muted (2 ms timeout): the launch ends, the sticky word is set and
    what the stalled workgroups wrote is NaN; a healthy launch afterwards gives the healthy result again"""
    T, B, S, fs = 4, 40, 128, 64
    ng = 4 if cell == 'lstm' else 3
    gen = torch.Generator().manual_seed(9)
    r = lambda *shape: torch.randn(*shape, generator=gen).cuda()      # noqa: E731
    gates = torch.sigmoid(r(T, B, ng * S))
    if cell == 'lstm':
        gates[:, :, 2 * S:3 * S] = torch.tanh(r(T, B, S))
    else:
        gates[:, :, 2 * S:] = torch.tanh(r(T, B, S))
    state, gh, x = r(T + 1, B, S) * 0.5, r(T, B, ng * S) * 0.5, torch.tanh(r(B, T * fs))
    dh_ext, dx_ext = r(T, B, S) * 0.1, r(B, T * fs) * 0.1
    whh, wih, wp = r(ng * S, S) * 0.1, r(ng * S, fs + 8) * 0.1, r(fs, S) * 0.1
    wx = wih[:, :fs]

    def run():
        dgs, dgh, dxt = torch.empty(T, B, ng * S).cuda(), torch.empty(T, B, ng * S).cuda(), torch.empty(T, B, fs).cuda()
        if cell == 'lstm':
            K.gfront_bwd_persist(gates, state, x, dh_ext, dx_ext, whh, wx, wp, dgs, dxt)
        else:
            K.grufront_bwd_persist(gates, state, gh, x, dh_ext, dx_ext, whh, wx, wp, dgs, dgh, dxt)
        torch.cuda.synchronize()
        return dgs, dxt

    assert K.gfront_bwd_persist_ok(B, S, fs, torch.device('cuda', 0))
    K.lstm_persist_status(reset=True)
    ok = run()
    assert K.lstm_persist_status() == 0 and all(bool(torch.isfinite(t).all()) for t in ok)
    try:
        K.persist_debug(timeout_ticks=200000, mute_block=0)          # 2 ms; block 0 (an h tile) never raises its flag
        bad = run()
    finally:
        K.persist_debug(0, -1)
    st = K.lstm_persist_status()
    assert st & 0x80000000, hex(st)
    assert bool(torch.isnan(bad[0]).any()) and bool(torch.isnan(bad[1]).any()), 'workgroups that gave up must poison their outputs'
    again = run()
    assert all(torch.equal(a, b) for a, b in zip(again, ok)) and K.lstm_persist_status() == st
    K.lstm_persist_status(reset=True)


def test_persistent_workspace_is_never_freed(K):
    """a larger request after a buffer was handed out must not free that buffer (a captured graph keeps its pointer)"""
    dev = torch.device('cuda', torch.cuda.current_device())
    a = K._persist_workspace(dev, 1024)
    b = K._persist_workspace(dev, a.numel() + 4096)
    assert b.numel() >= a.numel() + 4096 and a.data_ptr() != b.data_ptr()
    assert any(t is a for t in K._persist_ws[dev]) and K._persist_ws[dev][0] is b
    assert K._persist_workspace(dev, 1024) is b


@pytest.mark.parametrize('S,fs,B,T', [(128, 64, 7, 1), (128, 64, 5, 3), (128, 64, 40, 6), (1024, 256, 64, 4), (1024, 256, 33, 3)])
def test_grufront_persistent_launch(K, S, fs, B, T):
    """the GRU-front generator's frame loop (BASELINE configs[3]) as ONE persistent launch vs one launch per operation:
    frames, stop logits and every gradient (the backward runs on the saved gates / hidden states of either forward)"""
    from audiogan_amd import ops
    assert K.lib.ag_gfront_persist_ok(B, S, fs, 256)
    gen = torch.Generator().manual_seed(29)
    Fz = 24
    front = ops.GRUFront(fs, S)
    shapes = [(3 * S, fs + Fz), (3 * S, S), (3 * S,), (3 * S,), (fs, S), (fs,), (1, S), (1,)]
    params = []
    for shp in shapes:
        v = torch.nn.Parameter((torch.randn(shp, generator=gen) / (shp[-1] ** 0.5 if len(shp) > 1 else 4.0)).cuda())
        gq = torch.nn.Parameter((v.detach().reshape(shp[0], -1).norm(dim=1) if len(shp) > 1 else v.detach().abs())
                                .view([shp[0]] + [1] * (len(shp) - 1)).clone())
        front.group.add(v, gq)
        params += [v, gq]
    zc = torch.randn(T, B, Fz, generator=gen).cuda()
    gx, gs = torch.randn(B, T * fs, generator=gen).cuda(), torch.randn(B, T, generator=gen).cuda()
    outs = []
    old = K.PERSIST[0]
    try:
        for persist in (False, True):
            K.PERSIST[0] = persist
            for q in params:
                q.grad = None
            x, s = ops.GRUFrontFn.apply(zc, front, *front.group.params())
            ((x * gx).sum() + (s * gs).sum()).backward()
            torch.cuda.synchronize()
            assert K.lstm_persist_status() == 0
            outs.append((x.detach().clone(), s.detach().clone(), [q.grad.clone() for q in params]))
    finally:
        K.PERSIST[0] = old
    close(outs[1][0], outs[0][0], rtol=1e-4, atol=1e-6)
    close(outs[1][1], outs[0][1], rtol=1e-4, atol=1e-5)
    for a, b in zip(outs[1][2], outs[0][2]):
        close(a, b, rtol=1e-3, atol=1e-5 * max(1.0, float(b.abs().max())))


@pytest.mark.parametrize('k,s,p,cout,lin,B', [(7, 2, 3, 16, 8192, 3), (7, 2, 3, 16, 1001, 2), (7, 2, 3, 5, 64, 1),
                                             (17, 8, 8, 128, 8192, 2), (17, 8, 8, 128, 1000, 3), (17, 8, 8, 33, 40, 1)])
def test_single_input_channel_conv_kernels(K, k, s, p, cout, lin, B):
    """D1 (one input channel, k7 s2): the streaming kernels of conv_c1.hip (G1.conv's shape stays on the engine; its cases
    here check that the dispatch leaves it alone) - forward with bias + LeakyReLU + length
    mask, backward-data (plain and accumulating), backward-weight - against float64 convolutions, and against the general
    engine (AG_CONV_C1=0) which must agree to fp32 rounding"""
    from audiogan_amd import ops
    gen = torch.Generator().manual_seed(61)
    w = torch.randn(cout, 1, k, generator=gen) / k ** 0.5
    x = torch.randn(B, 1, lin, generator=gen)
    spec = ops.ConvSpec('conv', 1, cout, k, s, p)
    lout = spec.out_len(lin)
    bias = torch.randn(cout, generator=gen)
    lens = torch.randint(1, lout + 1, (B,), generator=gen)
    lens[0] = lout
    dy = torch.randn(B, cout, lout, generator=gen)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    lin_out = F.conv1d(xr, wr, None, s, p)
    ref = F.leaky_relu(lin_out + bias.double().view(1, -1, 1), K.LEAKY_SLOPE) * \
        (torch.arange(lout).view(1, 1, -1) < lens.view(B, 1, 1)).double()
    lin_out.backward(dy.double())
    prep = ops.Prepared(w=None, wpa=torch.zeros(K.wpa_numel(cout, 1, k)).cuda(), wpb=torch.zeros(K.wpb_numel(cout, 1, k, s)).cuda(), pad=p)
    K.prep_conv_weight(w.cuda(), prep.wpa, prep.wpb, s, p)
    dx0 = torch.randn(B, 1, lin, generator=gen)
    res = {}
    for sw in ('1', '0'):
        os.environ['AG_CONV_C1'] = sw
        try:
            y = torch.full((B, cout, lout), float('nan')).cuda()
            K.conv_engine(x.cuda(), prep.wpa, y, k, s, p, 0, bias=bias.cuda(), lens=lens.cuda(), act=K.ACT_LEAKY)
            dx = torch.full((B, 1, lin), float('nan')).cuda()
            ops.conv_bwd_data(spec, prep, dy.cuda(), dx)
            dxa = dx0.cuda().clone()
            ops.conv_bwd_data(spec, prep, dy.cuda(), dxa, accumulate=True)
            dw = torch.zeros(cout, 1, k).cuda()
            ops.conv_wgrad(spec, x.cuda(), dy.cuda(), dw, None)
            res[sw] = (y.cpu(), dx.cpu(), dxa.cpu(), dw.cpu())
        finally:
            os.environ.pop('AG_CONV_C1', None)
    y, dx, dxa, dw = res['1']
    close(y, ref, rtol=1e-4, atol=1e-5, msg='fwd')
    close(dx, xr.grad, rtol=1e-4, atol=1e-4 * float(xr.grad.abs().max()), msg='bwd-data')
    close(dxa, dx0.double() + xr.grad, rtol=1e-4, atol=1e-4 * float(xr.grad.abs().max()), msg='bwd-data accumulate')
    close(dw, wr.grad, rtol=1e-4, atol=1e-4 * float(wr.grad.abs().max()), msg='bwd-weight')
    for a, b, n in zip(res['1'], res['0'], ('fwd', 'bwd-data', 'bwd-data acc', 'bwd-weight')):
        close(a, b, rtol=1e-4, atol=1e-4 * max(1.0, float(b.abs().max())), msg='vs engine: ' + n)


def test_input_assembly_kernels(K):
    """ag_build_zc / ag_critic_batch (one launch each) against the torch expressions they replace (audiogan.py:433-439,
    :724-728, :749-751, :533, :844), incl. row-pitched inputs, missing noise / lengths and the one-network form"""
    gen = torch.Generator().manual_seed(61)
    B, T, ns, es, L = 5, 7, 6, 3, 1000
    z, c = torch.randn(B, T, ns, generator=gen).cuda(), torch.randn(B, es, generator=gen).cuda()
    ref = torch.cat([z, c.unsqueeze(1).expand(B, T, es)], 2).transpose(0, 1).contiguous()
    assert torch.equal(K.build_zc(z, c), ref)
    wide = torch.randn(B, 3, L, generator=gen).cuda()
    real, nr = wide[:, 1], torch.randn(B, L, generator=gen).cuda() * 0.01          # a row-pitched view
    fake, nf = torch.randn(4, L, generator=gen).cuda(), torch.randn(4, L, generator=gen).cuda() * 0.01
    la, lb = torch.tensor([1000, 3, 999, 512, 1]).cuda(), torch.tensor([1000, 1000, 77, 640]).cuda()
    cb = torch.randn(4, es, generator=gen).cuda()
    prods = [2, 4, 8, 16, 32, 64]
    x, lens, c2 = K.critic_batch(real, nr, fake, nf, la, lb, prods, c, cb)
    assert torch.equal(x, torch.cat([real + nr, fake + nf], 0))
    ln = torch.cat([la, lb])
    assert torch.equal(lens, torch.stack([(ln + p - 1) // p for p in prods]))
    assert torch.equal(c2, torch.cat([c, cb], 0))
    x, lens, c2 = K.critic_batch(fake, None, None, None, None, None, prods[:3])
    assert torch.equal(x, fake) and c2 is None
    assert torch.equal(lens, torch.stack([torch.full((4,), (L + p - 1) // p, dtype=torch.long) for p in prods[:3]]).cuda())
    odd = torch.randn(3, 1001, generator=gen).cuda()                                 # rows that are not 16-byte multiples
    x, lens, _ = K.critic_batch(odd[:, 1:], odd[:, :1000])
    assert torch.equal(x, odd[:, 1:] + odd[:, :1000]) and lens is None


def test_deferred_split_k_weight_gradient_is_bitwise_the_same(K):
    """a split-K weight-gradient product inside a deferral scope with defer=True hands its second stage to the scope's one
    launch (nothing written before the flush); the result equals the stand-alone product bit for bit, also when it
    accumulates (beta = 1); without defer=True, or with an epilogue, the product is complete at once inside the scope"""
    gen = torch.Generator().manual_seed(62)
    M, N, Kd = 512, 640, 8192
    a, b = torch.randn(Kd, M, generator=gen).cuda(), torch.randn(Kd, N, generator=gen).cuda()
    assert K.lib.ag_gemm_ws_numel(M, N, Kd, 0) > 0, 'shape must take the split-K path'
    base = torch.randn(M, N, generator=gen).cuda()
    ref0, ref1 = torch.empty(M, N).cuda(), base.clone()
    K.gemm(a, b, ref0, ta=True)
    K.gemm(a, b, ref1, ta=True, beta=1.0)
    got0, got1, now = torch.zeros(M, N).cuda(), base.clone(), torch.zeros(M, N).cuda()
    with K.deferred_reduces():
        K.gemm(a, b, got0, ta=True, defer=True)
        K.gemm(a, b, got1, ta=True, beta=1.0, defer=True)
        K.gemm(a, b, now, ta=True)
        torch.cuda.synchronize()
        assert not bool(got0.any()) and torch.equal(got1, base), 'deferred second stages ran before the flush'
        assert torch.equal(now, ref0), 'a product without defer=True must be complete inside the scope'
    torch.cuda.synchronize()
    assert torch.equal(got0, ref0) and torch.equal(got1, ref1)
    close(ref0, a.t().double().cpu() @ b.double().cpu(), rtol=1e-4, atol=1e-2)


def test_no_float_atomic_path_is_left(K):
    """round 4: a cross-workgroup sum without a bound workspace is an error, not an order-dependent float-atomic sum"""
    import ctypes as C
    x = torch.randn(4, 64, 4096).cuda()
    db = torch.zeros(64).cuda()
    K.lib.ag_bind_workspace(None, 0)
    rc = K.lib.ag_channel_sum(C.c_void_p(x.data_ptr()), x.stride(0), x.stride(1), C.c_void_p(db.data_ptr()), 4, 64, 4096, 1,
                              C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == -1 and b'workspace' in K.lib.ag_last_error()
    K.channel_sum(x, db)
    close(db, x.double().sum((0, 2)).cpu(), rtol=1e-5, atol=1e-3)
