"""-m gpu: seeded random shapes through the conv engine and the weight-gradient kernel, fp32 and bf16 mode, against torch
convolutions in float64 (bf16 mode: on operands rounded to bf16, which the kernels' contract defines).  The fixed layer
tables in test_gpu_kernels.py / test_bf16.py pin the C2 shapes; this one walks the dispatch space around them - chunk
counts of the register-pipelined staging, tiles at the signal edges, strides that are not powers of two, lengths that are
not multiples of 4 (generic staging / element-wise epilogue), strided views, every epilogue option.
Tolerance: 2e-4 of the largest reference value (fp32 summation order only)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def K():
    assert torch.cuda.is_available(), 'gpu tests need a GPU'
    import audiogan_amd.kernels as K_
    return K_


def _rnd(t):
    return t.to(torch.bfloat16).to(torch.float32)


def _cases(n, seed):
    gen = torch.Generator().manual_seed(seed)

    def ri(lo, hi):
        return int(torch.randint(lo, hi + 1, (1,), generator=gen))
    out = []
    while len(out) < n:
        kind = 'conv' if ri(0, 1) == 0 else 'convT'
        s = [1, 2, 2, 3, 4, 4, 8][ri(0, 6)]
        k = ri(max(1, s - 1), 2 * s + 3)
        p = ri(0, k - 1) if kind == 'conv' else ri(0, min(k - 1, 5))
        cin = [1, 3, 16, 17, 24, 40, 64, 96, 130][ri(0, 8)]
        cout = [1, 5, 16, 32, 33, 70, 128, 200][ri(0, 7)]
        lin = [7, 64, 100, 256, 333, 512, 1000, 1024][ri(0, 7)]
        B = ri(1, 3)
        lout = (lin + 2 * p - k) // s + 1 if kind == 'conv' else (lin - 1) * s - 2 * p + k
        if lout < 1 or (kind == 'conv' and lin + 2 * p < k):
            continue
        opts = dict(bias=ri(0, 1), res=ri(0, 1), lens=ri(0, 1), act=ri(0, 2), acc=ri(0, 1), view=ri(0, 1))
        if opts['act'] == 2:
            opts['res'] = 1          # act 2 = ACT_LEAKY_GATE: `res` is the saved activation that gates the result
        out.append((kind, cin, cout, k, s, p, lin, B, opts))
    return out


@pytest.mark.parametrize('prec', ['f32', 'bf16'])
def test_conv_engine_and_wgrad_random_shapes(K, prec):
    from audiogan_amd import ops
    old = K.set_precision(prec)
    try:
        for ci, (kind, cin, cout, k, s, p, lin, B, o) in enumerate(_cases(40, 77)):
            tag = (ci, kind, cin, cout, k, s, p, lin, B, o)
            gen = torch.Generator().manual_seed(1000 + ci)
            w = torch.randn((cout, cin, k) if kind == 'conv' else (cin, cout, k), generator=gen) / (cin * k) ** 0.5
            x = torch.randn(B, cin, lin, generator=gen)
            spec = ops.ConvSpec(kind, cin, cout, k, s, p)
            lout = spec.out_len(lin)
            dy = torch.randn(B, cout, lout, generator=gen)
            bias = torch.randn(cout, generator=gen) if o['bias'] else None
            res = torch.randn(B, cout, lout, generator=gen) if o['res'] else None
            lens = torch.randint(1, lout + 1, (B,), generator=gen) if o['lens'] else None
            y0 = torch.randn(B, cout, lout, generator=gen)
            r = _rnd if prec == 'bf16' else (lambda t: t)
            xr, wr = r(x).double().requires_grad_(True), r(w).double().requires_grad_(True)
            lin_out = F.conv1d(xr, wr, None, s, p) if kind == 'conv' else F.conv_transpose1d(xr, wr, None, s, p)
            v = lin_out
            if bias is not None:
                v = v + bias.double().view(1, -1, 1)
            if res is not None and o['act'] == 2:
                v = v * torch.where(res > 0, torch.ones_like(res), torch.full_like(res, K.LEAKY_SLOPE)).double()
            elif res is not None:
                v = v + res.double()
            if o['act'] == 1:
                v = F.leaky_relu(v, K.LEAKY_SLOPE)
            if lens is not None:
                v = v * (torch.arange(lout).view(1, 1, -1) < lens.view(B, 1, 1)).double()
            ref = v + y0.double() if o['acc'] else v
            d0, d1, _ = w.shape
            prep = ops.Prepared(w=w.cuda(), wpa=torch.zeros(K.wpa_numel(d0, d1, k)).cuda(),
                                wpb=torch.zeros(K.wpb_numel(d0, d1, k, s)).cuda(), pad=p)
            K.prep_conv_weight(prep.w, prep.wpa, prep.wpb, s, p)
            xg = x.cuda()
            if o['view']:            # a channel slice of a wider slab with a padded row pitch: strided batch / channel strides
                slab = torch.zeros(B, cin + 2, lin + 12).cuda()
                slab[:, 1:cin + 1, 4:lin + 4] = xg
                xg = slab[:, 1:cin + 1, 4:lin + 4]
            y = y0.cuda().clone()
            K.conv_engine(xg, prep.wpa if kind == 'conv' else prep.wpb, y, k, s, p, 0 if kind == 'conv' else 1,
                          bias=bias.cuda() if bias is not None else None, res=res.cuda() if res is not None else None,
                          lens=lens.cuda() if lens is not None else None, act=(K.ACT_NONE, K.ACT_LEAKY, K.ACT_LEAKY_GATE)[o['act']],
                          accumulate=bool(o['acc']), wp_pad=p)
            scale = max(1.0, float(ref.detach().abs().max()))
            err = float((y.cpu().double() - ref.detach()).abs().max())
            assert err <= 2e-4 * scale, ('forward', tag, err)
            # backward-data and backward-weight of the plain convolution
            dyr = r(dy).double()
            lin_out.backward(dyr)
            dx = torch.full((B, cin, lin), float('nan')).cuda()
            ops.conv_bwd_data(spec, prep, dy.cuda(), dx)
            err = float((dx.cpu().double() - xr.grad).abs().max())
            assert err <= 2e-4 * max(1.0, float(xr.grad.abs().max())), ('backward-data', tag, err)
            dw = torch.zeros_like(w).cuda()
            ops.conv_wgrad(spec, xg, dy.cuda(), dw, None)
            err = float((dw.cpu().double() - wr.grad).abs().max())
            assert err <= 2e-4 * max(1.0, float(wr.grad.abs().max())), ('backward-weight', tag, err)
    finally:
        K.set_precision(old)
