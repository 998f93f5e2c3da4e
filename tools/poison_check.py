"""fill the caching allocator's free blocks with a poison value, then run a network forward + backward: any kernel that
reads memory it (or a predecessor) never wrote shows up as NaN / huge values (python tools/poison_check.py [bf16])"""
import sys
import torch
sys.path.insert(0, '.')
import audiogan_amd as A
from audiogan_amd import kernels as K
from oracle import audiogan_oracle as O
from tests.test_bf16 import _small_models, rel_l2

prec = sys.argv[1] if len(sys.argv) > 1 else 'f32'
K.set_precision(prec)

# every torch.empty / empty_like of a floating CUDA tensor comes back NaN-filled: a kernel that reads what nobody wrote
# turns the results NaN, whatever the allocator's history
_empty, _empty_like = torch.empty, torch.empty_like


def _nan_empty(*a, **k):
    t = _empty(*a, **k)
    if t.is_cuda and t.is_floating_point():
        t.fill_(float('nan'))
    return t


def _nan_empty_like(*a, **k):
    t = _empty_like(*a, **k)
    if t.is_cuda and t.is_floating_point():
        t.fill_(float('nan'))
    return t


torch.empty, torch.empty_like = _nan_empty, _nan_empty_like


def poison(val):
    blocks = [torch.full((64 << 20,), val, device='cuda') for _ in range(8)]      # 2 GB
    blocks += [torch.full((1 << 20,), val, device='cuda') for _ in range(256)]    # and many small blocks
    blocks += [torch.full((16 << 10,), val, device='cuda') for _ in range(2048)]
    torch.cuda.synchronize()
    del blocks


for val in (float('nan'), 3.0e4):
    go, do, g, d = _small_models(A)
    B, T, fs = 16, 16, 64
    gen = torch.Generator().manual_seed(32)
    z, c = torch.randn(B, T, 16, generator=gen), torch.randn(B, 16, generator=gen)
    lens = torch.randint(300, T * fs + 1, (B,), generator=gen)
    lens[0] = T * fs
    wl = torch.randn(B, T * fs // 16, generator=gen)
    stop = torch.zeros(B, T, dtype=torch.long)
    ctx = O.bf16_mode() if prec == 'bf16' else torch.enable_grad()
    with ctx:
        xo = go(z=z, c=c, stop=stop)[0]
        lo = do(xo, lens, c)[0]
        (lo * wl).sum().backward()
    zc, cc, lc, wc = z.cuda(), c.cuda(), lens.cuda(), wl.cuda()
    poison(val)
    x = g(z=zc, c=cc, stop='never')[0]
    l = d(x, lc, cc)[0]
    (l * wc).sum().backward()
    torch.cuda.synchronize()
    print('poison %r: x rel %.2e  logits rel %.2e' % (val, rel_l2(x, xo), rel_l2(l, lo)))
    for mod, ref in ((g, go), (d, do)):
        rp = dict(ref.named_parameters())
        for k, q in mod.named_parameters():
            if k.split('.')[-1].startswith('bias') and k.endswith('_v'):
                continue
            r = rp[k].grad if rp[k].grad is not None else torch.zeros_like(rp[k])
            e = float((q.grad.cpu() - r).abs().max()) / max(float(r.abs().max()), 1e-12)
            bad = (not torch.isfinite(q.grad).all()) or e > 0.2
            if bad:
                print('   BAD %-45s max err / max|ref| = %.3e   finite %s' % (k, e, bool(torch.isfinite(q.grad).all())))

# the canonical train step on the same small models
from audiogan_amd import optim, train
go, do, g, d = _small_models(A)
B, T, fs = 8, 16, 64
gen = torch.Generator().manual_seed(33)
real = (torch.rand(B, T * fs, generator=gen) * 2 - 1).cuda()
rl = torch.full((B,), T * fs, dtype=torch.long).cuda()
c, z = torch.randn(B, 16, generator=gen).cuda(), torch.randn(B, T, 16, generator=gen).cuda()
nr, nf = (torch.randn(B, T * fs, generator=gen) * 0.01).cuda(), (torch.randn(B, T * fs, generator=gen) * 0.01).cuda()
opt_d, opt_g = optim.make_optimizer(list(d.parameters()), 'adam', 1e-4), optim.make_optimizer(list(g.parameters()), 'adam', 1e-4)
for it in range(2):
    l = train.gd_step(g, d, opt_g, opt_d, real, rl, c, z, nr, nf, check=True)
    print('train step %d losses %.6f %.6f' % (it, float(l[0]), float(l[1])))
print('params finite:', all(bool(torch.isfinite(p).all()) for p in list(g.parameters()) + list(d.parameters())))
print('done')
