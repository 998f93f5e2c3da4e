"""how exact CAN a gradient of the full-size critic iteration be in fp32?  The critic iteration of bench.py's workload (B = 64)
on the HIP path, on the fp32 oracle and on the oracle in float64; per parameter tensor: max-abs and relative-L2 error of the
two fp32 computations against float64, in units of the tensor's largest entry / norm (python tools/diag_grad.py > gpurun_out/...)"""
import copy
import os
import sys
import torch
sys.path.insert(0, '.')
import audiogan_amd as A
import bench
from audiogan_amd import train, optim
from oracle import audiogan_oracle as O

torch.set_num_threads(min(32, os.cpu_count() or 1))
torch.manual_seed(0)
go = O.Generator(frame_size=256, embed_size=100, noise_size=100, state_size=1024)
do = O.Discriminator(state_size=1024, embed_size=100)
g, d = A.Generator(frame_size=256, embed_size=100, noise_size=100, state_size=1024), A.Discriminator(state_size=1024, embed_size=100)
g.load_state_dict(go.state_dict()); d.load_state_dict(do.state_dict())
g.cuda(); d.cuda()
B = 64
b = bench.synthetic_batch(B, torch.device('cpu'), 1000)
cu = {k: t.cuda() for k, t in b.items()}
stop = torch.zeros(B, bench.L // bench.FRAME, dtype=torch.long)


def oracle_grads(dtype):
    g2 = O.Generator(frame_size=256, embed_size=100, noise_size=100, state_size=1024)
    d2 = O.Discriminator(state_size=1024, embed_size=100)
    g2.load_state_dict(go.state_dict()); d2.load_state_dict(do.state_dict())
    g2, d2 = g2.to(dtype), d2.to(dtype)
    torch.set_default_dtype(dtype)          # (the oracle's forward creates its initial states with the default dtype)
    bb = {k: (t.to(dtype) if t.is_floating_point() else t) for k, t in b.items()}
    with torch.no_grad():
        fake, _, _, fl = g2(z=bb['z'], c=bb['c'], stop=stop)
        fake = fake + bb['noise_fake']
    cd, _, _, nd = d2(bb['real'] + bb['noise_real'], bb['real_len'], bb['c'])
    cg, _, _, ng = d2(fake, fl, bb['c'])
    bce = O.binary_cross_entropy_with_logits_per_sample
    loss = (bce(cd, torch.full_like(cd, 0.9), weight=O.length_mask(cd.size(), nd)) / nd.to(dtype)).mean() + \
        (bce(cg, torch.zeros_like(cg), weight=O.length_mask(cg.size(), ng)) / ng.to(dtype)).mean()
    loss.backward()
    torch.set_default_dtype(torch.float32)
    return {k: p.grad.double() for k, p in d2.named_parameters()}, float(loss)


o64, l64 = oracle_grads(torch.float64)
o32, l32 = oracle_grads(torch.float32)
opt_d = optim.make_optimizer(list(d.parameters()), 'adam', 1e-4)
l = train.d_backward(g, d, opt_d, cu['real'], cu['real_len'], cu['c'], cu['z'], cu['noise_real'], cu['noise_fake'])
torch.cuda.synchronize()
print('loss  hip %.8f  o32 %.8f  o64 %.8f' % (float(l), l32, l64))
print('%-44s %10s %10s | %10s %10s | where the HIP error peaks' % ('tensor', 'hip max', 'o32 max', 'hip l2', 'o32 l2'))
for k, p in d.named_parameters():
    if k.split('.')[-1].startswith('bias') and k.endswith('_v'):
        continue
    r = o64[k]
    gh, g3 = p.grad.detach().cpu().double(), o32[k]
    mx, nr = float(r.abs().max()), float(r.norm())
    eh, e3 = (gh - r), (g3 - r)
    idx = int(eh.abs().flatten().argmax())
    pos = tuple(int(v) for v in torch.unravel_index(torch.tensor(idx), eh.shape)) if eh.dim() else ()
    print('%-44s %10.2e %10.2e | %10.2e %10.2e | %s' % (k, float(eh.abs().max()) / mx, float(e3.abs().max()) / mx,
                                                      float(eh.norm()) / nr, float(e3.norm()) / nr, pos))


# ---- the generator iteration (boundary-seeking loss through the frozen critic) ----------------------------------------------
def oracle_g_grads(dtype):
    g2 = O.Generator(frame_size=256, embed_size=100, noise_size=100, state_size=1024)
    d2 = O.Discriminator(state_size=1024, embed_size=100)
    g2.load_state_dict(go.state_dict()); d2.load_state_dict(do.state_dict())
    g2, d2 = g2.to(dtype), d2.to(dtype)
    for p in d2.parameters():
        p.requires_grad_(False)
    torch.set_default_dtype(dtype)
    bb = {k: (t.to(dtype) if t.is_floating_point() else t) for k, t in b.items()}
    fake, _, _, fl = g2(z=bb['z'], c=bb['c'], stop=stop)
    cg, _, _, ng = d2(fake + bb['noise_fake'], fl, bb['c'])
    bce = O.binary_cross_entropy_with_logits_per_sample
    loss = (bce(cg, torch.full_like(cg, 0.5), weight=O.length_mask(cg.size(), ng)) / ng.to(dtype)).mean()
    loss.backward()
    torch.set_default_dtype(torch.float32)
    return {k: p.grad.double() for k, p in g2.named_parameters() if p.grad is not None}, float(loss)


d.load_state_dict(do.state_dict())
o64, l64 = oracle_g_grads(torch.float64)
o32, l32 = oracle_g_grads(torch.float32)
opt_g = optim.make_optimizer(list(g.parameters()), 'adam', 1e-4)
l = train.g_backward(g, d, opt_g, cu['c'], cu['z'], cu['noise_fake'])
torch.cuda.synchronize()
print()
print('GENERATOR iteration: loss  hip %.8f  o32 %.8f  o64 %.8f' % (float(l), l32, l64))
print('%-44s %10s %10s | %10s %10s | %10s' % ('tensor', 'hip max', 'o32 max', 'hip l2', 'o32 l2', '|grad|'))
for k, p in g.named_parameters():
    if (k.split('.')[-1].startswith('bias') and k.endswith('_v')) or k not in o64 or p.grad is None:
        continue
    r = o64[k]
    gh, g3 = p.grad.detach().cpu().double(), o32[k]
    mx, nr = float(r.abs().max()), float(r.norm())
    eh, e3 = (gh - r), (g3 - r)
    print('%-44s %10.2e %10.2e | %10.2e %10.2e | %10.2e' % (k, float(eh.abs().max()) / mx, float(e3.abs().max()) / mx,
                                                          float(eh.norm()) / nr, float(e3.norm()) / nr, nr))
