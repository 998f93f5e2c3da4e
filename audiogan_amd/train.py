"""The canonical G+D step: audiogan.py:706-788 (critic) and :816-921
(generator) minus the out-of-scope extras.  All stochastic inputs are arguments so the same
numbers can be fed to the CPU oracle."""
import torch

from .common import frozen
from .losses import masked_bce_mean

_SIDE = {}


def gd_step(g, d, opt_g, opt_d, real, real_len, c, z, noise_real, noise_fake, dgradclip=1.0, ggradclip=0.1,
            overlap=True, hook_d=None, hook_g=None, check=False):
    """critic iteration + generator iteration (the canonical step).  ``overlap``: the generator iteration's G forward
    depends on neither D nor the critic iteration, so it is enqueued on a second stream beside the critic iteration
    (one fork, one join; a parallel branch when the step is captured into a hipGraph): its latency-bound frame loop
    then shares the GPU with the critic's equally latency-bound biLSTM chains.  Same work, same results: G's
    weight-normed weights are materialised on the main stream BEFORE the fork, so neither branch rewrites a buffer
    the other one reads.  Returns (loss_d, loss_g)."""
    pre = None
    if overlap and real.is_cuda:
        dev = real.device
        side = _SIDE.get(dev)
        if side is None:
            side = _SIDE[dev] = torch.cuda.Stream(device=dev)
        g.prepare_weights()
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            pre = g(z=z, c=c, stop='never')
    loss_d = d_step(g, d, opt_d, real, real_len, c, z, noise_real, noise_fake, dgradclip, grad_hook=hook_d,
                    check=check)[0]
    if pre is not None:
        torch.cuda.current_stream().wait_stream(side)
    loss_g = g_step(g, d, opt_g, c, z, noise_fake, ggradclip, grad_hook=hook_g, check=check, pre=pre)[0]
    return loss_d, loss_g


def d_step(g, d, opt_d, real, real_len, c, z, noise_real, noise_fake, dgradclip=1.0, stop='never',
           grad_hook=None, check=False, batch_real_fake=True):
    """One critic iteration: G forward without grad, D(real) vs 0.9, D(fake) vs 0
    (audiogan.py:723-728, 739-740, 748-751, 761-766, 780-788)."""
    with torch.no_grad():
        fake, _, _, fake_len = g(z=z, c=c, stop=stop)
        fake = fake + noise_fake
    B = real.size(0)
    if batch_real_fake and fake.size(1) == real.size(1):
        # D has no cross-sample op, so D(cat(real, fake)) == cat(D(real), D(fake)): one pass over
        # 2B clips halves the number of strictly sequential biLSTM steps of the critic iteration
        cls, _, _, nf = d(torch.cat([real + noise_real, fake], 0), torch.cat([real_len.to(fake_len.device), fake_len], 0),
                          torch.cat([c, c], 0))
        cls_d, cls_g, nf_d, nf_g = cls[:B], cls[B:], nf[:B], nf[B:]
    else:
        cls_d, _, _, nf_d = d(real + noise_real, real_len, c)
        cls_g, _, _, nf_g = d(fake, fake_len, c)
    loss_d, _ = masked_bce_mean(cls_d, 0.9, nf_d.contiguous())
    loss_g, _ = masked_bce_mean(cls_g, 0.0, nf_g.contiguous())
    loss = loss_d + loss_g
    opt_d.zero_grad()
    loss.backward()
    scale = grad_hook() if grad_hook is not None else 1.0
    opt_d.step(clip_norm=dgradclip, grad_scale=scale, check=check)
    return loss.detach(), cls_d.detach(), cls_g.detach()


def g_step(g, d, opt_g, c, z, noise_fake, ggradclip=0.1, g_optim='boundary_seeking', stop='never',
           grad_hook=None, check=False, pre=None):
    """One generator iteration: loss through D into G (audiogan.py:841-845, 857-864, 897,
    902-903, 909-921).  ``pre``: the result of ``g(z=z, c=c, stop=stop)`` when the caller has already run the
    generator's forward (it does not involve D, so it may be enqueued on a second stream beside the critic
    iteration - bench.py does that)."""
    with frozen(d):
        fake, _, _, fake_len = pre if pre is not None else g(z=z, c=c, stop=stop)
        cls_g, _, _, nf_g = d(fake + noise_fake, fake_len, c)
        tgt = 0.5 if g_optim == 'boundary_seeking' else 0.0
        loss, _ = masked_bce_mean(cls_g, tgt, nf_g)
        opt_g.zero_grad()
        loss.backward()
    scale = grad_hook() if grad_hook is not None else 1.0
    opt_g.step(clip_norm=ggradclip, grad_scale=scale, check=check)
    return loss.detach(), fake.detach(), cls_g.detach()


# --------------------------------------------------------------------------------------
# The same two iterations cut at the gradient all-reduce, so that each phase can be captured
# into its own hipGraph while the RCCL collective between them stays an ordinary stream op:
#   d_backward -> [all-reduce D grads] -> opt_d.step -> g_backward -> [all-reduce G grads] -> opt_g.step
# --------------------------------------------------------------------------------------
def d_backward(g, d, opt_d, real, real_len, c, z, noise_real, noise_fake, stop='never',
               batch_real_fake=True):
    """forward + backward of the critic iteration; gradients are left in ``.grad``"""
    with torch.no_grad():
        fake, _, _, fake_len = g(z=z, c=c, stop=stop)
        fake = fake + noise_fake
    B = real.size(0)
    if batch_real_fake and fake.size(1) == real.size(1):
        cls, _, _, nf = d(torch.cat([real + noise_real, fake], 0),
                          torch.cat([real_len.to(fake_len.device), fake_len], 0), torch.cat([c, c], 0))
        cls_d, cls_g, nf_d, nf_g = cls[:B], cls[B:], nf[:B], nf[B:]
    else:
        cls_d, _, _, nf_d = d(real + noise_real, real_len, c)
        cls_g, _, _, nf_g = d(fake, fake_len, c)
    loss_d, _ = masked_bce_mean(cls_d, 0.9, nf_d.contiguous())
    loss_g, _ = masked_bce_mean(cls_g, 0.0, nf_g.contiguous())
    loss = loss_d + loss_g
    opt_d.zero_grad()
    loss.backward()
    return loss.detach()


def g_backward(g, d, opt_g, c, z, noise_fake, g_optim='boundary_seeking', stop='never', pre=None):
    """forward + backward of the generator iteration (through D, whose weights get no gradient);
    ``pre``: the generator's forward result when the caller already ran it (see g_step)"""
    with frozen(d):
        fake, _, _, fake_len = pre if pre is not None else g(z=z, c=c, stop=stop)
        cls_g, _, _, nf_g = d(fake + noise_fake, fake_len, c)
        loss, _ = masked_bce_mean(cls_g, 0.5 if g_optim == 'boundary_seeking' else 0.0, nf_g)
        opt_g.zero_grad()
        loss.backward()
    return loss.detach()


def d_backward_early(g, d, opt_d, real, real_len, c, z, noise_real, noise_fake, keep, stop='never'):
    """critic forward + backward DOWN TO the conv features: afterwards the gradients of the heads and the
    biLSTM (93 % of D's parameters, ``d.early_params()``) are final, so their all-reduce can run while
    ``d_backward_late`` pushes the gradient through the conv stack.  ``keep`` carries the cut tensors."""
    with torch.no_grad():
        fake, _, _, fake_len = g(z=z, c=c, stop=stop)
        fake = fake + noise_fake
    B = real.size(0)
    x = torch.cat([real + noise_real, fake], 0)
    lens = torch.cat([real_len.to(fake_len.device), fake_len], 0)
    acts, lens_list = d.features(x, lens)
    a_cut = acts[-1].detach().requires_grad_(True)
    nf = lens_list[-1]
    cls = d.classify(a_cut, nf, torch.cat([c, c], 0))
    loss_d, _ = masked_bce_mean(cls[:B], 0.9, nf[:B].contiguous())
    loss_g, _ = masked_bce_mean(cls[B:], 0.0, nf[B:].contiguous())
    loss = loss_d + loss_g
    opt_d.zero_grad()
    loss.backward()
    keep['acts'], keep['a_cut'] = acts, a_cut
    return loss.detach()


def d_backward_late(keep):
    """the conv stack's backward (weight gradients of D's cnn) from the cut gradient"""
    keep['acts'][-1].backward(keep['a_cut'].grad)


def g_backward_early(g, d, opt_g, c, z, noise_fake, keep, g_optim='boundary_seeking', stop='never', pre=None):
    """generator iteration, forward + backward DOWN TO the frames the recurrent front produced: afterwards the
    conv trunk's gradients (``g.early_params()``) are final and their all-reduce can run while ``g_backward_late``
    runs the front's frame-by-frame backward.  ``pre``: result of ``g(z=z, c=c, stop=stop, cut=keep)`` when the
    caller already ran the generator's forward (with the same ``keep``)."""
    with frozen(d):
        fake, _, _, fake_len = pre if pre is not None else g(z=z, c=c, stop=stop, cut=keep)
        cls_g, _, _, nf_g = d(fake + noise_fake, fake_len, c)
        loss, _ = masked_bce_mean(cls_g, 0.5 if g_optim == 'boundary_seeking' else 0.0, nf_g)
        opt_g.zero_grad()
        loss.backward()
    return loss.detach()


def g_backward_late(keep):
    """the recurrent front's backward from the cut gradient (weight gradients of rnn / proj / stopper)"""
    keep['x'].backward(keep['x_cut'].grad)
