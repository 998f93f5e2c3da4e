"""The canonical G+D step: audiogan.py:706-788 (critic) and :816-921
(generator) minus the out-of-scope extras.  All stochastic inputs are arguments so the same
numbers can be fed to the CPU oracle."""
import torch

from . import kernels as K
from .common import frozen, network_backward
from .ops import CriticInputFn
from .extras import adversarial_movement_d, adversarially_sample_z, calc_dists, feature_penalty, feature_penalty_fused  # noqa: F401
from .losses import length_mask, masked_bce_mean, only_stopper_trains, real_fake_targets, stopper_surrogate_loss

_SIDE = {}
_ONE = {}


def _backward(t, grad=None, **kw):
    """``t.backward(grad)`` inside a network-level scope (common.network_backward): the second stages of every deferred
    reduction of the whole backward run as ONE launch and the weight-norm backward of every block as ONE launch.  The root
    gradient of a scalar loss is a cached ones tensor (autograd would fill a fresh one: a launch per backward)."""
    if grad is None:
        key = (t.device, t.dtype)
        grad = _ONE.get(key)
        if grad is None:
            grad = _ONE[key] = torch.ones((), device=t.device, dtype=t.dtype)
        assert t.dim() == 0, 'a root gradient is needed for a non-scalar tensor'
    with network_backward():
        t.backward(grad, **kw)


def _critic_inputs(d, real, noise_real, fake, noise_fake, real_len, fake_len, c):
    """the critic iteration's minibatch [real + noise ; fake + noise], the rows' lengths after every conv layer and the
    conditioning rows [c ; c] - ONE launch (kernels.critic_batch) instead of two adds, three cats and the length
    arithmetic; plain torch when the tensors are not on the GPU (host-logic tests)"""
    if real.is_cuda and hasattr(d, 'stride_products') and real.dtype == torch.float32:
        return K.critic_batch(real, noise_real, fake, noise_fake, real_len.to(real.device).long().contiguous(),
                              fake_len.contiguous(), d.stride_products(), c.contiguous(), c.contiguous())
    x = torch.cat([real + noise_real, fake + noise_fake if noise_fake is not None else fake], 0)
    return x, None, torch.cat([c, c], 0)


def _noisy(d, fake, noise, fake_len):
    """generator iteration: (fake + noise, lengths table or None), the gradient passing through to ``fake``"""
    if fake.is_cuda and hasattr(d, 'stride_products') and fake.dtype == torch.float32:
        return CriticInputFn.apply(fake, noise, fake_len.contiguous(), d.stride_products())
    return fake + noise, None


def gd_step(g, d, opt_g, opt_d, real, real_len, c, z, noise_real, noise_fake, dgradclip=1.0, ggradclip=0.1,
            overlap=True, hook_d=None, hook_g=None, check=False):
    """critic iteration + generator iteration (the canonical step).  ``overlap``: the generator iteration's G forward
    depends on neither D nor the critic iteration, so it is enqueued on a second stream beside the critic iteration
    (one fork, one join; a parallel branch when the step is captured into a hipGraph): its latency-bound frame loop
    then shares the GPU with the critic's equally latency-bound biLSTM chains.  Same work, same results: G's
    weight-normed weights are materialised on the main stream BEFORE the fork, so neither branch rewrites a buffer
    the other one reads.  Returns (loss_d, loss_g)."""
    pre = None
    if overlap and real.is_cuda and g.front_is_persistent(z.size(0), real.device):
        # a persistent launch occupies every CU and must be the only one in flight: nothing to gain from a second stream
        # (and two persistent launches racing for the CUs would stall each other)
        overlap = False
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


def _single_gpu(opt):
    """no gradient bucket on the optimiser = no collective can run beside this network's backward: the recurrent front's
    backward may then be a persistent launch (kernels.PERSIST_FRONT_BWD)"""
    return getattr(opt, 'bucket', None) is None


def d_step(g, d, opt_d, real, real_len, c, z, noise_real, noise_fake, dgradclip=1.0, stop='never',
           grad_hook=None, check=False, batch_real_fake=True):
    """One critic iteration: G forward without grad, D(real) vs 0.9, D(fake) vs 0
    (audiogan.py:723-728, 739-740, 748-751, 761-766, 780-788)."""
    with torch.no_grad():
        fake, _, _, fake_len = g(z=z, c=c, stop=stop)
    B = real.size(0)
    if batch_real_fake and fake.size(1) == real.size(1):
        # D has no cross-sample op, so D(cat(real, fake)) == cat(D(real), D(fake)): one pass over
        # 2B clips halves the number of strictly sequential biLSTM steps of the critic iteration
        x2, tab, c2 = _critic_inputs(d, real, noise_real, fake, noise_fake, real_len, fake_len, c)
        cls, _, _, nf = d(x2, torch.cat([real_len.to(fake_len.device), fake_len], 0) if tab is None else None, c2,
                          lens_all=tab)
        cls_d, cls_g = cls[:B], cls[B:]
        # mean over the real clips + mean over the fake clips = (1 / B) * the sum over all 2B rows, targets 0.9 / 0 per row:
        # ONE loss launch forward and backward (no slices for autograd to reassemble)
        loss, _ = masked_bce_mean(cls, real_fake_targets(B, cls.device), nf, scale=1.0 / B)
    else:
        cls_d, _, _, nf_d = d(real + noise_real, real_len, c)
        cls_g, _, _, nf_g = d(fake + noise_fake, fake_len, c)
        loss_d, _ = masked_bce_mean(cls_d, 0.9, nf_d.contiguous())
        loss_g, _ = masked_bce_mean(cls_g, 0.0, nf_g.contiguous())
        loss = loss_d + loss_g
    opt_d.zero_grad()
    _backward(loss)
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
        noisy, tab = _noisy(d, fake, noise_fake, fake_len)
        cls_g, _, _, nf_g = d(noisy, fake_len, c, lens_all=tab)
        tgt = 0.5 if g_optim == 'boundary_seeking' else 0.0
        loss, _ = masked_bce_mean(cls_g, tgt, nf_g)
        opt_g.zero_grad()
        with K.front_bwd_persist(_single_gpu(opt_g)):
            _backward(loss)
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
    B = real.size(0)
    if batch_real_fake and fake.size(1) == real.size(1):
        x2, tab, c2 = _critic_inputs(d, real, noise_real, fake, noise_fake, real_len, fake_len, c)
        cls, _, _, nf = d(x2, torch.cat([real_len.to(fake_len.device), fake_len], 0) if tab is None else None, c2,
                          lens_all=tab)
        loss, _ = masked_bce_mean(cls, real_fake_targets(B, cls.device), nf, scale=1.0 / B)     # (see d_step)
    else:
        cls_d, _, _, nf_d = d(real + noise_real, real_len, c)
        cls_g, _, _, nf_g = d(fake + noise_fake, fake_len, c)
        loss_d, _ = masked_bce_mean(cls_d, 0.9, nf_d.contiguous())
        loss_g, _ = masked_bce_mean(cls_g, 0.0, nf_g.contiguous())
        loss = loss_d + loss_g
    opt_d.zero_grad()
    _backward(loss)
    return loss.detach()


def g_backward(g, d, opt_g, c, z, noise_fake, g_optim='boundary_seeking', stop='never', pre=None):
    """forward + backward of the generator iteration (through D, whose weights get no gradient);
    ``pre``: the generator's forward result when the caller already ran it (see g_step)"""
    with frozen(d):
        fake, _, _, fake_len = pre if pre is not None else g(z=z, c=c, stop=stop)
        noisy, tab = _noisy(d, fake, noise_fake, fake_len)
        cls_g, _, _, nf_g = d(noisy, fake_len, c, lens_all=tab)
        loss, _ = masked_bce_mean(cls_g, 0.5 if g_optim == 'boundary_seeking' else 0.0, nf_g)
        opt_g.zero_grad()
        with K.front_bwd_persist(_single_gpu(opt_g)):
            _backward(loss)
    return loss.detach()


def d_backward_early(g, d, opt_d, real, real_len, c, z, noise_real, noise_fake, keep, stop='never'):
    """critic forward + backward DOWN TO the conv features: afterwards the gradients of the heads and the
    biLSTM (93 % of D's parameters, ``d.early_params()``) are final, so their all-reduce can run while
    ``d_backward_late`` pushes the gradient through the conv stack.  ``keep`` carries the cut tensors."""
    with torch.no_grad():
        fake, _, _, fake_len = g(z=z, c=c, stop=stop)
    B = real.size(0)
    x, tab, c2 = _critic_inputs(d, real, noise_real, fake, noise_fake, real_len, fake_len, c)
    lens = torch.cat([real_len.to(fake_len.device), fake_len], 0) if tab is None else None
    acts, lens_list = d.features(x, lens, tab)
    a_cut = acts[-1].detach().requires_grad_(True)
    nf = lens_list[-1]
    cls = d.classify(a_cut, nf, c2)
    loss, _ = masked_bce_mean(cls, real_fake_targets(B, cls.device), nf, scale=1.0 / B)     # (see d_step)
    opt_d.zero_grad()
    _backward(loss)
    keep['acts'], keep['a_cut'] = acts, a_cut
    return loss.detach()


def d_backward_late(keep):
    """the conv stack's backward (weight gradients of D's cnn) from the cut gradient"""
    _backward(keep['acts'][-1], keep['a_cut'].grad)


def g_backward_early(g, d, opt_g, c, z, noise_fake, keep, g_optim='boundary_seeking', stop='never', pre=None):
    """generator iteration, forward + backward DOWN TO the frames the recurrent front produced: afterwards the
    conv trunk's gradients (``g.early_params()``) are final and their all-reduce can run while ``g_backward_late``
    runs the front's frame-by-frame backward.  ``pre``: result of ``g(z=z, c=c, stop=stop, cut=keep)`` when the
    caller already ran the generator's forward (with the same ``keep``)."""
    with frozen(d):
        fake, _, _, fake_len = pre if pre is not None else g(z=z, c=c, stop=stop, cut=keep)
        noisy, tab = _noisy(d, fake, noise_fake, fake_len)
        cls_g, _, _, nf_g = d(noisy, fake_len, c, lens_all=tab)
        loss, _ = masked_bce_mean(cls_g, 0.5 if g_optim == 'boundary_seeking' else 0.0, nf_g)
        opt_g.zero_grad()
        _backward(loss)
    return loss.detach()


def g_backward_late(keep, persist=False):
    """the recurrent front's backward from the cut gradient (weight gradients of rnn / proj / stopper).  ``persist``: may
    this backward be ONE persistent launch?  Default False: the phase-split form exists to run beside the asynchronous
    all-reduce of the trunk's gradients, and a collective's kernels hold compute units for as long as their peers need
    while a persistent launch wants all of its workgroups resident - pass True only when no collective is in flight
    (``GraphedStep`` does when it has no gradient bucket)."""
    with K.front_bwd_persist(bool(persist)):
        _backward(keep['x'], keep['x_cut'].grad)


# --------------------------------------------------------------------------------------
# The reference's CURRENT iterations: the classic step plus the FGSM-style passes (audiogan.py:99-150), the
# feature-matching penalty (:847-855) and the REINFORCE update of the stop head (:444-460, :866-908).  Same arguments
# and results as oracle.audiogan_oracle.d_step_full / g_step_full (every random quantity is an argument).
# --------------------------------------------------------------------------------------
def _acc(cls, nframes, positive, host=True):
    w = length_mask(cls.size(), nframes)
    hit = (cls > 0) if positive else (cls < 0)
    a = (hit.float() * w).sum() / w.sum()
    return float(a) if host else a


def d_step_full(g, d, e_g, e_d, opt_d, dis_iter, real, real_len, cs, cl, cs2, cl2, z, noise_real, noise_fake,
                dgradclip=1.0, stop=None, check=True, host=True):
    """critic iteration ``dis_iter`` of audiogan.py:706-788.  Even iterations: instance noise on the real and the
    generated clips (:724-728, :749-751).  Odd iterations: clean real clips (the FGSM perturbation of :735-736 is
    computed after ``cls_d`` and never reaches the loss, so it is not computed here) and generated clips moved by
    +-1e-3 along the sign of the critic's input gradient (:752-759, ``extras.adversarial_movement_d``: an
    input-gradient pass that leaves every ``.grad`` untouched).  ``opt_d`` holds the parameters of d and e_d (:691).
    ``host=False``: the accuracies stay device scalars - with ``check=False`` and ``stop='never'`` the iteration then issues no
    host read at all and can be captured into a hipGraph (loop.TrainLoop(graphed=True))."""
    even = dis_iter % 2 == 0
    embed_real = e_d(cs, cl)
    cls_d, _, _, nf_d = d(real + noise_real if even else real, real_len, embed_real)
    loss_d, _ = masked_bce_mean(cls_d, 0.9, nf_d.contiguous())
    with torch.no_grad():
        embed_g = e_g(cs2, cl2)
        fake, _, _, fake_len = g(z=z, c=embed_g, stop=stop)
    embed_d = e_d(cs2, cl2)
    if even:
        fake = fake + noise_fake[:, :fake.size(1)]
    else:
        fake = fake + adversarial_movement_d(fake, fake_len, embed_d.detach(), 0.0, None, d, scale=1e-3)
    cls_g, _, _, nf_g = d(fake, fake_len, embed_d)
    loss_g, _ = masked_bce_mean(cls_g, 0.0, nf_g.contiguous())
    loss = loss_d + loss_g
    opt_d.zero_grad()
    _backward(loss)
    opt_d.step(clip_norm=dgradclip, check=check)
    return dict(loss=loss.detach(), loss_d=loss_d.detach(), loss_g=loss_g.detach(), cls_d=cls_d.detach(),
                cls_g=cls_g.detach(), grad_norm=opt_d.last_norm_sum,
                acc_d=_acc(cls_d.detach(), nf_d, True, host), acc_g=_acc(cls_g.detach(), nf_g, False, host))


def g_step_full(g, d, e_g, e_d, opt_g, real, real_len, cs, cl, z0, noise_real, noise_adv, noise_fake, stop_adv, stop,
                baseline=None, ggradclip=0.1, g_optim='boundary_seeking', lambda_fp=1.0, check=True, host=True):
    """generator iteration of audiogan.py:816-921: adversarial z (:836), generator + critic forward (:841-847), feature
    penalty against the critic's statistics on the real clips (:847-855), BCE towards 0.5 (:857-864), reward / baseline
    (:873-887), loss + penalty (:897), REINFORCE of the stop draws into the stop head only (:898-908), per-parameter clip
    and step over the parameters of g and e_g (:909-921).  Returns a dict with the new ``baseline``.  ``baseline``: None (first
    iteration), a float, or a device scalar; ``host=False``: it is kept / returned as a device scalar (no host read: the
    iteration can be captured into a hipGraph, see d_step_full)."""
    B = real.size(0)
    fs, ns = g._frame_size, g._noise_size
    embed_g = e_g(cs, cl)
    with frozen(d, e_d):
        embed_d = e_d(cs, cl).detach()
        z = adversarially_sample_z(g, d, B, z0.size(1), ns, z0.size(1) * fs, embed_g.detach(), 0.0, embed_d,
                                   g_optim=g_optim, scale=1e-2, z=z0, noise=noise_adv, stop=stop_adv)
        fake, s, stop_list, fake_len = g(z=z, c=embed_g, stop=stop)
        fake = fake + noise_fake[:, :fake.size(1)]
        cls_g, hs_g, hl_g, nf_g = d(fake, fake_len, embed_d)
        with torch.no_grad():
            _, hs_d, hl_d, _ = d(real + noise_real, real_len, embed_d)
        pen = feature_penalty_fused(hs_d, hl_d, hs_g, hl_g, B)
        bce, per = masked_bce_mean(cls_g, 0.5 if g_optim == 'boundary_seeking' else 0.0, nf_g.contiguous())
        reward = -(per / nf_g.float())                       # per-sample loss, a constant for the stop head
        rmean = reward.mean().detach()
        if host:
            rmean = float(rmean)
            if torch.is_tensor(baseline):
                baseline = float(baseline)
        baseline = rmean if baseline is None else baseline * 0.5 + rmean * 0.5
        frames = fake_len // fs
        # (the frames that were generated: max_b frames[b] == s.size(1) by construction of Generator.forward - no host read)
        weight_r = length_mask((B, s.size(1)), frames)
        reward = (reward - baseline).unsqueeze(1) * weight_r
        loss = bce + pen * lambda_fp
        opt_g.zero_grad()
        _backward(loss, retain_graph=True)
        with only_stopper_trains(g, e_g):
            _backward(stopper_surrogate_loss(s, stop_list, reward))
    opt_g.step(clip_norm=ggradclip, check=check)
    return dict(loss=loss.detach(), bce=bce.detach(), feature_penalty=pen.detach(), z=z, fake=fake.detach(),
                fake_len=fake_len, s=s.detach(), baseline=baseline, grad_norm=opt_g.last_norm_sum)


# --------------------------------------------------------------------------------------
# BASELINE configs[3] (C4) and configs[4] (C5): the step with a conv critic that scores a whole clip
# (``convnets.ConvPoolCritic``: the a4 conv stack + average pool + Linear(1)).
#   c4_step       GRU-front generator (modules.GRUGenerator) + conv critic, BCE-with-logits adversarial loss
#   wgan_gp_step  WGAN-GP (modeltf.py:460-469, lambda 10): critic loss with the gradient penalty's double backward,
#                 generator loss -D(G(z))
# Same oracle-side statements: oracle.audiogan_oracle.c4_step / wgan_gp_step.
# --------------------------------------------------------------------------------------
def c4_step(g, critic, opt_g, opt_d, real, c, z, noise_real, noise_fake, dgradclip=1.0, ggradclip=0.1, check=False,
            hook_d=None, hook_g=None):
    with torch.no_grad():
        fake = g(z=z, c=c, stop='never')[0] + noise_fake
    B = real.size(0)
    cls = critic(torch.cat([real + noise_real, fake], 0)).view(2 * B, 1)
    loss_d = masked_bce_mean(cls[:B], 0.9, None)[0] + masked_bce_mean(cls[B:], 0.0, None)[0]
    opt_d.zero_grad()
    _backward(loss_d)
    opt_d.step(clip_norm=dgradclip, grad_scale=hook_d() if hook_d is not None else 1.0, check=check)
    with frozen(critic):
        fake = g(z=z, c=c, stop='never')[0]
        loss_g = masked_bce_mean(critic(fake + noise_fake).view(B, 1), 0.5, None)[0]
        opt_g.zero_grad()
        _backward(loss_g)
    opt_g.step(clip_norm=ggradclip, grad_scale=hook_g() if hook_g is not None else 1.0, check=check)
    return loss_d.detach(), loss_g.detach()


def wgan_gp_step(g, critic, opt_g, opt_d, real, c, z, eps, lam=10.0, dgradclip=0.0, ggradclip=0.0, check=False,
                 hook_d=None, hook_g=None):
    from .convnets import wgan_gp_d_loss, wgan_g_loss
    with torch.no_grad():
        fake = g(z=z, c=c, stop='never')[0]
    loss_d = wgan_gp_d_loss(critic, real, fake, eps, lam)
    opt_d.zero_grad()
    _backward(loss_d)
    opt_d.step(clip_norm=dgradclip, grad_scale=hook_d() if hook_d is not None else 1.0, check=check)
    with frozen(critic):
        loss_g = wgan_g_loss(critic, g(z=z, c=c, stop='never')[0])
        opt_g.zero_grad()
        _backward(loss_g)
    opt_g.step(clip_norm=ggradclip, grad_scale=hook_g() if hook_g is not None else 1.0, check=check)
    return loss_d.detach(), loss_g.detach()


# --------------------------------------------------------------------------------------
# The canonical step as replayed hipGraphs (what bench.py times).  ~1000 launches per step would otherwise be paced by
# the Python interpreter, not by the GPU.
#   single     one graph for the whole G+D step (one GPU)
#   phased     six graphs - critic forward + backward of heads / biLSTM | critic conv-stack backward | opt_d | generator
#              forward + backward down to the front's frames | front backward | opt_g - with the gradient all-reduces
#              (``ddp.GradBucket``) as ordinary stream operations between them; the first all-reduce of each network
#              runs on RCCL's stream WHILE the next graph does the rest of that network's backward
# --------------------------------------------------------------------------------------
class GraphedStep(object):
    def __init__(self, g, d, opt_g, opt_d, batch, phased=False, bucket_d=None, bucket_g=None, world=1, overlap=True,
                 dgradclip=1.0, ggradclip=0.1):
        """``batch``: dict(real, real_len, c, z, noise_real, noise_fake) of device tensors that stay where they are (a
        replay re-reads them: refill them in place for new data).  Capturing records the step WITHOUT executing it: the
        model and optimiser state are untouched, and every lazily created resource (optimiser state, persistent-kernel
        workspace, gradient buckets) must exist already - run at least one eager ``gd_step`` with the same shapes first.
        Raises whatever the capture raises; the caller may then fall back to ``gd_step``."""
        from . import common, kernels as K
        self.g, self.d, self.opt_g, self.opt_d, self.b = g, d, opt_g, opt_d, batch
        self.phased, self.bd, self.bg = phased, bucket_d, bucket_g
        self.losses, self.keep, self.gkeep = {}, {}, {}
        self.graph = self.phases = None
        b = batch
        dev = b['real'].device
        K.reserve_table_arena()
        mark = K.capture_mark()

        def capture(fn):
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            common.new_capture()      # every weight group materialises at its first use INSIDE this graph
            # thread_local: other threads (e.g. the RCCL watchdog) may keep calling HIP while we capture
            with torch.cuda.graph(gr, capture_error_mode='thread_local'):
                fn()
            return gr

        try:
            if not phased:
                def whole():
                    self.losses['d'], self.losses['g'] = gd_step(
                        g, d, opt_g, opt_d, b['real'], b['real_len'], b['c'], b['z'], b['noise_real'], b['noise_fake'],
                        dgradclip, ggradclip, overlap=overlap)
                self.graph = capture(whole)
            else:
                scale = 1.0 / world
                # (no second stream when the generator's frame loop is a persistent launch: one at a time per device)
                side_ok = overlap and not g.front_is_persistent(b['z'].size(0), dev)
                self.side_ok = side_ok
                keep, gkeep = self.keep, self.gkeep

                def critic_early():
                    # the G forward of the generator iteration as a parallel branch of this graph: G's weights are
                    # materialised on the main stream, then fork, run it beside the critic, join before the capture ends
                    if side_ok:
                        side = _SIDE.get(dev)
                        if side is None:
                            side = _SIDE[dev] = torch.cuda.Stream(device=dev)
                        g.prepare_weights()
                        side.wait_stream(torch.cuda.current_stream())
                        with torch.cuda.stream(side):
                            keep['pre'] = g(z=b['z'], c=b['c'], stop='never', cut=gkeep)
                    else:
                        keep['pre'] = None
                    self.losses['d'] = d_backward_early(g, d, opt_d, b['real'], b['real_len'], b['c'], b['z'],
                                                        b['noise_real'], b['noise_fake'], keep)
                    if side_ok:
                        torch.cuda.current_stream().wait_stream(_SIDE[dev])

                def gen_early():
                    self.losses['g'] = g_backward_early(g, d, opt_g, b['c'], b['z'], b['noise_fake'], gkeep,
                                                        pre=keep.get('pre'))

                g1a = capture(critic_early)
                g1b = capture(lambda: d_backward_late(keep))
                g2 = capture(lambda: opt_d.step(clip_norm=dgradclip, grad_scale=scale))
                g3a = capture(gen_early)
                # g3b runs beside the all-reduce of the trunk's gradients: no persistent launch in it (a collective's
                # kernels hold CUs for as long as their peers need; the per-frame form shares the chip with them)
                g3b = capture(lambda: g_backward_late(gkeep, persist=self.bg is None))
                g4 = capture(lambda: opt_g.step(clip_norm=ggradclip, grad_scale=scale))
                self.phases = (g1a, g1b, g2, g3a, g3b, g4)
            torch.cuda.synchronize()
        except Exception:
            K.drop_captured_tables(mark)      # their device copies were never executed
            torch.cuda.synchronize()
            raise

    def status(self):
        """sticky status word of the persistent recurrent launches on this device (0 = every launch of every replay so
        far completed; one host sync).  Replays cannot raise from inside the graph: call this (or ``check()``) every
        N steps / at the end of a run."""
        from . import kernels as K
        return K.lstm_persist_status(self.b['real'].device)

    def check(self):
        """raise ``kernels.PersistentLaunchError`` if a persistent launch of any replay gave up"""
        from . import kernels as K
        K.check_persist_status(self.b['real'].device)

    def step(self, check=False):
        """one G+D step; returns (loss_d, loss_g) - device scalars overwritten by the next step.  ``check``: synchronise
        and verify the persistent launches' status afterwards (see ``status``)."""
        out = self._step()
        if check:
            self.check()
        return out

    def _step(self):
        if self.graph is not None:
            self.graph.replay()
        else:
            ph, bd, bg = self.phases, self.bd, self.bg
            ph[0].replay()                                  # critic fwd + bwd of heads / biLSTM
            if bd is not None:
                bd.all_reduce(async_op=True, part='early')  # overlaps ...
            ph[1].replay()                                  # ... the conv-stack backward
            if bd is not None:
                bd.wait()
                bd.all_reduce(part='late')
            ph[2].replay()                                  # opt_d
            ph[3].replay()                                  # generator fwd + bwd down to the front's frames
            if bg is not None:
                bg.all_reduce(async_op=True, part='early')  # conv trunk's gradients; overlaps ...
            ph[4].replay()                                  # ... the recurrent front's backward
            if bg is not None:
                bg.wait()
                bg.all_reduce(part='late')
            ph[5].replay()                                  # opt_g
        return self.losses['d'], self.losses['g']


# --------------------------------------------------------------------------------------
# Training feed: dataset.dataloader minibatches -> the static device tensors a GraphedStep replays on
# (audiogan.py:94-97 ``tovar``, :714-716, :823-828: the reference converts and uploads every minibatch synchronously
# inside the iteration).  Here the upload of minibatch i+1 runs on a copy stream WHILE step i is replaying:
#   host arrays -> pinned staging (slot i & 1) --H2D, copy stream--> device staging (slot i & 1)
#   step boundary, main stream: wait for that H2D, static tensors <- device staging (a few MB of D2D), replay
# The replayed graph reads only the static tensors; they are rewritten only between replays, on the stream the
# replays run on, so no replay ever sees a half-written batch.
# --------------------------------------------------------------------------------------
def batch_from_loader(item, device_noise=True):
    """the reference loader's ``next()`` list (dataset.py:91: [epoch, batch, samples f64, lengths, keys, cseq i32, clen])
    as the host half of a feed: float32 clips (``tovar``, audiogan.py:95) and int64 lengths / character sequences"""
    import numpy as np
    _, _, samples, lengths, _, cseq, clen = item
    return dict(real=np.ascontiguousarray(samples, dtype=np.float32), real_len=np.asarray(lengths, dtype=np.int64),
                cs=np.asarray(cseq, dtype=np.int64), cl=np.asarray(clen, dtype=np.int64))


class Feeder(object):
    def __init__(self, static, keys=None, device_fill=None):
        """``static``: dict of the step's static device tensors (``GraphedStep.b``).  ``keys``: the entries that come
        from the host (default: all of ``static``); the others stay as they are unless ``device_fill`` (dict key ->
        callable(tensor), e.g. ``lambda t: t.normal_().mul_(0.01)`` for instance noise / z) refreshes them on the device
        at every step boundary."""
        self.static = static
        self.keys = list(keys) if keys is not None else list(static.keys())
        self.device_fill = dict(device_fill or {})
        for k in self.keys + list(self.device_fill):
            if k not in static:
                raise KeyError('Feeder: %r is not one of the step\'s static tensors %r' % (k, sorted(static)))
        any_t = static[self.keys[0]]
        self.cuda = any_t.is_cuda
        self.n_staged = self.n_fed = 0
        self.host_ms = dict(sync=0.0, pin=0.0, sync_max=0.0, pin_max=0.0)     # where stage() spends host time
        self._ready = [None, None]
        if self.cuda:
            self.copy_stream = torch.cuda.Stream(device=any_t.device)
            self.pinned = [{k: torch.empty(static[k].shape, dtype=static[k].dtype).pin_memory() for k in self.keys}
                           for _ in range(2)]
            self.pinned_np = [{k: v.numpy() for k, v in sl.items()} for sl in self.pinned]
            self.dev = [{k: torch.empty_like(static[k]) for k in self.keys} for _ in range(2)]
            self._free = [None, None]            # event: the device staging slot has been consumed (D2D done)
        else:
            self.dev = [{k: torch.empty_like(static[k]) for k in self.keys} for _ in range(2)]

    def _check(self, host):
        for k in self.keys:
            if k not in host:
                raise KeyError('Feeder.stage: the host batch lacks %r' % k)
            t = host[k] if torch.is_tensor(host[k]) else torch.from_numpy(host[k])
            if tuple(t.shape) != tuple(self.static[k].shape):
                raise ValueError('Feeder.stage: %r has shape %r, the captured step expects %r (a hipGraph replays fixed '
                                 'shapes: pad / crop the minibatch, e.g. dataloader(..., maxlen=, frame_size=))'
                                 % (k, tuple(t.shape), tuple(self.static[k].shape)))
            if t.dtype != self.static[k].dtype:
                raise TypeError('Feeder.stage: %r is %s, expected %s' % (k, t.dtype, self.static[k].dtype))
            yield k, t

    def stage(self, host):
        """start moving the NEXT minibatch (dict key -> numpy array / CPU tensor) towards the device; returns at once.
        At most one batch can be staged ahead of the one being fed."""
        if self.n_staged - self.n_fed >= 2:
            raise RuntimeError('Feeder.stage: two minibatches are already staged; call feed() first')
        slot = self.n_staged & 1
        items = list(self._check(host))
        if not self.cuda:
            for k, t in items:
                self.dev[slot][k].copy_(t)
        else:
            import time as _t
            t0 = _t.perf_counter()
            if self._free[slot] is not None:
                self._free[slot].synchronize()       # the pinned + device slot were consumed two batches ago
            t1 = _t.perf_counter()
            # (numpy's single-threaded memcpy, not Tensor.copy_: a CPU tensor copy of a few MB fans out over every core the
            # machine has and the pool then spins through its idle time; under a container's CPU quota that gets the whole
            # process throttled for the rest of the scheduler period - 90 ms stalls once per pass were measured)
            for k, t in items:
                self.pinned_np[slot][k][...] = t.numpy()
            t2 = _t.perf_counter()
            self.host_ms['sync'] += (t1 - t0) * 1e3; self.host_ms['pin'] += (t2 - t1) * 1e3
            self.host_ms['sync_max'] = max(self.host_ms['sync_max'], (t1 - t0) * 1e3)
            self.host_ms['pin_max'] = max(self.host_ms['pin_max'], (t2 - t1) * 1e3)
            with torch.cuda.stream(self.copy_stream):
                if self._free[slot] is not None:
                    self.copy_stream.wait_event(self._free[slot])
                for k, _ in items:
                    self.dev[slot][k].copy_(self.pinned[slot][k], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self.copy_stream)
                self._ready[slot] = ev
        self.n_staged += 1

    def feed(self):
        """step boundary: make the oldest staged minibatch the step's input (enqueued on the current stream, behind the
        previous replay): static tensors <- device staging, device-side refreshes; then call ``GraphedStep.step()``"""
        if self.n_fed >= self.n_staged:
            raise RuntimeError('Feeder.feed: nothing staged')
        slot = self.n_fed & 1
        if self.cuda:
            torch.cuda.current_stream().wait_event(self._ready[slot])
        for k in self.keys:
            self.static[k].copy_(self.dev[slot][k], non_blocking=True)
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self._free[slot] = ev
        for k, fn in self.device_fill.items():
            fn(self.static[k])
        self.n_fed += 1

    def run(self, graphed, host_batches, on_step=None):
        """feed ``host_batches`` (an iterable of host dicts, e.g. ``map(batch_from_loader, gen_train)``) through
        ``graphed`` with the upload of batch i+1 overlapping step i; returns the number of steps"""
        it = iter(host_batches)
        try:
            self.stage(next(it))
        except StopIteration:
            return 0
        n = 0
        while self.n_fed < self.n_staged:
            self.feed()
            out = graphed.step()
            try:
                self.stage(next(it))          # overlaps the replay just enqueued
            except StopIteration:
                pass
            n += 1
            if on_step is not None:
                on_step(n, out)
        return n
