"""The reference's OUTER training loop (audiogan.py:703-940) around ``train.d_step_full`` / ``train.g_step_full``:

    while True:
        for j in range(critic_iter):          # :710
            <critic iteration dis_iter>       # :711-788   odd: FGSM branch, even: instance noise
            if acc_d > require_acc and acc_g > require_acc: break      # :813
        for _ in range(gencatchup):           # :820
            <generator iteration gen_iter>    # :822-921
            if gen_iter % 500 == 0: save d, g, e_g, e_d               # :932-939

Data comes from the reference's loader interface (``dataset.dataloader`` generator + ``dataset.pick_words``, :714-716,
:823-828); instance noise, z and the stop draws are drawn on the device (the reference draws them on the host with
``RNG.randn`` / ``T.randn`` and uploads them, :724, :749, :825).  The accuracy test of :813 needs the two accuracies on the
host after every critic iteration, exactly as the reference reads them back for its summaries (:790-792): that host read
stays (``fixed_critic_iter`` replaces it by a fixed count, for benchmarks).  Checkpoints are ``state_dict`` files under the
reference's names (``checkpoint.save``)."""
import numpy as np
import torch

from . import checkpoint, train


class TrainLoop(object):
    def __init__(self, g, d, e_g, e_d, opt_g, opt_d, loader, pick_words, batch_size, maxlen, device, noisescale=0.01,
                 critic_iter=100, require_acc=0.5, gencatchup=1, dgradclip=1.0, ggradclip=0.1, g_optim='boundary_seeking',
                 checkpoint_every=500, checkpoint_prefix=None, fixed_critic_iter=None, stop=None, check=True):
        """``loader``: the generator ``dataset.dataloader`` returns (``next()`` -> [epoch, batch, samples, lengths, keys, cseq,
        clen], dataset.py:91); ``pick_words``: a callable () -> (cseq, clen) numpy arrays for ``batch_size`` random words
        (``dataset.pick_words(..., skip_samples=True)[1:3]``, audiogan.py:715-716); ``stop``: None = Bernoulli stop draws
        like the reference (a host sync per generator forward), 'never' = fixed-length clips.  ``opt_d`` holds the parameters
        of d and e_d, ``opt_g`` those of g and e_g (:690-691)."""
        self.g, self.d, self.e_g, self.e_d, self.opt_g, self.opt_d = g, d, e_g, e_d, opt_g, opt_d
        self.loader, self.pick_words = loader, pick_words
        self.B, self.maxlen, self.dev = batch_size, maxlen, torch.device(device)
        self.noisescale, self.critic_iter, self.require_acc, self.gencatchup = noisescale, critic_iter, require_acc, gencatchup
        self.dgradclip, self.ggradclip, self.g_optim = dgradclip, ggradclip, g_optim
        self.checkpoint_every, self.prefix = checkpoint_every, checkpoint_prefix
        self.fixed_critic_iter, self.stop, self.check = fixed_critic_iter, stop, check
        self.dis_iter = self.gen_iter = 0
        self.baseline = None
        self.log = []
        fs = g._frame_size
        self.nframes = (maxlen + fs - 1) // fs
        self.L = self.nframes * fs

    # ---- minibatch pieces ---------------------------------------------------------------
    def _up(self, a, dtype):
        t = torch.from_numpy(np.ascontiguousarray(a)).to(dtype)
        return t.pin_memory().to(self.dev, non_blocking=True) if self.dev.type == 'cuda' else t

    def _real(self):
        """``tovar`` of the loader's next minibatch (audiogan.py:94-97, :714): float32 clips padded to the frame grid"""
        _, _, samples, lengths, _, cseq, clen = next(self.loader)
        x = np.zeros((self.B, self.L), dtype=np.float32)
        n = min(self.L, samples.shape[1])
        x[:, :n] = samples[:, :n]
        return (self._up(x, torch.float32), self._up(lengths, torch.long), self._up(cseq, torch.long), self._up(clen, torch.long))

    def _words(self):
        cs, cl = self.pick_words()
        return self._up(cs, torch.long), self._up(cl, torch.long)

    def _noise(self):
        return torch.randn(self.B, self.L, device=self.dev) * self.noisescale

    def _stop_arg(self, nframes):
        return None if self.stop is None else self.stop

    # ---- iterations -----------------------------------------------------------------------
    def d_iteration(self):
        self.dis_iter += 1
        real, real_len, cs, cl = self._real()
        cs2, cl2 = self._words()
        z = torch.randn(self.B, self.nframes, self.g._noise_size, device=self.dev)
        even = self.dis_iter % 2 == 0       # (odd iterations take the FGSM branch: no instance noise is drawn, :729-736, :752-759)
        r = train.d_step_full(self.g, self.d, self.e_g, self.e_d, self.opt_d, self.dis_iter, real, real_len, cs, cl, cs2, cl2, z,
                              self._noise() if even else None, self._noise() if even else None, self.dgradclip,
                              stop=self._stop_arg(self.nframes), check=self.check)
        return r

    def g_iteration(self):
        self.gen_iter += 1
        real, real_len, _, _ = self._real()
        cs, cl = self._words()
        z0 = torch.randn(self.B, self.nframes, self.g._noise_size, device=self.dev)
        r = train.g_step_full(self.g, self.d, self.e_g, self.e_d, self.opt_g, real, real_len, cs, cl, z0, self._noise(),
                              self._noise(), self._noise(), self._stop_arg(self.nframes), self._stop_arg(self.nframes),
                              self.baseline, self.ggradclip, self.g_optim, check=self.check)
        self.baseline = r['baseline']
        if self.prefix is not None and self.checkpoint_every and self.gen_iter % self.checkpoint_every == 0:
            checkpoint.save(self.prefix, self.gen_iter, d=self.d, g=self.g, e_g=self.e_g, e_d=self.e_d, opt_d=self.opt_d,
                            opt_g=self.opt_g, extra=dict(dis_iter=self.dis_iter, gen_iter=self.gen_iter,
                                                         baseline=self.baseline))
        return r

    def outer(self):
        """one pass of the ``while True`` body; returns (critic iterations run, last critic result, last generator result)"""
        n_d = self.fixed_critic_iter if self.fixed_critic_iter is not None else self.critic_iter
        rd = rg = None
        ran = 0
        for _ in range(n_d):
            rd = self.d_iteration()
            ran += 1
            self.log.append(('D', self.dis_iter, float(rd['loss']), rd['acc_d'], rd['acc_g']))
            if self.fixed_critic_iter is None and rd['acc_d'] > self.require_acc and rd['acc_g'] > self.require_acc:
                break
        for _ in range(self.gencatchup):
            rg = self.g_iteration()
            self.log.append(('G', self.gen_iter, float(rg['loss']), float(rg['feature_penalty'])))
        return ran, rd, rg

    def run(self, n_outer):
        for _ in range(n_outer):
            self.outer()
        return self.log

    def resume(self, iteration):
        """load the checkpoint written at generator iteration ``iteration`` (audiogan.py:696-701) and continue from there"""
        extra = checkpoint.load(self.prefix, iteration, d=self.d, g=self.g, e_g=self.e_g, e_d=self.e_d, opt_d=self.opt_d,
                                opt_g=self.opt_g) or {}
        self.dis_iter, self.gen_iter = int(extra.get('dis_iter', 0)), int(extra.get('gen_iter', iteration))
        self.baseline = extra.get('baseline', None)
        return extra


def words_picker(dataset_module, batch_size, maxlen, h5, keys, args, frame_size=None):
    """() -> (cseq, clen): ``dataset.pick_words(..., skip_samples=True)`` for a batch of random words (audiogan.py:715-716)"""
    maxchar = max(len(k) for k in keys)

    def pick():
        out = dataset_module.pick_words(batch_size, maxlen, h5, keys, maxchar, args, frame_size=frame_size, skip_samples=True)
        return out[1], out[2]
    return pick
