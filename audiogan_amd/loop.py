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
reference's names (``checkpoint.save``).

``graphed=True`` (needs ``stop='never'``, a CUDA device): the three iteration bodies - odd critic iteration, even critic
iteration, generator iteration - are captured ONCE into three hipGraphs over static input tensors and replayed; a minibatch
is uploaded into the static tensors and z / the instance noise are drawn in place on the device before each replay.  The
iterations run with ``host=False, check=False`` (no host read inside a graph): accuracies and the REINFORCE baseline are
device scalars; the accuracy test of :813 reads two of them back after the replay, as the reference does after its
iteration.  ~3000 launches per pass are then paced by the GPU, not by the Python interpreter."""
import time

import numpy as np
import torch

from . import checkpoint, train


class TrainLoop(object):
    def __init__(self, g, d, e_g, e_d, opt_g, opt_d, loader, pick_words, batch_size, maxlen, device, noisescale=0.01,
                 critic_iter=100, require_acc=0.5, gencatchup=1, dgradclip=1.0, ggradclip=0.1, g_optim='boundary_seeking',
                 checkpoint_every=500, checkpoint_prefix=None, fixed_critic_iter=None, stop=None, check=True, graphed=False,
                 host=None):
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
        self.graphed = bool(graphed)
        # host=False: the iterations return device scalars and keep the REINFORCE baseline on the device (fp32) - what the
        # captured iterations do; an eager loop with host=False runs the very same arithmetic (tests compare the two bit for bit)
        self.host = (not self.graphed) if host is None else bool(host)
        self._graphs = None
        # host milliseconds spent per phase of the captured iterations (enqueueing a replay, the loader, staging the upload,
        # feeding the static inputs), summed since construction: bench.py --workload full reports them per pass
        self.host_ms = dict(replay=0.0, loader=0.0, stage=0.0, inputs=0.0)
        self.gpu_timeline = None         # set to a list to record (key, start event, end event, host time) per replay
        if self.graphed:
            assert stop == 'never' and self.dev.type == 'cuda', "graphed=True needs stop='never' and a CUDA device"

    # ---- minibatch pieces ---------------------------------------------------------------
    def _up(self, a, dtype):
        a = np.asarray(a)
        if self.dev.type != 'cuda':
            return torch.from_numpy(np.ascontiguousarray(a)).to(dtype)
        # (a pinned buffer filled by numpy's single-threaded copy / conversion: Tensor.to / pin_memory fan a 2 MB copy out over
        # every core and the idle pool's spinning eats a container's CPU quota - see train.Feeder.stage)
        t = torch.empty(a.shape, dtype=dtype, pin_memory=True)
        t.numpy()[...] = a
        return t.to(self.dev, non_blocking=True)

    def _real(self):
        """``tovar`` of the loader's next minibatch (audiogan.py:94-97, :714): float32 clips padded to the frame grid"""
        _, _, samples, lengths, _, cseq, clen = next(self.loader)
        x = np.zeros((self.B, self.L), dtype=np.float32)
        n = min(self.L, samples.shape[1])
        x[:, :n] = samples[:, :n]
        return (self._up(x, torch.float32), self._up(lengths, torch.long), self._up(cseq, torch.long), self._up(clen, torch.long))

    def _words(self):
        cs, cl = self.pick_words()
        self._cwidth = int(np.asarray(cs).shape[1])
        return self._up(cs, torch.long), self._up(cl, torch.long)

    def _noise(self):
        return torch.randn(self.B, self.L, device=self.dev) * self.noisescale

    def _stop_arg(self, nframes):
        return None if self.stop is None else self.stop

    # ---- captured iterations ---------------------------------------------------------------
    def _capture(self):
        """eager warm-up of the three bodies (every lazily created resource - optimiser state, workspaces, descriptor tables,
        gradient buffers - must exist before a capture), then one capture each.  Capturing records without executing: the
        warm-up iterations are real training iterations (they count), the captures change nothing."""
        from . import common, kernels as K
        dev, B, L, T = self.dev, self.B, self.L, self.nframes
        ns = self.g._noise_size
        # one eager pass of each body on real data
        self.graphed = False
        try:
            for _ in range(2):
                self.d_iteration()
            self.g_iteration()
            self.d_iteration(); self.d_iteration()          # (.grad buffers flattened by the second zero_grad: stable from here)
            self.g_iteration()
        finally:
            self.graphed = True
        mc = max(self._cwidth, 1)          # character matrices are maxchar wide (dataset.word_to_seq), the same every batch
        st = dict(real=torch.zeros(B, L, device=dev), real_len=torch.full((B,), L, dtype=torch.long, device=dev),
                  # lcs / lcl: the loader batch's words (the critic iteration's matching text), wcs / wcl: the pick_words batch
                  # (the critic's mismatched text, the generator iteration's text)
                  lcs=torch.zeros(B, mc, dtype=torch.long, device=dev), lcl=torch.ones(B, dtype=torch.long, device=dev),
                  wcs=torch.zeros(B, mc, dtype=torch.long, device=dev), wcl=torch.ones(B, dtype=torch.long, device=dev),
                  z=torch.zeros(B, T, ns, device=dev), n1=torch.zeros(B, L, device=dev), n2=torch.zeros(B, L, device=dev),
                  n3=torch.zeros(B, L, device=dev),
                  baseline=torch.zeros((), device=dev) if self.baseline is None else
                  (self.baseline.detach().clone().float().reshape(()) if torch.is_tensor(self.baseline) else
                   torch.as_tensor(float(self.baseline), device=dev, dtype=torch.float32)))
        self._static, self._out = st, {}
        self._feeder = train.Feeder(st, keys=['real', 'real_len', 'lcs', 'lcl', 'wcs', 'wcl'])
        K.reserve_table_arena()
        mark = K.capture_mark()

        def cap(fn):
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            common.new_capture()
            with torch.cuda.graph(gr, capture_error_mode='thread_local'):
                fn()
            return gr

        def d_body(parity):
            def body():
                even = parity == 0
                self._out['d%d' % parity] = train.d_step_full(
                    self.g, self.d, self.e_g, self.e_d, self.opt_d, 2 if even else 1, st['real'], st['real_len'], st['lcs'],
                    st['lcl'], st['wcs'], st['wcl'], st['z'], st['n1'] if even else None, st['n2'] if even else None,
                    self.dgradclip, stop='never', check=False, host=False)
            return body

        def g_body():
            r = train.g_step_full(self.g, self.d, self.e_g, self.e_d, self.opt_g, st['real'], st['real_len'], st['wcs'], st['wcl'],
                                  st['z'], st['n1'], st['n2'], st['n3'], 'never', 'never', st['baseline'], self.ggradclip,
                                  self.g_optim, check=False, host=False)
            st['baseline'].copy_(r['baseline'])           # the running baseline lives on the device, updated by the graph
            self._out['g'] = r

        try:
            self._graphs = dict(d1=cap(d_body(1)), d0=cap(d_body(0)), g=cap(g_body))
            torch.cuda.synchronize()
        except Exception:
            K.drop_captured_tables(mark)
            torch.cuda.synchronize()
            raise

    def _host_pair(self):
        """what every iteration takes from the host, in the order the eager iterations draw it: the loader's next minibatch
        (:714) and one pick_words batch (:715-716), at the static shapes / dtypes"""
        _, _, samples, lengths, _, cseq, clen = next(self.loader)
        x = np.zeros((self.B, self.L), dtype=np.float32)
        n = min(self.L, samples.shape[1])
        x[:, :n] = samples[:, :n]
        ws, wl = self.pick_words()
        i64 = lambda v: np.ascontiguousarray(np.asarray(v), dtype=np.int64)   # noqa: E731
        return dict(real=x, real_len=i64(lengths), lcs=i64(cseq), lcl=i64(clen), wcs=i64(ws), wcl=i64(wl))

    def _next_inputs(self, noises):
        """step boundary of a captured iteration: the staged minibatch becomes the graphs' input (train.Feeder: pinned ->
        device staging on a copy stream while the previous replay runs, device -> static tensors here), then z and the
        instance noise are drawn in place - the same draws, in the same order, as the eager iteration's torch.randn calls"""
        f, st = self._feeder, self._static
        t0 = time.perf_counter()
        if f.n_fed >= f.n_staged:
            f.stage(self._host_pair())
        f.feed()
        st['z'].normal_()
        for k in ('n1', 'n2', 'n3')[:noises]:
            st[k].normal_().mul_(self.noisescale)
        self.host_ms['inputs'] += (time.perf_counter() - t0) * 1e3

    def _replay(self, key):
        t0 = time.perf_counter()
        if self.gpu_timeline is not None:     # (bench.py: GPU duration of each replay and the idle time between two)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        self._graphs[key].replay()
        if self.gpu_timeline is not None:
            e1.record()
            self.gpu_timeline.append((key, e0, e1, t0))
        t1 = time.perf_counter()
        # host work + upload of the NEXT minibatch overlap the replay just enqueued
        pair = self._host_pair()
        t2 = time.perf_counter()
        self._feeder.stage(pair)
        t3 = time.perf_counter()
        h = self.host_ms
        h['replay'] += (t1 - t0) * 1e3; h['loader'] += (t2 - t1) * 1e3; h['stage'] += (t3 - t2) * 1e3
        h['loader_max'] = max(h.get('loader_max', 0.0), (t2 - t1) * 1e3); h['replay_max'] = max(h.get('replay_max', 0.0), (t1 - t0) * 1e3)

    # ---- iterations -----------------------------------------------------------------------
    def d_iteration(self):
        if self.graphed:
            if self._graphs is None:
                self._capture()
            self.dis_iter += 1
            even = self.dis_iter % 2 == 0
            self._next_inputs(2 if even else 0)
            self._replay('d0' if even else 'd1')
            return self._out['d0' if even else 'd1']
        self.dis_iter += 1
        real, real_len, cs, cl = self._real()
        cs2, cl2 = self._words()
        z = torch.randn(self.B, self.nframes, self.g._noise_size, device=self.dev)
        even = self.dis_iter % 2 == 0       # (odd iterations take the FGSM branch: no instance noise is drawn, :729-736, :752-759)
        r = train.d_step_full(self.g, self.d, self.e_g, self.e_d, self.opt_d, self.dis_iter, real, real_len, cs, cl, cs2, cl2, z,
                              self._noise() if even else None, self._noise() if even else None, self.dgradclip,
                              stop=self._stop_arg(self.nframes), check=self.check, host=self.host)
        return r

    def _maybe_checkpoint(self):
        if self.prefix is not None and self.checkpoint_every and self.gen_iter % self.checkpoint_every == 0:
            b = self.baseline
            checkpoint.save(self.prefix, self.gen_iter, d=self.d, g=self.g, e_g=self.e_g, e_d=self.e_d, opt_d=self.opt_d,
                            opt_g=self.opt_g, extra=dict(dis_iter=self.dis_iter, gen_iter=self.gen_iter,
                                                         baseline=float(b) if b is not None else None))

    def g_iteration(self):
        if self.graphed:
            if self._graphs is None:
                self._capture()
            self.gen_iter += 1
            self._next_inputs(3)
            self._replay('g')
            self.baseline = self._static['baseline']
            self._maybe_checkpoint()
            return self._out['g']
        self.gen_iter += 1
        real, real_len, _, _ = self._real()
        cs, cl = self._words()
        z0 = torch.randn(self.B, self.nframes, self.g._noise_size, device=self.dev)
        r = train.g_step_full(self.g, self.d, self.e_g, self.e_d, self.opt_g, real, real_len, cs, cl, z0, self._noise(),
                              self._noise(), self._noise(), self._stop_arg(self.nframes), self._stop_arg(self.nframes),
                              self.baseline, self.ggradclip, self.g_optim, check=self.check, host=self.host)
        self.baseline = r['baseline']
        self._maybe_checkpoint()
        return r

    def outer(self):
        """one pass of the ``while True`` body; returns (critic iterations run, last critic result, last generator result)"""
        n_d = self.fixed_critic_iter if self.fixed_critic_iter is not None else self.critic_iter
        rd = rg = None
        ran = 0
        for _ in range(n_d):
            rd = self.d_iteration()
            ran += 1
            if self.fixed_critic_iter is None:
                # the accuracy test of :813: two host reads per critic iteration, as the reference
                self.log.append(('D', self.dis_iter, float(rd['loss']), float(rd['acc_d']), float(rd['acc_g'])))
                if float(rd['acc_d']) > self.require_acc and float(rd['acc_g']) > self.require_acc:
                    break
            elif self.host:
                self.log.append(('D', self.dis_iter, float(rd['loss']), float(rd['acc_d']), float(rd['acc_g'])))
        for _ in range(self.gencatchup):
            rg = self.g_iteration()
            if self.host:
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
        if self.graphed and self._graphs is not None and self.baseline is not None:
            self._static['baseline'].fill_(float(self.baseline))
            self.baseline = self._static['baseline']
        return extra


def words_picker(dataset_module, batch_size, maxlen, h5, keys, args, frame_size=None):
    """() -> (cseq, clen): ``dataset.pick_words(..., skip_samples=True)`` for a batch of random words (audiogan.py:715-716)"""
    maxchar = max(len(k) for k in keys)

    def pick():
        out = dataset_module.pick_words(batch_size, maxlen, h5, keys, maxchar, args, frame_size=frame_size, skip_samples=True)
        return out[1], out[2]
    return pick
