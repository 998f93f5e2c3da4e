#!/usr/bin/env python
"""bench.py -- audio-samples/sec per G+D train step (BASELINE.json metric) on N x MI355X.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (config.workload): BASELINE configs[1] "C2" -- the full audiogan.py Generator +
Discriminator (default structs, audiogan.py:368,:476; embed 100, noise 100, state 1024,
frame_size 256 -> T=32), batch 64 PER GPU, 8192-sample synthetic white-noise clips, fp32.
One "step" = one canonical G+D step: critic iteration (G fwd, D(real),
D(fake), BCE, backward, per-parameter clip, optimiser) + generator iteration (G fwd, D(fake),
BCE, backward through D into G, clip, optimiser).  Inputs are resident in HBM before the
timed region.  N > 1: pure data parallelism, one process per GPU, one RCCL all-reduce per
network per step (weak scaling: per-GPU batch fixed).

Rank 0 prints ONE JSON line (contract in the task description) with two extra objects:
  roofline     -- the kernel class with the largest share of GPU time, timed live with HIP
                  events on the launch stream during the timed steps
  cpu_baseline -- the CPU oracle (oracle/audiogan_oracle.py, kind "port") running the same
                  step on a bounded sample of the workload on this host's cores
"""
import argparse
import json
import os
import sys
import time
import warnings

warnings.filterwarnings('ignore', category=FutureWarning)
ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

L = 8192
FRAME = 256
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense fp32 matrix peak (spec)
PEAK_BF16_MFMA_TFLOPS = 2500.0 # MI355X_MICROARCH.md: dense bf16 matrix peak (spec)
PEAK_HBM_GBS = 8000.0


def build_models(A, dev, opt_kind, seed=0, workload='c2'):
    """c2: the audiogan.py Generator + Discriminator (BASELINE configs[1]/[2]).  c4: GRU-front generator + conv critic
    (configs[3]).  c5: the audiogan.py Generator + conv critic trained with WGAN-GP (configs[4])."""
    from audiogan_amd import optim
    torch.manual_seed(seed)
    if workload == 'c4':
        g = A.GRUGenerator(frame_size=FRAME, embed_size=100, noise_size=100, state_size=1024).to(dev)
    else:
        g = A.Generator(frame_size=FRAME, embed_size=100, noise_size=100, state_size=1024).to(dev)
    if workload == 'c2':
        d = A.Discriminator(state_size=1024, embed_size=100).to(dev)
    else:
        d = A.ConvPoolCritic().to(dev)
    opt_g = optim.make_optimizer(list(g.parameters()), opt_kind, 1e-4)
    opt_d = optim.make_optimizer(list(d.parameters()), opt_kind, 1e-4)
    return g, d, opt_g, opt_d


def synthetic_batch(batch, dev, seed):
    """white-noise clips U(-1,1), peak-normalised like dataset.py:68-71; z, c, instance noise from
    a seeded host generator"""
    rs = np.random.RandomState(seed)
    x = rs.uniform(-1, 1, size=(batch, L))
    x = (x / np.abs(x).max(1, keepdims=True)).astype(np.float32)
    gen = torch.Generator().manual_seed(seed)
    T = L // FRAME
    out = dict(real=torch.from_numpy(x), real_len=torch.full((batch,), L, dtype=torch.long),
               c=torch.randn(batch, 100, generator=gen), z=torch.randn(batch, T, 100, generator=gen),
               noise_real=torch.randn(batch, L, generator=gen) * 0.01,
               noise_fake=torch.randn(batch, L, generator=gen) * 0.01,
               eps=torch.rand(batch, 1, generator=gen))
    return {k: v.to(dev) for k, v in out.items()}


_SIDE = [None]


WORKLOAD = ['c2']
WORKLOAD_TEXT = {
    'c2': '%s: full audiogan.py Conv1d/LSTM G + D, batch %d per GPU, 8192-sample white-noise clips, frame_size 256 '
          '(T=32), canonical G+D step, %s, per-parameter clip d=1 g=0.1',
    'c4': '%s: GRU-front generator (the audiogan.py Generator with a GRU cell in its frame loop) + conv critic (a4 conv '
          'stack + average pool + Linear), batch %d per GPU, 8192-sample white-noise clips, frame_size 256 (T=32), BCE '
          'G+D step, %s, per-parameter clip d=1 g=0.1',
    'c5': '%s: audiogan.py Generator + conv critic (a4 conv stack + average pool + Linear), WGAN-GP (lambda 10, gradient '
          'penalty by a hand-written double backward), batch %d per GPU, 8192-sample white-noise clips, frame_size 256 '
          '(T=32), critic + generator iteration, %s',
}


def one_step(train, g, d, opt_g, opt_d, b, hook_d=None, hook_g=None, overlap=False):
    """the canonical step (train.gd_step).  overlap=True: the generator iteration's G forward runs on a second
    stream beside the critic iteration (see train.gd_step).  Returns (loss_d, loss_g)."""
    if WORKLOAD[0] == 'c4':
        return train.c4_step(g, d, opt_g, opt_d, b['real'], b['c'], b['z'], b['noise_real'], b['noise_fake'], 1.0, 0.1,
                             hook_d=hook_d, hook_g=hook_g)
    if WORKLOAD[0] == 'c5':
        return train.wgan_gp_step(g, d, opt_g, opt_d, b['real'], b['c'], b['z'], b['eps'], 10.0,
                                  hook_d=hook_d, hook_g=hook_g)
    return train.gd_step(g, d, opt_g, opt_d, b['real'], b['real_len'], b['c'], b['z'], b['noise_real'],
                         b['noise_fake'], 1.0, 0.1, overlap=overlap, hook_d=hook_d, hook_g=hook_g)


def pmc_traffic(kernel, dtype='f32'):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/r04_pmc_traffic.json, for
    --dtype bf16 profiles/r04_pmc_traffic_bf16.json; an earlier round's file when the newest has no entry: FETCH_SIZE and
    WRITE_SIZE in separate passes, gfx950 x2 read correction applied); the instantiation if it was profiled under that
    name, else the kernel class; None when neither was."""
    suffix = '_bf16' if dtype == 'bf16' else ''
    for rnd in ('r04', 'r03', 'r02'):
        path = os.path.join(ROOT, 'profiles', '%s_pmc_traffic%s.json' % (rnd, suffix))
        if not os.path.exists(path):
            continue
        tab = json.load(open(path))
        name = kernel.replace(' ', '')
        ent = tab.get(name) or tab.get(name.split('<')[0])
        if ent:
            return ent['bytes_per_launch']
    return None


def cpu_baseline(batch, steps, opt_kind, threads=0, budget_s=110.0, warm=2, workload='c2'):
    """the oracle's canonical step on `batch` clips of the same workload on the host cores (the full
    C2 batch, 2 warm-up + >= 5 timed steps, median).  Bounded: it keeps timing steps only while the total stays
    under `budget_s`; a slow host therefore reports from fewer steps (and says so) instead of stalling the bench."""
    from oracle import audiogan_oracle as O
    if threads:
        torch.set_num_threads(threads)
    torch.manual_seed(0)
    if workload == 'c4':
        g = O.GRUGenerator(frame_size=FRAME, embed_size=100, noise_size=100, state_size=1024)
    else:
        g = O.Generator(frame_size=FRAME, embed_size=100, noise_size=100, state_size=1024)
    if workload == 'c2':
        d = O.Discriminator(state_size=1024, embed_size=100)
    else:
        d = O.Conv1DDiscriminator(config=[(16, 7, 2), (32, 7, 2), (64, 7, 2), (128, 7, 2), (256, 7, 2), (512, 7, 2)])
    og, od = O.make_optimizer(list(g.parameters()), opt_kind, 1e-4), O.make_optimizer(list(d.parameters()), opt_kind, 1e-4)
    b = synthetic_batch(batch, torch.device('cpu'), 0)
    stop = torch.zeros(batch, L // FRAME, dtype=torch.long)
    cores = torch.get_num_threads()
    times, t_start = [], time.perf_counter()
    for i in range(steps + warm):
        t0 = time.perf_counter()
        if workload == 'c4':
            O.c4_step(g, d, og, od, b['real'], b['c'], b['z'], b['noise_real'], b['noise_fake'], 1.0, 0.1, stop=stop)
        elif workload == 'c5':
            O.wgan_gp_step(g, d, og, od, b['real'], b['c'], b['z'], b['eps'], 10.0, stop=stop)
        else:
            O.d_step(g, d, od, b['real'], b['real_len'], b['c'], b['z'], b['noise_real'], b['noise_fake'], 1.0, stop=stop)
            O.g_step(g, d, og, b['c'], b['z'], b['noise_fake'], 0.1, stop=stop)
        times.append(time.perf_counter() - t0)
        sys.stderr.write('cpu_baseline: step %d of %d took %.1f s\n' % (i, steps + warm, times[-1]))
        sys.stderr.flush()
        if time.perf_counter() - t_start + times[-1] > budget_s:
            break
    nwarm = min(warm, len(times) - 1)                   # drop the warm-up steps when there are others
    timed = times[nwarm:]
    t = float(np.median(timed))
    short = '' if len(timed) >= steps else ' (time budget of %.0f s reached: %d of the %d timed steps asked for)' % (
        budget_s, len(timed), steps)
    return dict(value=batch * L / t, unit='audio-samples/sec', cores=cores, host_cores=os.cpu_count() or 0, kind='port',
                sample='%d of the 64 clips per step, %d timed step(s) after %d warm-up%s (median %.2f s/step), '
                       'torch %s CPU, %d threads, %s' % (batch, len(timed), nwarm, short, t, torch.__version__,
                                                         cores, opt_kind))


class _AltGraph(object):
    """the c4 / c5 step as one replayed hipGraph (same capture recipe as train.GraphedStep)"""

    def __init__(self, train, g, d, opt_g, opt_d, batch):
        from audiogan_amd import common, kernels as K
        self.losses, self.phases = {}, None
        K.reserve_table_arena()

        def whole():
            self.losses['d'], self.losses['g'] = one_step(train, g, d, opt_g, opt_d, batch)

        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        common.new_capture()
        with torch.cuda.graph(self.graph, capture_error_mode='thread_local'):
            whole()

    def step(self):
        self.graph.replay()
        return self.losses['d'], self.losses['g']


def _full_setup(M, optim_mod, dev, batch, opt_kind, seed=0):
    """the reference's CURRENT iteration at the C2 widths (audiogan.py:620-637: Generator, Discriminator, the two text
    Embedders; :690-694: one optimiser over g + e_g, one over d + e_d), on loader data: ``M`` = audiogan_amd or the oracle"""
    import types
    from audiogan_amd import dataset as D
    torch.manual_seed(seed)
    g = M.Generator(frame_size=FRAME, embed_size=100, noise_size=100, state_size=1024).to(dev)
    d = M.Discriminator(state_size=1024, embed_size=100).to(dev)
    e_g = M.Embedder(output_size=100, char_embed_size=50, num_layers=1, num_chars=256).to(dev)
    e_d = M.Embedder(output_size=100, char_embed_size=50, num_layers=1, num_chars=256).to(dev)
    opt_g = optim_mod.make_optimizer(list(g.parameters()) + list(e_g.parameters()), opt_kind, 1e-4)
    opt_d = optim_mod.make_optimizer(list(d.parameters()) + list(e_d.parameters()), opt_kind, 1e-4)
    words = ['word%02d' % i + 'x' * (i % 5) for i in range(40)]
    ds = D.SyntheticWordDataset(words, n_per_word=8, min_len=L // 2, max_len=L, kind='noise', seed=seed)
    a = types.SimpleNamespace(conditional=True, dataset=ds, minwordlen=1, subset=None, amplitudes=0)
    np.random.seed(seed)
    h5, maxlen, gen_train, _, keys_train, _ = D.dataloader(batch, a, maxlen=L, frame_size=FRAME)
    return (g, d, e_g, e_d, opt_g, opt_d), (D, h5, maxlen, gen_train, keys_train, a)


def run_full(args, dev):
    """``--workload full``: what audiogan.py executes per pass of its ``while True`` body (:703-940) - here with a DECLARED fixed
    critic_iter of 2 (one odd = FGSM critic iteration, one even = instance-noise critic iteration; the reference runs up to
    100 and stops once both accuracies pass 0.5, a data-dependent count) and gencatchup 1: adversarial movement of the
    generated clips (an input-gradient pass through D), adversarial z (a gradient pass through D and G), the feature-matching
    penalty over the critic's six activations, the two Embedder biLSTMs, the REINFORCE surrogate of the stop head; ragged
    real clips from the loader interface, fixed-length generated clips (stop='never': an untrained stop head would end every
    clip after a frame or two), RMSprop + per-parameter clip as the reference.
    ``--full-launch graph`` (default): the three iteration bodies captured once (loop.TrainLoop(graphed=True)) and replayed
    over static inputs, minibatches staged by train.Feeder behind the replays, accuracies / baseline kept on the device - with
    a fixed critic_iter nothing in the pass needs the host (tests/test_loop.py: bit-identical to the eager loop);
    ``--full-launch eager``: Python-issued launches with the host reads of the reference's iteration."""
    import audiogan_amd as A
    from audiogan_amd import kernels as K, loop, optim
    K.set_precision(args.dtype)
    mods, (D, h5, maxlen, gen_train, keys_train, a) = _full_setup(A, optim, dev, args.batch, 'rmsprop')
    g, d, e_g, e_d, opt_g, opt_d = mods
    pick = loop.words_picker(D, args.batch, maxlen, h5, keys_train, a, frame_size=FRAME)
    lp = loop.TrainLoop(g, d, e_g, e_d, opt_g, opt_d, gen_train, pick, args.batch, maxlen, dev, fixed_critic_iter=2,
                        gencatchup=1, stop='never', checkpoint_every=0, check=False, graphed=args.full_launch == 'graph')
    for _ in range(max(args.warmup, 1)):
        lp.outer()
    torch.cuda.synchronize()
    if lp.graphed:
        lp.gpu_timeline = []
    h0 = dict(lp.host_ms)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        lp.outer()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    host_ms = {k: ((lp.host_ms[k] - h0.get(k, 0.0)) / args.steps if not k.endswith('_max') else lp.host_ms[k]) for k in lp.host_ms}
    # the same pass with device time only (launches enqueued back to back inside one profiler window): how much of the wall
    # time is the host issuing ~3000 launches
    prof, dev_ms = {}, None
    if not lp.graphed:
        K.Profiler.start(only=None)
        lp.outer()
        prof = K.Profiler.stop()
        dev_ms = sum(v['ms'] for v in prof.values())
    status = K.lstm_persist_status(dev)
    finite = all(np.isfinite(v) for rec in lp.log for v in rec[2:])
    extra = {}
    if lp.graphed:
        # where a pass's wall time goes: the three replays alone (inputs left as they are), and the host's share (loader +
        # pick_words for three minibatches), which the Feeder overlaps with the replays
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(5):
            for k in ("d1", "d0", "g"):
                lp._graphs[k].replay()
        torch.cuda.synchronize()
        extra['replay_only_ms_per_step'] = (time.perf_counter() - t1) / 5 * 1e3
        t1 = time.perf_counter()
        for _ in range(6):
            lp._host_pair()
        extra['host_batches_ms_per_step'] = (time.perf_counter() - t1) / 2 * 1e3
        extra['host_ms_per_step'] = dict(host_ms, issue_total=t_issue / args.steps * 1e3)
        extra['feeder_host_ms_total'] = dict(lp._feeder.host_ms)
        evs = lp.gpu_timeline
        if evs:       # GPU duration of every replay and the idle time in front of it (last 12 replays)
            tl = []
            for (k0, a0, b0, h0_), (k1, a1, b1, h1_) in zip(evs[-13:-1], evs[-12:]):
                tl.append((k1, round(a1.elapsed_time(b1), 2), round(b0.elapsed_time(a1), 2), round((h1_ - h0_) * 1e3, 2)))
            extra['replay_timeline'] = {'columns': ['graph', 'gpu_ms', 'gpu_idle_before_ms', 'host_interval_ms'], 'rows': tl}
    if lp.graphed:       # the captured iterations keep their scalars on the device: read the last ones once, after the timed region
        ran, rd, rg = lp.outer()
        finite = finite and all(np.isfinite(float(v)) for v in (rd['loss'], rd['acc_d'], rd['acc_g'], rg['loss'],
                                                                 rg['feature_penalty'], rg['baseline']))
    ms = dt / args.steps * 1e3
    out = {
        'metric': 'audio-samples/sec per pass of the reference\'s training loop body (2 critic iterations + 1 generator iteration)',
        'value': args.batch * L / (dt / args.steps), 'unit': 'audio-samples/sec', 'n_gpus': 1, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': ms, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': args.dtype, 'data': 'synthetic',
        'config': {'workload': 'FULL: audiogan.py:703-940 loop body at the C2 widths, batch %d, ragged 4096-8192-sample loader '
                               'clips (SyntheticWordDataset through dataset.dataloader / pick_words), declared fixed critic_iter 2 '
                               '(odd = FGSM branch, even = instance noise) + gencatchup 1, both Embedders, adversarial z, feature '
                               'penalty, REINFORCE surrogate, stop=never, rmsprop, clip d=1 g=0.1' % args.batch,
                   'global_batch': args.batch, 'clip_len': L, 'parallelism': 'dp1',
                   'launch': ('three hipGraphs (odd / even critic iteration, generator iteration) replayed over static inputs; '
                              'loader batches uploaded and z / instance noise drawn between replays; no host read in the pass')
                   if lp.graphed else 'eager (host reads of accuracies / baseline every iteration, as the reference)'},
        'persist_status': status, 'losses_finite': bool(finite),
    }
    if dev_ms:
        out.update({'kernel_ms_per_step': dev_ms, 'launches_per_step': int(sum(v['n'] for v in prof.values())),
                    'kernel_table': [{'kernel': k, 'share': round(v['ms'] / dev_ms, 4), 'launches': v['n']}
                                     for k, v in sorted(prof.items(), key=lambda kv: -kv[1]['ms'])[:10]]})
    out.update({k: v for k, v in extra.items() if v is not None})
    if not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline_full(min(args.cpu_batch, 8), 'rmsprop', min(32, os.cpu_count() or 1))
    print(json.dumps(out), flush=True)
    if status != 0 or not finite:
        sys.exit(2)


def cpu_baseline_full(batch, opt_kind, threads):
    """the oracle's restatement of the same pass (O.d_step_full x 2 + O.g_step_full) on `batch` clips, once after one warm-up"""
    from oracle import audiogan_oracle as O
    if threads:
        torch.set_num_threads(threads)

    class _OptMod(object):
        make_optimizer = staticmethod(O.make_optimizer)
    (g, d, e_g, e_d, opt_g, opt_d), (D, h5, maxlen, gen_train, keys_train, a) = _full_setup(O, _OptMod, torch.device('cpu'), batch,
                                                                                       opt_kind)
    T = maxlen // FRAME
    maxchar = max(len(k) for k in keys_train)
    times = []
    gen = torch.Generator().manual_seed(0)
    for it in range(2):
        t0 = time.perf_counter()
        for dis_iter in (1, 2):
            _, _, samples, lengths, _, cseq, clen = next(gen_train)
            w2 = D.pick_words(batch, maxlen, h5, keys_train, maxchar, a, frame_size=FRAME, skip_samples=True)
            real = torch.from_numpy(np.ascontiguousarray(samples[:, :maxlen], dtype=np.float32))
            stop = torch.zeros(batch, T, dtype=torch.long)
            O.d_step_full(g, d, e_g, e_d, opt_d, dis_iter, real, torch.from_numpy(np.asarray(lengths)).long(),
                          torch.from_numpy(np.asarray(cseq)).long(), torch.from_numpy(np.asarray(clen)).long(),
                          torch.from_numpy(np.asarray(w2[1])).long(), torch.from_numpy(np.asarray(w2[2])).long(),
                          torch.randn(batch, T, 100, generator=gen), torch.randn(batch, maxlen, generator=gen) * 0.01,
                          torch.randn(batch, maxlen, generator=gen) * 0.01, 1.0, stop=stop)
        _, _, samples, lengths, _, _, _ = next(gen_train)
        w2 = D.pick_words(batch, maxlen, h5, keys_train, maxchar, a, frame_size=FRAME, skip_samples=True)
        real = torch.from_numpy(np.ascontiguousarray(samples[:, :maxlen], dtype=np.float32))
        stop = torch.zeros(batch, T, dtype=torch.long)
        O.g_step_full(g, d, e_g, e_d, opt_g, real, torch.from_numpy(np.asarray(lengths)).long(),
                      torch.from_numpy(np.asarray(w2[1])).long(), torch.from_numpy(np.asarray(w2[2])).long(),
                      torch.randn(batch, T, 100, generator=gen), torch.randn(batch, maxlen, generator=gen) * 0.01,
                      torch.randn(batch, maxlen, generator=gen) * 0.01, torch.randn(batch, maxlen, generator=gen) * 0.01, stop, stop, None)
        times.append(time.perf_counter() - t0)
        sys.stderr.write('cpu_baseline(full): pass %d took %.1f s\n' % (it, times[-1]))
    t = times[-1]
    return dict(value=batch * L / t, unit='audio-samples/sec', cores=torch.get_num_threads(), host_cores=os.cpu_count() or 0,
                kind='port', sample='%d clips per pass, 1 timed pass after 1 warm-up (%.2f s/pass), torch %s CPU, %d threads, %s'
                % (batch, t, torch.__version__, torch.get_num_threads(), opt_kind))


class _StdoutToStderr(object):
    """RCCL prints a version banner on fd 1 when a communicator is created; the bench's stdout must stay
    ONE JSON line, so fd 1 points at fd 2 while the process group comes up"""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=64, help='clips per GPU')
    ap.add_argument('--opt', default='adam', choices=['adam', 'rmsprop'],
                    help='adam = north_star; rmsprop = audiogan.py:693-694')
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16', 'f32x3'],
                    help='f32 = BASELINE configs[1] (the headline); bf16 = configs[2]: every contraction rounds its '
                         'operands to bfloat16 and accumulates in fp32, gradients cross ranks as bfloat16; f32x3 = an EXPERIMENT, never '
                         'the headline: the large GEMMs on three bf16 MFMAs per product of bf16 hi + lo operand parts')
    ap.add_argument('--full-launch', default='graph', choices=['eager', 'graph'],
                    help='--workload full only: Python-issued launches, or the three captured iteration graphs')
    ap.add_argument('--workload', default='c2', choices=['c2', 'c4', 'c5', 'full'],
                    help='c2 = the headline (BASELINE configs[1]; with --dtype bf16: configs[2]); c4 = GRU-front '
                         'generator + conv critic (configs[3]); c5 = WGAN-GP with the conv critic (configs[4]): extra '
                         'lines, never the headline; full = the reference\'s current loop body (FGSM passes, adversarial z, '
                         'feature penalty, Embedders, REINFORCE) at the C2 widths, eager: an extra line')
    ap.add_argument('--cpu-batch', type=int, default=64)
    ap.add_argument('--cpu-steps', type=int, default=5)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--force-phases', action='store_true',
                    help='use the multi-GPU code path (gradient buckets + 5 graphs per step) on one GPU')
    ap.add_argument('--no-overlap', action='store_true',
                    help='keep the G forward of the generator iteration on the main stream (default: on a second '
                         'stream beside the critic iteration; single-GPU graph only)')
    ap.add_argument('--no-graph', action='store_true',
                    help='enqueue every kernel from Python each step instead of replaying a hipGraph')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        raise SystemExit('--gpus %d but WORLD_SIZE is %d; launch with: python -m torch.distributed.run --nnodes=1 '
                         '--nproc-per-node %d --master-addr 127.0.0.1 --master-port 29500 bench.py --gpus %d ...'
                         % (args.gpus, world, args.gpus, args.gpus))
    assert torch.cuda.is_available(), 'bench.py needs a GPU (no CPU fallback)'
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    # AG_REHEARSE_RCCL=1 (with --force-phases): a 1-rank RCCL group whose all-reduces are really issued, to
    # rehearse the N>1 sequence (communicator set-up, async collective between graph replays) on one GPU
    rehearse = os.environ.get('AG_REHEARSE_RCCL') == '1' and world == 1
    if world > 1 or rehearse:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        with _StdoutToStderr():
            dist.init_process_group('nccl')
            # the communicator itself is created lazily by the first collective: do that here, not in the step
            t_ = torch.ones(1, device=dev)
            dist.all_reduce(t_)
            torch.cuda.synchronize()
        assert dist.get_world_size() == (args.gpus if world > 1 else 1), 'process group size != --gpus'
        assert int(t_.item()) == dist.get_world_size(), 'RCCL all-reduce did not see every rank'

    if args.workload == 'full':
        assert world == 1, '--workload full is a single-GPU line'
        return run_full(args, dev)

    import audiogan_amd as A
    from audiogan_amd import train, ddp, kernels as K

    K.set_precision(args.dtype)
    WORKLOAD[0] = args.workload
    alt = args.workload != 'c2'
    g, d, opt_g, opt_d = build_models(A, dev, args.opt, workload=args.workload)
    hook_d = hook_g = None
    multi = world > 1 or args.force_phases
    if multi:
        ddp.broadcast_parameters(g)
        ddp.broadcast_parameters(d)
        bd = ddp.GradBucket(list(d.parameters()), early=None if alt else d.early_params(), force_collective=rehearse,
                            comm_dtype='bf16' if args.dtype == 'bf16' else 'f32')
        bg = ddp.GradBucket(list(g.parameters()), early=g.early_params(), force_collective=rehearse,
                            comm_dtype='bf16' if args.dtype == 'bf16' else 'f32')
        opt_d.bucket, opt_g.bucket = bd, bg
        hook_d, hook_g = bd.all_reduce, bg.all_reduce
    batch = synthetic_batch(args.batch, dev, seed=1000 + rank)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # ---- warm-up; the first warm-up step doubles as the discovery pass for the roofline kernel
    dominant, disc = None, None
    for i in range(max(args.warmup, 1)):
        prof = (i == max(args.warmup, 1) - 1) and not args.no_roofline
        if prof:
            torch.cuda.synchronize()
            K.Profiler.start(only=None)
        one_step(train, g, d, opt_g, opt_d, batch, hook_d, hook_g)
        if prof:
            disc = K.Profiler.stop()
            dominant = max(disc.items(), key=lambda kv: kv[1]['ms'])[0]
    # ---- hipGraph capture: ~1000 launches per step would otherwise be paced by the Python
    # interpreter, not by the GPU.  One GPU: the whole G+D step is ONE graph.  N GPUs: five graphs
    # (critic fwd + head/biLSTM bwd | critic conv-stack bwd | opt_d | generator fwd+bwd | opt_g) with the RCCL
    # gradient all-reduces as ordinary stream operations between them.
    gs = None
    capture_error = None
    losses = {}
    if not args.no_graph and not (alt and multi):
        try:
            if alt:
                gs = _AltGraph(train, g, d, opt_g, opt_d, batch)
            else:
                gs = train.GraphedStep(g, d, opt_g, opt_d, batch, phased=multi, bucket_d=bd if multi else None,
                                       bucket_g=bg if multi else None, world=world, overlap=not args.no_overlap)
            gs.step()
            torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001
            sys.stderr.write('hipGraph capture failed (%s: %s); running eagerly\n' % (type(e).__name__, e))
            gs = None
            capture_error = '%s: %s' % (type(e).__name__, str(e)[:200])
            torch.cuda.synchronize()
    graph = gs if (gs is not None and gs.graph is not None) else None
    phases = gs.phases if (gs is not None and gs.graph is None) else None

    def run_step():
        if gs is not None:
            losses['d'], losses['g'] = gs.step()
        else:
            losses['d'], losses['g'] = one_step(train, g, d, opt_g, opt_d, batch, hook_d, hook_g)

    graphed = graph is not None or phases is not None
    # ---- timed region: exactly K steps between barrier+synchronize on both sides
    if dominant is not None and not graphed:
        K.Profiler.start(only=dominant)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_step()
    sync()
    dt = time.perf_counter() - t0
    rec = K.Profiler.stop() if (dominant is not None and not graphed) else {}
    # sticky status word of the persistent recurrent launches: non-zero = some launch of the warm-up or the timed region
    # gave up waiting for its group (its results are NaN) - the number above would then be meaningless
    persist_status = K.lstm_persist_status(dev)

    # ---- replay check (outside the timed region): one more replay of the captured step against one EAGER step from
    # the same parameters / optimiser state on the same batch; loss_d and loss_g must agree
    replay = None
    if graphed:
        from audiogan_amd import common
        nets = list(g.parameters()) + list(d.parameters())
        snap = [p_.detach().clone() for p_ in nets]
        sog, sod = opt_g.state_dict(), opt_d.state_dict()
        run_step()
        torch.cuda.synchronize()
        got = (float(losses['d']), float(losses['g']))
        with torch.no_grad():
            for p_, s_ in zip(nets, snap):
                p_.copy_(s_)
        common.bump_param_epoch(nets)
        opt_g.load_state_dict(sog)
        opt_d.load_state_dict(sod)
        ld, lg_ = one_step(train, g, d, opt_g, opt_d, batch, hook_d, hook_g)
        torch.cuda.synchronize()
        ref = (float(ld), float(lg_))
        rel = max(abs(a - b) / max(abs(b), 1e-12) for a, b in zip(got, ref))
        replay = dict(ok=bool(rel <= 2e-4 and all(np.isfinite(got))), max_rel_diff=rel,
                      loss_d=[got[0], ref[0]], loss_g=[got[1], ref[1]])
    persist_status |= K.lstm_persist_status(dev)          # ... and of the replay check's launches
    if graphed and dominant is not None:
        # per-kernel HIP-event timing is impossible inside a graph replay: time the dominant
        # kernel class in eager steps right after the timed region (same kernels, same shapes)
        K.Profiler.start(only=dominant)
        for _ in range(2):
            one_step(train, g, d, opt_g, opt_d, batch, hook_d, hook_g)
        rec = K.Profiler.stop()
        for r_ in rec.values():
            r_['per_step_div'] = 2
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * args.batch * L / (dt / args.steps)
        out = {
            'metric': 'audio-samples/sec per G+D train step (8 kHz, 8192-sample clips)',
            'value': value, 'unit': 'audio-samples/sec', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': ms, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': WORKLOAD_TEXT[args.workload] % (
                                       ('C2' if args.dtype == 'f32' else 'C2 with the large GEMMs as split-bf16 (3 MFMAs per '
                                        'product) - an experiment, not the headline' if args.dtype == 'f32x3' else
                                        'C3 (the C2 models with bf16 contractions)')
                                       if args.workload == 'c2' else args.workload.upper(), args.batch, args.opt) + (
                                       '' if args.dtype != 'bf16' else '; every contraction rounds both operands to '
                                       'bfloat16 and accumulates in fp32, gradient all-reduce in bfloat16'),
                       'global_batch': world * args.batch, 'clip_len': L,
                       'parallelism': 'dp%d' % world,
                       'launch': ('hipGraph replay (1 graph per step%s)' % ('' if (args.no_overlap or alt or
                                                                            g.front_is_persistent(args.batch, dev)) else
                                  '; generator forward of the G iteration on a second stream beside the critic '
                                  'iteration') if graph is not None else
                                  'hipGraph replay (6 graphs per step; D all-reduce overlaps the conv-stack backward, '
                                  "G trunk all-reduce overlaps the front's backward)"
                                  if phases is not None
                                  else 'eager' + (' (capture failed: %s)' % capture_error if capture_error else ''))},
        }
        out['persist_status'] = persist_status
        out['table_uploads_in_graph'] = dict(count=K.TABLE_STATS['uploads_in_capture'], kinds=K.TABLE_STATS['kinds'][:8])
        if replay is not None:
            out['replay_check'] = 'ok' if replay['ok'] else 'FAILED'
            out['replay_check_detail'] = {k: v for k, v in replay.items() if k != 'ok'}
        if world > 1 or rehearse:
            out['rccl_ranks'] = dist.get_world_size()
        if dominant is not None and dominant in rec:
            r = rec[dominant]
            avg_ms = r['ms'] / r['n']
            tf = r['flops'] / r['n'] / (avg_ms * 1e-3) / 1e12
            gbs = r['bytes'] / r['n'] / (avg_ms * 1e-3) / 1e9
            share = disc[dominant]['ms'] / sum(v['ms'] for v in disc.values())
            peak_tf = PEAK_BF16_MFMA_TFLOPS if 'bf16' in dominant else PEAK_F32_MFMA_TFLOPS
            mfma_bound = tf / peak_tf >= gbs / PEAK_HBM_GBS
            out['roofline'] = {
                'kernel': dominant, 'bound': 'mfma' if mfma_bound else 'hbm',
                'achieved': tf if mfma_bound else gbs,
                'peak': peak_tf if mfma_bound else PEAK_HBM_GBS,
                'unit': 'TFLOP/s' if mfma_bound else 'GB/s',
                'frac': (tf / peak_tf) if mfma_bound else (gbs / PEAK_HBM_GBS),
                'traffic': pmc_traffic(dominant, args.dtype), 'launches_per_step': r['n'] / r.get('per_step_div', args.steps),
                'avg_launch_us': avg_ms * 1e3,
                'share_of_gpu_time': share,
                'algorithmic_per_launch': {'flops': r['flops'] / r['n'], 'bytes': r['bytes'] / r['n']},
            }
            # SURVEY §8(d) grades the conv stacks (G trunk + D stack) as a whole as well: every conv kernel
            # class of the discovery step together (forward, backward-data, backward-weight)
            cv = [v for k, v in disc.items() if k.startswith('conv_')]
            if cv:
                cms = sum(v['ms'] for v in cv)
                ctf = sum(v['flops'] for v in cv) / (cms * 1e-3) / 1e12
                out['conv_stack'] = {'bound': 'mfma', 'achieved': ctf, 'peak': PEAK_F32_MFMA_TFLOPS,
                                     'unit': 'TFLOP/s', 'frac': ctf / PEAK_F32_MFMA_TFLOPS,
                                     'gpu_ms_per_step': cms, 'flops_per_step': sum(v['flops'] for v in cv),
                                     'alg_gbs': sum(v['bytes'] for v in cv) / (cms * 1e-3) / 1e9,
                                     'share_of_gpu_time': cms / sum(v['ms'] for v in disc.values())}
            # the dense products as a class (Linear / LSTM input projections / their gradients): every fp32 GEMM launch of
            # the step together, and the heaviest instantiation (a row of profiles/*_bench_kernel_stats.csv) by itself
            gm = {k: v for k, v in disc.items() if k.startswith(('gemm_tile_kernel', 'gemm_kernel', 'gemm_bf16'))}
            if gm:
                gms = sum(v['ms'] for v in gm.values())
                gtf = sum(v['flops'] for v in gm.values()) / (gms * 1e-3) / 1e12
                gpeak = PEAK_BF16_MFMA_TFLOPS if args.dtype == 'bf16' else PEAK_F32_MFMA_TFLOPS
                hk, hv = max(gm.items(), key=lambda kv: kv[1]['ms'])
                out['gemm_class'] = {'bound': 'mfma', 'achieved': gtf, 'peak': gpeak, 'unit': 'TFLOP/s', 'frac': gtf / gpeak,
                                     'gpu_ms_per_step': gms, 'launches_per_step': sum(v['n'] for v in gm.values()),
                                     'share_of_gpu_time': gms / sum(v['ms'] for v in disc.values()),
                                     'heaviest': {'kernel': hk, 'launches_per_step': hv['n'],
                                                  'avg_launch_us': hv['ms'] / hv['n'] * 1e3,
                                                  'achieved': hv['flops'] / (hv['ms'] * 1e-3) / 1e12,
                                                  'frac': hv['flops'] / (hv['ms'] * 1e-3) / 1e12 / gpeak}}
            tot = sum(x['ms'] for x in disc.values())
            out['kernel_table'] = [
                {'kernel': k, 'share': round(v['ms'] / tot, 4), 'launches': v['n'],
                 'avg_us': round(v['ms'] / v['n'] * 1e3, 2),
                 'tflops': round(v['flops'] / (v['ms'] * 1e-3) / 1e12, 2) if v['flops'] else None,
                 'alg_gbs': round(v['bytes'] / (v['ms'] * 1e-3) / 1e9, 1) if v['bytes'] else None}
                for k, v in sorted(disc.items(), key=lambda kv: -kv[1]['ms'])[:14]]
        if world == 1 and not args.no_cpu_baseline:
            # torch CPU does not scale past ~32 threads on this model (128 threads measured slower
            # than 8): use min(32, cores); `cores` in the result is the thread count actually used, `host_cores` what the box has
            out['cpu_baseline'] = cpu_baseline(args.cpu_batch if not alt else min(args.cpu_batch, 16),
                                               args.cpu_steps if not alt else min(args.cpu_steps, 3), args.opt,
                                               min(32, os.cpu_count() or 1), workload=args.workload)
        print(json.dumps(out), flush=True)
    if world > 1 or rehearse:
        dist.barrier()
        dist.destroy_process_group()
    if persist_status != 0:
        sys.stderr.write('a persistent recurrent launch timed out (status 0x%08x): results invalid\n' % persist_status)
        sys.exit(2)
    if replay is not None and not replay['ok']:
        sys.stderr.write('replay check FAILED: %r\n' % (replay,))
        sys.exit(1)


if __name__ == '__main__':
    main()
