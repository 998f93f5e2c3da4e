"""N>1 path on CPU: 2 and 4 processes, gloo backend, kernels replaced by the torch model.  N ranks that
each see 1/N of the batch must end up with the same parameters as one process that sees all of it
(flat gradient bucket -> sum all-reduce per network -> 1/world folded into the optimiser), also on the
phase-split path of bench.py (critic: heads/biLSTM early, conv stack late; generator: conv trunk early,
recurrent front late, each early part all-reduced asynchronously beside the late backward)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _setup_models():
    from oracle import audiogan_oracle as O  # noqa: F401
    import audiogan_amd as A
    torch.manual_seed(5)
    g = A.Generator(16, 6, 5, 24, 1, struct=[[9, 4, 8, 4], [9, 4, 8, 4]])
    d = A.Discriminator(16, 6, 1, cnn_struct=[[7, 2, 4], [7, 2, 8]])
    return A, g, d


def _batch(B):
    gen = torch.Generator().manual_seed(9)
    T, fs = 4, 16
    return dict(real=torch.randn(B, T * fs, generator=gen), real_len=torch.full((B,), T * fs, dtype=torch.long),
                c=torch.randn(B, 6, generator=gen), z=torch.randn(B, T, 5, generator=gen),
                nr=torch.randn(B, T * fs, generator=gen) * 0.01, nf=torch.randn(B, T * fs, generator=gen) * 0.01)


def _install_model():
    import audiogan_amd.kernels as K
    from tests import kernel_model as KM
    for n in KM.ALL:
        setattr(K, n, getattr(KM, n))


def _steps(A, g, d, b, buckets=False):
    from audiogan_amd import optim, train, ddp
    og, od = optim.RMSprop(list(g.parameters()), lr=1e-3), optim.RMSprop(list(d.parameters()), lr=1e-3)
    hd = hg = None
    if buckets:
        ddp.broadcast_parameters(g); ddp.broadcast_parameters(d)
        bd = ddp.GradBucket(list(d.parameters()), early=d.early_params())
        bg = ddp.GradBucket(list(g.parameters()), early=g.early_params())
        assert 0 < bd.n_early < bd.flat.numel() and 0 < bg.n_early < bg.flat.numel()
        od.bucket, og.bucket = bd, bg
        hd, hg = bd.all_reduce, bg.all_reduce
    for it in range(2):
        if buckets and it == 1:
            # the phase-split critic iteration of bench.py's multi-GPU path: all-reduce of the heads /
            # biLSTM gradients issued asynchronously before the conv-stack backward runs
            keep = {}
            train.d_backward_early(g, d, od, b['real'], b['real_len'], b['c'], b['z'], b['nr'], b['nf'], keep)
            bd.all_reduce(async_op=True, part='early')
            train.d_backward_late(keep)
            bd.wait()
            scale = bd.all_reduce(part='late')
            od.step(clip_norm=1.0, grad_scale=scale)
            gkeep = {}
            train.g_backward_early(g, d, og, b['c'], b['z'], b['nf'], gkeep)
            bg.all_reduce(async_op=True, part='early')
            train.g_backward_late(gkeep)
            bg.wait()
            og.step(clip_norm=0.1, grad_scale=bg.all_reduce(part='late'))
        else:
            train.d_step(g, d, od, b['real'], b['real_len'], b['c'], b['z'], b['nr'], b['nf'], 1.0, grad_hook=hd)
            train.g_step(g, d, og, b['c'], b['z'], b['nf'], 0.1, grad_hook=hg)
        if buckets:
            assert bd.check_views() and bg.check_views(), 'autograd must accumulate into the flat bucket'
    return {k: v.clone() for k, v in list(g.state_dict().items()) + list(d.state_dict().items())}


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import warnings
    warnings.filterwarnings('ignore')
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    _install_model()
    A, g, d = _setup_models()
    if rank == 1:      # rank 1 starts from different weights: broadcast must fix that
        with torch.no_grad():
            for p in list(g.parameters()) + list(d.parameters()):
                p.add_(0.1)
    full = _batch(4)
    per = 4 // world
    half = {k: v[rank * per:(rank + 1) * per] for k, v in full.items()}
    sd = _steps(A, g, d, half, buckets=True)
    torch.save(sd, os.path.join(out, 'rank%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.parametrize('world', [2, 4])
def test_ranks_equal_one_process(tmp_path, world):
    port = 29500 + (os.getpid() % 2000) + world
    mp.start_processes(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method='spawn')
    r0 = torch.load(os.path.join(tmp_path, 'rank0.pt'))
    for r in range(1, world):
        rr = torch.load(os.path.join(tmp_path, 'rank%d.pt' % r))
        for k in r0:
            np.testing.assert_array_equal(r0[k].numpy(), rr[k].numpy(), err_msg='ranks diverged: ' + k)
    sys.path.insert(0, ROOT)
    _install_model_local = _install_model
    import audiogan_amd.kernels as K
    saved = {n: getattr(K, n) for n in dir(K)}
    try:
        _install_model_local()
        A, g, d = _setup_models()
        ref = _steps(A, g, d, _batch(4), buckets=False)
    finally:
        for n, v in saved.items():
            setattr(K, n, v)
    for k in ref:
        if k.split('.')[-1].startswith('bias') and k.endswith('_v'):
            continue   # rounding-noise gradients, see DESIGN.md section 2
        np.testing.assert_allclose(r0[k].numpy(), ref[k].numpy(), rtol=2e-3, atol=2e-5, err_msg=k)
