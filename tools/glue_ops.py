"""which torch (aten) ops does one canonical step issue OUTSIDE the library's kernels?  Runs on the CPU: the package's host
logic over tests/kernel_model.py (the torch model of the C ABI) under a TorchDispatchMode that logs every aten op whose
Python stack does not pass through the kernel model, grouped by the audiogan_amd source line that issued it.  Every op
listed that touches tensor data is one (or more) device launches / graph nodes per step on the GPU.

    python tools/glue_ops.py [c2|c4|c5|full] [small]        (full: one pass of the reference's loop body, loop.TrainLoop)
"""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

NO_LAUNCH = ('aten.view', 'aten._unsafe_view', 'aten.as_strided', 'aten.slice', 'aten.select', 'aten.transpose', 'aten.t.',
             'aten.expand', 'aten.unsqueeze', 'aten.squeeze', 'aten.detach', 'aten.alias', 'aten.empty', 'aten.permute',
             'aten.unbind', 'aten.split', 'aten._reshape_alias', 'aten.reshape', 'aten.empty_like', 'aten.new_empty',
             'aten.is_same_size', 'aten.stride', 'aten.size', 'aten.numel', 'aten.empty_strided', 'aten.lift_fresh',
             'aten._local_scalar_dense')


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.rows = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(NO_LAUNCH):
            st = traceback.extract_stack()
            if not any('kernel_model.py' in f.filename for f in st):
                where = '?'
                for f in reversed(st):
                    if '/audiogan_amd/' in f.filename or f.filename.endswith('bench.py'):
                        where = '%s:%d %s' % (os.path.basename(f.filename), f.lineno, f.name)
                        break
                self.rows[(where, name)] += 1
        return func(*args, **(kwargs or {}))


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] in ('c2', 'c4', 'c5', 'full') else 'c2'
    small = 'small' in sys.argv

    class MP(object):
        def setattr(self, obj, name, val):
            setattr(obj, name, val)

    from tests import kernel_model
    kernel_model.install(MP())
    import audiogan_amd as A
    from audiogan_amd import train
    import bench
    if wl == 'full':
        from audiogan_amd import loop, optim
        torch.set_num_threads(8)
        dev = torch.device('cpu')
        B = 2
        mods, (D, h5, maxlen, gen_train, keys_train, a) = bench._full_setup(A, optim, dev, B, 'rmsprop')
        g, d, e_g, e_d, opt_g, opt_d = mods
        pick = loop.words_picker(D, B, maxlen, h5, keys_train, a, frame_size=bench.FRAME)
        lp = loop.TrainLoop(g, d, e_g, e_d, opt_g, opt_d, gen_train, pick, B, maxlen, dev, fixed_critic_iter=2, gencatchup=1,
                            stop='never', checkpoint_every=0, check=False, host=False)
        lp.outer()
        with Log() as lg:
            lp.outer()
        tot = sum(lg.rows.values())
        print('%d aten ops outside the kernels in one eager pass of the full loop body' % tot)
        by = collections.Counter()
        for (where, name), n in lg.rows.items():
            by[where] += n
        for where, n in by.most_common(60):
            print('%4d  %s' % (n, where))
        return
    bench.WORKLOAD[0] = wl
    if small:
        bench.L, bench.FRAME = 1024, 256
    dev = torch.device('cpu')
    g, d, og, od = bench.build_models(A, dev, 'adam', workload=wl)
    b = bench.synthetic_batch(2, dev, 0)
    for _ in range(2):
        bench.one_step(train, g, d, og, od, b)
    with Log() as lg:
        bench.one_step(train, g, d, og, od, b)
    tot = sum(lg.rows.values())
    print('%d aten ops outside the kernels in one eager %s step' % (tot, wl))
    for (where, name), n in sorted(lg.rows.items(), key=lambda kv: (kv[0][0].split(':')[0], int(kv[0][0].split(':')[1].split()[0]) if ':' in kv[0][0] else 0)):
        print('%3d  %-46s %s' % (n, where, name))


if __name__ == '__main__':
    main()
