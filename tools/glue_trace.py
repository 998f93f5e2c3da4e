"""where does the step's glue come from?  One eager C2 step (batch 64) under torch.profiler with Python stacks: every
device kernel that is NOT one of the library's contraction kernels, grouped by the audiogan_amd source line that launched
it (python tools/glue_trace.py [bf16] > gpurun_out/glue.txt)"""
import collections
import sys
import torch
sys.path.insert(0, '.')
import audiogan_amd as A
from audiogan_amd import kernels as K, train
import bench

prec = sys.argv[1] if len(sys.argv) > 1 else 'f32'
K.set_precision(prec)
dev = torch.device('cuda')
g, d, opt_g, opt_d = bench.build_models(A, dev, 'adam')
b = bench.synthetic_batch(64, dev, 0)
for _ in range(2):
    bench.one_step(train, g, d, opt_g, opt_d, b)
torch.cuda.synchronize()

from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    bench.one_step(train, g, d, opt_g, opt_d, b)
    torch.cuda.synchronize()


def where(stack):
    for fr in stack:
        if 'audiogan_amd/' in fr or 'bench.py' in fr:
            return fr.split('audiogan_amd/')[-1].strip()
    return '?'


rows = collections.defaultdict(lambda: [0, 0.0])
big = ('gemm_', 'conv_engine', 'conv_wgrad', 'lstm_persist', 'gfront_persist', 'conv_o1', 'conv_c1')
tot_all = 0.0
for e in prof.events():
    ks = getattr(e, 'kernels', None)
    if not ks:
        continue
    for k in ks:
        tot_all += k.duration
        if any(k.name.startswith(p) or ('void ' + p) in k.name for p in big):
            continue
        key = (where(e.stack or []), e.name, k.name.split('(')[0][:60])
        rows[key][0] += 1
        rows[key][1] += k.duration
tot = sum(v[1] for v in rows.values())
print('glue kernels: %d launches, %.1f us of %.1f us device time in one eager step' % (sum(v[0] for v in rows.values()), tot, tot_all))
for key, v in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    print('%4d  %8.1f us  %-44s %-22s %s' % (v[0], v[1], key[0][:44], key[1][:22], key[2]))
