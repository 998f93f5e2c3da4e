"""which small launches does a replayed step still contain?  Reads a rocprofv3 --kernel-trace CSV of `bench.py` and lists,
for one replayed step, every launch under 30 us with the launch that follows it (python tools/small_launches.py
gpurun_out/prof_stats/stats_kernel_trace.csv)"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], int(r['Grid_Size_X'])) for r in rows)
names = [e[2] for e in ev]
idx = [i for i, n in enumerate(names) if n.startswith('opt_step')]
a, b = idx[-9], idx[-7]                      # one replayed step: between two optimiser launches of the critic
seg = ev[a + 1:b + 1]
short = lambda n: re.sub(r'\(.*$', '', re.sub(r'^void ', '', n))[:60]      # noqa: E731
print('launches in one replayed step: %d, wall %.3f ms' % (len(seg), (seg[-1][1] - seg[0][0]) / 1e6))
cnt, tot = collections.Counter(), collections.Counter()
for i, (s, e, n, g) in enumerate(seg):
    if e - s < 30000:
        cnt[short(n)] += 1
        tot[short(n)] += e - s
        if len(sys.argv) > 2:
            print('%3d %-62s grid %8d %5.1f us -> %s' % (i, short(n), g, (e - s) / 1e3, short(seg[i + 1][2]) if i + 1 < len(seg) else ''))
print('small launches: %d, %.2f ms' % (sum(cnt.values()), sum(tot.values()) / 1e6))
for k, v in cnt.most_common():
    print('  %-62s %3d  %7.1f us' % (k, v, tot[k] / 1e3))
