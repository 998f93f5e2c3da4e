"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md prescribes)
of `python bench.py --steps 2 --warmup 1 --no-graph --no-roofline --no-cpu-baseline` into
profiles/rNN_pmc_traffic.json: fabric bytes per launch for every kernel class.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -o f -- python bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -o w -- python bench.py ...
    python tools/pmc_traffic.py gpurun_out/pmc_f/f_counter_collection.csv gpurun_out/pmc_w/w_counter_collection.csv
"""
import collections
import csv
import json
import re
import sys


def short(name):
    n = re.sub(r'^void ', '', name)
    n = re.sub(r'\(.*$', '', n)          # drop the argument list
    return n.replace(' ', '')


def load(path, counter):
    tot, cnt = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        k = short(r['Kernel_Name'])
        for key in {k, k.split('<')[0]}:      # the instantiation and the kernel class
            tot[key] += float(r['Counter_Value'])
            cnt[key] += 1
    return tot, cnt


def main():
    f_tot, f_cnt = load(sys.argv[1], 'FETCH_SIZE')
    w_tot, w_cnt = load(sys.argv[2], 'WRITE_SIZE')
    out = {'_note': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over `python bench.py --steps 2 '
                    '--warmup 1 --no-graph --no-roofline`; values are per launch, averaged over all launches of the '
                    'kernel in the step mix. bytes = 1024 * (2 * FETCH_SIZE + WRITE_SIZE): on gfx950 FETCH_SIZE reports '
                    'half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact. '
                    'Counts fabric requests, Infinity-Cache hits included.'}
    for k in sorted(f_tot):
        n = f_cnt[k]
        fk = f_tot[k] / n
        wk = w_tot.get(k, 0.0) / max(w_cnt.get(k, 1), 1)
        out[k] = {'launches': n, 'fetch_kb': round(fk, 1), 'write_kb': round(wk, 1),
                  'bytes_per_launch': int(1024 * (2 * fk + wk))}
    dst = sys.argv[3] if len(sys.argv) > 3 else 'profiles/r04_pmc_traffic.json'
    json.dump(out, open(dst, 'w'), indent=1)
    print('wrote %s with %d kernels' % (dst, len(out) - 1))


if __name__ == '__main__':
    main()
