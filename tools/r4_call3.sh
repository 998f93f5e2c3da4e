# round 4, GPU call 3: suite with bf16 storage (stage 1: the critic's sequence path), fp32 + bf16 benches (storage on / off), diag
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q --maxfail=15 > gpurun_out/r4_t3.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4_t3.log
tail -30 gpurun_out/r4_t3.log
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r4_bench3.json 2> gpurun_out/r4_bench3.err || tail -5 gpurun_out/r4_bench3.err
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --dtype bf16 > gpurun_out/r4_bench3_bf16.json 2> gpurun_out/r4_bench3_bf16.err || tail -5 gpurun_out/r4_bench3_bf16.err
AG_BF16_STORE=0 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --dtype bf16 > gpurun_out/r4_bench3_bf16_nostore.json 2> gpurun_out/r4_bench3_bf16_nostore.err || tail -5 gpurun_out/r4_bench3_bf16_nostore.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4_prof3_bf16 -o stats -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --dtype bf16 > gpurun_out/r4_prof3_bf16.log 2>&1
timeout -k 10 600 python tools/diag_grad.py > gpurun_out/r4_diag_grad.txt 2> gpurun_out/r4_diag_grad.err
for f in r4_bench3 r4_bench3_bf16 r4_bench3_bf16_nostore; do python -c "import json,sys; d=json.load(open('gpurun_out/$f.json')); print('$f', d['ms_per_step'], d.get('replay_check'), d.get('persist_status'), d['roofline']['kernel'], round(d['roofline']['frac'],3))"; done
