cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_loop.py -m gpu -q -x > gpurun_out/r4_t8a.log 2>&1; echo "loop rc $?"; tail -15 gpurun_out/r4_t8a.log
timeout -k 10 200 python bench.py --workload full --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r4_full_eager.json 2> gpurun_out/r4_full_eager.err || tail -5 gpurun_out/r4_full_eager.err
timeout -k 10 200 python bench.py --workload full --full-launch graph --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r4_full_graph.json 2> gpurun_out/r4_full_graph.err || tail -15 gpurun_out/r4_full_graph.err
for f in r4_full_eager r4_full_graph; do python -c "import json,sys; d=json.load(open('gpurun_out/$f.json')); print('$f', d['ms_per_step'], d['kernel_ms_per_step'], d['launches_per_step'], d['persist_status'], d['losses_finite'])"; done
timeout -k 10 700 python -m pytest tests -m gpu -q --maxfail=10 > gpurun_out/r4_t8.log 2>&1; echo "pytest rc $?"; tail -25 gpurun_out/r4_t8.log
