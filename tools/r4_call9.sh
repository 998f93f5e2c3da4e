cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_loop.py tests/test_gpu_modules.py -m gpu -q -x -k "loop or captured or full_batch" > gpurun_out/r4_t9.log 2>&1; echo "tests rc $?"; tail -15 gpurun_out/r4_t9.log
timeout -k 10 200 python bench.py --workload full --full-launch graph --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r4_full_graph.json 2> gpurun_out/r4_full_graph.err || tail -15 gpurun_out/r4_full_graph.err
python -c "import json,sys; d=json.load(open('gpurun_out/r4_full_graph.json')); print(d['ms_per_step'], d.get('replay_only_ms_per_step'), d.get('host_batches_ms_per_step'), d['persist_status'], d['losses_finite'])"
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r4_prof_full -o stats -- python $GRAFT_REPO_ROOT/bench.py --workload full --full-launch graph --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r4_prof_full.log 2>&1; echo "prof rc $?"
