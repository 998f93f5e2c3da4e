cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python tools/diag_bf16_xgrad.py > gpurun_out/r4_diag_bf16_xgrad.txt 2> gpurun_out/r4_diag_bf16_xgrad.err; echo "diag rc $?"; cat gpurun_out/r4_diag_bf16_xgrad.txt
timeout -k 10 400 python -m pytest tests/test_extras.py tests/test_full_step.py tests/test_loop.py -m gpu -q -x > gpurun_out/r4_t25.log 2>&1; echo "tests rc $?"; tail -3 gpurun_out/r4_t25.log
timeout -k 10 200 python bench.py --workload full --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r4_full_graph3.json 2> gpurun_out/r4_full_graph3.err || tail -15 gpurun_out/r4_full_graph3.err
python -c "import json; d=json.load(open('gpurun_out/r4_full_graph3.json')); print('full graph', d['ms_per_step'], d.get('replay_only_ms_per_step'), d.get('replay_timeline', {}).get('rows', [])[-3:])"
timeout -k 10 200 python bench.py --workload full --full-launch eager --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r4_full_eager3.json 2> gpurun_out/r4_full_eager3.err || tail -5 gpurun_out/r4_full_eager3.err
python -c "import json,sys; d=json.load(open('gpurun_out/r4_full_eager3.json')); print('eager', d['ms_per_step'], d['kernel_ms_per_step'], d['launches_per_step'])"
