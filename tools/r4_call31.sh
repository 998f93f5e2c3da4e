cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r4_bench31.json 2> gpurun_out/r4_bench31.err || tail -5 gpurun_out/r4_bench31.err
python -c "import json,sys; d=json.load(open('gpurun_out/r4_bench31.json')); print('bench31', d['ms_per_step'], d.get('replay_check'), d.get('persist_status'), d['conv_stack']['frac'], d['conv_stack']['gpu_ms_per_step'])"
AG_CONV_SOLO=0 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r4_bench31b.json 2> gpurun_out/r4_bench31b.err || tail -5 gpurun_out/r4_bench31b.err
python -c "import json,sys; d=json.load(open('gpurun_out/r4_bench31b.json')); print('bench31 solo off', d['ms_per_step'], d.get('replay_check'), d.get('persist_status'), d['conv_stack']['frac'], d['conv_stack']['gpu_ms_per_step'])"
timeout -k 10 900 python -m pytest tests -m gpu -q --maxfail=10 > gpurun_out/r4_t31.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r4_t31.log
