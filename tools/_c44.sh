cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py -m gpu -q -x -k "front or gru or persist or generator or step" 2>&1 | tail -2
for a in "" "--workload c4"; do python bench.py --steps 20 --warmup 3 --no-cpu-baseline $a 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('bench $a', d['ms_per_step'], d.get('replay_check'), d.get('persist_status')); [print('  ',k['kernel'], k['launches'], k['avg_us']) for k in d['kernel_table'] if 'front' in k['kernel']]"; done
