cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_bf16.py -m gpu -q -x > gpurun_out/r4_t20.log 2>&1; echo "bf16 tests rc $?"; tail -4 gpurun_out/r4_t20.log
timeout -k 10 300 python tools/prof_gemm_h.py > gpurun_out/r4_gemmh_tile.txt 2>&1; grep -v amdgpu.ids gpurun_out/r4_gemmh_tile.txt | tail -14
python bench.py --dtype bf16 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r4_bench20_bf16.json 2> gpurun_out/r4_bench20_bf16.err || tail -5 gpurun_out/r4_bench20_bf16.err
python -c "import json,sys; d=json.load(open('gpurun_out/r4_bench20_bf16.json')); print('bench20 bf16', d['ms_per_step'], d.get('replay_check'), d.get('persist_status'), d['roofline']['kernel'], d.get('gemm_class')); [print('  ',k) for k in d['kernel_table'][:12]]"
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r4_bench20.json 2> gpurun_out/r4_bench20.err || tail -5 gpurun_out/r4_bench20.err
python -c "import json,sys; d=json.load(open('gpurun_out/r4_bench20.json')); print('bench20', d['ms_per_step'], d.get('replay_check'), d.get('persist_status'), d['roofline'], d.get('gemm_class')); [print('  ',k) for k in d['kernel_table'][:12]]"
