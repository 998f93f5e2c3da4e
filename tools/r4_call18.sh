cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/prof_gemm.py > gpurun_out/r4_gemm_f32_tiles4.txt 2>&1; grep -v amdgpu.ids gpurun_out/r4_gemm_f32_tiles4.txt
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r4_bench18.json 2> gpurun_out/r4_bench18.err || tail -5 gpurun_out/r4_bench18.err
python -c "import json,sys; d=json.load(open('gpurun_out/r4_bench18.json')); print('bench18', d['ms_per_step'], d.get('replay_check'), d.get('persist_status'), d['roofline']['kernel'], round(d['roofline']['frac'],3), d['roofline']['avg_launch_us']); [print('  ',k) for k in d['kernel_table'][:4]]"
python bench.py --dtype bf16 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r4_bench18_bf16.json 2> gpurun_out/r4_bench18_bf16.err || tail -5 gpurun_out/r4_bench18_bf16.err
python -c "import json,sys; d=json.load(open('gpurun_out/r4_bench18_bf16.json')); print('bench18 bf16', d['ms_per_step'], d.get('replay_check'), d.get('persist_status'))"
timeout -k 10 200 python bench.py --workload full --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r4_full_graph2.json 2> gpurun_out/r4_full_graph2.err || tail -15 gpurun_out/r4_full_graph2.err
python -c "import json; d=json.load(open('gpurun_out/r4_full_graph2.json')); print('full graph', d['ms_per_step'], d.get('replay_only_ms_per_step'), d.get('host_ms_per_step'))"
timeout -k 10 900 python -m pytest tests -m gpu -q --maxfail=10 > gpurun_out/r4_t18.log 2>&1; echo "pytest rc $?"; tail -12 gpurun_out/r4_t18.log
