cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
LAB_DBG=1 ./tools/gemm_lab > gpurun_out/r4_gemm_lab4.txt 2>&1; grep -v "^peak" gpurun_out/r4_gemm_lab4.txt
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "gemm" > gpurun_out/r4_t14.log 2>&1; echo "gemm tests rc $?"; tail -3 gpurun_out/r4_t14.log
python tools/prof_gemm.py > gpurun_out/r4_gemm_f32_tiles2.txt 2>&1; grep -v amdgpu.ids gpurun_out/r4_gemm_f32_tiles2.txt
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r4_bench14.json 2> gpurun_out/r4_bench14.err || tail -5 gpurun_out/r4_bench14.err
python -c "import json,sys; d=json.load(open('gpurun_out/r4_bench14.json')); print('bench14', d['ms_per_step'], d.get('replay_check'), d.get('persist_status'), d['roofline']['kernel'], round(d['roofline']['frac'],3), d['roofline']['avg_launch_us']); [print('  ',k) for k in d['kernel_table'][:4]]"
AG_LOOP_EXP=ev timeout -k 10 200 python bench.py --workload full --full-launch graph --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r4_full_graph_ev.json 2> gpurun_out/r4_full_graph_ev.err || tail -15 gpurun_out/r4_full_graph_ev.err
python -c "
import json; d=json.load(open('gpurun_out/r4_full_graph_ev.json')); print('ev', d['ms_per_step'], d.get('replay_only_ms_per_step'), d.get('host_ms_per_step'), d.get('feeder_host_ms_total'))
for k,v in d.items():
    if k.startswith('replay_timeline'): print(k); [print('   ', x) for x in v]
"
