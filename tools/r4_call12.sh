cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
LAB_LIB_ONLY=1 ./tools/gemm_lab > gpurun_out/r4_gemm_lab2.txt 2>&1; grep -v "^peak" gpurun_out/r4_gemm_lab2.txt
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "gemm" > gpurun_out/r4_t12.log 2>&1; echo "gemm tests rc $?"; tail -5 gpurun_out/r4_t12.log
python tools/prof_gemm.py > gpurun_out/r4_gemm_f32_tiles.txt 2>&1; grep -v amdgpu.ids gpurun_out/r4_gemm_f32_tiles.txt
AG_GEMM_TILE=0 python tools/prof_gemm.py > gpurun_out/r4_gemm_f32_tile0.txt 2>&1; grep -v amdgpu.ids gpurun_out/r4_gemm_f32_tile0.txt
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r4_bench12.json 2> gpurun_out/r4_bench12.err || tail -5 gpurun_out/r4_bench12.err
python -c "import json,sys; d=json.load(open('gpurun_out/r4_bench12.json')); print('bench12', d['ms_per_step'], d.get('replay_check'), d.get('persist_status'), d['roofline']['kernel'], round(d['roofline']['frac'],3), d['roofline']['avg_launch_us']); [print('  ',k) for k in d['kernel_table'][:8]]"
timeout -k 10 500 python -m pytest tests/test_gpu_modules.py tests/test_full_step.py -m gpu -q --maxfail=5 > gpurun_out/r4_t12b.log 2>&1; echo "module tests rc $?"; tail -8 gpurun_out/r4_t12b.log
