cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/prof_gemm.py > gpurun_out/r4_gemm_f32_cap4.txt 2>&1; grep -v amdgpu.ids gpurun_out/r4_gemm_f32_cap4.txt
AG_GEMM_LDS_PAD=21000 python tools/prof_gemm.py > gpurun_out/r4_gemm_f32_cap3.txt 2>&1; grep -v amdgpu.ids gpurun_out/r4_gemm_f32_cap3.txt
AG_GEMM_LDS_PAD=8000 python tools/prof_gemm.py > gpurun_out/r4_gemm_f32_pad8k.txt 2>&1; grep -v amdgpu.ids gpurun_out/r4_gemm_f32_pad8k.txt
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r4_bench7.json 2> gpurun_out/r4_bench7.err || tail -5 gpurun_out/r4_bench7.err
AG_GEMM_LDS_PAD=21000 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r4_bench7_cap3.json 2> gpurun_out/r4_bench7_cap3.err || tail -5 gpurun_out/r4_bench7_cap3.err
for f in r4_bench7 r4_bench7_cap3; do python -c "import json,sys; d=json.load(open('gpurun_out/$f.json')); print('$f', d['ms_per_step'], d.get('replay_check'), d.get('persist_status'), d['roofline']['kernel'], round(d['roofline']['frac'],3), d['roofline']['avg_launch_us'])"; done
timeout -k 10 600 python -m pytest tests/test_gpu_modules.py tests/test_gpu_kernels.py -m gpu -q --maxfail=10 -k "gemm or full_batch" > gpurun_out/r4_t7.log 2>&1; tail -5 gpurun_out/r4_t7.log
