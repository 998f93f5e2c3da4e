cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q --maxfail=15 > gpurun_out/r4_t6.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4_t6.log; tail -8 gpurun_out/r4_t6.log
python tools/prof_gemm.py > gpurun_out/r4_gemm_f32.txt 2>&1; grep -v amdgpu.ids gpurun_out/r4_gemm_f32.txt
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/r4_pmc_gemm -o g -- python tools/prof_gemm.py > gpurun_out/r4_pmc_gemm.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4_prof6 -o stats -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/r4_prof6.log 2>&1
python tools/small_launches.py gpurun_out/r4_prof6/stats_kernel_trace.csv v > gpurun_out/r4_small6.txt 2>&1; sed -n '/^small launches/,$p' gpurun_out/r4_small6.txt | head -40; head -1 gpurun_out/r4_small6.txt
