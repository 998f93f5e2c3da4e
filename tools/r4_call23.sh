cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/r4_pmc_gemm2 -o g -- python tools/prof_gemm.py > gpurun_out/r4_pmc_gemm2.log 2>&1; echo "pmc rc $?"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4_prof_full2 -o stats -- python bench.py --workload full --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r4_prof_full2.log 2>&1; echo "prof full rc $?"
python bench.py > gpurun_out/r4_bench_default.json 2> gpurun_out/r4_bench_default.err; echo "default bench rc $?"; python tools/show_bench.py gpurun_out/r4_bench_default.json | head -30
timeout -k 10 300 python bench.py --workload full --steps 10 --warmup 2 > gpurun_out/r4_full_with_cpu.json 2> gpurun_out/r4_full_with_cpu.err; echo "full+cpu rc $?"; python -c "import json; d=json.load(open('gpurun_out/r4_full_with_cpu.json')); print(d['ms_per_step'], d.get('cpu_baseline'))"
