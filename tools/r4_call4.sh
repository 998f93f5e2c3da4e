cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in 0 1 2 3 4; do AG_GEMMH_VARIANT=$v python tools/prof_gemm_h.py > gpurun_out/r4_gemmh_v$v.txt 2>&1 || { tail -5 gpurun_out/r4_gemmh_v$v.txt; exit 1; }; tail -1 gpurun_out/r4_gemmh_v$v.txt; done
timeout -k 10 600 python -m pytest tests/test_bf16.py tests/test_gpu_modules.py tests/test_loop.py -m gpu -q --maxfail=15 > gpurun_out/r4_t4.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4_t4.log; tail -12 gpurun_out/r4_t4.log
timeout -k 10 600 python tools/diag_grad.py > gpurun_out/r4_diag_grad.txt 2> gpurun_out/r4_diag_grad.err; cat gpurun_out/r4_diag_grad.txt
timeout -k 10 900 python bench.py --workload full --steps 3 --warmup 1 > gpurun_out/r4_bench_full.json 2> gpurun_out/r4_bench_full.err; tail -3 gpurun_out/r4_bench_full.err; cut -c1-600 gpurun_out/r4_bench_full.json
