# round 4, GPU call 1: the test suite (incl. the new parity tests), the headline bench, a kernel trace of the replayed step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r4_t1.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4_t1.log
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r4_bench1.json 2> gpurun_out/r4_bench1.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4_prof1 -o stats -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/r4_prof1.log 2>&1 || exit 1
python tools/small_launches.py gpurun_out/r4_prof1/stats_kernel_trace.csv v > gpurun_out/r4_small1.txt 2>&1
python tools/glue_trace.py > gpurun_out/r4_glue1.txt 2>gpurun_out/r4_glue1.err
tail -5 gpurun_out/r4_t1.log; cat gpurun_out/r4_bench1.json | cut -c1-600
