# round 4, GPU call 2: suite after the node-stripping changes, headline bench, kernel trace of a replayed step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --maxfail=12 > gpurun_out/r4_t2.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4_t2.log
tail -25 gpurun_out/r4_t2.log
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r4_bench2.json 2> gpurun_out/r4_bench2.err || { tail -20 gpurun_out/r4_bench2.err; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4_prof2 -o stats -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/r4_prof2.log 2>&1 || exit 1
python tools/small_launches.py gpurun_out/r4_prof2/stats_kernel_trace.csv v > gpurun_out/r4_small2.txt 2>&1
cat gpurun_out/r4_bench2.json | cut -c1-400
