cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q --maxfail=10 > gpurun_out/r4_t26.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r4_t26.log
bash tools/r4_extra_lines.sh 2>&1 | tail -14
timeout -k 10 300 python bench.py --workload full --steps 10 --warmup 2 > gpurun_out/r4_full_with_cpu.json 2> gpurun_out/r4_full_with_cpu.err; echo "full+cpu rc $?"
