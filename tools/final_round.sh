# everything the round's profiles/ and the docs quote, on the final HEAD (run from the repo root on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q --maxfail=10 > gpurun_out/final_tests.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/final_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout -k 10 1100 bash tools/profile_round.sh > gpurun_out/pr.log 2>&1; echo "profile_round rc $?"; tail -2 gpurun_out/pr.log
bash tools/r4_extra_lines.sh 2>&1 | tail -12
timeout -k 10 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "default bench rc $?"
python tools/show_bench.py gpurun_out/bench_default.json | head -8
timeout -k 10 300 python bench.py --workload full --steps 10 --warmup 2 > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err; echo "full bench rc $?"
