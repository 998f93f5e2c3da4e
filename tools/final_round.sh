timeout -k 10 1000 bash tools/profile_round.sh > gpurun_out/pr.log 2>&1; tail -1 gpurun_out/pr.log
timeout -k 10 600 python tools/prof_layers.py 64 > gpurun_out/layers64.txt 2>/dev/null; tail -2 gpurun_out/layers64.txt
timeout -k 10 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
python tools/show_bench.py gpurun_out/bench_default.json
