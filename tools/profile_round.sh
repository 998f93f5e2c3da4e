# rocprofv3 evidence for profiles/: kernel stats of the replayed step + the two PMC traffic passes (separate runs,
# counters never combined with sys/hip traces).  Run from the repo root on the GPU box.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -o stats -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/prof_stats.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -o f -- python bench.py --steps 2 --warmup 1 --no-graph --no-roofline --no-cpu-baseline > gpurun_out/pmc_f.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -o w -- python bench.py --steps 2 --warmup 1 --no-graph --no-roofline --no-cpu-baseline > gpurun_out/pmc_w.log 2>&1 || exit 1
python tools/pmc_traffic.py gpurun_out/pmc_f/f_counter_collection.csv gpurun_out/pmc_w/w_counter_collection.csv profiles/r04_pmc_traffic.json
cp profiles/r04_pmc_traffic.json gpurun_out/r04_pmc_traffic.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats_bf16 -o stats -- python bench.py --dtype bf16 --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/prof_stats_bf16.log 2>&1 || exit 1
python tools/prof_layers.py 64 > gpurun_out/r04_conv_layers_b64.txt 2>/dev/null || exit 1
python tools/prof_layers.py 64 bf16 > gpurun_out/r04_conv_layers_b64_bf16.txt 2>/dev/null || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f16 -o f -- python bench.py --dtype bf16 --steps 2 --warmup 1 --no-graph --no-roofline --no-cpu-baseline > gpurun_out/pmc_f16.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w16 -o w -- python bench.py --dtype bf16 --steps 2 --warmup 1 --no-graph --no-roofline --no-cpu-baseline > gpurun_out/pmc_w16.log 2>&1 || exit 1
python tools/pmc_traffic.py gpurun_out/pmc_f16/f_counter_collection.csv gpurun_out/pmc_w16/w_counter_collection.csv profiles/r04_pmc_traffic_bf16.json
cp profiles/r04_pmc_traffic_bf16.json gpurun_out/r04_pmc_traffic_bf16.json
