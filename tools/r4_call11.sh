cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
./tools/gemm_lab > gpurun_out/r4_gemm_lab1.txt 2>&1; cat gpurun_out/r4_gemm_lab1.txt
for e in nofeed mainstream noload; do
AG_LOOP_EXP=$e timeout -k 10 200 python bench.py --workload full --full-launch graph --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r4_full_graph_$e.json 2> gpurun_out/r4_full_graph_$e.err || tail -15 gpurun_out/r4_full_graph_$e.err
python -c "import json,sys; d=json.load(open('gpurun_out/r4_full_graph_$e.json')); print('$e', d['ms_per_step'], d.get('replay_only_ms_per_step'), d.get('host_ms_per_step'))"
done
