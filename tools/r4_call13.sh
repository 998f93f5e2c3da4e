cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
LAB_DBG=1 ./tools/gemm_lab > gpurun_out/r4_gemm_lab3.txt 2>&1; grep -v "^peak" gpurun_out/r4_gemm_lab3.txt
AG_LOOP_EXP=ev timeout -k 10 200 python bench.py --workload full --full-launch graph --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r4_full_graph_ev.json 2> gpurun_out/r4_full_graph_ev.err || tail -15 gpurun_out/r4_full_graph_ev.err
python -c "
import json; d=json.load(open('gpurun_out/r4_full_graph_ev.json')); print('ev', d['ms_per_step'], d.get('replay_only_ms_per_step'), d.get('host_ms_per_step'))
for k,v in d.items():
    if k.startswith('replay_timeline'): print(k); [print('   ', x) for x in v]
"
