cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 200 python bench.py --workload full --full-launch graph --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r4_full_graph.json 2> gpurun_out/r4_full_graph.err || tail -15 gpurun_out/r4_full_graph.err
python -c "import json,sys; d=json.load(open('gpurun_out/r4_full_graph.json')); print(d['ms_per_step'], d.get('replay_only_ms_per_step'), d.get('host_batches_ms_per_step'), d.get('host_ms_per_step'), d['persist_status'], d['losses_finite'])"
timeout -k 10 400 python tools/diag_grad.py > gpurun_out/r4_diag_grad2.txt 2> gpurun_out/r4_diag_grad2.err; echo "diag rc $?"; tail -45 gpurun_out/r4_diag_grad2.txt
