cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in 1 0; do AG_GEMMH_VARIANT=$v python tools/prof_gemm_h.py > gpurun_out/r4_gemmh2_v$v.txt 2>&1 || { tail -5 gpurun_out/r4_gemmh2_v$v.txt; exit 1; }; cat gpurun_out/r4_gemmh2_v$v.txt | grep -v amdgpu.ids; done
timeout -k 10 900 python -m pytest tests -m gpu -q --maxfail=15 > gpurun_out/r4_t5.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4_t5.log; tail -12 gpurun_out/r4_t5.log
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r4_bench5.json 2> gpurun_out/r4_bench5.err || tail -5 gpurun_out/r4_bench5.err
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --dtype bf16 > gpurun_out/r4_bench5_bf16.json 2> gpurun_out/r4_bench5_bf16.err || tail -5 gpurun_out/r4_bench5_bf16.err
AG_BF16_STORE=0 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --dtype bf16 > gpurun_out/r4_bench5_bf16_nostore.json 2> gpurun_out/r4_bench5_bf16_nostore.err || tail -5 gpurun_out/r4_bench5_bf16_nostore.err
for f in r4_bench5 r4_bench5_bf16 r4_bench5_bf16_nostore; do python -c "import json,sys; d=json.load(open('gpurun_out/$f.json')); print('$f', d['ms_per_step'], d.get('replay_check'), d.get('persist_status'), d['roofline']['kernel'], round(d['roofline']['frac'],3))"; done
timeout -k 10 600 python tools/diag_grad.py > gpurun_out/r4_diag_grad.txt 2> gpurun_out/r4_diag_grad.err; cat gpurun_out/r4_diag_grad.txt; tail -2 gpurun_out/r4_diag_grad.err
