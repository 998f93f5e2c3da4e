cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for wp in 0 1; do
for b in 128 64; do
AG_LSTM_WAVE_POLL=$wp timeout -k 10 120 python tools/prof_lstm.py 128 $b 512 2>&1 | grep "persist=1\|max" | tail -2 | sed "s/^/wave_poll=$wp  /"
done
done
AG_LSTM_WAVE_POLL=1 timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "lstm or persist" > gpurun_out/r4_t27.log 2>&1; echo "lstm tests (wave poll) rc $?"; tail -3 gpurun_out/r4_t27.log
AG_LSTM_WAVE_POLL=1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r4_bench27.json 2> gpurun_out/r4_bench27.err || tail -5 gpurun_out/r4_bench27.err
python -c "import json,sys; d=json.load(open('gpurun_out/r4_bench27.json')); print('bench27 wave_poll=1', d['ms_per_step'], d.get('replay_check'), d.get('persist_status')); [print('  ',k) for k in d['kernel_table'][:4]]"
AG_LSTM_WAVE_POLL=1 python bench.py --dtype bf16 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r4_bench27b.json 2> gpurun_out/r4_bench27b.err || tail -5 gpurun_out/r4_bench27b.err
python -c "import json,sys; d=json.load(open('gpurun_out/r4_bench27b.json')); print('bench27 bf16 wave_poll=1', d['ms_per_step'], d.get('replay_check'), d.get('persist_status')); [print('  ',k) for k in d['kernel_table'][:3]]"
