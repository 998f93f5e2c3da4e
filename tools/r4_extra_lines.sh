# the bench lines that are never the headline, of the final HEAD: one JSON object per line into gpurun_out/r04_extra_lines.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/r04_extra_lines.json
: > $out
run() { echo "== $*" >&2; timeout -k 10 400 python bench.py "$@" --no-cpu-baseline 2> gpurun_out/extra_last.err | tail -1 >> $out || { echo "FAILED: $*" >&2; tail -5 gpurun_out/extra_last.err >&2; }; }
run --steps 20 --warmup 3
run --steps 20 --warmup 3 --dtype bf16
run --steps 20 --warmup 3 --dtype f32x3
run --steps 20 --warmup 3 --workload c4
run --steps 20 --warmup 3 --workload c4 --dtype bf16
run --steps 20 --warmup 3 --workload c5
run --steps 20 --warmup 3 --workload c5 --dtype bf16
run --steps 20 --warmup 3 --force-phases
run --steps 20 --warmup 3 --opt rmsprop
run --workload full --steps 10 --warmup 2
run --workload full --steps 5 --warmup 2 --full-launch eager
python - <<'PY'
import json
for l in open('gpurun_out/r04_extra_lines.json'):
    d = json.loads(l)
    print('%-8s %-6s %8.2f ms  %s  %s  %s' % (d['config'].get('workload', '')[:8], d['dtype'], d['ms_per_step'], d.get('replay_check'), d.get('persist_status'), d['config'].get('launch', '')[:40]))
PY
