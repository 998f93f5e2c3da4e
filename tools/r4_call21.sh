cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 bash tools/profile_round.sh > gpurun_out/pr.log 2>&1; echo "profile_round rc $?"; tail -3 gpurun_out/pr.log
ls gpurun_out/prof_stats gpurun_out/prof_stats_bf16 2>/dev/null | head
