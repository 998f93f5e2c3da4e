cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python tools/diag_bf16_xgrad.py > gpurun_out/r4_diag_bf16_xgrad.txt 2> gpurun_out/r4_diag_bf16_xgrad.err; echo "diag rc $?"; cat gpurun_out/r4_diag_bf16_xgrad.txt; tail -3 gpurun_out/r4_diag_bf16_xgrad.err
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
