cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for cfg in "conv 256 512 7 2 3 256 0" "conv 256 512 7 2 3 256 1" "conv 64 128 7 2 3 1024 1" "convt 64 32 8 4 2 2048 0" "conv 49 64 9 4 4 8192 1"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_conv -o c$i -- python tools/prof_conv.py $cfg > gpurun_out/pc$i.log 2>&1 || exit 1
done
