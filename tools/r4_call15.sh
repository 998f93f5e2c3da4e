cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
./tools/gemm_lab > gpurun_out/r4_gemm_lab5.txt 2>&1; cat gpurun_out/r4_gemm_lab5.txt
LAB_DBG=1 ./tools/gemm_lab > gpurun_out/r4_gemm_lab5b.txt 2>&1; grep -v "^peak" gpurun_out/r4_gemm_lab5b.txt
