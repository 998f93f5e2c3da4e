cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 ./tools/gemm_lab_h > gpurun_out/r4_gemm_lab_h1.txt 2>&1; cat gpurun_out/r4_gemm_lab_h1.txt
timeout -k 10 300 python tools/prof_gemm_h.py > gpurun_out/r4_gemmh_base.txt 2>&1; grep -v amdgpu.ids gpurun_out/r4_gemmh_base.txt | tail -16
