cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/prof_gemm2.py 2>&1 | grep -v amdgpu.ids
LAB_LIB_ONLY=1 ./tools/gemm_lab 2>&1 | grep "256x256\|V128"
