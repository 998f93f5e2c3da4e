cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/prof_thin_gemm.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_thin_gemm.txt
for t in 0 1 2; do echo "AG_GEMM_TILE=$t"; AG_GEMM_TILE=$t python tools/prof_thin_gemm.py 2>&1 | grep -v amdgpu.ids | cut -c1-75; done
timeout -k 10 300 python -m pytest tests/test_bf16.py -m gpu -q -x -k "networks" 2>&1 | tail -2
