cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
AG_CONV_SOLO=1 timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_conv_fuzz.py -m gpu -q -x -k "conv" > gpurun_out/r4_t30.log 2>&1; echo "conv tests (solo+dma forced) rc $?"; tail -5 gpurun_out/r4_t30.log
AG_CONV_SOLO=0 python tools/prof_layers.py 64 > gpurun_out/r4_layers_solo0.txt 2>/dev/null; echo "solo0 rc $?"
AG_CONV_SOLO=1 python tools/prof_layers.py 64 > gpurun_out/r4_layers_dma1.txt 2>/dev/null; echo "dma1 rc $?"
AG_CONV_SOLO=1 AG_CONV_DMA=0 python tools/prof_layers.py 64 > gpurun_out/r4_layers_solo1.txt 2>/dev/null; echo "solo1 rc $?"
paste gpurun_out/r4_layers_solo0.txt gpurun_out/r4_layers_solo1.txt gpurun_out/r4_layers_dma1.txt | awk 'NR>2 {printf "%-10s %-6s %8s %6s   |  solo %8s %6s  | solo+dma %8s %6s\n", $1,$2,$3,$4,$9,$10,$15,$16}'
