cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for abl in 0 1 2 4 6 7; do echo "AG_CONV_ABL=$abl (1 no epilogue, 2 no MFMA, 4 no LDS reads)"; AG_CONV_ABL=$abl python tools/prof_one_layer.py 2>&1 | grep -v amdgpu.ids; done
echo "register staging:"; AG_CONV_DMA=0 python tools/prof_one_layer.py 2>&1 | grep -v amdgpu.ids
echo "8-wave form:"; AG_CONV_SOLO=0 python tools/prof_one_layer.py 2>&1 | grep -v amdgpu.ids
for abl in 1 2 7; do echo "8-wave form AG_CONV_ABL=$abl"; AG_CONV_SOLO=0 AG_CONV_ABL=$abl python tools/prof_one_layer.py 2>&1 | grep -v amdgpu.ids; done
