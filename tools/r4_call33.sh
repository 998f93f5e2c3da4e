cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "NO-EPILOGUE BUILD"; python tools/prof_one_layer.py G1.deconv G2.deconv G4.deconv G3.conv G4.conv 2>&1 | grep -v amdgpu.ids
echo "8-wave form:"; AG_CONV_SOLO=0 python tools/prof_one_layer.py G1.deconv G2.deconv G4.deconv G3.conv G4.conv 2>&1 | grep -v amdgpu.ids
echo "register staging:"; AG_CONV_DMA=0 python tools/prof_one_layer.py G2.deconv G4.conv 2>&1 | grep -v amdgpu.ids
