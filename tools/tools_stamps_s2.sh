cd $GRAFT_REPO_ROOT
for so in lib_EXP_STAMP libs_EXP_NO_A_STAGE; do echo "=== $so"; python tools/tools_stamps2.py ms-nets_amd/$so.so s2_32_64 40 41 2>&1 | grep -v amdgpu | head -6; done
