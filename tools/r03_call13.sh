#!/bin/bash
# round 3, call 13: do VGPR bank relations between a VOP2's first source and its destination cost anything in k_lanczos3_x2?
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r03_call13
rm -rf $out && mkdir -p $out
cd $root
timeout -k 10 400 python3 tools/lz_variants.py --rounds 7 cur=nu_scaler_amd/lib/libnuscaler_hip.so same=tools/_ablate/lib_asmsame.so avoid=tools/_ablate/lib_bankavoid.so seek=tools/_ablate/lib_bankseek.so > $out/bank_swap_ab.txt 2>&1; grep -v amdgpu $out/bank_swap_ab.txt
