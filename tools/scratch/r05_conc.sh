#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
echo "== new lib, 8 concurrent processes"
for i in 0 1 2 3 4 5 6 7; do python tools/scratch/r05_conc.py new$i 2>&1 | grep -v amdgpu.ids & done; wait
echo "== old lib (round 4), 8 concurrent processes"
for i in 0 1 2 3 4 5 6 7; do MARLIN_AMD_LIB=marlin_amd/csrc/build_alt/libmarlin_hip_r04.so python tools/scratch/r05_conc.py old$i 2>&1 | grep -v amdgpu.ids & done; wait
