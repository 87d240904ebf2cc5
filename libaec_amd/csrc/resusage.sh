#!/bin/bash
# prints sgpr / vgpr / scratch / occupancy per kernel of a .hip file (compiler resource-usage remarks)
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -c "$1" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "Function Name|    VGPRs:|ScratchSize|Occupancy|TotalSGPRs" | sed -E 's/.*remark: *//; s/ \[-Rpass-analysis=kernel-resource-usage\]//' | paste - - - - - \
 | sed -E 's/Function Name: _ZN3aec12_GLOBAL__N_1[0-9]+//; s/(ILi[0-9]+ELi[0-9]+E)[A-Za-z0-9_]*/\1/; s/E[PN][A-Za-z0-9_]*//; s/TotalSGPRs: /sgpr=/; s/VGPRs: /vgpr=/; s/ScratchSize \[bytes\/lane\]: /scratch=/; s/Occupancy \[waves\/SIMD\]: /occ=/'
