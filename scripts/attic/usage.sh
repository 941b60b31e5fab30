#!/bin/bash
# Compact register / spill table of the kernels (hipcc -Rpass-analysis=kernel-resource-usage).
cd "$(dirname "$0")/../hicpeaks_amd/csrc"
make usage EXTRA="$EXTRA" 2>&1 | sed 's/ \[-Rpass-analysis=kernel-resource-usage\]//' | awk '
/Function Name:/ {name=$NF}
/TotalSGPRs:/ {sg=$NF}
/ VGPRs:/ {vg=$NF}
/ScratchSize/ {sc=$NF}
/Occupancy/ {oc=$NF}
/SGPRs Spill/ {ss=$NF}
/VGPRs Spill/ {vs=$NF; printf "%-66s sgpr %3s vgpr %3s scratch %4s occ %s sgpr-spill %3s vgpr-spill %3s\n", substr(name,1,66), sg, vg, sc, oc, ss, vs}
'
