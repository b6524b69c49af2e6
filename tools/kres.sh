#!/bin/sh
# register / scratch / occupancy summary of the kernels of one translation unit:  sh tools/kres.sh igd_sweep.hip [name filter] [extra flags]
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 -O3 -std=c++17 $3 -I include -c gtars_amd/csrc/$1 -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 |
  grep -E "Function Name|VGPRs:|ScratchSize|Occupancy" | sed 's/.*remark: [^ ]* //; s/\[-Rpass.*//; s/.*hip:[0-9]*:[0-9]*: //' | paste - - - - |
  sed 's/Function Name: //; s/EEEv[A-Za-z0-9_]*//' | grep -E "${2:-.}" | while read l; do n=$(echo "$l" | cut -f1 | c++filt 2>/dev/null | cut -c1-70); echo "$n |$(echo "$l" | cut -f2-)"; done
