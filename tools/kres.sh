#!/bin/sh
# register / spill / LDS summary per kernel of one HIP source: sh tools/kres.sh gtars_amd/csrc/tokenize_lds.hip [filter]
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -c "$1" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
  awk '/Function Name:/ {name=$0; sub(/.*Function Name: /,"",name); sub(/ \[.*/,"",name)}
       /VGPRs:/ && !/Agprs|AGPRs/ {v=$0; sub(/.*VGPRs: /,"",v); sub(/ .*/,"",v)}
       /TotalSGPRs:/ {s=$0; sub(/.*TotalSGPRs: /,"",s); sub(/ .*/,"",s)}
       /VGPR Spill:/ {sp=$0; sub(/.*VGPR Spill: /,"",sp); sub(/ .*/,"",sp)}
       /ScratchSize/ {sc=$0; sub(/.*ScratchSize \[bytes\/lane\]: /,"",sc); sub(/ .*/,"",sc)}
       /Occupancy/ {o=$0; sub(/.*Occupancy \[waves\/SIMD\]: /,"",o); sub(/ .*/,"",o)}
       /LDS Size/ {l=$0; sub(/.*LDS Size \[bytes\/block\]: /,"",l); sub(/ .*/,"",l); print "vgpr",v,"sgpr",s,"spill",sp,"scratch",sc,"occ",o,"lds",l,name}' |
  (if [ -n "$2" ]; then grep "$2"; else cat; fi) | while read line; do set -- $line; last=$(eval echo \${$#}); echo "$line" | sed "s/$last//"; echo "   $(echo $last | c++filt | cut -c1-110)"; done
