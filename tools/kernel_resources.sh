#!/bin/bash
# Per-kernel register / spill / LDS usage of one csrc file (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel.
#   bash tools/kernel_resources.sh csrc/gemm4.hip [extra flags]
cd "$(dirname "$(readlink -f "$0")")/../dinov2.cpp_amd"
f=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -Wno-unused-lambda-capture "$@" -x hip -c "$f" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
  awk '/Function Name|remark:.* Name:/{n=$NF; sub(/\[.*/,"",n); name=$0; sub(/.*Name: /,"",name); sub(/ \[.*/,"",name)}
       / VGPRs:/{v=$0; sub(/.* VGPRs: /,"",v); sub(/ .*/,"",v)} / AGPRs:/{a=$0; sub(/.* AGPRs: /,"",a); sub(/ .*/,"",a)}
       /VGPRs Spill/{s=$0; sub(/.*Spill: /,"",s); sub(/ .*/,"",s)} /ScratchSize/{sc=$0; sub(/.*: /,"",sc); sub(/ .*/,"",sc)}
       /Occupancy/{o=$0; sub(/.*: /,"",o); sub(/ .*/,"",o)} /LDS Size/{l=$0; sub(/.*: /,"",l); sub(/ .*/,"",l); printf "%-70s vgpr %3s agpr %3s spill %3s scratch %4s occ %s lds %s\n", name, v, a, s, sc, o, l}' | c++filt | sed 's/dinov2:://g; s/(dinov2::GemmArgs[^)]*)//'
