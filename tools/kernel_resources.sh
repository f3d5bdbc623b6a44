#!/bin/bash
# Register / scratch / occupancy summary of the kernels of one translation unit, as the compiler reports them.
# usage: tools/kernel_resources.sh srsran_project_23.5_amd/csrc/ldpc_decode_pk.hip [extra hipcc flags]
f=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage "$@" -c "$f" -o /dev/null 2>&1 |
  grep -E "Function Name|VGPRs:|AGPRs|Spill|ScratchSize|Occupancy|LDS Size" | sed 's/.*remark: [^ ]* *//; s/ *\[-Rpass.*//' |
  awk '/Name:/ {if (line) print line; n=$0; sub(/.*Name: (_ZN12_GLOBAL__N_1)?/,"",n); line=substr(n,1,48)} !/Name:/ {gsub(/^ +/,""); gsub(/ \[[^]]*\]/,""); line=line" | "$0} END {print line}'
