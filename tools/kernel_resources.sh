#!/bin/bash
# usage: tools/kernel_resources.sh <file.hip> [name filter]  -> one line per kernel: VGPRs, occupancy, LDS, spills
F=$1; PAT=${2:-.}
EXTRA=""
[ "$(basename $F)" = "match.hip" ] && EXTRA="-mllvm -amdgpu-mfma-vgpr-form=1"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math $EXTRA -I/root/repo/include -I/root/repo/vulkansift_amd/csrc \
  -I/root/repo/vulkansift_amd/csrc/host -c $F -o /tmp/_kr.o -Rpass-analysis=kernel-resource-usage 2>&1 \
  | grep -E "remark: +(Function Name|VGPRs:|AGPRs:|Occupancy|VGPRs Spill|LDS Size|ScratchSize)" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' \
  | paste - - - - - - - | while read l; do n=$(echo "$l" | sed -E 's/Function Name: ([^ \t]+).*/\1/' | c++filt | cut -c 1-70); echo "$n | $(echo "$l" | cut -f2-)"; done | grep -E "$PAT"
