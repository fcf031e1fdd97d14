#!/bin/bash
# Developer tool: register / spill / scratch numbers of the kernels in a built libomgx.so (code object notes).
LIB=${1:-$(dirname $0)/../omg-tools_amd/csrc/libomgx.so}
T=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy -O binary --only-section=.hip_fatbin "$LIB" $T/fat.bin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/co.o
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/co.o | grep -E "^ +(- )?\.(name|vgpr_count|agpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|group_segment_fixed_size):" | sed 's/^ *-\? *//' | awk '/^\.agpr_count/{if(rec)print rec; rec=""} {rec=rec" "$0} END{print rec}' | sed 's/_ZN4omgx//; s/_Z//' | grep -E "${2:-.}" | sed -E "s/EvN4omgx.*Pyi//; s/ +/ /g" | cut -c1-260
ls -l $T/co.o | awk '{print "code object bytes:", $5}'
rm -rf $T
