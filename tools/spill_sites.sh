#!/bin/bash
# Developer tool: where the solve kernel of /tmp/libomgx_hl.so (tools/hl_build.sh) touches scratch memory -- line numbers of
# the scratch instructions within the kernel's disassembly, and the kernel's length
T=/tmp/hl_dis; mkdir -p $T
/opt/rocm/lib/llvm/bin/llvm-objcopy -O binary --only-section=.hip_fatbin ${1:-/tmp/libomgx_hl.so} $T/fat.bin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/co.o
/opt/rocm/lib/llvm/bin/llvm-objdump -d --mcpu=gfx950 $T/co.o > $T/dis.s
s=$(grep -n "<_Z16ipm_solve_kernel" $T/dis.s | head -1 | cut -d: -f1); e=$(grep -n "<_Z13sample_kernelIfE" $T/dis.s | head -1 | cut -d: -f1)
sed -n "${s},${e}p" $T/dis.s > $T/solve.s
echo "kernel lines: $(wc -l < $T/solve.s); scratch ops at: $(grep -n 'scratch_store\|scratch_load' $T/solve.s | awk -F: '{print $1}' | tr '\n' ' ')"
echo "v_writelane $(grep -c v_writelane $T/solve.s) v_readlane $(grep -c v_readlane $T/solve.s) s_barrier $(grep -c s_barrier $T/solve.s)"
