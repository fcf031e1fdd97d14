#!/bin/bash
# Developer tool: compile only the solve kernel of the benchmark class (extra -D flags as arguments) and print its register numbers
cd $(dirname $0)/../omg-tools_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-but-set-variable -DOMGX_ONLY_HEADLINE "$@" -shared -o /tmp/libomgx_hl.so omgx.hip 2>&1 | grep -E "error|warning: v" | head
../../tools/kernel_resources.sh /tmp/libomgx_hl.so ipm_solve
