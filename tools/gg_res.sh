#!/bin/bash
# register / scratch use of one .hip file's kernels (default: the resident GV kernel)
cd "$(dirname "$0")/../jbonsai_amd/csrc"
f=${1:-jb_gv_gang.hip}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-value -Rpass-analysis=kernel-resource-usage -x hip -c $f -o /tmp/res_check.o 2>&1 | grep -i "error\|warning:\|Function Name\|VGPRs\|Scratch\|SGPRs\|Occupancy\|LDS"
