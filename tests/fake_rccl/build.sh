#!/bin/bash
# builds the RCCL test double (tests/fake_rccl/fake_rccl.cpp) -> tests/fake_rccl/build/librccl_double.so
set -euo pipefail
cd "$(dirname "$0")"
mkdir -p build
g++ -O2 -std=c++17 -shared -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include fake_rccl.cpp -o build/librccl_double.so \
    -L/opt/rocm/lib -lamdhip64 -lrt -pthread -Wl,-rpath,/opt/rocm/lib
echo "$(pwd)/build/librccl_double.so"
