#!/bin/bash
# Assembly of kernels.hip with the Makefile's flags -> build_ab/kernels.s (extra -D flags may follow)
cd "$(dirname "$0")/../renderer_amd/csrc"
mkdir -p ../../build_ab
/opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -Xclang -target-feature -Xclang -packed-fp32-ops "$@" -S --cuda-device-only -o ../../build_ab/kernels.s kernels.hip 2> >(grep -v "is not a recognized feature\|hip-link" >&2)
