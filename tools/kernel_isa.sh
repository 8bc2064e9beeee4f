#!/bin/bash
# ISA of one kernel instantiation of a unit (product flags + $PN_DIAG_FLAGS): tools/kernel_isa.sh <mangled-name prefix> [unit] > out.s
# e.g. tools/kernel_isa.sh _ZN2pn18bf16_filter_kernelILi8ELi1ELb0ELb1ELi2ELb0EEE
R=$(cd $(dirname $0)/.. && pwd)
U=${2:-$R/petal-neighbors_amd/csrc/bf16_filter.hip}
T=$(mktemp -d)
(cd $T && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt $PN_DIAG_FLAGS -x hip --cuda-device-only -S $U -o $T/u.s 2>/dev/null)
awk -v n="$1" 'index($0, n) == 1 && /:/ {p = 1} p {print} p && /s_endpgm/ {exit}' $T/u.s
rm -rf $T
