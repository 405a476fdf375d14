#!/bin/bash
# Measurement builds of librvll.so with compile-time switches (A/B on one box): evidence_amd/diag/librvll_<name>.so,
# selected at run time with RVLL_LIBRARY=<path>.  Usage: scripts/build_variants.sh name "-DFLAG ..." [name2 "-D..."] ...
set -e
cd "$(dirname "$0")/../evidence_amd/csrc"
mkdir -p ../diag ../../build/var
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -fvisibility=hidden -Wall -Wno-unused-function -I../../include -I."
while [ $# -ge 2 ]; do
  name=$1; defs=$2; shift 2
  B=../../build/var/$name; mkdir -p $B
  /opt/rocm/bin/hipcc $FLAGS $defs -c rvll_kernels.hip -o $B/rvll_kernels.o &
  /opt/rocm/bin/hipcc $FLAGS $defs -mllvm -disable-machine-licm -c rvll_walk.hip -o $B/rvll_walk.o &
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o ../diag/librvll_$name.so ../../build/obj/rvll_api.o ../../build/obj/rvll_walk_host.o ../../build/obj/rvll_comm.o $B/rvll_kernels.o $B/rvll_walk.o ../../build/obj/rvll_fip.o ../../build/obj/rvll_live.o -ldl
  echo "built evidence_amd/diag/librvll_$name.so ($defs)"
done
