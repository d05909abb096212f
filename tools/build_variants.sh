#!/bin/bash
# A/B builds of the library: the "large" MSM variant with 2^18 instead of 2^19 buckets -> build/variants/libplonk_nb18.so
# (run after python __graft_entry__.py; tools/variant_bench.sh or PLONK_HIP_LIB=... selects it)
set -e
cd "$(dirname "$0")/.."
mkdir -p build/variants build/obj
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-pass-failed"
for bits in "$@"; do
  hipcc $F -DPLONK_MSM_NB_BITS=$bits -c plonk_amd/csrc/msm.hip -o build/obj/msm.hip.nbl$bits.o &
  hipcc $F -DPLONK_MSM_NB_BITS=$bits -c plonk_amd/csrc/msm_sort.hip -o build/obj/msm_sort.hip.nbl$bits.o &
  wait
  objs=$(ls build/obj/*.hip.o | grep -v nbl)
  hipcc --offload-arch=gfx950 -shared -fPIC $objs build/obj/msm.hip.nbl$bits.o build/obj/msm_sort.hip.nbl$bits.o -o build/variants/libplonk_nb$bits.so
  echo "built build/variants/libplonk_nb$bits.so"
done
