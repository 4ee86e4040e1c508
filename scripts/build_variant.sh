#!/bin/bash
# usage: scripts/build_variant.sh NAME "-DFOO=1 ..."  -> gpurun_out/variants/NAME.so (experiments only)
set -e
cd "$(dirname "$0")/../step_amd/csrc"
mkdir -p ../../variants /tmp/var_$1
for f in bh_dft_dense bh_dft_sep bh_dft_sep3 bh_closure mri_fft mcmc_update; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $2 -c $f.hip -o /tmp/var_$1/$f.o &
done
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $2 -x hip -c tables.cpp -o /tmp/var_$1/tables.o &
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../variants/$1.so /tmp/var_$1/*.o
