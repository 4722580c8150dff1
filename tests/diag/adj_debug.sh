#!/bin/bash
# private debug build with -DGODE_ADJ_DEBUG (trial-step log of the adaptive adjoint), never part of libgode.so
set -e
D=/tmp/gode_dbg; rm -rf $D; mkdir -p $D
cp -r gan-ode_amd include scripts oracle tests gan_ode_amd.py $D/
cd $D/gan-ode_amd/csrc
for f in igemm wgrad ode ode_valu odernn adj_adaptive elementwise api; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -DGODE_ADJ_DEBUG -c $f.hip -o ../lib/$f.o &
done; wait
hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libgode.so ../lib/*.o
cd $D && sed -i 's#/root/repo#/tmp/gode_dbg#g' tests/diag/dbg_adj.py && python3 tests/diag/dbg_adj.py
