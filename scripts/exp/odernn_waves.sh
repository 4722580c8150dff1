#!/bin/bash
# Is the VALU ODE-RNN kernel issue-bound (two waves per SIMD share the vector ALU) or latency-bound (dependent chains)?
# Private build with 4 waves per workgroup (16 trajectories, one wave per SIMD) timed at N = 16 against the product build
# (8 waves, 32 trajectories) at N = 32.   bash scripts/exp/odernn_waves.sh
set -e
D=/tmp/gode_w4; rm -rf $D; mkdir -p $D
cp -r gan-ode_amd include scripts gan_ode_amd.py $D/
cd $D/gan-ode_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -DRV_WAVES=4 -c odernn_valu.hip -o ../lib/odernn_valu.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libgode.so ../lib/*.o
cd $D && python3 scripts/bench_odernn.py 16
