#!/bin/bash
# timing-only ablation build of the GEMM kernels (WRONG results): gan-ode_amd/lib/libgode_abl.so, selected per run with
# GODE_IGEMM_STAGGER=-1..-4 and GODE_AB_LIB=gan-ode_amd/lib/libgode_abl.so (scripts/exp/ab_igemm.py)
set -e
cd "$(dirname "$0")/../.."
L=gan-ode_amd/lib
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -DGODE_ABLATION_BUILD -c gan-ode_amd/csrc/igemm.hip -o /tmp/igemm_abl.o
OBJS=$(ls $L/*.o | grep -v igemm.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libgode_abl.so /tmp/igemm_abl.o $OBJS
echo built $L/libgode_abl.so
