#!/bin/bash
# Diagnostic build of the VALU ODE-RNN forward kernel with s_memtime stamps (never part of libgode.so): a private copy of the
# library with -DGODE_ODE_STAMPS under /tmp, run on the N = 32, T = 16 solve.   bash scripts/exp/odernn_stamps.sh
set -e
D=/tmp/gode_stamps; rm -rf $D; mkdir -p $D
cp -r gan-ode_amd include scripts gan_ode_amd.py $D/
cd $D/gan-ode_amd/csrc
for f in igemm conv_patch wgrad ode ode_valu odernn odernn_valu adj_adaptive elementwise api; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -DGODE_ODE_STAMPS -c $f.hip -o ../lib/$f.o &
done; wait
hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libgode.so ../lib/*.o
cd $D && python3 - <<'PY'
import sys; sys.path.insert(0, "/tmp/gode_stamps")
import torch, gan_ode_amd._lib as L
def st(): return torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
f = torch.nn.Sequential(torch.nn.Linear(16, 16), torch.nn.Tanh(), torch.nn.Linear(16, 16)); gru = torch.nn.GRUCell(16, 16)
P = [p.detach().cuda() for p in list(f.parameters()) + [gru.weight_ih, gru.weight_hh, gru.bias_ih, gru.bias_hh]]
prm = L.OdeRnnParams(*[p.data_ptr() for p in P]); N, T = 32, 16
noise = torch.randn(T + 1, N, 16, device="cuda"); z = torch.empty(N * T, 96, device="cuda"); hp = torch.empty(N, T, 16, device="cuda")
op = L.OdeRnnFwdOp(p=prm, noise=noise.data_ptr(), content=None, sel_t=None, z=z.data_ptr(), hs=None, hp=hp.data_ptr(), nsteps=None,
                   N=N, T=T, rtol=1e-7, atol=1e-9, zcols=96, sync=None)
# keep the clocks up with some GEMM load first, as inside a training iteration
a = torch.randn(4096, 4096, device="cuda")
for _ in range(20): a @ a
for _ in range(2):
    L.run_one(op, st()); torch.cuda.synchronize()
PY
