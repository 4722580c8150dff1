#!/bin/bash
# Diagnostic build of the VALU adjoint kernel with s_memtime stamps (never part of libgode.so): builds a private copy of
# the library with -DGODE_ODE_STAMPS under /tmp and runs the ODE micro-benchmark against it.   bash scripts/exp/ode_stamps.sh
set -e
D=/tmp/gode_stamps; rm -rf $D; mkdir -p $D
cp -r gan-ode_amd include scripts gan_ode_amd.py $D/
cd $D/gan-ode_amd/csrc
for f in igemm wgrad ode ode_valu odernn elementwise api; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -DGODE_ODE_STAMPS -c $f.hip -o ../lib/$f.o &
done; wait
hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libgode.so ../lib/*.o
cd $D && python3 - <<'PY'
import sys; sys.path.insert(0, "/tmp/gode_stamps")
import torch, gan_ode_amd._lib as L
def st(): return torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
P = [torch.randn(s, device="cuda") * 0.3 for s in ((64, 16), (64,), (16, 64), (16,), (16, 16), (16,), (16, 16), (16,))]
op = L.OdeParams(*[p.data_ptr() for p in P]); T = 16
tt = torch.linspace(0, 1, T); dt = (tt[1:] - tt[:-1]).cuda()
for N in (32, 1024):
    x = torch.randn(N, 16, device="cuda"); z = torch.empty(N * T, 72, device="cuda"); traj = torch.empty(N, T, 16, device="cuda")
    gz = torch.randn(N * T, 72, device="cuda"); grads = torch.empty(L.ODE_NPARAM, device="cuda")
    work = torch.empty(L.lib().gode_ode_bwd_work_size(N), device="cuda")
    f2 = L.OdeFwdOp(p=op, x=x.data_ptr(), content=None, dt=dt.data_ptr(), sel_t=None, z=z.data_ptr(), traj=traj.data_ptr(), N=N, T=T, substeps=1, prenet=1, zcols=72)
    b = L.OdeBwdOp(p=op, x=x.data_ptr(), traj=traj.data_ptr(), dt=dt.data_ptr(), sel_t=None, gz=gz.data_ptr(), work=work.data_ptr(), grads=grads.data_ptr(), N=N, T=T, substeps=1, prenet=1, accumulate=0, zcols=72)
    L.run_one(f2, st())
    for _ in range(4):
        L.run_one(b, st()); torch.cuda.synchronize()
PY
