"""ODE kernels at the config sizes and at a saturating batch: achieved GB/s on the algorithmic
1,088 B/trajectory (64 B read + 16x16 fp32 written) and TFLOP/s on 61.4 kFLOP/trajectory (+ 960 tanh).
    python scripts/bench_ode.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gan_ode_amd._lib as L

def st(): return torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
P = [torch.randn(s, device="cuda") * 0.3 for s in ((64, 16), (64,), (16, 64), (16,), (16, 16), (16,), (16, 16), (16,))]
op = L.OdeParams(*[p.data_ptr() for p in P])
T = 16
tt = torch.linspace(0, 1, T)
dt = (tt[1:] - tt[:-1]).cuda()
for N in (32, 1024, 1 << 16, 1 << 20):
    x = torch.randn(N, 16, device="cuda")
    z = torch.empty(N * T, 72, device="cuda")
    traj = torch.empty(N, T, 16, device="cuda")
    gz = torch.randn(N * T, 72, device="cuda")
    grads = torch.empty(L.ODE_NPARAM, device="cuda")
    work = torch.empty(L.lib().gode_ode_bwd_work_size(N), device="cuda")
    f = L.OdeFwdOp(p=op, x=x.data_ptr(), content=None, dt=dt.data_ptr(), sel_t=None, z=z.data_ptr(), traj=None, N=N, T=T, substeps=1, prenet=1, zcols=72)
    f2 = L.OdeFwdOp(p=op, x=x.data_ptr(), content=None, dt=dt.data_ptr(), sel_t=None, z=z.data_ptr(), traj=traj.data_ptr(), N=N, T=T, substeps=1, prenet=1, zcols=72)
    b = L.OdeBwdOp(p=op, x=x.data_ptr(), traj=traj.data_ptr(), dt=dt.data_ptr(), sel_t=None, gz=gz.data_ptr(), work=work.data_ptr(), grads=grads.data_ptr(), N=N, T=T, substeps=1, prenet=1, accumulate=0, zcols=72)
    L.run_one(f2, st())
    for name, o, flop, byt in (("fwd", f, 61.4e3 + 4.1e3, 1088), ("bwd(adjoint)", b, 4 * 61.4e3, 2 * 1024 + 64)):
        reps = 20 if N < (1 << 18) else 5
        for _ in range(2): L.run_one(o, st())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): L.run_one(o, st())
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(f"N={N:8d} {name:13s} {ms*1e3:10.1f} us  {N*byt/ms/1e6:9.2f} GB/s ({N*byt/ms/1e6/8000*100:5.2f}% of 8 TB/s)  {N*flop/ms/1e9:7.2f} TFLOP/s ({N*flop/ms/1e9/157.3*100:5.1f}% of fp32 peak)  {N/ms/1e3:9.2f} M traj/s")
