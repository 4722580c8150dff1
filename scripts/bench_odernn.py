"""ODE-RNN latent kernels on their own (HIP events): forward at N = 32 (one solve, six solves in one launch), N = 1024
(32 workgroups exchanging the whole-batch norm), the adaptive adjoint (one / two solves per launch).
   python scripts/bench_odernn.py"""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gan_ode_amd._lib as L


def st():
    return torch.cuda.current_stream().cuda_stream


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


torch.manual_seed(0)
f = torch.nn.Sequential(torch.nn.Linear(16, 16), torch.nn.Tanh(), torch.nn.Linear(16, 16))
gru = torch.nn.GRUCell(16, 16)
P = [p.detach().cuda() for p in list(f.parameters()) + [gru.weight_ih, gru.weight_hh, gru.bias_ih, gru.bias_hh]]
prm = L.OdeRnnParams(*[p.data_ptr() for p in P])
T = 16
lib = L.lib()


def make(N):
    d = dict(noise=torch.randn(T + 1, N, 16, device="cuda"), content=torch.randn(N, 50, device="cuda"),
             z=torch.empty(N * T, 96, device="cuda"), hp=torch.empty(N, T, 16, device="cuda"),
             nst=torch.zeros(T, dtype=torch.int32, device="cuda"), nstb=torch.zeros(T, dtype=torch.int32, device="cuda"),
             gz=torch.randn(N * T, 96, device="cuda"), grads=torch.empty(L.ODERNN_NPARAM, device="cuda"),
             work=torch.empty(lib.gode_odernn_bwd_work_size(N), device="cuda"))
    ns = lib.gode_odernn_sync_size(N)
    d["sync"] = torch.zeros(max(ns, 1), dtype=torch.int32, device="cuda")
    d["syncb"] = torch.zeros(max(ns, 1), dtype=torch.int32, device="cuda")
    sp = d["sync"].data_ptr() if ns else None
    spb = d["syncb"].data_ptr() if ns else None
    d["fop"] = L.OdeRnnFwdOp(p=prm, noise=d["noise"].data_ptr(), content=d["content"].data_ptr(), sel_t=None, z=d["z"].data_ptr(),
                             hs=None, hp=d["hp"].data_ptr(), nsteps=d["nst"].data_ptr(), N=N, T=T, rtol=1e-7, atol=1e-9, zcols=96, sync=sp)
    d["bop"] = L.OdeRnnBwdOp(p=prm, noise=d["noise"].data_ptr(), hp=d["hp"].data_ptr(), sel_t=None, gz=d["gz"].data_ptr(),
                             work=d["work"].data_ptr(), grads=d["grads"].data_ptr(), N=N, T=T, substeps=0, accumulate=0, zcols=96,
                             rtol=1e-7, atol=1e-9, sync=spb, nsteps=d["nstb"].data_ptr())
    return d


def multi(kind, ds):
    arr_t = (L.OdeRnnFwdOp if kind == "f" else L.OdeRnnBwdOp) * len(ds)
    arr = arr_t(*[d["fop" if kind == "f" else "bop"] for d in ds])
    fn = lib.gode_odernn_fwd_multi if kind == "f" else lib.gode_odernn_bwd_multi
    return lambda: L.check(fn(arr, len(ds), st()))


only = [int(v) for v in sys.argv[1:]]
for N in (only or (32, 64, 1024)):
    d = make(N)
    us = timed(lambda: L.run_one(d["fop"], st()))
    print(f"forward N={N:5d}: {us:8.1f} us   trial steps/frame {d['nst'].cpu().tolist()}")
    us = timed(lambda: L.run_one(d["bop"], st()), reps=5)
    print(f"adjoint N={N:5d}: {us:8.1f} us   trial steps/frame {d['nstb'].cpu().tolist()}")
if only:
    sys.exit(0)
ds = [make(32) for _ in range(6)]
print(f"forward 6 x N=32 in one launch: {timed(multi('f', ds)):8.1f} us")
print(f"adjoint 2 x N=32 in one launch: {timed(multi('b', ds[:2]), reps=5):8.1f} us")
