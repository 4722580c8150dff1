"""A/B of two libgode builds on the GEMM shapes of the three configs (GPU box):
    GODE_AB_LIB=gan-ode_amd/lib/libgode_base.so python scripts/exp/ab_igemm.py ; python scripts/exp/ab_igemm.py
prints us / TFLOP/s per (shape, direction) at the cost model's own choice, and a checksum of the output."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import gan_ode_amd._lib as L
if os.environ.get("GODE_AB_LIB"):
    L.LIB_PATH = os.path.abspath(os.environ["GODE_AB_LIB"])
from gan_ode_amd.engine import make_geom, conv_out, stream_ptr
lib = L.lib()


def g3(N, Ci, Co, xi, k, s, p):
    yo = tuple(conv_out(xi[a], k[a], s[a], p[a]) for a in range(3))
    return make_geom(N, Ci, Co, xi, yo, k, s, p)


cases = [("dec L1 N=512", make_geom(512, 256, 512, (1, 8, 8), (1, 4, 4), (1, 4, 4), (1, 2, 2), (0, 1, 1))),
         ("dec L2 N=512", make_geom(512, 128, 256, (1, 16, 16), (1, 8, 8), (1, 4, 4), (1, 2, 2), (0, 1, 1))),
         ("dec L3 N=512", make_geom(512, 64, 128, (1, 32, 32), (1, 16, 16), (1, 4, 4), (1, 2, 2), (0, 1, 1))),
         ("dec L3 N=544", make_geom(544, 64, 128, (1, 32, 32), (1, 16, 16), (1, 4, 4), (1, 2, 2), (0, 1, 1))),
         ("ucf dec L2 N=256", make_geom(256, 128, 256, (1, 16, 16), (1, 8, 8), (1, 4, 4), (1, 2, 2), (0, 1, 1))),
         ("mnist vidD L1 N=64", g3(64, 64, 128, (15, 15, 15), (2, 2, 2), (1, 2, 2), (0, 1, 1))),
         ("mnist vidD L2 N=64", g3(64, 128, 256, (14, 8, 8), (2, 2, 2), (1, 2, 2), (0, 1, 1))),
         ("mnist vidD L3 N=64", g3(64, 256, 512, (13, 5, 5), (2, 2, 2), (1, 2, 2), (0, 1, 1))),
         ("imgD L1 N=64", make_geom(64, 64, 128, (1, 14, 14), (1, 7, 7), (1, 4, 4), (1, 2, 2), (0, 1, 1))),
         ("ucf vidD L1", g3(16, 64, 128, (13, 32, 32), (4, 4, 4), (1, 2, 2), (0, 1, 1))),
         ("ucf vidD L2", g3(16, 128, 256, (10, 16, 16), (4, 4, 4), (1, 2, 2), (0, 1, 1))),
         ("ucf vidD L3", g3(16, 256, 512, (7, 8, 8), (4, 4, 4), (1, 2, 2), (0, 1, 1)))]
torch.manual_seed(0)
# (the first ~25 ms of GEMM load in a process run 8-15 % slow while the clock ramps: warm up before the first timed case)
_w = torch.randn(4096, 4096, device="cuda")
for _ in range(40):
    _w = (_w @ _w) * 1e-4
torch.cuda.synchronize()
tot = 0.0
for name, g in cases:
    for d, dn in ((L.FPROP, "fprop"), (L.DGRAD, "dgrad")):
        src_dims = (g.N, g.Do, g.Ho, g.Wo, g.Co) if d == L.DGRAD else (g.N, g.Di, g.Hi, g.Wi, g.Ci)
        out_dims = (g.N, g.Di, g.Hi, g.Wi, g.Ci) if d == L.DGRAD else (g.N, g.Do, g.Ho, g.Wo, g.Co)
        src = torch.randn(src_dims, device="cuda")
        w = torch.randn(g.Co, g.Ci, g.kd, g.kh, g.kw, device="cuda") * 0.05
        wp = torch.empty(lib.gode_pack_size(C.byref(g), d), device="cuda")
        L.check(lib.gode_pack_weights(C.byref(g), d, w.data_ptr(), wp.data_ptr(), None, 0, stream_ptr()))
        out = torch.zeros(out_dims, device="cuda")
        fl = 2.0 * g.N * g.Do * g.Ho * g.Wo * g.Co * g.Ci * g.kd * g.kh * g.kw
        op = L.IgemmOp(g=g, dir=d, act=L.ACT_NONE, epilogue=L.EPI_RAW, tile=0, src=src.data_ptr(), wpack=wp.data_ptr(), out=out.data_ptr())
        ws = lib.gode_igemm_work_size(C.byref(op))
        work = torch.empty(max(ws, 1), device="cuda")
        op.work = work.data_ptr()
        st = stream_ptr()
        for _ in range(3):
            L.run_one(op, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            L.run_one(op, st)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        tot += us
        print(f"{name:20s} {dn}  {us:7.1f} us  {fl/us/1e6:6.1f} TF   sum {float(out.double().sum()):.6e} sq {float((out.double()**2).sum()):.6e}", flush=True)
print(f"total {tot:.1f} us   lib {L.LIB_PATH}")
tot = 0.0
wcases = [(n, g) for n, g in cases if not n.startswith("dec L3 N=544")] + [("dec L2 N=544", make_geom(544, 128, 256, (1, 16, 16), (1, 8, 8), (1, 4, 4), (1, 2, 2), (0, 1, 1)))]
for name, g in wcases:
    x = torch.randn(g.N, g.Di, g.Hi, g.Wi, g.Ci, device="cuda")
    y = torch.randn(g.N, g.Do, g.Ho, g.Wo, g.Co, device="cuda")
    dw = torch.zeros(g.Co, g.Ci, g.kd, g.kh, g.kw, device="cuda")
    op = L.WgradOp(g=g, act=L.ACT_NONE, xform_on_y=0, splits=0, accumulate=0, x=x.data_ptr(), y=y.data_ptr(), scale=None, shift=None, dw=dw.data_ptr())
    work = torch.empty(lib.gode_wgrad_work_size(C.byref(op)), device="cuda")
    op.work = work.data_ptr()
    st = stream_ptr()
    for _ in range(3):
        L.run_one(op, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        L.run_one(op, st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    tot += us
    fl = 2.0 * g.N * g.Do * g.Ho * g.Wo * g.Co * g.Ci * g.kd * g.kh * g.kw
    print(f"{name:20s} wgrad  {us:7.1f} us  {fl/us/1e6:6.1f} TF   sum {float(dw.double().sum()):.6e} sq {float((dw.double()**2).sum()):.6e}", flush=True)
print(f"wgrad total {tot:.1f} us")
