"""Times wgrad for forced position splits next to the automatic choice (MNIST video-D / decoder shapes)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import gan_ode_amd._lib as L
from gan_ode_amd.engine import make_geom, conv_out, stream_ptr
lib = L.lib()

def g3(N, Ci, Co, xi, k, s, p):
    yo = tuple(conv_out(xi[a], k[a], s[a], p[a]) for a in range(3))
    return make_geom(N, Ci, Co, xi, yo, k, s, p)

cases = [("vidD L1 N=64", g3(64, 64, 128, (15, 15, 15), (2, 2, 2), (1, 2, 2), (0, 1, 1))),
         ("vidD L2 N=64", g3(64, 128, 256, (14, 8, 8), (2, 2, 2), (1, 2, 2), (0, 1, 1))),
         ("vidD L3 N=64", g3(64, 256, 512, (13, 5, 5), (2, 2, 2), (1, 2, 2), (0, 1, 1))),
         ("dec L1 N=512", make_geom(512, 256, 512, (1, 8, 8), (1, 4, 4), (1, 4, 4), (1, 2, 2), (0, 1, 1))),
         ("dec L2 N=512", make_geom(512, 128, 256, (1, 16, 16), (1, 8, 8), (1, 4, 4), (1, 2, 2), (0, 1, 1))),
         ("dec L3 N=512", make_geom(512, 64, 128, (1, 32, 32), (1, 16, 16), (1, 4, 4), (1, 2, 2), (0, 1, 1))),
         ("imgD L1 N=64", make_geom(64, 64, 128, (1, 14, 14), (1, 7, 7), (1, 4, 4), (1, 2, 2), (0, 1, 1))),
         ("imgD L2 N=64", make_geom(64, 128, 256, (1, 7, 7), (1, 3, 3), (1, 4, 4), (1, 2, 2), (0, 1, 1))),
         ("dec L1 N=544", make_geom(544, 256, 512, (1, 8, 8), (1, 4, 4), (1, 4, 4), (1, 2, 2), (0, 1, 1))),
         ("dec L2 N=544", make_geom(544, 128, 256, (1, 16, 16), (1, 8, 8), (1, 4, 4), (1, 2, 2), (0, 1, 1))),
         ("dec L3 N=544", make_geom(544, 64, 128, (1, 32, 32), (1, 16, 16), (1, 4, 4), (1, 2, 2), (0, 1, 1))),
         ("ucf vidD L1 N=32", g3(32, 64, 128, (13, 32, 32), (4, 4, 4), (1, 2, 2), (0, 1, 1))),
         ("ucf vidD L2 N=32", g3(32, 128, 256, (10, 16, 16), (4, 4, 4), (1, 2, 2), (0, 1, 1))),
         ("ucf vidD L3 N=32", g3(32, 256, 512, (7, 8, 8), (4, 4, 4), (1, 2, 2), (0, 1, 1))),
         ("ucf dec L1 N=272", make_geom(272, 256, 512, (1, 8, 8), (1, 4, 4), (1, 4, 4), (1, 2, 2), (0, 1, 1))),
         ("ucf dec L2 N=272", make_geom(272, 128, 256, (1, 16, 16), (1, 8, 8), (1, 4, 4), (1, 2, 2), (0, 1, 1))),
         ("ucf dec L3 N=272", make_geom(272, 64, 128, (1, 32, 32), (1, 16, 16), (1, 4, 4), (1, 2, 2), (0, 1, 1)))]
for name, g in cases:
    x = torch.randn(g.N, g.Di, g.Hi, g.Wi, g.Ci, device="cuda")
    y = torch.randn(g.N, g.Do, g.Ho, g.Wo, g.Co, device="cuda")
    dw = torch.empty(g.Co, g.Ci, g.kd, g.kh, g.kw, device="cuda")
    fl = 2.0 * g.N * g.Do * g.Ho * g.Wo * g.Co * g.Ci * g.kd * g.kh * g.kw
    res = []
    auto = lib.gode_wgrad_auto_splits(C.byref(g))
    for sp in [0, 1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 128]:
        op = L.WgradOp(g=g, act=L.ACT_NONE, xform_on_y=0, splits=sp, accumulate=0, x=x.data_ptr(), y=y.data_ptr(), dw=dw.data_ptr())
        ws = lib.gode_wgrad_work_size(C.byref(op))
        if ws * 4 > 600e6:
            continue
        work = torch.empty(max(ws, 1), device="cuda"); op.work = work.data_ptr()
        st = stream_ptr()
        try:
            for _ in range(2): L.run_one(op, st)
        except RuntimeError:
            continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): L.run_one(op, st)
        e1.record(); torch.cuda.synchronize()
        res.append((e0.elapsed_time(e1) / 10 * 1e3, sp))
    a = [r for r in res if r[1] == 0][0]
    best = min(r for r in res if r[1] != 0)
    print(f"{name:14s} auto({auto}) {a[0]:7.1f} us {fl/a[0]/1e6:6.1f} TF | best s={best[1]} {best[0]:7.1f} us {fl/best[0]/1e6:6.1f} TF | " + " ".join(f"{s}:{t:.0f}" for t, s in sorted(res)[:6]), flush=True)
