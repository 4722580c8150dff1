"""Calibration sweep (GPU box): time GEMM (+ split-K reduce) for forced (tile, ksplit) choices on the mid-size layer
shapes, next to what the cost model picks.   GODE_IGEMM_SWEEP=1 python scripts/sweep_tiles.py"""
import ctypes as C, os, sys
os.environ["GODE_IGEMM_SWEEP"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gan_ode_amd._lib as L
from gan_ode_amd.engine import make_geom, conv_out, stream_ptr
lib = L.lib()


def g3(N, Ci, Co, xi, k, s, p):
    yo = tuple(conv_out(xi[a], k[a], s[a], p[a]) for a in range(3))
    return make_geom(N, Ci, Co, xi, yo, k, s, p)


cases = [("dec L1 N=512", make_geom(512, 256, 512, (1, 8, 8), (1, 4, 4), (1, 4, 4), (1, 2, 2), (0, 1, 1))),
         ("dec L2 N=512", make_geom(512, 128, 256, (1, 16, 16), (1, 8, 8), (1, 4, 4), (1, 2, 2), (0, 1, 1))),
         ("dec L3 N=512", make_geom(512, 64, 128, (1, 32, 32), (1, 16, 16), (1, 4, 4), (1, 2, 2), (0, 1, 1))),
         ("ucf dec L1 N=256", make_geom(256, 256, 512, (1, 8, 8), (1, 4, 4), (1, 4, 4), (1, 2, 2), (0, 1, 1))),
         ("ucf dec L2 N=256", make_geom(256, 128, 256, (1, 16, 16), (1, 8, 8), (1, 4, 4), (1, 2, 2), (0, 1, 1))),
         ("mnist vidD L1 N=64", g3(64, 64, 128, (15, 15, 15), (2, 2, 2), (1, 2, 2), (0, 1, 1))),     # paired D(real)+D(fake) passes
         ("mnist vidD L2 N=64", g3(64, 128, 256, (14, 8, 8), (2, 2, 2), (1, 2, 2), (0, 1, 1))),
         ("mnist vidD L3 N=64", g3(64, 256, 512, (13, 5, 5), (2, 2, 2), (1, 2, 2), (0, 1, 1))),
         ("imgD L1 N=64", make_geom(64, 64, 128, (1, 14, 14), (1, 7, 7), (1, 4, 4), (1, 2, 2), (0, 1, 1))),
         ("imgD L2 N=64", make_geom(64, 128, 256, (1, 7, 7), (1, 3, 3), (1, 4, 4), (1, 2, 2), (0, 1, 1))),
         ("mnist vidD L1", g3(32, 64, 128, (15, 15, 15), (2, 2, 2), (1, 2, 2), (0, 1, 1))),
         ("mnist vidD L2", g3(32, 128, 256, (14, 8, 8), (2, 2, 2), (1, 2, 2), (0, 1, 1))),
         ("mnist vidD L3", g3(32, 256, 512, (13, 5, 5), (2, 2, 2), (1, 2, 2), (0, 1, 1))),
         ("ucf vidD L1", g3(16, 64, 128, (13, 32, 32), (4, 4, 4), (1, 2, 2), (0, 1, 1))),
         ("ucf vidD L2", g3(16, 128, 256, (10, 16, 16), (4, 4, 4), (1, 2, 2), (0, 1, 1))),
         ("ucf vidD L3", g3(16, 256, 512, (7, 8, 8), (4, 4, 4), (1, 2, 2), (0, 1, 1))),
         ("ucf vidD L3 N=32", g3(32, 256, 512, (7, 8, 8), (4, 4, 4), (1, 2, 2), (0, 1, 1))),
         ("ucf vidD L1 N=32", g3(32, 64, 128, (13, 32, 32), (4, 4, 4), (1, 2, 2), (0, 1, 1))),
         ("ucf vidD L2 N=32", g3(32, 128, 256, (10, 16, 16), (4, 4, 4), (1, 2, 2), (0, 1, 1))),
         ("img convT1 N=32", make_geom(32, 256, 512, (1, 8, 8), (1, 4, 4), (1, 4, 4), (1, 2, 2), (0, 1, 1))),
         ("img convT2 N=32", make_geom(32, 128, 256, (1, 16, 16), (1, 8, 8), (1, 4, 4), (1, 2, 2), (0, 1, 1))),
         ("img convT3 N=32", make_geom(32, 64, 128, (1, 32, 32), (1, 16, 16), (1, 4, 4), (1, 2, 2), (0, 1, 1))),
         ("imgD L1 N=32", make_geom(32, 64, 128, (1, 14, 14), (1, 7, 7), (1, 4, 4), (1, 2, 2), (0, 1, 1))),
         ("imgD L2 N=32", make_geom(32, 128, 256, (1, 7, 7), (1, 3, 3), (1, 4, 4), (1, 2, 2), (0, 1, 1)))]
import os as _os
_f = _os.environ.get('GODE_SWEEP_FILTER')
if _f:
    cases = [c for c in cases if _f in c[0]]
for name, g in cases:
    for d, dn in ((L.FPROP, "fprop"), (L.DGRAD, "dgrad")):
        src_dims = (g.N, g.Do, g.Ho, g.Wo, g.Co) if d == L.DGRAD else (g.N, g.Di, g.Hi, g.Wi, g.Ci)
        out_dims = (g.N, g.Di, g.Hi, g.Wi, g.Ci) if d == L.DGRAD else (g.N, g.Do, g.Ho, g.Wo, g.Co)
        src = torch.randn(src_dims, device="cuda")
        w = torch.randn(g.Co, g.Ci, g.kd, g.kh, g.kw, device="cuda") * 0.05
        wp = torch.empty(lib.gode_pack_size(C.byref(g), d), device="cuda")
        L.check(lib.gode_pack_weights(C.byref(g), d, w.data_ptr(), wp.data_ptr(), None, 0, stream_ptr()))
        out = torch.empty(out_dims, device="cuda")
        fl = 2.0 * g.N * g.Do * g.Ho * g.Wo * g.Co * g.Ci * g.kd * g.kh * g.kw
        res = []
        for force in ["model"] + [f"{t},{k}" for t in (1, 2, 4, 5, 6) for k in (1, 2, 3, 4, 8, 16)]:
            if force == "model":
                os.environ.pop("GODE_IGEMM_FORCE", None)
            else:
                os.environ["GODE_IGEMM_FORCE"] = force
            op = L.IgemmOp(g=g, dir=d, act=L.ACT_NONE, epilogue=L.EPI_RAW, tile=0, src=src.data_ptr(), wpack=wp.data_ptr(), out=out.data_ptr())
            ws = lib.gode_igemm_work_size(C.byref(op))
            if ws < 0 or ws * 4 > 400e6:
                continue
            work = torch.empty(max(ws, 1), device="cuda")
            op.work = work.data_ptr()
            rows = lib.gode_igemm_stats_rows(C.byref(op))
            # (full-K DGRAD geometries -- 1x1x1 output, kernel = input extent -- have taps * Ci statistics columns, not Ci)
            fullk = d == L.DGRAD and (g.Do, g.Ho, g.Wo) == (1, 1, 1) and (g.Di, g.Hi, g.Wi) == (g.kd, g.kh, g.kw) and g.kd * g.kh * g.kw > 1
            ncols = g.kd * g.kh * g.kw * g.Ci if fullk else out_dims[-1]
            stats = torch.empty(max(rows, 1) * 2 * ncols + 16, device="cuda")
            op.stats = stats.data_ptr()
            st = stream_ptr()
            try:
                for _ in range(2):
                    L.run_one(op, st)
            except RuntimeError:
                continue
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                L.run_one(op, st)
            e1.record(); torch.cuda.synchronize()
            res.append((e0.elapsed_time(e1) / 10 * 1e3, force, ws * 4 / 1e6))
        model = [r for r in res if r[1] == "model"][0]
        best = min(r for r in res if r[1] != "model")
        print(f"{name:16s} {dn}  model {model[0]:7.1f} us (work {model[2]:5.1f} MB)  best {best[1]:>5s} {best[0]:7.1f} us  {fl/best[0]/1e6:6.1f} TF | " +
              " ".join(f"{f}:{t:.0f}" for t, f, _ in sorted(res)[:6]), flush=True)
