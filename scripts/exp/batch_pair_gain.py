"""Would batching D(real) and D(fake) into one N=2B pass pay?  Times the video-D / image-D GEMMs (fprop, dgrad, wgrad incl.
its reduce) at N=32 (x2 launches) vs N=64 (x1).   python scripts/exp/batch_pair_gain.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import gan_ode_amd._lib as L
from gan_ode_amd.engine import make_geom, conv_out, stream_ptr
lib = L.lib()
def g3(N, Ci, Co, xi, k, s, p):
    yo = tuple(conv_out(xi[a], k[a], s[a], p[a]) for a in range(3))
    return make_geom(N, Ci, Co, xi, yo, k, s, p)
def timeit(op, reps=20):
    st = stream_ptr()
    for _ in range(3): L.run_one(op, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): L.run_one(op, st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
shapes = [("vidD L1 64->128", 64, 128, (15, 15, 15), (2, 2, 2), (1, 2, 2), (0, 1, 1)),
          ("vidD L2 128->256", 128, 256, (14, 8, 8), (2, 2, 2), (1, 2, 2), (0, 1, 1)),
          ("vidD L3 256->512", 256, 512, (13, 5, 5), (2, 2, 2), (1, 2, 2), (0, 1, 1)),
          ("imgD L1 64->128", 64, 128, (1, 14, 14), (1, 4, 4), (1, 2, 2), (0, 1, 1)),
          ("imgD L2 128->256", 128, 256, (1, 7, 7), (1, 4, 4), (1, 2, 2), (0, 1, 1))]
tot = {32: 0.0, 64: 0.0}
for name, Ci, Co, xi, k, s, p in shapes:
    row = []
    for N in (32, 64):
        g = g3(N, Ci, Co, xi, k, s, p)
        fl = 2.0 * g.N * g.Do * g.Ho * g.Wo * g.Co * g.Ci * g.kd * g.kh * g.kw
        x = torch.randn(g.N, g.Di, g.Hi, g.Wi, g.Ci, device="cuda"); y = torch.randn(g.N, g.Do, g.Ho, g.Wo, g.Co, device="cuda")
        w = torch.randn(g.Co, g.Ci, g.kd, g.kh, g.kw, device="cuda") * 0.05
        ts = []
        for d, src, out in ((L.FPROP, x, y), (L.DGRAD, y, x)):
            wp = torch.empty(lib.gode_pack_size(C.byref(g), d), device="cuda")
            L.check(lib.gode_pack_weights(C.byref(g), d, w.data_ptr(), wp.data_ptr(), None, 0, stream_ptr()))
            o = torch.empty_like(out)
            op = L.IgemmOp(g=g, dir=d, act=L.ACT_NONE, epilogue=L.EPI_RAW, tile=0, src=src.data_ptr(), wpack=wp.data_ptr(), out=o.data_ptr())
            work = torch.empty(max(lib.gode_igemm_work_size(C.byref(op)), 1), device="cuda"); op.work = work.data_ptr()
            if d == L.FPROP:
                stats = torch.empty(lib.gode_igemm_stats_rows(C.byref(op)) * 2 * g.Co + 16, device="cuda"); op.stats = stats.data_ptr()
            ts.append(timeit(op))
        dw = torch.empty_like(w)
        wop = L.WgradOp(g=g, act=L.ACT_NONE, xform_on_y=0, splits=0, accumulate=0, x=x.data_ptr(), y=y.data_ptr(), dw=dw.data_ptr())
        wk = torch.empty(lib.gode_wgrad_work_size(C.byref(wop)), device="cuda"); wop.work = wk.data_ptr()
        ts.append(timeit(wop))
        row.append((N, fl, ts))
    (n1, f1, t1), (n2, f2, t2) = row
    print(f"{name:18s} N=32: fprop {t1[0]:6.1f} dgrad {t1[1]:6.1f} wgrad {t1[2]:6.1f} us ({f1/t1[0]/1e6:5.1f}/{f1/t1[1]/1e6:5.1f}/{f1/t1[2]/1e6:5.1f} TF)   "
          f"N=64: fprop {t2[0]:6.1f} dgrad {t2[1]:6.1f} wgrad {t2[2]:6.1f} us ({f2/t2[0]/1e6:5.1f}/{f2/t2[1]/1e6:5.1f}/{f2/t2[2]/1e6:5.1f} TF)   "
          f"2x32 = {2*sum(t1):6.1f} us vs 1x64 = {sum(t2):6.1f} us")
    tot[32] += 2 * sum(t1); tot[64] += sum(t2)
print(f"sum over shapes: two N=32 passes {tot[32]:.1f} us, one N=64 pass {tot[64]:.1f} us")
