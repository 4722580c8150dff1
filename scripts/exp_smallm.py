"""Experiment (GPU box): the small-M GEMMs of the image path (N=32 frames) under different tile / split choices."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gan_ode_amd._lib as L
from gan_ode_amd.engine import make_geom, stream_ptr
lib = L.lib()
N = int(os.environ.get("N", "32"))
cases = [("fprop 256->512 8x8->4x4", make_geom(N, 256, 512, (1, 8, 8), (1, 4, 4), (1, 4, 4), (1, 2, 2), (0, 1, 1)), L.FPROP),
         ("dgrad 256->512 8x8->4x4", make_geom(N, 256, 512, (1, 8, 8), (1, 4, 4), (1, 4, 4), (1, 2, 2), (0, 1, 1)), L.DGRAD),
         ("fprop 128->256 16x16->8x8", make_geom(N, 128, 256, (1, 16, 16), (1, 8, 8), (1, 4, 4), (1, 2, 2), (0, 1, 1)), L.FPROP),
         ("dgrad 128->256 16x16->8x8", make_geom(N, 128, 256, (1, 16, 16), (1, 8, 8), (1, 4, 4), (1, 2, 2), (0, 1, 1)), L.DGRAD),
         ("fprop 64->128 32x32->16x16", make_geom(N, 64, 128, (1, 32, 32), (1, 16, 16), (1, 4, 4), (1, 2, 2), (0, 1, 1)), L.FPROP),
         ("dgrad 64->128 32x32->16x16", make_geom(N, 64, 128, (1, 32, 32), (1, 16, 16), (1, 4, 4), (1, 2, 2), (0, 1, 1)), L.DGRAD),
         ("fprop 512->96 4x4->1x1", make_geom(N * 16, 512, 96, (1, 4, 4), (1, 1, 1), (1, 4, 4), (1, 1, 1), (0, 0, 0)), L.FPROP),
         ("imgD fprop 128->256 7x7->3x3", make_geom(N, 128, 256, (1, 7, 7), (1, 3, 3), (1, 4, 4), (1, 2, 2), (0, 1, 1)), L.FPROP),
         ("imgD fprop 64->128 14x14->7x7", make_geom(N, 64, 128, (1, 14, 14), (1, 7, 7), (1, 4, 4), (1, 2, 2), (0, 1, 1)), L.FPROP),
         ("imgD dgrad 128->256 7x7->3x3", make_geom(N, 128, 256, (1, 7, 7), (1, 3, 3), (1, 4, 4), (1, 2, 2), (0, 1, 1)), L.DGRAD),
         ("imgD dgrad 64->128 14x14->7x7", make_geom(N, 64, 128, (1, 14, 14), (1, 7, 7), (1, 4, 4), (1, 2, 2), (0, 1, 1)), L.DGRAD)]
for name, g, d in cases:
    src_dims = (g.N, g.Do, g.Ho, g.Wo, g.Co) if d == L.DGRAD else (g.N, g.Di, g.Hi, g.Wi, g.Ci)
    out_dims = (g.N, g.Di, g.Hi, g.Wi, g.Ci) if d == L.DGRAD else (g.N, g.Do, g.Ho, g.Wo, g.Co)
    src = torch.randn(src_dims, device="cuda")
    w = torch.randn(g.Co, g.Ci, g.kd, g.kh, g.kw, device="cuda") * 0.05
    wp = torch.empty(lib.gode_pack_size(C.byref(g), d), device="cuda")
    L.check(lib.gode_pack_weights(C.byref(g), d, w.data_ptr(), wp.data_ptr(), None, 0, stream_ptr()))
    out = torch.empty(out_dims, device="cuda")
    fl = 2.0 * g.N * g.Do * g.Ho * g.Wo * g.Co * g.Ci * g.kd * g.kh * g.kw
    for tile in [int(t) for t in os.environ.get("TILES", "0").split(",")]:
        op = L.IgemmOp(g=g, dir=d, act=L.ACT_NONE, epilogue=L.EPI_RAW, tile=tile, src=src.data_ptr(), wpack=wp.data_ptr(),
                       out=out.data_ptr())
        work = torch.empty(max(lib.gode_igemm_work_size(C.byref(op)), 1), device="cuda")
        op.work = work.data_ptr()
        rows = lib.gode_igemm_stats_rows(C.byref(op))
        stats = torch.empty(rows * 2 * out_dims[-1] + 16, device="cuda")
        op.stats = stats.data_ptr()
        st = stream_ptr()
        for _ in range(3):
            L.run_one(op, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            L.run_one(op, st)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"{name:32s} tile={tile} work={work.numel()*4/1e6:6.1f}MB {ms*1e3:8.1f} us {fl/ms/1e9:7.1f} TF", flush=True)
