"""Times the input-gradient GEMMs of the UCF video discriminator (k=4 temporal taps) and of the MNIST decoder."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import gan_ode_amd._lib as L
from gan_ode_amd.engine import make_geom, stream_ptr
lib = L.lib()

def timeit(op, reps=20):
    st = stream_ptr()
    for _ in range(3): L.run_one(op, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): L.run_one(op, st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

def run(name, g, direction):
    src_dims = (g.N, g.Do, g.Ho, g.Wo, g.Co) if direction == L.DGRAD else (g.N, g.Di, g.Hi, g.Wi, g.Ci)
    out_dims = (g.N, g.Di, g.Hi, g.Wi, g.Ci) if direction == L.DGRAD else (g.N, g.Do, g.Ho, g.Wo, g.Co)
    src = torch.randn(src_dims, device="cuda")
    w = torch.randn(g.Co, g.Ci, g.kd, g.kh, g.kw, device="cuda") * 0.05
    wp = torch.empty(lib.gode_pack_size(C.byref(g), direction), device="cuda")
    L.check(lib.gode_pack_weights(C.byref(g), direction, w.data_ptr(), wp.data_ptr(), None, 0, stream_ptr()))
    out = torch.empty(out_dims, device="cuda")
    op = L.IgemmOp(g=g, dir=direction, act=L.ACT_NONE, epilogue=L.EPI_RAW, tile=0, src=src.data_ptr(), wpack=wp.data_ptr(), out=out.data_ptr())
    work = torch.empty(max(lib.gode_igemm_work_size(C.byref(op)), 1), device="cuda")
    op.work = work.data_ptr()
    flop = 2.0 * g.N * g.Do * g.Ho * g.Wo * g.Co * g.Ci * g.kd * g.kh * g.kw
    ms = timeit(op)
    print(f"{name:40s} {ms*1e3:9.1f} us {flop/ms/1e9:7.1f} TF", flush=True)

for N in (32, 16):
    run(f"ucf dv L1 dgrad N={N}", make_geom(N, 64, 128, (13, 32, 32), (10, 16, 16), (4, 4, 4), (1, 2, 2), (0, 1, 1)), L.DGRAD)
    run(f"ucf dv L2 dgrad N={N}", make_geom(N, 128, 256, (10, 16, 16), (7, 8, 8), (4, 4, 4), (1, 2, 2), (0, 1, 1)), L.DGRAD)
    run(f"ucf dv L3 dgrad N={N}", make_geom(N, 256, 512, (7, 8, 8), (4, 4, 4), (4, 4, 4), (1, 2, 2), (0, 1, 1)), L.DGRAD)
run("ucf dv L1 fprop N=32", make_geom(32, 64, 128, (13, 32, 32), (10, 16, 16), (4, 4, 4), (1, 2, 2), (0, 1, 1)), L.FPROP)
run("dec L1 convT 512->256 fwd", make_geom(512, 256, 512, (1, 8, 8), (1, 4, 4), (1, 4, 4), (1, 2, 2), (0, 1, 1)), L.DGRAD)
run("dec L3 convT 128->64 fwd", make_geom(512, 64, 128, (1, 32, 32), (1, 16, 16), (1, 4, 4), (1, 2, 2), (0, 1, 1)), L.DGRAD)
run("dec L3 bwd (fprop)", make_geom(512, 64, 128, (1, 32, 32), (1, 16, 16), (1, 4, 4), (1, 2, 2), (0, 1, 1)), L.FPROP)
run("mnist dv L2 dgrad N=64", make_geom(64, 128, 256, (14, 8, 8), (13, 5, 5), (2, 2, 2), (1, 2, 2), (0, 1, 1)), L.DGRAD)
