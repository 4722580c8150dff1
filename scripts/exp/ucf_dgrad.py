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

def run_wgrad(name, g, on_y=False, strided=False):
    if strided:   # [N, D, C, H, W] memory (the real-video tensor), logical strides N, D, H, W, C
        xm = torch.randn(g.N, g.Di, g.Ci, g.Hi, g.Wi, device="cuda")
        xs = (xm.stride(0), xm.stride(1), xm.stride(3), xm.stride(4), xm.stride(2))
    else:
        xm = torch.randn(g.N, g.Di, g.Hi, g.Wi, g.Ci, device="cuda"); xs = None
    y = torch.randn(g.N, g.Do, g.Ho, g.Wo, g.Co, device="cuda")
    dw = torch.empty(g.Co, g.Ci, g.kd, g.kh, g.kw, device="cuda")
    sc, sh = torch.rand(g.Co, device="cuda") + 0.5, torch.randn(g.Co, device="cuda")
    op = L.WgradOp(g=g, act=L.ACT_RELU if on_y else L.ACT_NONE, xform_on_y=1 if on_y else 0, splits=0, accumulate=0, x=xm.data_ptr(), y=y.data_ptr(),
                   scale=sc.data_ptr() if on_y else None, shift=sh.data_ptr() if on_y else None, dw=dw.data_ptr())
    if xs:
        for i in range(5): op.xs[i] = xs[i]
    work = torch.empty(lib.gode_wgrad_work_size(C.byref(op)), device="cuda")
    op.work = work.data_ptr()
    flop = 2.0 * g.N * g.Do * g.Ho * g.Wo * g.Co * g.Ci * g.kd * g.kh * g.kw
    ms = timeit(op)
    print(f"{name:40s} {ms*1e3:9.1f} us {flop/ms/1e9:7.1f} TF", flush=True)

run_wgrad("ucf dv L0 wgrad N=16 (channels-last)", make_geom(16, 3, 64, (16, 64, 64), (13, 32, 32), (4, 4, 4), (1, 2, 2), (0, 1, 1)))
run_wgrad("ucf dv L0 wgrad N=16 (NDCHW video)", make_geom(16, 3, 64, (16, 64, 64), (13, 32, 32), (4, 4, 4), (1, 2, 2), (0, 1, 1)), strided=True)
run_wgrad("ucf G head wgrad N=256", make_geom(256, 3, 64, (1, 64, 64), (1, 32, 32), (1, 4, 4), (1, 2, 2), (0, 1, 1)), on_y=True)
run_wgrad("ucf di L0 wgrad N=16", make_geom(16, 3, 64, (1, 64, 64), (1, 32, 32), (1, 4, 4), (1, 2, 2), (0, 1, 1)))
run_wgrad("mnist dv L0 wgrad N=32", make_geom(32, 1, 64, (16, 28, 28), (15, 15, 15), (2, 2, 2), (1, 2, 2), (0, 1, 1)), strided=True)
run_wgrad("mnist di L0 wgrad N=32", make_geom(32, 1, 64, (1, 28, 28), (1, 14, 14), (1, 4, 4), (1, 2, 2), (0, 1, 1)))
run_wgrad("mnist G head wgrad N=512", make_geom(512, 1, 64, (1, 28, 28), (1, 32, 32), (1, 1, 1), (1, 1, 1), (0, 2, 2)), on_y=True)

def run_fprop_strided(name, g, ndchw):
    if ndchw:   # [N, D, C, H, W] memory (the real-video tensor)
        xm = torch.randn(g.N, g.Di, g.Ci, g.Hi, g.Wi, device="cuda")
        xs = (xm.stride(0), xm.stride(1), xm.stride(3), xm.stride(4), xm.stride(2))
    else:
        xm = torch.randn(g.N, g.Di, g.Hi, g.Wi, g.Ci, device="cuda"); xs = None
    w = torch.randn(g.Co, g.Ci, g.kd, g.kh, g.kw, device="cuda") * 0.05
    wp = torch.empty(lib.gode_pack_size(C.byref(g), L.FPROP), device="cuda")
    L.check(lib.gode_pack_weights(C.byref(g), L.FPROP, w.data_ptr(), wp.data_ptr(), None, 0, stream_ptr()))
    out = torch.empty(g.N, g.Do, g.Ho, g.Wo, g.Co, device="cuda")
    op = L.IgemmOp(g=g, dir=L.FPROP, act=L.ACT_NONE, epilogue=L.EPI_RAW, tile=0, src=xm.data_ptr(), wpack=wp.data_ptr(), out=out.data_ptr())
    if xs:
        for i in range(5): op.gs[i] = xs[i]
    work = torch.empty(max(lib.gode_igemm_work_size(C.byref(op)), 1), device="cuda"); op.work = work.data_ptr()
    flop = 2.0 * g.N * g.Do * g.Ho * g.Wo * g.Co * g.Ci * g.kd * g.kh * g.kw
    ms = timeit(op)
    print(f"{name:40s} {ms*1e3:9.1f} us {flop/ms/1e9:7.1f} TF", flush=True)

run_fprop_strided("ucf dv L0 fprop N=16 (channels-last)", make_geom(16, 3, 64, (16, 64, 64), (13, 32, 32), (4, 4, 4), (1, 2, 2), (0, 1, 1)), False)
run_fprop_strided("ucf dv L0 fprop N=16 (NDCHW video)", make_geom(16, 3, 64, (16, 64, 64), (13, 32, 32), (4, 4, 4), (1, 2, 2), (0, 1, 1)), True)
run_fprop_strided("ucf di L0 fprop N=16", make_geom(16, 3, 64, (1, 64, 64), (1, 32, 32), (1, 4, 4), (1, 2, 2), (0, 1, 1)), False)
run_fprop_strided("mnist dv L0 fprop N=32 (NDCHW)", make_geom(32, 1, 64, (16, 28, 28), (15, 15, 15), (2, 2, 2), (1, 2, 2), (0, 1, 1)), True)
run_fprop_strided("mnist di L0 fprop N=32", make_geom(32, 1, 64, (1, 28, 28), (1, 14, 14), (1, 4, 4), (1, 2, 2), (0, 1, 1)), False)
