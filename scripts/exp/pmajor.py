"""Border-tap skipping with position-major rows on the decoder GEMMs: time per (tile, ksplit) with and without.
Record of an experiment ("Tried and dropped"): the GODE_IGEMM_PMAJOR hook it drove was removed from
igemm.hip again, so today both halves of the output are the plain kernel; GODE_STATS=0 drops the BatchNorm partial sums."""
import ctypes as C, os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) < 2:
    for pm in ("0", "1"):
        for force in ("",):
            env = dict(os.environ, GODE_IGEMM_PMAJOR=pm)
            if force:
                env.update(GODE_IGEMM_SWEEP="1", GODE_IGEMM_FORCE=force)
            print(f"== PMAJOR={pm} FORCE={force or 'model'}", flush=True)
            subprocess.run([sys.executable, __file__, "run"], env=env, check=True)
    sys.exit(0)
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

def run(name, g, direction, stats):
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
    if stats:
        rows = lib.gode_igemm_stats_rows(C.byref(op))
        st = torch.empty(rows * 2 * out_dims[-1] + 16, device="cuda"); op.stats = st.data_ptr()
    flop = 2.0 * g.N * g.Do * g.Ho * g.Wo * g.Co * g.Ci * g.kd * g.kh * g.kw
    ms = timeit(op)
    print(f"  {name:34s} {ms*1e3:9.1f} us {flop/ms/1e9:7.1f} TF", flush=True)

for N in (512, 256):
    run(f"dec L1 fwd 512->256 4->8 N={N}", make_geom(N, 256, 512, (1, 8, 8), (1, 4, 4), (1, 4, 4), (1, 2, 2), (0, 1, 1)), L.DGRAD, os.environ.get("GODE_STATS", "1") != "0")
    run(f"dec L2 fwd 256->128 8->16 N={N}", make_geom(N, 128, 256, (1, 16, 16), (1, 8, 8), (1, 4, 4), (1, 2, 2), (0, 1, 1)), L.DGRAD, os.environ.get("GODE_STATS", "1") != "0")
    run(f"dec L3 fwd 128->64 16->32 N={N}", make_geom(N, 64, 128, (1, 32, 32), (1, 16, 16), (1, 4, 4), (1, 2, 2), (0, 1, 1)), L.DGRAD, os.environ.get("GODE_STATS", "1") != "0")
run("dec L1 bwd-data N=512", make_geom(512, 256, 512, (1, 8, 8), (1, 4, 4), (1, 4, 4), (1, 2, 2), (0, 1, 1)), L.FPROP, False)
run("dec L2 bwd-data N=512", make_geom(512, 128, 256, (1, 16, 16), (1, 8, 8), (1, 4, 4), (1, 2, 2), (0, 1, 1)), L.FPROP, False)
