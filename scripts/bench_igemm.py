"""Micro-benchmark (GPU box): times single igemm / wgrad launches on the layer shapes of the MNIST config with
HIP events and prints achieved fp32 TFLOP/s (algorithmic FLOPs / launch time) per tile choice.
    python scripts/bench_igemm.py [--reps 20] [--only dec|disc|wgrad]"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import gan_ode_amd._lib as L
from gan_ode_amd.engine import conv_out, make_geom, stream_ptr

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--only", default="")
ap.add_argument("--tiles", default="0")
ap.add_argument("--noxf", action="store_true", help="forward GEMMs on materialised (already activated) inputs")
a = ap.parse_args()
lib = L.lib()
R = int(os.environ.get("GODE_BENCH_ROWS", "512"))


def timeit(op):
    st = stream_ptr()
    for _ in range(3):
        L.run_one(op, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        L.run_one(op, st)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.reps


def run_igemm(name, g, direction, flop, xform=True, tiles=(0,)):
    src_dims = (g.N, g.Do, g.Ho, g.Wo, g.Co) if direction == L.DGRAD else (g.N, g.Di, g.Hi, g.Wi, g.Ci)
    out_dims = (g.N, g.Di, g.Hi, g.Wi, g.Ci) if direction == L.DGRAD else (g.N, g.Do, g.Ho, g.Wo, g.Co)
    src = torch.randn(src_dims, device="cuda")
    w = torch.randn(g.Co, g.Ci, g.kd, g.kh, g.kw, device="cuda") * 0.05
    wp = torch.empty(lib.gode_pack_size(C.byref(g), direction), device="cuda")
    L.check(lib.gode_pack_weights(C.byref(g), direction, w.data_ptr(), wp.data_ptr(), None, 0, stream_ptr()))
    out = torch.empty(out_dims, device="cuda")
    Cg = src_dims[-1]
    sc, sh = torch.rand(Cg, device="cuda") + 0.5, torch.randn(Cg, device="cuda")
    for tile in tiles:
        op = L.IgemmOp(g=g, dir=direction, act=L.ACT_RELU if xform else L.ACT_NONE, epilogue=L.EPI_RAW, tile=tile, src=src.data_ptr(),
                       wpack=wp.data_ptr(), out=out.data_ptr(), scale=sc.data_ptr() if xform else None,
                       shift=sh.data_ptr() if xform else None)
        work = torch.empty(max(lib.gode_igemm_work_size(C.byref(op)), 1), device="cuda")
        op.work = work.data_ptr()
        rows = lib.gode_igemm_stats_rows(C.byref(op))
        stats = torch.empty(rows * 2 * out_dims[-1] * 16 + 16, device="cuda")
        op.stats = stats.data_ptr()
        ms = timeit(op)
        print(f"{name:34s} tile={tile} {ms*1e3:9.1f} us  {flop/ms/1e9:7.2f} TFLOP/s  ({flop/1e9:.2f} GFLOP)")


def run_wgrad(name, g, flop, on_y):
    x = torch.randn(g.N, g.Di, g.Hi, g.Wi, g.Ci, device="cuda")
    y = torch.randn(g.N, g.Do, g.Ho, g.Wo, g.Co, device="cuda")
    dw = torch.empty(g.Co, g.Ci, g.kd, g.kh, g.kw, device="cuda")
    Cx = g.Co if on_y else g.Ci
    sc, sh = torch.rand(Cx, device="cuda") + 0.5, torch.randn(Cx, device="cuda")
    op = L.WgradOp(g=g, act=L.ACT_NONE if a.noxf else L.ACT_RELU, xform_on_y=on_y, splits=0, accumulate=0,
                   x=x.data_ptr(), y=y.data_ptr(), scale=None if a.noxf else sc.data_ptr(),
                   shift=None if a.noxf else sh.data_ptr(), dw=dw.data_ptr())
    work = torch.empty(lib.gode_wgrad_work_size(C.byref(op)), device="cuda")
    op.work = work.data_ptr()
    ms = timeit(op)
    print(f"{name:34s} splits={lib.gode_wgrad_auto_splits(C.byref(g))} {ms*1e3:9.1f} us  {flop/ms/1e9:7.2f} TFLOP/s (incl. reduce)")


tiles = tuple(int(t) for t in a.tiles.split(","))
dec = [("dec L1 convT 512->256 4->8", make_geom(R, 256, 512, (1, 8, 8), (1, 4, 4), (1, 4, 4), (1, 2, 2), (0, 1, 1))),
       ("dec L2 convT 256->128 8->16", make_geom(R, 128, 256, (1, 16, 16), (1, 8, 8), (1, 4, 4), (1, 2, 2), (0, 1, 1))),
       ("dec L3 convT 128->64 16->32", make_geom(R, 64, 128, (1, 32, 32), (1, 16, 16), (1, 4, 4), (1, 2, 2), (0, 1, 1)))]
flop_dec = 2.0 * 33554432 * R
if a.only in ("", "dec"):
    for n, g in dec:
        run_igemm(n + " fwd(DGRAD)", g, L.DGRAD, flop_dec, xform=not a.noxf, tiles=tiles)
    for n, g in dec:
        run_igemm(n + " bwd-data(FPROP)", g, L.FPROP, flop_dec, xform=False, tiles=tiles)
if a.only in ("", "wgrad"):
    for n, g in dec:
        run_wgrad(n + " wgrad", g, flop_dec, 1)
if a.only in ("", "disc"):
    B, D, H = 32, 16, 28
    chans = [1, 64, 128, 256, 512]
    for i in range(4):
        Do, Ho = conv_out(D, 2, 1, 0), conv_out(H, 2, 2, 1)
        g = make_geom(B, chans[i], chans[i + 1], (D, H, H), (Do, Ho, Ho), (2, 2, 2), (1, 2, 2), (0, 1, 1))
        flop = 2.0 * B * Do * Ho * Ho * chans[i + 1] * chans[i] * 8
        run_igemm(f"vidD L{i} conv3d {chans[i]}->{chans[i+1]} fwd", g, L.FPROP, flop, xform=i > 0, tiles=tiles)
        if i > 0:
            run_igemm(f"vidD L{i} bwd-data(DGRAD)", g, L.DGRAD, flop, xform=False, tiles=tiles)
            run_wgrad(f"vidD L{i} wgrad", g, flop, 0)
        D, H = Do, Ho
