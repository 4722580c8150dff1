import ctypes as C, os, sys
sys.path.insert(0, "/root/repo")
import torch
import gan_ode_amd._lib as L
from gan_ode_amd.engine import make_geom, stream_ptr
lib = L.lib()
R = 512
_w = torch.randn(8192, 8192, device="cuda")
for _ in range(40):      # ~100 ms of load first: the clock ramps for the first tens of ms of a process
    _w = _w @ _w * 1e-4
torch.cuda.synchronize()
for cin in (512, 256, 128, 64, 64, 128, 256, 512):     # the first pass also warms the clock; read the second
    # convT cin->64 k4 s2 16x16 -> 32x32 as the conv whose dgrad it is: Ci=64, Co=cin
    g = make_geom(R, 64, cin, (1, 32, 32), (1, 16, 16), (1, 4, 4), (1, 2, 2), (0, 1, 1))
    src = torch.randn(R, 1, 16, 16, cin, device="cuda")
    w = torch.randn(cin, 64, 1, 4, 4, device="cuda") * 0.05
    wp = torch.empty(lib.gode_pack_size(C.byref(g), L.DGRAD), device="cuda")
    L.check(lib.gode_pack_weights(C.byref(g), L.DGRAD, w.data_ptr(), wp.data_ptr(), None, 0, stream_ptr()))
    out = torch.empty(R, 1, 32, 32, 64, device="cuda")
    op = L.IgemmOp(g=g, dir=L.DGRAD, act=0, epilogue=0, tile=int(os.environ.get("TILE", "2")), src=src.data_ptr(), wpack=wp.data_ptr(), out=out.data_ptr())
    rows = lib.gode_igemm_stats_rows(C.byref(op))
    stats = torch.empty(rows * 2 * 64 + 16, device="cuda"); op.stats = stats.data_ptr()
    st = stream_ptr()
    for _ in range(5): L.run_one(op, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): L.run_one(op, st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 30
    fl = 2.0 * R * 16 * 16 * 64 * cin * 16
    print(f"cin={cin:4d} K/phase={4*cin:5d} slabs={4*cin//32:3d}  {ms*1e3:8.1f} us  {fl/ms/1e9:6.1f} TF")
