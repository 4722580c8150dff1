"""Does the placement of the GEMM's buffers matter?  ConvT 128->64 (the dominant kernel) with the output / source
tensors carved out of one big allocation at different byte offsets (GPU box)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gan_ode_amd._lib as L
from gan_ode_amd.engine import make_geom, stream_ptr
lib = L.lib()
R, cin = 512, 128
g = make_geom(R, 64, cin, (1, 32, 32), (1, 16, 16), (1, 4, 4), (1, 2, 2), (0, 1, 1))
n_src, n_out = R * 16 * 16 * cin, R * 32 * 32 * 64
pool = torch.empty(n_src + n_out + (64 << 20), device="cuda")        # floats
base = pool.data_ptr()
w = torch.randn(cin, 64, 1, 4, 4, device="cuda") * 0.05
wp = torch.empty(lib.gode_pack_size(C.byref(g), L.DGRAD), device="cuda")
L.check(lib.gode_pack_weights(C.byref(g), L.DGRAD, w.data_ptr(), wp.data_ptr(), None, 0, stream_ptr()))
print("pool base %#x  (mod 2MiB = %d KiB)" % (base, (base % (2 << 20)) >> 10))
for src_off_kb, gap_kb in [(0, 0), (0, 4), (0, 64), (0, 256), (0, 1024), (0, 2048), (0, 4096 + 4), (64, 0), (1024, 1024)]:
    so = src_off_kb * 256                                   # floats
    oo = so + n_src + gap_kb * 256
    src = pool[so:so + n_src].view(R, 1, 16, 16, cin); src.normal_()
    out = pool[oo:oo + n_out].view(R, 1, 32, 32, 64)
    op = L.IgemmOp(g=g, dir=L.DGRAD, act=0, epilogue=0, tile=0, src=src.data_ptr(), wpack=wp.data_ptr(), out=out.data_ptr())
    rows = lib.gode_igemm_stats_rows(C.byref(op))
    stats = torch.empty(rows * 2 * 64 + 16, device="cuda"); op.stats = stats.data_ptr()
    st = stream_ptr()
    for _ in range(5): L.run_one(op, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): L.run_one(op, st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 30
    print(f"src +{src_off_kb:5d} KiB, out = src_end + {gap_kb:5d} KiB  (out-src = {(out.data_ptr()-src.data_ptr())/2**20:8.3f} MiB)  {ms*1e3:7.1f} us  {34.36/ms:6.1f} TF")
