"""Time per 10 training iterations next to the losses and gradient magnitudes (is the slow stretch data-dependent?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gan_ode_amd as G
G.limit_host_threads()
torch.manual_seed(0); np.random.seed(0)
gen, dv, di = G.build_mnist(); gen.cuda(); dv.cuda(); di.cuda()
B = 32
g = torch.Generator().manual_seed(1)
imgs = [torch.rand(B, 1, 28, 28, generator=g).cuda() for _ in range(2)]
vids = [torch.rand(B, 16, 1, 28, 28, generator=g).cuda() for _ in range(2)]
tr = G.GanTrainer(gen, dv, di)
for _ in range(3): tr.step(imgs, vids)
import gc
if os.environ.get("GC") == "freeze":
    gc.collect(); gc.freeze()
if os.environ.get("GC") == "off":
    gc.disable()
gc.callbacks.append(lambda phase, info: print(f"   [gc {phase} gen{info['generation']} collected={info.get('collected')}]") if phase == "stop" and info["generation"] == 2 else None)
import glob
def cg():
    try:
        d = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat").read().strip().splitlines())
        return int(d.get("nr_throttled", 0)), int(d.get("throttled_usec", 0)) / 1e3
    except Exception:
        return (0, 0.0)
def ticks():
    out = {}
    for p in glob.glob("/proc/self/task/*/stat"):
        try:
            f = open(p).read().rsplit(")", 1)[1].split()
            out[p.split("/")[4]] = (open(p.replace("/stat", "/comm")).read().strip(), int(f[11]) + int(f[12]))
        except Exception:
            pass
    return out
for blk in range(16):
    torch.cuda.synchronize(); c0 = cg(); k0 = ticks(); t0 = time.perf_counter()
    worst = 0.0
    for _ in range(10):
        a0 = time.perf_counter(); li, lv, lg = tr.step(imgs, vids); worst = max(worst, time.perf_counter() - a0)
    torch.cuda.synchronize(); dt = 1e2 * (time.perf_counter() - t0); c1 = cg(); k1 = ticks()
    busy = sorted(((k1[t][1] - k0.get(t, ("", 0))[1], k1[t][0]) for t in k1), reverse=True)[:4]
    print(f"   throttled +{c1[0]-c0[0]} periods +{c1[1]-c0[1]:.1f} ms; longest step() issue {worst*1e3:.1f} ms; threads {len(k1)}; busiest (ticks,comm) {busy}")
    gmax = max(float(p.grad.abs().max()) for p in gen.parameters() if p.grad is not None)
    gmin = min(float(p.grad.abs()[p.grad != 0].min()) if (p.grad != 0).any() else 1.0 for p in gen.parameters() if p.grad is not None)
    dgmax = max(float(p.grad.abs().max()) for p in dv.parameters() if p.grad is not None)
    den = sum(int(((p.grad != 0) & (p.grad.abs() < 1.2e-38)).sum()) for m in (gen, dv, di) for p in m.parameters() if p.grad is not None)
    print(f"it {3+10*(blk+1):4d} {dt:6.2f} ms/it  lossD_img {float(li):.3e} lossD_vid {float(lv):.3e} lossG {float(lg):.3e} "
          f"|gG|max {gmax:.2e} min {gmin:.2e} |gDv|max {dgmax:.2e} denormal grads {den}", flush=True)
