import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gan_ode_amd as G
G.limit_host_threads()
torch.manual_seed(0); np.random.seed(0)
gen = G.VideoGeneratorMNISTODERNN(1, 50, 0, 16, 16).cuda()
B, T = 32, 16
def pieces():
    t = [time.perf_counter()]
    content, noise = gen._draw(B * T * 2, T); t.append(time.perf_counter())
    j = np.sort(np.random.choice(B * T * 2 * T, B, replace=False)).astype(np.int64); t.append(time.perf_counter())
    traj = torch.from_numpy(j // T); sel = torch.from_numpy((j % T).astype(np.int32)); t.append(time.perf_counter())
    a = noise[:, traj].contiguous(); t.append(time.perf_counter())
    b = content[traj].contiguous(); t.append(time.perf_counter())
    with torch.no_grad():
        h = gen._run(B, T, True, a, b, sel); t.append(time.perf_counter())
    torch.cuda.synchronize(); t.append(time.perf_counter())
    return [round((t[i + 1] - t[i]) * 1e3, 2) for i in range(len(t) - 1)]
for mode in ("run1", "run2"):
    print(mode)
    for i in range(12):
        print("  draw, choice, from_numpy, noise[:,traj], content[traj], _run, sync =", pieces())
import glob
def threads():
    out = {}
    for p in glob.glob("/proc/self/task/*/stat"):
        try:
            f = open(p).read().split()
            out[f[0]] = (open(p.replace("stat", "comm")).read().strip(), int(f[13]) + int(f[14]))
        except Exception:
            pass
    return out
t0 = threads(); w0 = time.perf_counter()
for i in range(40): pieces()
t1 = threads(); w1 = time.perf_counter()
print("wall", round(w1 - w0, 3), "s; threads:", len(t1), "torch threads", torch.get_num_threads())
for tid, (comm, ticks) in sorted(t1.items(), key=lambda kv: -(kv[1][1] - t0.get(kv[0], ("", 0))[1]))[:8]:
    print("  ", tid, comm, "cpu ticks", ticks - t0.get(tid, ("", 0))[1])
try:
    print(open("/sys/fs/cgroup/cpu.max").read().strip(), "|", open("/sys/fs/cgroup/cpu.stat").read().replace("\n", " "))
except Exception as e:
    print("cgroup", e)
