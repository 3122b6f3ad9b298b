import os, time, torch
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: pass
print("torch threads", torch.get_num_threads())
for n in (8, 16, 32, 64):
    torch.set_num_threads(n)
    a = torch.randn(2048, 2048); b = torch.randn(2048, 2048)
    a @ b
    t0 = time.time()
    for _ in range(5): a @ b
    dt = (time.time() - t0) / 5
    print(n, "threads: %.1f GFLOP/s" % (2 * 2048 ** 3 / dt / 1e9))
import subprocess
print(subprocess.run("lscpu | head -20; free -g | head -2", shell=True, capture_output=True, text=True).stdout)
