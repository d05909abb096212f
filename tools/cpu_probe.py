import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
print("nproc", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
os.system("grep -m1 'model name' /proc/cpuinfo; cat /proc/loadavg")
from oracle import cbind
def rand_fr(n, seed):
    rng = np.random.default_rng(seed); a = rng.integers(0,256,size=(n,32),dtype=np.uint8); a[:,31] &= 0x3f; return a.tobytes()
L = 17
d = rand_fr(1<<L, L)
for th in (1, 4, 8, 16, 32, 64, 128, 256):
    best = 1e9
    for _ in range(2):
        t=time.perf_counter(); cbind.ntt_bytes(d, L, False, False, 1<<L, th); best=min(best,time.perf_counter()-t)
    print("ntt 2^17 threads", th, round(best*1e3,1), "ms", flush=True)
