import os, time, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try:
        print(p, open(p).read().strip())
    except Exception as e:
        print(p, "n/a")
import torch
print("torch threads", torch.get_num_threads(), "interop", torch.get_num_interop_threads())
print("OMP_NUM_THREADS", os.environ.get("OMP_NUM_THREADS"))
import torch.nn.functional as F
x = torch.randn(1, 32, 8, 40, 160); w = torch.randn(32, 32, 3, 3, 3)
for n in (None, 8, 16, 32):
    if n: torch.set_num_threads(n)
    F.conv3d(x, w, padding=1)
    t0 = time.perf_counter()
    for _ in range(3): F.conv3d(x, w, padding=1)
    print("threads", torch.get_num_threads(), "conv3d 32->32 8x40x160:", (time.perf_counter() - t0) / 3 * 1e3, "ms", flush=True)
