"""How many host cores this process may really use (TEST / MEASUREMENT INFRASTRUCTURE, like the rest of oracle/).

torch sizes its thread pool from the machine's core count; on a GPU box whose container has a CPU QUOTA (cgroup cpu.max: 16 cores of a
256-thread host) that oversubscribes the quota eightfold and the CPU oracle runs 2.2 x SLOWER than with one thread per usable core
(measured: actor_forward on 8 items x 20 tags 4.51 s at 128 threads, 2.01 s at 16; tools/dbg/oracle_threads.py)."""
import math
import os


def usable_cores() -> int:
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, math.ceil(int(txt[0]) / int(txt[1]))))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, math.ceil(quota / period)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def fit_torch_threads() -> int:
    """torch.set_num_threads(usable_cores()) when torch's pool is larger; -> the thread count in effect."""
    import torch
    n = usable_cores()
    if torch.get_num_threads() > n:
        torch.set_num_threads(n)
    return torch.get_num_threads()
