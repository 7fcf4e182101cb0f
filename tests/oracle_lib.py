"""Loader for the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may use it."""
import ctypes
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None


def load():
    global _LIB
    if _LIB is None:
        path = os.path.join(ROOT, "oracle", "liboracle.so")
        srcs = [os.path.join(ROOT, "oracle", f) for f in os.listdir(os.path.join(ROOT, "oracle")) if f.endswith((".cpp", ".h"))]
        srcs.append(os.path.join(ROOT, "include", "hrcore.h"))
        if not os.path.exists(path) or any(os.path.getmtime(s) > os.path.getmtime(path) for s in srcs):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"], stdout=subprocess.DEVNULL)
        _LIB = ctypes.CDLL(path)
    return _LIB


def usable_cpus():
    """CPUs this process may actually use: the affinity mask, capped by the cgroup CPU quota (a GPU box shows 256 CPUs but
    grants 16: 256 OpenMP threads then share the quota and spend their time being throttled)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def engine(threads=None, **kw):
    from heatray_amd._ffi import Engine
    eng = Engine(load(), "ora_", **kw)
    load().ora_set_threads(eng._ctx, int(threads) if threads else usable_cpus())
    return eng
