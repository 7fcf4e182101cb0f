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


_NATIVE = None


def _host_stamp():
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith(("model name", "flags")):
                model += line
                if line.startswith("flags"):
                    break
    except OSError:
        pass
    import hashlib
    return hashlib.sha1(model.encode()).hexdigest()


def load_native():
    """The -O3 -march=native build of the same sources (oracle/Makefile `native`): the CPU baseline BASELINE.md §2 asks for.
    Timing only — never a checker.  Rebuilt whenever the sources changed or it was built on a different CPU."""
    global _NATIVE
    if _NATIVE is None:
        odir = os.path.join(ROOT, "oracle")
        path, stamp = os.path.join(odir, "liboracle_native.so"), os.path.join(odir, "liboracle_native.stamp")
        srcs = [os.path.join(odir, f) for f in os.listdir(odir) if f.endswith((".cpp", ".h"))] + [os.path.join(ROOT, "include", "hrcore.h")]
        want = _host_stamp()
        have = open(stamp).read().strip() if os.path.exists(stamp) else ""
        if not os.path.exists(path) or have != want or any(os.path.getmtime(s) > os.path.getmtime(path) for s in srcs):
            if os.path.exists(path):
                os.remove(path)
            subprocess.check_call(["make", "-C", odir, "native"], stdout=subprocess.DEVNULL)
            open(stamp, "w").write(want)
        _NATIVE = ctypes.CDLL(path)
    return _NATIVE


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


def engine(threads=None, native=False, **kw):
    from heatray_amd._ffi import Engine
    lib = load_native() if native else load()
    eng = Engine(lib, "ora_", **kw)
    lib.ora_set_threads(eng._ctx, int(threads) if threads else usable_cpus())
    return eng
