"""tools/r4_soak_bisect.py [workload] — GPU box: digests of the frame at checkpoints for two trace modes (child processes), to find the first pass at which they differ"""
import hashlib, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
marks = [16, 32, 48, 64, 96, 128, 160, 192, 256, 320, 384, 512]
if len(sys.argv) > 2:  # child
    import numpy as np, bench
    from heatray_amd import core
    sc = bench.build_scene(wl, int(os.environ.get("W", "0")), int(os.environ.get("H", "0")), marks[-1])
    e = core.create_engine()
    sc.apply(e)
    for i in range(marks[-1]):
        e.render_pass(sc.options.pass_params(i))
        if i + 1 in marks:
            fr = e.readback()
            print("DIGEST", i + 1, hashlib.sha256(np.ascontiguousarray(fr).tobytes()).hexdigest()[:16], flush=True)
    sys.exit(0)
res = {}
for tune in ("packets=1,corun=0", "packets=0"):
    r = subprocess.run([sys.executable, __file__, wl, "child"], env=dict(os.environ, HR_TUNE=tune), capture_output=True, text=True, timeout=900)
    res[tune] = {int(l.split()[1]): l.split()[2] for l in r.stdout.splitlines() if l.startswith("DIGEST")}
    if not res[tune]: print("FAILED", tune, r.stderr[-400:])
for m in marks:
    a, b = res["packets=1,corun=0"].get(m), res["packets=0"].get(m)
    print(m, a, b, "same" if a == b else "DIFFERENT")
