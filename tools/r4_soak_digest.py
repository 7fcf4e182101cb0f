"""tools/r4_soak_digest.py [workload] [passes] — GPU box: the frame after `passes` passes with the camera rays as packets beside k_trace (the selector's
choice, forced here from the first pass) and with one ray per lane everywhere: the two digests must be equal (hits never depend on how rays are grouped)."""
import hashlib, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 512
if len(sys.argv) > 3:  # child
    import numpy as np, bench
    from heatray_amd import core
    sc = bench.build_scene(wl, int(os.environ.get("W", "0")), int(os.environ.get("H", "0")), passes)
    e = core.create_engine()
    sc.apply(e)
    for i in range(passes):
        e.render_pass(sc.options.pass_params(i))
    fr = e.readback()
    print("DIGEST", hashlib.sha256(np.ascontiguousarray(fr).tobytes()).hexdigest(), float(fr[..., 3].max()))
    sys.exit(0)
out = {}
for tune in ("packets=1,corun=2", "packets=1,corun=0", "packets=0"):
    r = subprocess.run([sys.executable, __file__, wl, str(passes), "child"], env=dict(os.environ, HR_TUNE=tune), capture_output=True, text=True, timeout=900)
    line = [l for l in r.stdout.splitlines() if l.startswith("DIGEST")]
    out[tune] = line[0] if line else "FAILED: " + r.stderr[-300:]
    print(f"{wl} {passes} passes, HR_TUNE={tune}: {out[tune]}")
print("EQUAL" if len(set(out.values())) == 1 and not any(v.startswith("FAILED") for v in out.values()) else "DIFFERENT")
