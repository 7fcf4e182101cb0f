"""Register budgets of the hot kernels, checked at build time (no GPU needed: hipcc cross-compiles gfx950).

k_trace sits EXACTLY at the VGPR count that lets five waves share a SIMD (96); one more register group and the kernel runs four
waves per SIMD, 4.5 % slower (measured in round 4, when an inlined prologue did just that: profiles/r4n_trace_occupancy.txt).
The compiler's remarks are the cheapest guard."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "heatray_amd", "csrc")


def _resources(src):
    flags = subprocess.run(["make", "-s", "-C", CSRC, "print-flags"], capture_output=True, text=True, check=True).stdout.split()
    out = subprocess.run(["/opt/rocm/bin/hipcc", *flags, "-c", os.path.join(CSRC, src), "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage"],
                         capture_output=True, text=True, cwd=CSRC)
    assert out.returncode == 0, out.stderr[-2000:]
    res, name = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            res[name] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and name:
            res[name][m.group(1).strip()] = int(m.group(2))
    return res


def test_trace_and_shade_kernels_keep_their_occupancy():
    res = _resources("hr_render.hip")
    trace = [v for k, v in res.items() if "k_trace<false>" in k]
    assert len(trace) == 1, list(res)
    assert trace[0]["Occupancy"] >= 5 and trace[0]["VGPRs"] <= 96 and trace[0]["VGPRs Spill"] == 0, trace[0]
    hits = [v for k, v in res.items() if "k_shade_hit<0, 0>" in k]
    assert len(hits) == 1 and hits[0]["Occupancy"] >= 4 and hits[0]["VGPRs"] <= 128, hits
    sort = [v for k, v in res.items() if "k_shade_sort" in k]
    assert len(sort) == 1 and sort[0]["Occupancy"] >= 8, sort
    # the packet kernel hides the latency of its scalar node fetches behind other waves: eight per SIMD, nothing in scratch
    pkt = [v for k, v in res.items() if "k_raygen_packets<false, true>" in k]   # (pass parameters through the scalar cache: the usual batch)
    assert len(pkt) == 1 and pkt[0]["Occupancy"] >= 8 and pkt[0]["ScratchSize"] == 0, pkt
    pkt = [v for k, v in res.items() if "k_raygen_packets<false, false>" in k]  # (every lane its own pass's parameters)
    assert len(pkt) == 1 and pkt[0]["Occupancy"] >= 6 and pkt[0]["ScratchSize"] == 0, pkt
