"""tools/valu_mix.py [out.json] — cycles a SIMD needs per VALU wave-instruction of each hot kernel, from the kernel's own ISA (no GPU needed).

`SQ_ACTIVE_INST_VALU x 4` — round 4's "VALU busy" — charges every VALU instruction four cycles.  That holds for most of what a node
step is made of (conversions, min / max, compares, bit fields: 1.72 ns = 4 cycles per wave64 instruction, tools/calib_ops.hip), but
v_fma / v_mul / v_add / v_sub_f32, v_add_u32, v_and_b32, v_mov_b32 and v_cndmask_b32_e32 on VGPR operands issue at the double rate
(0.94-1.14 ns = 2.2-2.6 cycles) and v_rcp / v_sqrt / v_rsq / v_exp / v_log take 8: for the packet kernel, which is mostly FMAs, the
x 4 convention gave a busy "fraction" of 1.3.  This tool prices every VALU instruction of a kernel's disassembly with the
measured cost of its class and prints the mean: bench.py multiplies SQ_INSTS_VALU (wave-instructions issued) by it,

    valu_busy = SQ_INSTS_VALU x cycles_per_instruction / (SIMDs x kernel cycles)   <= 1 by construction of the prices (they are
                                                                                  throughputs measured with the SIMD saturated).

The mix is STATIC (every instruction of the kernel once, whatever its execution count): k_trace is six unrolled copies of the node step
and the triangle test, the packet kernel one node step and one triangle test — the hot loops are most of each kernel's text.  The
JSON is committed (profiles/r5_valu_mix.json) and re-made whenever the kernels change (tests/test_kernel_resources.py compares)."""
import json, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "heatray_amd", "csrc")
FAST = {"v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32",
        "v_mov_b32", "v_cndmask_b32", "v_mac_f32", "v_madak_f32", "v_madmk_f32", "v_fmaak_f32", "v_fmamk_f32", "v_not_b32"}
TRANS = {"v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_hi_i32", "v_mad_u64_u32", "v_mad_i64_i32"}
COST = {"fast": 2.3, "slow": 4.0, "trans": 8.0}  # cycles per wave64 instruction (calib_ops.hip / calib_valu.hip at ~2.3 GHz: 1.0 / 1.72 / 3.4 ns)
KERNELS = {"k_trace": "_ZN2hr7k_traceILb0EE", "k_raygen_packets": "_ZN2hr16k_raygen_packetsILb0ELb1EE", "k_shade_hit": "_ZN2hr11k_shade_hitILi0ELi0EE",
           "k_shade_sort": "_ZN2hr12k_shade_sortE", "k_raygen": "_ZN2hr8k_raygenE"}


def classify(line):
    m = re.match(r"\s+(v_[a-z0-9_]+)\s*(.*)", line)
    if not m:
        return None
    op, args = m.group(1), m.group(2)
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if base.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
        return "slow"
    if base in TRANS or base.startswith(("v_div_scale", "v_div_fmas", "v_div_fixup")):
        return "trans" if base in TRANS else "slow"
    if base in FAST:
        # the double rate needs VGPR / inline-constant operands: an SGPR or a 64-bit-encoded select costs four cycles (calib_ops.hip)
        scalar = re.search(r"(?<![a-z])(s\d+|s\[\d+:\d+\]|vcc|exec|m0)\b", args.split(";")[0])
        if op.endswith("_e64") or (scalar and not (base == "v_cndmask_b32" and op.endswith("_e32"))):
            return "slow"
        return "fast"
    return "slow"


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r5_valu_mix.json")
    flags = subprocess.run(["make", "-s", "-C", CSRC, "print-flags"], capture_output=True, text=True, check=True).stdout.split()
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "hr_render.s")
        subprocess.run(["/opt/rocm/bin/hipcc", *flags, "-S", "--cuda-device-only", os.path.join(CSRC, "hr_render.hip"), "-o", asm], check=True, capture_output=True, cwd=CSRC)
        text = open(asm).read().splitlines()
    res = {"cost_cycles": COST, "source": "static instruction mix of hr_render.hip's gfx950 ISA, priced with tools/calib_ops.hip's measured issue costs",
           "kernels": {}}
    for name, prefix in KERNELS.items():
        start = next((i for i, l in enumerate(text) if l.startswith(prefix) and l.split(";")[0].rstrip().endswith(":")), None)
        if start is None:
            continue
        counts = {"fast": 0, "slow": 0, "trans": 0}
        fast_f32 = 0  # of the fast ones: f32 add / sub / mul / fma — what the SQ's typed counters (SQ_INSTS_VALU_{ADD,MUL,FMA}_F32) see as a class of their own
        for l in text[start + 1:]:
            if "s_endpgm" in l:
                break
            c = classify(l)
            if c:
                counts[c] += 1
                if c == "fast" and re.match(r"\s+v_(fma|fmac|mul|add|sub|subrev|mac|madak|madmk|fmaak|fmamk)_f32", l):
                    fast_f32 += 1
        n = sum(counts.values())
        rest = n - fast_f32 - counts["trans"]
        res["kernels"][name] = {**counts, "fast_f32": fast_f32, "valu_instructions": n, "cycles_per_instruction": sum(COST[k] * v for k, v in counts.items()) / max(n, 1),
                                # of the instructions the typed counters leave in the "rest" class, the share that still issues at the double rate (v_mov, v_and, v_add_u32 ...)
                                "double_rate_share_of_rest": (counts["fast"] - fast_f32) / max(rest, 1)}
    json.dump(res, open(out_path, "w"), indent=1)
    for k, v in res["kernels"].items():
        print(f"{k:18s} {v['valu_instructions']:6d} VALU instructions: fast {v['fast']}, slow {v['slow']}, trans {v['trans']} -> {v['cycles_per_instruction']:.2f} cycles each")


if __name__ == "__main__":
    main()
