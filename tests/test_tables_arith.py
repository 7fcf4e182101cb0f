"""heatray_amd/csrc/hr_tables.h on the CPU: MT19937 twisted in the kernel's three rounds of 208 lanes, libstdc++'s uniform_real / uniform_int
distributions and BlueNoise.h's sign-extending FNV-1a, each against the host's own <random> / a literal restatement (no GPU).  The device
kernels built on the same header are checked against the reference's golden tables in tests/test_gpu_parity.py."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_table_arithmetic_matches_the_standard_library(tmp_path):
    exe = tmp_path / "tables_arith_test"
    # -ffp-contract=off like the library: the header's float lines must mean the same on both sides
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-Wall", "-I" + os.path.join(ROOT, "heatray_amd", "csrc"),
                           os.path.join(ROOT, "tests", "host", "tables_arith_test.cpp"), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "tables arith: ok" in out.stdout
