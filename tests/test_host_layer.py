"""The C++ drop-in layer (heatray_amd/host): builds standalone, builds against the reference's real glm /
OpenRL / Utility headers where the reference is present, and passes its host-only checks (no GPU)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "heatray_amd", "host")
EXE = os.path.join(ROOT, "tests", "host", "host_layer_test")


@pytest.fixture(scope="module")
def host_test_exe():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "heatray_amd", "csrc")], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", HOST, "test"], stdout=subprocess.DEVNULL)
    return EXE


def test_host_only_checks(host_test_exe):
    out = subprocess.run([host_test_exe, "--cpu-checks"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "cpu checks: ok" in out.stdout


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="needs the reference checkout (build container only)")
def test_layer_compiles_against_reference_third_party_headers(tmp_path):
    # the layer is written against glm / OpenRL's rl.h / the application's Utility headers exactly like the classes it
    # replaces: compile it with the REAL headers (libstdc++ needs the std::sqrtf shim the reference's Random.h assumes)
    srcs = []
    for d in ("HeatrayRenderer", "HeatrayRenderer/Scene", "HeatrayRenderer/Materials", "HeatrayRenderer/Lights"):
        srcs += [os.path.join(HOST, d, f) for f in os.listdir(os.path.join(HOST, d)) if f.endswith(".cpp")]
    cmd = ["g++", "-std=c++20", "-fsyntax-only", "-include", os.path.join(ROOT, "oracle", "ref", "compat_std_math.h"), "-I" + HOST,
           "-I" + os.path.join(ROOT, "include"), "-I/root/reference/3rdParty", "-I/root/reference/Source",
           "-I" + os.path.join(HOST, "standalone", "HeatrayRenderer", "Scene")] + srcs
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-3000:]


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="needs the reference checkout (build container only)")
def test_reference_texture_loader_compiles_unchanged_against_the_layer(tmp_path):
    # overlay: the application's tree with the layer's files in place of the ones they replace; the application's own
    # Utility/TextureLoader.cpp (the producer of openrl::Texture objects) must compile untouched
    ov = tmp_path / "overlay"
    for base, dirs, files in os.walk("/root/reference/Source"):
        rel = os.path.relpath(base, "/root/reference/Source")
        (ov / rel).mkdir(parents=True, exist_ok=True)
        for f in files:
            os.symlink(os.path.join(base, f), ov / rel / f)
    for sub in ("HeatrayRenderer", "RLWrapper"):
        for base, dirs, files in os.walk(os.path.join(HOST, sub)):
            rel = os.path.relpath(base, HOST)
            (ov / rel).mkdir(parents=True, exist_ok=True)
            for f in files:
                dst = ov / rel / f
                if dst.is_symlink() or dst.exists():
                    dst.unlink()
                os.symlink(os.path.join(base, f), dst)
    for gone in ("Buffer.h", "Program.h", "Shader.h", "Primitive.h", "Framebuffer.h", "Error.h"):
        (ov / "RLWrapper" / gone).unlink()
    # (-include: libstdc++ lacks the std::powf the application's code assumes; see oracle/ref/compat_std_math.h)
    cmd = ["g++", "-std=c++20", "-fsyntax-only", "-include", os.path.join(ROOT, "oracle", "ref", "compat_std_math.h"), "-I" + str(ov),
           "-I" + os.path.join(ROOT, "include"), "-I/root/reference/3rdParty", str(ov / "Utility" / "TextureLoader.cpp")]
    out = subprocess.run(cmd, capture_output=True, text=True, cwd=ov)
    assert out.returncode == 0, out.stderr[-3000:]
