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
    ov = _overlay_tree(tmp_path)
    # (-include: libstdc++ lacks the std::powf the application's code assumes; see oracle/ref/compat_std_math.h)
    cmd = ["g++", "-std=c++20", "-fsyntax-only", "-include", os.path.join(ROOT, "oracle", "ref", "compat_std_math.h"), "-I" + str(ov),
           "-I" + os.path.join(ROOT, "include"), "-I/root/reference/3rdParty", str(ov / "Utility" / "TextureLoader.cpp")]
    out = subprocess.run(cmd, capture_output=True, text=True, cwd=ov)
    assert out.returncode == 0, out.stderr[-3000:]
    # ... and in that tree the layer's MultiScatterUtil reads the reference's own input, Resources/multiscatter_lut.tiff, through that
    # kept loader (MultiScatterUtil.cpp:141-150 of the reference); device integration is only the fallback
    obj = tmp_path / "msu.o"
    cmd = ["g++", "-std=c++20", "-c", "-o", str(obj), "-include", os.path.join(ROOT, "oracle", "ref", "compat_std_math.h"), "-I" + str(ov),
           "-I" + os.path.join(ROOT, "include"), "-I/root/reference/3rdParty", "-I" + os.path.join(HOST, "standalone", "HeatrayRenderer", "Scene"),
           str(ov / "HeatrayRenderer" / "Materials" / "MultiScatterUtil.cpp")]
    out = subprocess.run(cmd, capture_output=True, text=True, cwd=ov)
    assert out.returncode == 0, out.stderr[-3000:]
    syms = subprocess.run(["nm", "-C", str(obj)], capture_output=True, text=True).stdout
    assert "U util::loadTexture" in syms and "hr_multiscatter_lut_generate" in syms, syms[-2000:]


def _overlay_tree(tmp_path):
    """The application's Source/ tree with the layer's files in place of the ones they replace (symlinks)."""
    ov = tmp_path / "tree" / "Source"   # (some application headers reach 3rdParty by "../../3rdParty/...": keep the tree's shape)
    ov.mkdir(parents=True)
    os.symlink("/root/reference/3rdParty", tmp_path / "tree" / "3rdParty")
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
    return ov


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="needs the reference checkout (build container only)")
def test_reference_viewer_compiles_unchanged_against_the_layer(tmp_path):
    # SURVEY 8(b)'s headline: the viewer (HeatrayRenderer.cpp, 2,010 lines: imgui UI, GL display, session, screenshots, every call the
    # application makes into PassGenerator / Scene / Lighting / materials / lights / PixelPackBuffer) compiles UNCHANGED against the
    # overlay — no -include, no edits.  What this image lacks and the viewer only names in #include lines is stood in for by three
    # declaration-only test aids (tests/host/viewer_stubs): the macOS OpenGL/gl3*.h pair (glew.h from 3rdParty declares the API) and
    # an assimp-free AssimpMeshProvider.h (the viewer never names the class; the loader is outside the boundary).
    ov = _overlay_tree(tmp_path)
    stubs = os.path.join(ROOT, "tests", "host", "viewer_stubs")
    amp = ov / "HeatrayRenderer" / "Scene" / "AssimpMeshProvider.h"
    amp.unlink()
    os.symlink(os.path.join(stubs, "AssimpMeshProvider.h"), amp)
    cmd = ["g++", "-std=c++20", "-fsyntax-only", "-DGLEW_NO_GLU", "-I" + str(ov), "-I" + str(ov / "HeatrayRenderer"), "-I" + stubs,
           "-I" + os.path.join(ROOT, "include"), "-I/root/reference/3rdParty", str(ov / "HeatrayRenderer" / "HeatrayRenderer.cpp")]
    out = subprocess.run(cmd, capture_output=True, text=True, cwd=ov)
    assert out.returncode == 0, out.stderr[-4000:]
    # the viewer's header on its own, too (HeatrayRenderer.h:159,177 use LOG_ERROR: Utility/Log.h reaches it through the layer's
    # Lights/Light.h, as it did through the reference's Light.h -> RLWrapper/Program.h)
    probe = tmp_path / "probe.cpp"
    probe.write_text('#include "HeatrayRenderer/HeatrayRenderer.h"\nint main() { return 0; }\n')
    out = subprocess.run(cmd[:-1] + [str(probe)], capture_output=True, text=True, cwd=ov)
    assert out.returncode == 0, out.stderr[-4000:]


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="needs the reference checkout (build container only)")
def test_reference_viewer_links_against_the_layer(tmp_path):
    # One step past "compiles": the viewer's OBJECT file, built from the reference's own HeatrayRenderer.cpp in the overlay tree, names
    # (as undefined symbols) every function of PassGenerator / Scene / Lighting / materials / lights it calls; each of them must be
    # DEFINED by the layer built against the same glm the viewer uses (make GLM=/root/reference/3rdParty) — same manglings, same
    # argument types.  (VERDICT r4 did this by hand: 22 symbols.)  What remains undefined outside those classes is the application's
    # own (imgui, GL, glfw, the loaders): outside the boundary.
    ov = _overlay_tree(tmp_path)
    stubs = os.path.join(ROOT, "tests", "host", "viewer_stubs")
    amp = ov / "HeatrayRenderer" / "Scene" / "AssimpMeshProvider.h"
    amp.unlink()
    os.symlink(os.path.join(stubs, "AssimpMeshProvider.h"), amp)
    obj = tmp_path / "HeatrayRenderer.o"
    cmd = ["g++", "-std=c++20", "-O0", "-c", "-DGLEW_NO_GLU", "-I" + str(ov), "-I" + str(ov / "HeatrayRenderer"), "-I" + stubs,
           "-I" + os.path.join(ROOT, "include"), "-I/root/reference/3rdParty", str(ov / "HeatrayRenderer" / "HeatrayRenderer.cpp"), "-o", str(obj)]
    out = subprocess.run(cmd, capture_output=True, text=True, cwd=ov)
    assert out.returncode == 0, out.stderr[-4000:]
    lib = tmp_path / "libheatrayhost_glm.so"
    out = subprocess.run(["make", "-C", HOST, "GLM=/root/reference/3rdParty", "OUT=" + str(lib)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-4000:]
    und = subprocess.run(["nm", "-u", "-C", str(obj)], capture_output=True, text=True, check=True).stdout
    defined = subprocess.run(["nm", "-D", "--defined-only", "-C", str(lib)], capture_output=True, text=True, check=True).stdout
    have = {l.split(None, 2)[2].strip() for l in defined.splitlines() if len(l.split(None, 2)) == 3}
    import re
    ours = re.compile(r"^(PassGenerator|Scene|Mesh|Lighting|Material|PhysicallyBasedMaterial|GlassMaterial|Light|DirectionalLight|PointLight|SpotLight|EnvironmentLight|"
                      r"openrl::\w+)::")
    wanted = [l.split(None, 1)[1].strip() for l in und.splitlines() if len(l.split(None, 1)) == 2]
    wanted = [w for w in wanted if ours.match(w) or any(w.startswith(p) for p in ("vtable for ", "typeinfo for ")) and ours.match(w.split(" for ", 1)[1] + "::")]
    assert len(wanted) >= 15, wanted                      # (the viewer really does call into the layer: 22 symbols when this was written)
    missing = [w for w in wanted if w not in have]
    assert not missing, missing


@pytest.mark.parametrize("san,env", [("tsan", {"TSAN_OPTIONS": "halt_on_error=1"}),
                                     ("asan", {"ASAN_OPTIONS": "detect_leaks=1", "UBSAN_OPTIONS": "halt_on_error=1"})])
def test_layer_threading_under_sanitizers(san, env):
    # the layer's job queue, callbacks and the pixel hand-off, on the CPU under ThreadSanitizer and
    # AddressSanitizer + UBSan, against the do-nothing C-ABI stub (tests/host/hrcore_stub.cpp; GPU sanitizers are not available)
    subprocess.check_call(["make", "-C", HOST, san], stdout=subprocess.DEVNULL)
    exe = os.path.join(ROOT, "tests", "host", f"host_threading_{san}")
    out = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, **env), timeout=300)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    assert "threading checks: ok" in out.stdout
    assert "WARNING: ThreadSanitizer" not in out.stderr and "ERROR: AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr
