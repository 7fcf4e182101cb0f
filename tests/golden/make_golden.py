#!/usr/bin/env python3
"""Regenerate tests/golden/ref_vectors.npz from the reference (run in the build container only).

TEST INFRASTRUCTURE.  Needs /root/reference: builds oracle/_ref/gen_golden (the
reference's own Random.h / OrbitCamera.h / *MeshProvider.h compiled where they
lie, see oracle/ref/gen_golden.cpp) via `make -C oracle ref`, runs it, and packs
its raw dumps plus the reference's shipped multiscatter LUT
(/root/reference/Resources/multiscatter_lut.tiff, a data file) into one .npz.
The .npz holds data only — inputs and expected outputs — never reference source.
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"


def main():
    if not os.path.isdir(REF):
        sys.exit("reference not present; golden vectors can only be regenerated in the build container")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    out = {}
    with tempfile.TemporaryDirectory() as d:
        subprocess.check_call([os.path.join(ROOT, "oracle", "_ref", "gen_golden"), d])
        for fn in sorted(os.listdir(d)):
            name = fn[:-4]
            raw = open(os.path.join(d, fn), "rb").read()
            if name.startswith("int_") or name.endswith("_ib"):
                arr = np.frombuffer(raw, dtype=np.uint32).copy()
            else:
                arr = np.frombuffer(raw, dtype=np.float32).copy()
            if name.split("_")[0] in ("sobol", "halton", "hammersley", "radialsobol", "bluenoise", "random",
                                      "polygon5", "polygon6", "polygon8", "seqoffsets"):
                arr = arr.reshape(-1, 2)
            if name == "orbit_matrices":
                arr = arr.reshape(-1, 16)
            if name == "orbit_params":
                arr = arr.reshape(-1, 6)
            out[name] = arr
    # The reference's only true known-answer fixture: the shipped LUT.
    from PIL import Image
    lut = np.array(Image.open(os.path.join(REF, "Resources", "multiscatter_lut.tiff")), dtype=np.float32)
    assert lut.shape == (128, 128)
    # File rows are top-down; the generator's row 0 (lowest roughness) is the bottom
    # scanline (FreeImage bottom-up bitmap), so flip to generator order.
    out["multiscatter_lut"] = lut[::-1].copy()
    path = os.path.join(ROOT, "tests", "golden", "ref_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
