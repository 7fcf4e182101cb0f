#!/usr/bin/env python3
"""Experiment (GPU box): which part of c3's lighting keeps passes-to-converge (BASELINE metric 2) high?
Variants of the c3 scene, each through bench.convergence_leg with a few runs.
    python tools/converge_variants.py [estimator: reference|env_mis|all_lights] [runs]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from heatray_amd import _ffi as ffi  # noqa: E402
from heatray_amd import core  # noqa: E402

est = {"reference": ffi.HR_ESTIMATOR_REFERENCE, "env_mis": ffi.HR_ESTIMATOR_ENV_MIS, "all_lights": ffi.HR_ESTIMATOR_ALL_LIGHTS}[sys.argv[1] if len(sys.argv) > 1 else "all_lights"]
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 2
only = sys.argv[3].split(",") if len(sys.argv) > 3 else None
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream(dev).cuda_stream
out = {}


def variant(name, edit):
    if only and name not in only:
        return
    sc = bench.build_scene("c3", 0, 0, 32)
    sc.options.estimator = est
    edit(sc)
    r = bench.convergence_leg(core, sc, dev, stream, 4096, runs)
    out[name] = {"p50": r["p50"], "runs": r["runs"], "err": r["median_err_at_pass"]}
    print(name, r["p50"], r["runs"], {k: round(v, 4) for k, v in r["median_err_at_pass"].items() if int(k) in (1, 16, 256, 1024)}, flush=True)


def no_env(sc):
    sc.env_pixels = None


def no_sun(sc):
    sc.lights.directional.clear()


def direct_only(sc):
    sc.options.max_ray_depth = 1


def flat_env(sc):
    import numpy as np
    sc.env_pixels = np.full((1, 1, 3), float(sc.env_pixels[..., 1].mean()), dtype=np.float32)


variant("full", lambda sc: None)
variant("no_env", no_env)
variant("flat_env", flat_env)
variant("direct_only", direct_only)
variant("no_sun", no_sun)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "converge_variants.json"), "w"), indent=1)
