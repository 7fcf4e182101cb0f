"""Metric 2 of BASELINE.json / SURVEY.md §8d: **passes-to-converge p50**.

For pass n, err(n) = || I_n - I_ref ||_2 / || I_ref ||_2 over the RGB/A image, I_ref = the 8192-pass render (the largest pass
count the reference's UI offers, HeatrayRenderer.cpp:982-983) of the same build; passes-to-converge = min n with
err(n) <= 0.02; p50 over 16 runs that differ only in the Sobol sequence index (0..15) of the SequenceOffsets table
(PassGenerator.cpp:150-159 uses index 0).  It is a property of the estimator, not of speed: the CPU oracle and the HIP core
must report the same numbers (tests/test_gpu_parity.py::test_passes_to_converge_agrees_with_the_oracle).

Works on any engine object of heatray_amd._ffi (the HIP core or, in the tests, the oracle)."""
import numpy as np

from . import _ffi as ffi

THRESHOLD = 0.02
N_RUNS = 16
REFERENCE_PASSES = 8192


def normalised(frame):
    """RGB / A of an accumulation buffer [H, W, 4] (numpy or torch); pixels without samples give 0."""
    a = frame[..., 3:4]
    if isinstance(frame, np.ndarray):
        return np.where(a > 0, frame[..., :3] / np.where(a > 0, a, 1), 0).astype(np.float32)
    import torch
    return torch.where(a > 0, frame[..., :3] / torch.where(a > 0, a, torch.ones_like(a)), torch.zeros_like(frame[..., :3]))


def rel_l2(img, ref):
    if isinstance(img, np.ndarray):
        return float(np.linalg.norm((img.astype(np.float64) - ref.astype(np.float64)).ravel()) / np.linalg.norm(ref.astype(np.float64).ravel()))
    return float(((img.double() - ref.double()).norm() / ref.double().norm()).item())


def first_pass_below(errors, threshold=THRESHOLD):
    """errors[n-1] = err(n); returns the 1-based pass count, or None if the run never got there."""
    for n, e in enumerate(errors, 1):
        if e <= threshold:
            return n
    return None


def p50(values, cap):
    """Median of the runs; a run that did not converge counts as `cap` (and is reported by the caller)."""
    v = sorted(cap if x is None else x for x in values)
    return 0.5 * (v[(len(v) - 1) // 2] + v[len(v) // 2])


def offsets_table(eng, sequence_index, width, height):
    """generateSequenceOffsets (PassGenerator.cpp:150-159) for Sobol sequence `sequence_index`."""
    return eng.qmc_generate(ffi.HR_SAMPLE_SOBOL, sequence_index, width * height)


def run_with_readback(eng, options, sequence_index, max_passes, ref_img, width, height, threshold=THRESHOLD):
    """Small-frame version (tests): one readback per pass.  Returns (passes-to-converge or None, [err(1), err(2), ...])."""
    eng.set_seq_offsets(offsets_table(eng, sequence_index, width, height))
    eng.clear()
    errs = []
    for n in range(max_passes):
        eng.render_pass(options.pass_params(n))
        errs.append(rel_l2(normalised(eng.readback()), ref_img))
        if errs[-1] <= threshold:
            break
    return first_pass_below(errs, threshold), errs


def reference_image(eng, options, passes, width, height):
    eng.set_seq_offsets(offsets_table(eng, 0, width, height))
    eng.clear()
    for n in range(passes):
        eng.render_pass(options.pass_params(n))
    return normalised(eng.readback())
