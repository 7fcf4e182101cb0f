"""tools/tables_time.py — generateRandomSequences(P, mode, shape) (16 sequences + 16 aperture tables, PassGenerator.cpp:603-684) through
hr_sequences_generate on the device, beside the checker's serial loops (the reference's algorithms on one host thread) for the serial kinds."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from heatray_amd import _ffi as ffi, core
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import oracle_lib

g = core.create_engine()
g.resize(64, 64)
o = oracle_lib.engine()
o.resize(64, 64)
names = {ffi.HR_SAMPLE_RANDOM: "random", ffi.HR_SAMPLE_BLUE_NOISE: "blue noise", ffi.HR_SAMPLE_SOBOL: "sobol"}
shapes = {ffi.HR_BOKEH_CIRCULAR: "circular", ffi.HR_BOKEH_PENTAGON: "pentagon"}
g.generate_sequences(ffi.HR_SAMPLE_SOBOL, ffi.HR_BOKEH_CIRCULAR, 32)
for P in (1024, 4096, 8192):
    for mode, shape in ((ffi.HR_SAMPLE_SOBOL, ffi.HR_BOKEH_CIRCULAR), (ffi.HR_SAMPLE_SOBOL, ffi.HR_BOKEH_PENTAGON), (ffi.HR_SAMPLE_RANDOM, ffi.HR_BOKEH_CIRCULAR),
                        (ffi.HR_SAMPLE_BLUE_NOISE, ffi.HR_BOKEH_CIRCULAR)):
        t0 = time.perf_counter()
        g.generate_sequences(mode, shape, P)
        g.synchronize()
        tg = time.perf_counter() - t0
        to = float("nan")
        if mode != ffi.HR_SAMPLE_BLUE_NOISE or P <= 4096:
            t0 = time.perf_counter()
            o.generate_sequences(mode, shape, P)
            to = time.perf_counter() - t0
        print(f"P={P:5d} {names[mode]:10s} {shapes[shape]:9s}: device {tg * 1e3:9.2f} ms   one host thread {to * 1e3:10.2f} ms", flush=True)
