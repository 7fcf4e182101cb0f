"""Pin the CPU oracle's sample-table generators to the reference's own code.

tests/golden/ref_vectors.npz holds outputs of the reference's Utility/Random.h compiled in the
build container (oracle/ref/gen_golden.cpp, tests/golden/make_golden.py) and the reference's
shipped Resources/multiscatter_lut.tiff.  Bit-exact unless stated.
"""
import ctypes as C

import numpy as np
import pytest

import oracle_lib
from heatray_amd import _ffi as ffi


@pytest.fixture(scope="module")
def ora():
    return oracle_lib.engine()


def test_integer_building_blocks(golden, oracle_lib):
    lib = oracle_lib
    for fn in (lib.ora_burley_hash, lib.ora_reverse_bits, lib.ora_laine_karras, lib.ora_nested_scramble):
        fn.restype = C.c_uint32
    x = golden["int_in"]
    seeds = [lib.ora_burley_hash(C.c_uint32(i + 1)) for i in range(len(x))]
    assert [lib.ora_burley_hash(C.c_uint32(int(v))) for v in x] == list(golden["int_burleyhash"])
    assert [lib.ora_reverse_bits(C.c_uint32(int(v))) for v in x] == list(golden["int_reversebits"])
    assert [lib.ora_laine_karras(C.c_uint32(int(v)), C.c_uint32(s)) for v, s in zip(x, seeds)] == list(golden["int_lainekarras"])
    assert [lib.ora_nested_scramble(C.c_uint32(int(v)), C.c_uint32(s)) for v, s in zip(x, seeds)] == list(golden["int_nestedscramble"])


@pytest.mark.parametrize("name,mode", [("sobol", ffi.HR_SAMPLE_SOBOL), ("halton", ffi.HR_SAMPLE_HALTON),
                                       ("hammersley", ffi.HR_SAMPLE_HAMMERSLEY)])
@pytest.mark.parametrize("P", [32, 1024])
def test_owen_scrambled_sequences_bit_exact(golden, ora, name, mode, P):
    for seq in range(16):
        got = ora.qmc_generate(mode, seq, P)
        want = golden[f"{name}_p{P}_s{seq}"]
        assert got.tobytes() == want.tobytes(), f"{name} P={P} seq={seq}"


def test_survey_probe_values(ora):
    # values probed from the compiled reference header
    s = ora.qmc_generate(ffi.HR_SAMPLE_SOBOL, 0, 32)
    assert np.allclose(s[:3], [(0.312489927, 0.743847013), (0.949272215, 0.30650726), (0.578496635, 0.94785428)], atol=1e-8)
    assert np.allclose(ora.qmc_generate(ffi.HR_SAMPLE_HALTON, 3, 32)[1], (0.486762732, 0.681400061), atol=1e-8)


@pytest.mark.parametrize("P", [32, 1024])
def test_radial_sobol(golden, ora, P):
    # libm sinf/cosf on the same toolchain -> bit-exact here; documented as libm-specific
    for seq in range(16):
        got = ora.qmc_generate(ffi.HR_SAMPLE_SOBOL, seq, P, radial=True)
        assert got.tobytes() == golden[f"radialsobol_p{P}_s{seq}"].tobytes()


def test_blue_noise_bit_exact(golden, ora):
    for seq in range(16):
        got = ora.qmc_generate(ffi.HR_SAMPLE_BLUE_NOISE, seq, 32)
        assert got.tobytes() == golden[f"bluenoise_p32_s{seq}"].tobytes()


def test_std_distribution_tables_same_toolchain(golden, ora, oracle_lib):
    # std::uniform_*_distribution are standard-library specific (SURVEY §4 caveat): these pin libstdc++ only
    for seq in range(16):
        got = ora.qmc_generate(ffi.HR_SAMPLE_RANDOM, seq, 32)
        assert got.tobytes() == golden[f"random_p32_s{seq}"].tobytes()
        for edges in (5, 6, 8):
            out = np.empty((32, 2), dtype=np.float32)
            oracle_lib.ora_qmc_polygon(None, C.c_uint32(edges), C.c_uint32(seq), C.c_uint32(32), out.ctypes.data_as(ffi.f32p))
            assert np.array_equal(out, golden[f"polygon{edges}_p32_s{seq}"])


def test_aperture_tables_entry_point(golden, ora):
    # ora_aperture_generate = the checker's side of hr_aperture_generate (PassGenerator.cpp:653-676)
    for seq in range(16):
        assert ora.aperture_generate(ffi.HR_BOKEH_CIRCULAR, seq, 1024).tobytes() == golden[f"radialsobol_p1024_s{seq}"].tobytes()
        for shape, edges in ((ffi.HR_BOKEH_PENTAGON, 5), (ffi.HR_BOKEH_HEXAGON, 6), (ffi.HR_BOKEH_OCTAGON, 8)):
            assert ora.aperture_generate(shape, seq, 32).tobytes() == golden[f"polygon{edges}_p32_s{seq}"].tobytes()
    with pytest.raises(ffi.EngineError):
        ora.aperture_generate(11, 0, 32)


def test_sequence_offsets_table(golden, ora):
    # PassGenerator::generateSequenceOffsets: sobol(W*H, sequence 0)
    ora.resize(64, 64)
    got = ora.qmc_generate(ffi.HR_SAMPLE_SOBOL, 0, 64 * 64)
    assert got.tobytes() == golden["seqoffsets_64x64"].tobytes()


def test_multiscatter_lut_matches_shipped_tiff(golden, ora):
    # The reference's only true known-answer fixture: Resources/multiscatter_lut.tiff.
    lut, _ = ora.generate_multiscatter_lut()
    want = golden["multiscatter_lut"]
    diff = np.abs(lut - want)
    rel_l2 = np.linalg.norm(lut - want) / np.linalg.norm(want)
    assert diff.max() < 2e-6, diff.max()
    assert rel_l2 < 1e-6, rel_l2
    assert (lut == want).mean() > 0.9
