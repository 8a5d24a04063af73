"""Pins the oracle's sampling routines: the reference's own rng.h (golden vectors minted from it + live oracle/_ref build),
the published PCG32 demo vector, and the Random123 Philox4x32-10 known-answer vectors."""
import os

import numpy as np
import pytest

import oracle_ffi
from conftest import GOLDEN


@pytest.fixture(scope="module")
def orc():
    return oracle_ffi.load()


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN, "ref_rng.npz"))


def test_pcg32_published_vector(orc):
    out = np.zeros(6, np.uint32)
    orc.orc_pcg32_seq(54, 42, 1, 6, out.ctypes.data, None)
    assert [hex(x) for x in out] == ["0xa15c02b7", "0x7b47f409", "0xba1d3330", "0x83d2f293", "0xbfa4784b", "0xcbed606e"]


def test_pcg32_matches_reference_golden(orc, gold):
    assert np.array_equal(gold["pcg_54_42"], np.array([0xa15c02b7, 0x7b47f409, 0xba1d3330, 0x83d2f293, 0xbfa4784b, 0xcbed606e], np.uint32))
    for i, s in enumerate(gold["pcg_seq"]):
        ints = np.zeros(16, np.uint32)
        flts = np.zeros(16, np.float32)
        orc.orc_pcg32_seq(int(s), 0, 0, 16, ints.ctypes.data, None)
        orc.orc_pcg32_seq(int(s), 0, 0, 16, None, flts.ctypes.data)
        assert np.array_equal(ints, gold["pcg_ints"][i]), f"sequence {s}"
        assert np.array_equal(flts, gold["pcg_floats"][i]), f"sequence {s}"
    # pixel 0's randomOffset / randomOffsetY (RayTracer.cpp:129-131)
    assert gold["pcg_floats"][0][0] == np.float32(0.89357424) and gold["pcg_floats"][0][1] == np.float32(0.21723014)


def test_halton_matches_reference_golden(orc, gold):
    for b in (2, 3, 5, 7):
        mine = np.array([orc.orc_halton(i, b) for i in range(4096)], np.float32)
        assert np.array_equal(mine, gold[f"halton_{b}"])
        assert np.array_equal(mine[:128], gold[f"haltonseq128_{b}"])
    assert orc.orc_halton(1, 2) == 0.5
    assert np.float32(orc.orc_halton(5, 3)) == np.float32(0.777777791)
    assert np.float32(orc.orc_halton(127, 7)) == np.float32(0.230320707)


def test_against_live_reference_build(orc):
    ref = oracle_ffi.load_ref_rng()
    if ref is None:
        pytest.skip("oracle/_ref/librefrng.so not built (reference checkout absent)")
    for seq in (0, 3, 77, 2**20 + 1):
        a = np.zeros(64, np.uint32)
        b = np.zeros(64, np.uint32)
        orc.orc_pcg32_seq(seq, 0, 0, 64, a.ctypes.data, None)
        ref.ref_pcg32_seq(seq, 0, 0, 64, b.ctypes.data, None)
        assert np.array_equal(a, b)
    for b in (2, 3, 5, 7, 11):
        for i in (0, 1, 2, 17, 999, 65535):
            assert orc.orc_halton(i, b) == ref.ref_halton(i, b)


PHILOX_KAT = [  # Random123 kat_vectors, philox4x32 10 rounds
    ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


@pytest.mark.parametrize("ctr,key,expect", PHILOX_KAT)
def test_philox_kat(orc, ctr, key, expect):
    c = np.array(ctr, np.uint32)
    k = np.array(key, np.uint32)
    o = np.zeros(4, np.uint32)
    orc.orc_philox4x32_10(c.ctypes.data, k.ctypes.data, o.ctypes.data)
    assert tuple(int(x) for x in o) == expect
