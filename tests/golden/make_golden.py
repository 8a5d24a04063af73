"""Generates tests/golden/ref_rng.npz from the REFERENCE's own rng.h, compiled unmodified into oracle/_ref/librefrng.so
(oracle/Makefile target `ref`; only possible where /root/reference exists).  The vectors — not the reference — travel:
PCG32 streams (ints and floats) for several (sequence, seed) pairs, Halton values, HaltonSeq<128> tables."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_ffi  # noqa: E402


def main():
    ref = oracle_ffi.load_ref_rng()
    if ref is None:
        raise SystemExit("oracle/_ref/librefrng.so missing: run `make -C oracle ref` where /root/reference exists")
    out = {}
    seqs = np.array([0, 1, 2, 54, 12345, 921599, 2**31 + 7], dtype=np.uint64)
    ints = np.zeros((len(seqs), 16), np.uint32)
    flts = np.zeros((len(seqs), 16), np.float32)
    for i, s in enumerate(seqs):
        ref.ref_pcg32_seq(int(s), 0, 0, 16, ints[i].ctypes.data, None)
        ref.ref_pcg32_seq(int(s), 0, 0, 16, None, flts[i].ctypes.data)
    out["pcg_seq"], out["pcg_ints"], out["pcg_floats"] = seqs, ints, flts
    seeded = np.zeros(6, np.uint32)
    ref.ref_pcg32_seq(54, 42, 1, 6, seeded.ctypes.data, None)
    out["pcg_54_42"] = seeded
    idx = np.arange(0, 4096, dtype=np.int32)
    for b in (2, 3, 5, 7):
        out[f"halton_{b}"] = np.array([ref.ref_halton(int(i), b) for i in idx], np.float32)
        t = np.zeros(128, np.float32)
        ref.ref_halton_seq128(b, t.ctypes.data)
        out[f"haltonseq128_{b}"] = t
    np.savez_compressed(os.path.join(HERE, "ref_rng.npz"), **out)
    print("wrote ref_rng.npz:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
