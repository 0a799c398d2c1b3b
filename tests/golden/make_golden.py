#!/usr/bin/env python3
"""Convert the reference's own Transform test patterns into compact .npz fixtures.

Run in the build container (needs /root/reference); the GPU box only ever sees the
.npz outputs.  Input files are DATA the reference's tests hold
(Testing/Patterns/DSP/Transform/Transform{F32,Q15,Q31}/*.txt, format described in
Testing/README.md:287-305: line 1 = word size W/H/B/D, line 2 = sample count, then
"// human value" + "0xHEX" line pairs).  Which file feeds which test:
Testing/Source/Tests/TransformCF32.cpp:48-492, TransformRF32.cpp, TransformCQ15.cpp,
TransformCQ31.cpp (forward: input=*InputSamples*, ref=*FFTSamples*; inverse:
input=*InputIFFTSamples*, ref=*InputSamples*).

Usage: python tests/golden/make_golden.py [/root/reference]
"""
import os
import re
import sys

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
PAT = os.path.join(REF, "Testing/Patterns/DSP/Transform")
OUT = os.path.dirname(os.path.abspath(__file__))

_WORD = {"W": (np.uint32, 32), "H": (np.uint16, 16), "B": (np.uint8, 8), "D": (np.uint64, 64)}


def read_pattern(path, kind):
    """kind: 'f32' | 'q15' | 'q31' -> numpy array of the stored samples (bit patterns)."""
    with open(path) as f:
        lines = [ln.strip() for ln in f if ln.strip()]
    utype, _bits = _WORD[lines[0]]
    n = int(lines[1])
    hexes = [ln for ln in lines[2:] if not ln.startswith("//")]
    assert len(hexes) == n, (path, len(hexes), n)
    raw = np.array([int(h, 16) for h in hexes], dtype=np.uint64).astype(utype)
    if kind == "f32":
        return raw.view(np.float32)
    if kind == "q15":
        return raw.view(np.int16)
    if kind == "q31":
        return raw.view(np.int32)
    raise ValueError(kind)


def convert(subdir, kind, out_name):
    d = os.path.join(PAT, subdir)
    arrays = {}
    rx = re.compile(r"^(Complex|Real)(InputSamples|FFTSamples|InputIFFTSamples)_(Noisy|Step)_(\d+)_(\d+)_%s\.txt$" % kind)
    for fn in sorted(os.listdir(d)):
        m = rx.match(fn)
        if not m:
            continue
        dom, what, sig, n, tid = m.groups()
        key = "%s_%s_%s_%s_%s" % (dom.lower(), {"InputSamples": "in", "FFTSamples": "fft", "InputIFFTSamples": "ifftin"}[what],
                                 sig.lower(), n, tid)
        arrays[key] = read_pattern(os.path.join(d, fn), kind)
    path = os.path.join(OUT, out_name)
    np.savez_compressed(path, **arrays)
    print("%s: %d arrays, %.1f KiB" % (out_name, len(arrays), os.path.getsize(path) / 1024.0))


if __name__ == "__main__":
    convert("TransformF32", "f32", "transform_f32.npz")
    convert("TransformQ15", "q15", "transform_q15.npz")
    convert("TransformQ31", "q31", "transform_q31.npz")
