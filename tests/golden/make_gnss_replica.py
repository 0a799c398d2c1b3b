#!/usr/bin/env python3
"""Golden vectors of the reference's make_replica / rotate_fwd (quasi_pilot_tests.c:153-181, :138-151).

Run in the build container: needs oracle/_ref/libquasi_ref.so, i.e. the reference's UNMODIFIED
Source/TransformFunctions/quasi_pilot_tests.c compiled where it lies by `make -C oracle ref`.  The outputs
written to tests/golden/gnss_replica.npz are what the reference's own object code returns here; the GPU box
only ever sees the .npz.  Inputs: C/A chips from the repo's getCode restatement (oracle/gnss_oracle.c, itself
pinned by the IS-GPS-200 first-ten-chips table), sample-and-hold at `spc` samples per chip, repeated / cut to
`size` samples -- stored in the fixture, so the file is self-contained.

Usage: python tests/golden/make_gnss_replica.py
"""
import ctypes
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import helpers as H  # noqa: E402

# (prn, samples per chip, size, doppler Hz, sample rate Hz): the GPS L1 geometry of CMSIS.cpp:28 (4.092 MHz), the fork's
# 16384-point Galileo-sized correlation, a short odd length, negative / off-grid / zero Doppler
CASES = [
    (1, 4, 4092, 0.0, 4.092e6),
    (3, 4, 4092, 1000.0, 4.092e6),
    (7, 4, 4092, -4250.0, 4.092e6),
    (11, 4, 4096, 4750.0, 4.092e6),
    (19, 4, 4092, 840.5, 4.092e6),
    (22, 2, 2046, -125.25, 2.046e6),
    (30, 1, 1023, 5000.0, 1.023e6),
    (32, 4, 16384, -2250.0, 16.384e6),
    (37, 4, 8184, 90.0, 4.092e6),
    (5, 4, 333, 250.0, 4.092e6),
]
ROTATIONS = [(4092, 0), (4092, 1), (4092, 1033), (4092, 4091), (4092, 4092), (17, 5)]


def main():
    Q = H.ref_quasi()
    assert Q is not None, "oracle/_ref/libquasi_ref.so missing: run `make -C oracle ref` where /root/reference is mounted"
    out = {}
    meta = np.zeros(len(CASES), dtype=[("prn", np.int32), ("spc", np.int32), ("size", np.int32), ("doppler", np.float32), ("samp_freq", np.float32)])
    for k, (prn, spc, size, dop, fs) in enumerate(CASES):
        code = np.resize(H.ca_code(prn, spc), size).astype(np.int32)
        rep = np.zeros((size, 2), np.float32)
        Q.make_replica(code.ctypes.data_as(ctypes.c_void_p), rep.ctypes.data_as(ctypes.c_void_p), dop, size, fs)
        meta[k] = (prn, spc, size, dop, fs)
        out["code_%d" % k] = code.astype(np.int8)
        out["replica_%d" % k] = rep
    out["cases"] = meta
    rot = np.zeros(len(ROTATIONS), dtype=[("size", np.int32), ("offset", np.int32)])
    for k, (size, off) in enumerate(ROTATIONS):
        arr = (np.arange(size, dtype=np.int32) * 7 + 3) % 1009 - 500
        got = arr.copy()
        Q.rotate_fwd(got.ctypes.data_as(ctypes.c_void_p), size, off)
        rot[k] = (size, off)
        out["rot_in_%d" % k] = arr
        out["rot_out_%d" % k] = got
    out["rotations"] = rot
    path = os.path.join(HERE, "gnss_replica.npz")
    np.savez_compressed(path, **out)
    print("wrote %s: %d make_replica cases, %d rotate_fwd cases, %d bytes" % (path, len(CASES), len(ROTATIONS), os.path.getsize(path)))


if __name__ == "__main__":
    main()
