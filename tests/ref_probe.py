"""Child process of tests/test_ref_pinned.py: the ONLY process that maps oracle/_ref/libddcmd_ref_small.so (the reference's own
crc32.c / primes.c / format.c / solve.c, compiled where they lie by oracle/Makefile, target `ref`).  The reference's code is untrusted content: it runs
here, in a short-lived child, and hands back plain numbers as one JSON line -- never inside the pytest process.

   python tests/ref_probe.py crc                 checksum_crc32_table / checksum_crc32 of the test's deterministic records
   python tests/ref_probe.py primes TASK NTASKS  prime_init(30000, task, ntasks), 700 x nextPrime()
   python tests/ref_probe.py formats             loopFormatInit(12), gidFormatInit("decimal")
   python tests/ref_probe.py solve < systems     solve(n, a, b, x) (solve.c:3-28: scaled partial pivoting, the linear solve of
                                                 solveConstraintMatrix, nglfconstraint.c:161) for a JSON list of [n, a (row-major), b]"""
import ctypes
import json
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "libddcmd_ref_small.so")


def records():
    """the records both sides checksum: the standard check string, then random bytes of every length up to 300, 1024, 4099"""
    rng = np.random.default_rng(7)
    return [b"123456789"] + [rng.integers(0, 256, n, dtype=np.uint8).tobytes() for n in list(range(1, 301)) + [1024, 4099]]


def main():
    L = ctypes.CDLL(REF)
    what = sys.argv[1]
    if what == "crc":
        L.checksum_crc32_table.restype = ctypes.c_uint
        L.checksum_crc32_table.argtypes = [ctypes.c_char_p, ctypes.c_uint]
        L.checksum_crc32.restype = ctypes.c_uint
        L.checksum_crc32.argtypes = [ctypes.c_char_p, ctypes.c_uint]
        out = [[int(L.checksum_crc32_table(b, len(b))), int(L.checksum_crc32(b, len(b)))] for b in records()]
    elif what == "primes":
        L.nextPrime.restype = ctypes.c_ulonglong
        L.prime_init.argtypes = [ctypes.c_uint, ctypes.c_uint, ctypes.c_uint]
        L.prime_init(30000, int(sys.argv[2]), int(sys.argv[3]))
        out = [int(L.nextPrime()) for _ in range(700)]
    elif what == "solve":
        dp = ctypes.POINTER(ctypes.c_double)
        L.solve.restype = None
        L.solve.argtypes = [ctypes.c_int, dp, dp, dp]
        out = []
        for n, a, b in json.load(sys.stdin):
            a = np.array(a, dtype=np.float64).reshape(n, n).copy()      # (solve overwrites a and b)
            b = np.array(b, dtype=np.float64).copy()
            x = np.zeros(n)
            L.solve(int(n), a.ctypes.data_as(dp), b.ctypes.data_as(dp), x.ctypes.data_as(dp))
            out.append(x.tolist())
    else:
        L.loopFormat.restype = ctypes.c_char_p
        L.gidFormat.restype = ctypes.c_char_p
        L.loopFormatInit(12)
        L.gidFormatInit(b"decimal")
        out = [L.loopFormat().decode(), L.gidFormat().decode()]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
