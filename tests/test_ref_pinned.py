"""Restatements pinned against the REAL reference where it can be built here: oracle/_ref/libddcmd_ref_small.so is
compiled by oracle/Makefile from /root/reference/src/{crc32,primes,format}.c where they lie (files that need nothing
but libc; ddcMD as a whole cannot be built: its util/ recbis/ cub/ submodules are empty).  The product's and the
oracle's versions of the same functions must agree with it bit for bit."""
import ctypes
import os
import numpy as np
import pytest

import pyoracle
import ddcmd_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "libddcmd_ref_small.so")
pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref not built (make -C oracle needs /root/reference)")


def _ref(*args):
    """what the reference's own code answers, from a child process (tests/ref_probe.py): the reference library is never
    mapped into this process; every call is a fresh process, so primes.c's file-scope search state starts clean"""
    import json
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "ref_probe.py")] + [str(a) for a in args], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.splitlines()[-1])


def test_record_checksum_is_the_references_crc32():
    """crc32.c:46-85 (checksum_crc32_table, what collection_write.c:122 stamps on every atoms record) against
    ddcmi_crc32 of the product (deck.c: reader's verification, plugin.c: writer) on the standard check value and on
    random records of every length up to 300"""
    from ref_probe import records
    want = _ref("crc")
    lib = ddcmd_amd.load_library()
    lib.ddcmi_crc32.restype = ctypes.c_uint32
    lib.ddcmi_crc32.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
    recs = records()
    assert len(want) == len(recs) == 303 and want[0] == [0xCBF43926, 0xCBF43926]
    for buf, (table, bitwise) in zip(recs, want):
        assert table == bitwise
        assert lib.ddcmi_crc32(buf, len(buf)) == table, len(buf)


@pytest.mark.parametrize("task,ntasks", [(0, 1), (3, 8), (7, 8), (1, 2)])
def test_prime_sequence_of_the_lcg64_defaults_is_the_references(tmp_path, task, ntasks):
    """primes.c (prime_init(30000, rank, size) in ddcMD.c:70, nextPrime per three particles in lcg64_default): the
    reference's own sequence for four (task, tasks) pairs against the oracle's restatement of its primality test and
    against the product's Miller-Rabin (deck.c, ddcmi_lcg64_default) -- 700 primes each, i.e. across block boundaries"""
    want = _ref("primes", task, ntasks)
    assert all(w % 2 == 1 for w in want) and len(set(want)) == 700
    labels = np.arange(1, 2101, dtype=np.uint64)
    o = pyoracle.lcg64_default(labels, task, ntasks)
    assert [int(x) for x in o["prime"][0::3]] == [w & 0xffffffff for w in want]
    lib = ddcmd_amd.load_library()
    u64p, u32p = ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint32)
    lib.ddcmi_lcg64_default.restype = None
    lib.ddcmi_lcg64_default.argtypes = [ctypes.c_int, u64p, ctypes.c_uint, ctypes.c_uint, u64p, u32p, u32p]
    st, mu, pr = np.zeros(2100, np.uint64), np.zeros(2100, np.uint32), np.zeros(2100, np.uint32)
    lib.ddcmi_lcg64_default(2100, labels.ctypes.data_as(u64p), task, ntasks, st.ctypes.data_as(u64p), mu.ctypes.data_as(u32p), pr.ctypes.data_as(u32p))
    assert (pr == o["prime"]).all() and (mu == o["multID"]).all() and (st == o["state"]).all()


def test_print_formats_of_loop_and_gid():
    """format.c: loopFormat (snapshot directory names, io.c:128-129) and the decimal gid format of the atoms records
    (collection_write.c:69) as the writer of the driver uses them"""
    loop_fmt, gid_fmt = _ref("formats")
    assert loop_fmt == "%12.12lu" and gid_fmt == "%12.12lu"      # PRIu64 on this platform
    src = open(os.path.join(ROOT, "ddcmd_amd", "csrc", "host", "plugin.c")).read()
    assert '"snapshot.%012" PRId64' in src and '%12.12" PRIu64' in src
    assert "snapshot.%012d" % 40 == "snapshot." + (loop_fmt.replace("lu", "d") % 40)
