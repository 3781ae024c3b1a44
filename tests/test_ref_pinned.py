"""Restatements pinned against the REAL reference where it can be built here: oracle/_ref/libddcmd_ref_small.so is
compiled by oracle/Makefile from /root/reference/src/{crc32,primes,format}.c where they lie (files that need nothing
but libc; ddcMD as a whole cannot be built: its util/ recbis/ cub/ submodules are empty).  The product's and the
oracle's versions of the same functions must agree with it bit for bit."""
import ctypes
import os
import shutil
import numpy as np
import pytest

import pyoracle
import ddcmd_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "libddcmd_ref_small.so")
pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref not built (make -C oracle needs /root/reference)")


def _ref(tmp_path=None, tag=""):
    """the reference library; a private copy when the caller needs fresh static state (primes.c keeps its search
    interval in file-scope statics)"""
    path = REF
    if tmp_path is not None:
        path = str(tmp_path / ("ref_%s.so" % tag))
        shutil.copy(REF, path)
    L = ctypes.CDLL(path)
    L.checksum_crc32_table.restype = ctypes.c_uint
    L.checksum_crc32_table.argtypes = [ctypes.c_char_p, ctypes.c_uint]
    L.checksum_crc32.restype = ctypes.c_uint
    L.checksum_crc32.argtypes = [ctypes.c_char_p, ctypes.c_uint]
    L.nextPrime.restype = ctypes.c_ulonglong
    L.prime_init.argtypes = [ctypes.c_uint, ctypes.c_uint, ctypes.c_uint]
    L.loopFormat.restype = ctypes.c_char_p
    L.gidFormat.restype = ctypes.c_char_p
    return L


def test_record_checksum_is_the_references_crc32():
    """crc32.c:46-85 (checksum_crc32_table, what collection_write.c:122 stamps on every atoms record) against
    ddcmi_crc32 of the product (deck.c: reader's verification, plugin.c: writer) on the standard check value and on
    random records of every length up to 300"""
    R = _ref()
    lib = ddcmd_amd.load_library()
    lib.ddcmi_crc32.restype = ctypes.c_uint32
    lib.ddcmi_crc32.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
    assert R.checksum_crc32_table(b"123456789", 9) == 0xCBF43926 == R.checksum_crc32(b"123456789", 9) == lib.ddcmi_crc32(b"123456789", 9)
    rng = np.random.default_rng(7)
    for n in list(range(1, 301)) + [1024, 4099]:
        buf = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        want = R.checksum_crc32_table(buf, n)
        assert want == R.checksum_crc32(buf, n)
        assert lib.ddcmi_crc32(buf, n) == want, n


@pytest.mark.parametrize("task,ntasks", [(0, 1), (3, 8), (7, 8), (1, 2)])
def test_prime_sequence_of_the_lcg64_defaults_is_the_references(tmp_path, task, ntasks):
    """primes.c (prime_init(30000, rank, size) in ddcMD.c:70, nextPrime per three particles in lcg64_default): the
    reference's own sequence for four (task, tasks) pairs against the oracle's restatement of its primality test and
    against the product's Miller-Rabin (deck.c, ddcmi_lcg64_default) -- 700 primes each, i.e. across block boundaries"""
    R = _ref(tmp_path, "%d_%d" % (task, ntasks))
    R.prime_init(30000, task, ntasks)
    want = [int(R.nextPrime()) for _ in range(700)]
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
    R = _ref()
    R.loopFormatInit(12)
    R.gidFormatInit(b"decimal")
    assert R.loopFormat() == b"%12.12lu" and R.gidFormat() == b"%12.12lu"      # PRIu64 on this platform
    src = open(os.path.join(ROOT, "ddcmd_amd", "csrc", "host", "plugin.c")).read()
    assert '"snapshot.%012" PRId64' in src and '%12.12" PRIu64' in src
    assert "snapshot.%012d" % 40 == "snapshot." + (R.loopFormat().decode().replace("lu", "d") % 40)
