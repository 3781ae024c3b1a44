"""ctypes loader for the product: libddcmi.so -- the drop-in boundary, HIP device code behind the C-ABI of
include/ddcmi.h and nothing else -- and libddcmi_host.so, the stand-alone host layer (object-file reader, units,
deck loader, plugin glue under ddcMD's own names).  Fails loudly when absent."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DDCMI_LIB: kernel-tuning builds of the device library (tools/variant.py)
LIB_PATH = os.environ.get("DDCMI_LIB") or os.path.join(_HERE, "lib", "libddcmi.so")
HOST_LIB_PATH = os.path.join(_HERE, "lib", "libddcmi_host.so")
# the second link of the same objects that also exports include/ddcmi_test.h (in-process domain groups, the halo planner's host
# logic, the branch census): tests/ and tools/ only.  A tuning build (DDCMI_LIB) carries both tables.
TEST_LIB_PATH = os.environ.get("DDCMI_LIB") or os.path.join(_HERE, "lib", "libddcmi_test.so")


class LibraryMissing(RuntimeError):
    pass


class _Libs(object):
    """one handle over both libraries: ddcmi_* of include/ddcmi.h from the device library, the rest from the host layer"""

    def __init__(self, dev, host):
        self.dev, self.host = dev, host

    def __getattr__(self, name):
        for lib in (self.dev, self.host):
            try:
                f = getattr(lib, name)
            except AttributeError:
                continue
            setattr(self, name, f)
            return f
        raise AttributeError(name)


_lib = None
_test_lib = None


def load_test_library():
    """libddcmi_test.so (+ the host layer): the product's objects with the test-only entry points of include/ddcmi_test.h exported
    as well.  A context belongs to the library that created it: drive it through this handle only."""
    global _test_lib
    if _test_lib is not None:
        return _test_lib
    base = load_library()
    if not os.path.exists(TEST_LIB_PATH):
        raise LibraryMissing("%s not found -- build it with `make -C ddcmd_amd/csrc`" % TEST_LIB_PATH)
    dev = base.dev if os.path.realpath(TEST_LIB_PATH) == os.path.realpath(LIB_PATH) else ctypes.CDLL(TEST_LIB_PATH, mode=ctypes.RTLD_LOCAL)
    _test_lib = _Libs(dev, base.host)
    _declare(_test_lib)
    return _test_lib


def load_library():
    """Return the handle of libddcmi.so + libddcmi_host.so; raise LibraryMissing if they are not built.

    There is deliberately no fallback: the HIP library *is* the implementation.
    """
    global _lib
    if _lib is not None:
        return _lib
    for path in (LIB_PATH, HOST_LIB_PATH):
        if not os.path.exists(path):
            raise LibraryMissing(
                "%s not found -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C ddcmd_amd/csrc`)" % path)
    # several processes sharing device memory (RCCL between the ranks of a node): this pool's host driver only supports
    # dmabuf IPC; the HIP runtime reads the switch when it initialises, i.e. after this line
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)      # (first, and global: the host layer's libddcmi.so dependency resolves to THIS copy by soname)
    host = ctypes.CDLL(HOST_LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    _lib = _Libs(dev, host)
    _declare(_lib)
    return _lib


c_double_p = ctypes.POINTER(ctypes.c_double)
c_int_p = ctypes.POINTER(ctypes.c_int)
c_u64_p = ctypes.POINTER(ctypes.c_uint64)
c_char_pp = ctypes.POINTER(ctypes.c_char_p)


class CSetup(ctypes.Structure):
    """Mirror of struct ddcmi_setup (ddcmd_amd/csrc/host/deck.h)."""
    _fields_ = [
        ("loop", ctypes.c_int64), ("maxloop", ctypes.c_int64), ("deltaloop", ctypes.c_int64),
        ("time", ctypes.c_double), ("dt", ctypes.c_double),
        ("printrate", ctypes.c_int), ("snapshotrate", ctypes.c_int), ("checkpointrate", ctypes.c_int),
        ("h", ctypes.c_double * 9),
        ("pbc", ctypes.c_int),
        ("deltaR", ctypes.c_double),
        ("updateRate", ctypes.c_int),
        ("lx", ctypes.c_int), ("ly", ctypes.c_int), ("lz", ctypes.c_int),
        ("rmax", ctypes.c_double), ("rcoulomb", ctypes.c_double), ("epsilon_r", ctypes.c_double),
        ("epsilon_rf", ctypes.c_double), ("krf", ctypes.c_double), ("crf", ctypes.c_double), ("keR", ctypes.c_double),
        ("excludePotentialTerm", ctypes.c_int), ("potentialShift", ctypes.c_int),
        ("nlj", ctypes.c_int),
        ("sigma", c_double_p), ("eps", c_double_p), ("shift", c_double_p),
        ("nspecies", ctypes.c_int),
        ("species_name", c_char_pp),
        ("mass", c_double_p), ("charge", c_double_p),
        ("ljtype", c_int_p), ("moltype", c_int_p), ("resitype", c_int_p), ("atomoffset", c_int_p),
        ("nmoltype", ctypes.c_int),
        ("mol_nspecies", c_int_p), ("bpair_off", c_int_p), ("bpairI", c_int_p), ("bpairJ", c_int_p),
        ("nresi", ctypes.c_int),
        ("resi_natoms", c_int_p),
        ("bond_off", c_int_p), ("bondI", c_int_p), ("bondJ", c_int_p),
        ("bond_kb", c_double_p), ("bond_b0", c_double_p),
        ("angle_off", c_int_p), ("angleI", c_int_p), ("angleJ", c_int_p), ("angleK", c_int_p), ("angle_func", c_int_p),
        ("angle_k", c_double_p), ("angle_t0", c_double_p),
        ("tors_off", c_int_p), ("torsI", c_int_p), ("torsJ", c_int_p), ("torsK", c_int_p), ("torsL", c_int_p),
        ("tors_func", c_int_p), ("tors_n", c_int_p),
        ("tors_k", c_double_p), ("tors_delta", c_double_p),
        ("ngroup", ctypes.c_int),
        ("group_name", c_char_pp),
        ("group_type", c_int_p),
        ("group_Teq", c_double_p), ("group_tau", c_double_p),
        ("group_interval", c_int_p),
        ("natoms", ctypes.c_int),
        ("rx", c_double_p), ("ry", c_double_p), ("rz", c_double_p),
        ("vx", c_double_p), ("vy", c_double_p), ("vz", c_double_p),
        ("gid", c_u64_p),
        ("species", c_int_p), ("group", c_int_p),
        ("nConstraints", ctypes.c_int),
        ("integrator_type", ctypes.c_char_p),
        ("has_accelerator", ctypes.c_int),
        ("accelerator_type", ctypes.c_char_p),
        ("u_pressure", ctypes.c_char_p), ("u_volume", ctypes.c_char_p), ("u_temperature", ctypes.c_char_p),
        ("u_energy", ctypes.c_char_p), ("u_time", ctypes.c_char_p), ("u_length", ctypes.c_char_p),
        ("rng_seed", ctypes.c_uint64),
        ("nrest", ctypes.c_int), ("rest_origin", ctypes.c_int), ("printMolecularPressure", ctypes.c_int),
        ("nresicons", ctypes.c_int),
        ("npt_T", ctypes.c_double), ("npt_P0", ctypes.c_double), ("npt_beta", ctypes.c_double), ("npt_tau", ctypes.c_double),
        ("rest_gid", c_u64_p), ("rest_fc", c_int_p), ("rest_r0", c_double_p), ("rest_kb", c_double_p),
        ("cons_off", c_int_p), ("consI", c_int_p), ("consJ", c_int_p), ("cons_grp", c_int_p), ("cons_r0", c_double_p),
        ("npt_isotropic", ctypes.c_int),
        ("printStress", ctypes.c_int), ("printHmatrix", ctypes.c_int), ("u_energyflux", ctypes.c_char_p),
        ("random_name", ctypes.c_char_p), ("random_lcg64", ctypes.c_int), ("lcg_from_file", ctypes.c_int),
        ("lcg_state", c_u64_p), ("lcg_multID", ctypes.POINTER(ctypes.c_uint32)), ("lcg_prime", ctypes.POINTER(ctypes.c_uint32)),
        ("group_vcm", c_double_p),
    ]


def _declare(lib):
    lib.ddcmi_deck_load_with.restype = ctypes.POINTER(CSetup)
    lib.ddcmi_deck_load_with.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int]
    lib.ddcmi_setup_free.restype = None
    lib.ddcmi_setup_free.argtypes = [ctypes.POINTER(CSetup)]
    lib.ddcmi_setup_sizeof.restype = ctypes.c_int
    lib.units_convert.restype = ctypes.c_double
    lib.units_convert.argtypes = [ctypes.c_double, ctypes.c_char_p, ctypes.c_char_p]
    lib.units_ke.restype = ctypes.c_double
    lib.units_kB.restype = ctypes.c_double
    lib.units_ddcmd_defaults.restype = None
    assert lib.ddcmi_setup_sizeof() == ctypes.sizeof(CSetup), "struct ddcmi_setup layout mismatch"
