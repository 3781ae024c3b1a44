"""Flat description of a Martini run (numpy mirror of struct ddcmi_setup).

`load_deck` calls the host C deck loader inside libddcmi.so (object.data / restart /
martini.data / atoms#000000, ddcMD formats) and copies the result into numpy arrays.
"""
import ctypes
import numpy as np
from . import _lib

_INT_ARRAYS_SPECIES = ("ljtype", "moltype", "resitype", "atomoffset")
GROUP_FREE, GROUP_BERENDSEN, GROUP_LANGEVIN, GROUP_OTHER = 0, 1, 2, 3


class Setup(object):
    """Everything the hot path needs, internal ddcMD units (bohr, fs, Ry, e; kB=1)."""

    scalar_fields = ("loop", "maxloop", "time", "dt", "printrate", "pbc", "deltaR", "updateRate",
                     "lx", "ly", "lz", "rmax", "rcoulomb", "epsilon_r", "epsilon_rf", "krf", "crf", "keR",
                     "excludePotentialTerm", "nlj", "nspecies", "nmoltype", "nresi", "ngroup", "natoms",
                     "nConstraints")

    def __init__(self):
        self.loop = 0
        self.maxloop = 0
        self.time = 0.0
        self.dt = 0.0
        self.printrate = 1
        self.h = np.zeros(9)
        self.pbc = 7
        self.deltaR = 0.0
        self.updateRate = 20
        self.lx = self.ly = self.lz = 0
        self.rmax = self.rcoulomb = 0.0
        self.epsilon_r = 15.0
        self.epsilon_rf = -1.0
        self.krf = self.crf = self.keR = 0.0
        self.excludePotentialTerm = 0
        self.nlj = 0
        self.sigma = self.eps = self.shift = np.zeros(0)
        self.nspecies = 0
        self.species_name = []
        self.mass = self.charge = np.zeros(0)
        self.ljtype = self.moltype = self.resitype = self.atomoffset = np.zeros(0, np.int32)
        self.nmoltype = 0
        self.mol_nspecies = np.zeros(0, np.int32)
        self.bpair_off = np.zeros(1, np.int32)
        self.bpairI = self.bpairJ = np.zeros(0, np.int32)
        self.nresi = 0
        self.resi_natoms = np.zeros(0, np.int32)
        self.bond_off = self.angle_off = self.tors_off = np.zeros(1, np.int32)
        self.bondI = self.bondJ = np.zeros(0, np.int32)
        self.bond_kb = self.bond_b0 = np.zeros(0)
        self.angleI = self.angleJ = self.angleK = self.angle_func = np.zeros(0, np.int32)
        self.angle_k = self.angle_t0 = np.zeros(0)
        self.torsI = self.torsJ = self.torsK = self.torsL = self.tors_func = self.tors_n = np.zeros(0, np.int32)
        self.tors_k = self.tors_delta = np.zeros(0)
        self.cons_off = np.zeros(1, np.int32)       # distance constraints per residue type (CONSLISTPARMS)
        self.consI = self.consJ = self.cons_grp = np.zeros(0, np.int32)
        self.cons_r0 = np.zeros(0)
        self.ngroup = 0
        self.group_name = []
        self.group_type = self.group_interval = np.zeros(0, np.int32)
        self.group_Teq = self.group_tau = np.zeros(0)
        self.natoms = 0
        self.rx = self.ry = self.rz = self.vx = self.vy = self.vz = np.zeros(0)
        self.gid = np.zeros(0, np.uint64)
        self.species = self.group = np.zeros(0, np.int32)
        self.nConstraints = 0
        self.integrator_type = "NGLF"
        self.has_accelerator = 0
        self.rng_seed = 0
        self.lcg64 = None             # RANDOM type=LCG64: records {state, multID, prime} per atom (file order), else None
        self.lcg_from_file = 0
        self.npt_T = self.npt_P0 = self.npt_beta = self.npt_tau = 0.0
        self.npt_isotropic = 0
        self.nresicons = 0
        self.nrest, self.rest_origin = 0, 0
        self.rest_gid = np.zeros(0, np.uint64)
        self.rest_fc = np.zeros((0, 3), np.int32)
        self.rest_r0 = np.zeros((0, 3))
        self.rest_kb = np.zeros(0)
        self.accelerator_type = "NONE"
        self.units = {}

    @property
    def box(self):
        return np.array([self.h[0], self.h[4], self.h[8]])

    @property
    def volume(self):
        return float(self.h[0] * self.h[4] * self.h[8])


def _arr(ptr, n, dtype):
    if n <= 0 or not ptr:
        return np.zeros(0, dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)


def load_deck(object_file, restart_file=None, extra_objects=None):
    """Load a ddcMD deck through the host C loader (deck.c)."""
    lib = _lib.load_library()
    err = ctypes.create_string_buffer(1024)
    p = lib.ddcmi_deck_load_with(object_file.encode(), restart_file.encode() if restart_file else None,
                                 extra_objects.encode() if extra_objects else None, err, 1024)
    if not p:
        raise RuntimeError("deck load failed: " + err.value.decode())
    c = p.contents
    s = Setup()
    try:
        for f in ("loop", "maxloop", "time", "dt", "printrate", "pbc", "deltaR", "updateRate", "lx", "ly", "lz",
                  "rmax", "rcoulomb", "epsilon_r", "epsilon_rf", "krf", "crf", "keR", "excludePotentialTerm",
                  "nlj", "nspecies", "nmoltype", "nresi", "ngroup", "natoms", "nConstraints", "has_accelerator"):
            setattr(s, f, getattr(c, f))
        s.h = np.array(list(c.h), dtype=np.float64)
        s.rng_seed = int(c.rng_seed)
        if int(c.random_lcg64) and c.natoms > 0:
            s.lcg64 = np.zeros(c.natoms, dtype=[("state", "<u8"), ("multID", "<u4"), ("prime", "<u4")])
            s.lcg64["state"] = _arr(c.lcg_state, c.natoms, np.uint64)
            s.lcg64["multID"] = _arr(c.lcg_multID, c.natoms, np.uint32)
            s.lcg64["prime"] = _arr(c.lcg_prime, c.natoms, np.uint32)
            s.lcg_from_file = int(c.lcg_from_file)
        s.npt_T, s.npt_P0, s.npt_beta, s.npt_tau = float(c.npt_T), float(c.npt_P0), float(c.npt_beta), float(c.npt_tau)
        s.npt_isotropic = int(c.npt_isotropic)
        s.nresicons = int(c.nresicons)
        s.nrest, s.rest_origin = int(c.nrest), int(c.rest_origin)
        s.rest_gid = _arr(c.rest_gid, s.nrest, np.uint64) if s.nrest else np.zeros(0, np.uint64)
        s.rest_fc = _arr(c.rest_fc, 3 * s.nrest, np.int32).reshape(-1, 3) if s.nrest else np.zeros((0, 3), np.int32)
        s.rest_r0 = _arr(c.rest_r0, 3 * s.nrest, np.float64).reshape(-1, 3) if s.nrest else np.zeros((0, 3))
        s.rest_kb = _arr(c.rest_kb, s.nrest, np.float64) if s.nrest else np.zeros(0)
        n2 = c.nlj * c.nlj
        s.sigma, s.eps, s.shift = (_arr(getattr(c, k), n2, np.float64) for k in ("sigma", "eps", "shift"))
        ns = c.nspecies
        s.species_name = [c.species_name[i].decode() for i in range(ns)]
        s.mass = _arr(c.mass, ns, np.float64)
        s.charge = _arr(c.charge, ns, np.float64)
        for k in _INT_ARRAYS_SPECIES:
            setattr(s, k, _arr(getattr(c, k), ns, np.int32))
        nm = c.nmoltype
        s.mol_nspecies = _arr(c.mol_nspecies, nm, np.int32)
        s.bpair_off = _arr(c.bpair_off, nm + 1, np.int32) if nm > 0 else np.zeros(1, np.int32)
        nb = int(s.bpair_off[-1]) if nm > 0 else 0
        s.bpairI = _arr(c.bpairI, nb, np.int32)
        s.bpairJ = _arr(c.bpairJ, nb, np.int32)
        nr = c.nresi
        s.resi_natoms = _arr(c.resi_natoms, nr, np.int32)
        s.bond_off = _arr(c.bond_off, nr + 1, np.int32)
        s.angle_off = _arr(c.angle_off, nr + 1, np.int32)
        s.tors_off = _arr(c.tors_off, nr + 1, np.int32)
        nbond, nang, ntor = int(s.bond_off[-1]), int(s.angle_off[-1]), int(s.tors_off[-1])
        for k in ("bondI", "bondJ"):
            setattr(s, k, _arr(getattr(c, k), nbond, np.int32))
        for k in ("bond_kb", "bond_b0"):
            setattr(s, k, _arr(getattr(c, k), nbond, np.float64))
        for k in ("angleI", "angleJ", "angleK", "angle_func"):
            setattr(s, k, _arr(getattr(c, k), nang, np.int32))
        for k in ("angle_k", "angle_t0"):
            setattr(s, k, _arr(getattr(c, k), nang, np.float64))
        for k in ("torsI", "torsJ", "torsK", "torsL", "tors_func", "tors_n"):
            setattr(s, k, _arr(getattr(c, k), ntor, np.int32))
        for k in ("tors_k", "tors_delta"):
            setattr(s, k, _arr(getattr(c, k), ntor, np.float64))
        s.cons_off = _arr(c.cons_off, nr + 1, np.int32)
        ncons = int(s.cons_off[-1]) if nr > 0 else 0
        for k in ("consI", "consJ", "cons_grp"):
            setattr(s, k, _arr(getattr(c, k), ncons, np.int32))
        s.cons_r0 = _arr(c.cons_r0, ncons, np.float64)
        ng = c.ngroup
        s.group_name = [c.group_name[i].decode() for i in range(ng)]
        s.group_type = _arr(c.group_type, ng, np.int32)
        s.group_interval = _arr(c.group_interval, ng, np.int32)
        s.group_Teq = _arr(c.group_Teq, ng, np.float64)
        s.group_tau = _arr(c.group_tau, ng, np.float64)
        s.group_vcm = _arr(c.group_vcm, 3 * ng, np.float64) if c.group_vcm else np.zeros(3 * ng)
        na = c.natoms
        for k in ("rx", "ry", "rz", "vx", "vy", "vz"):
            setattr(s, k, _arr(getattr(c, k), na, np.float64))
        s.gid = _arr(c.gid, na, np.uint64)
        s.species = _arr(c.species, na, np.int32)
        s.group = _arr(c.group, na, np.int32)
        s.integrator_type = c.integrator_type.decode()
        s.accelerator_type = c.accelerator_type.decode()
        s.units = {k: getattr(c, "u_" + k).decode() for k in ("pressure", "volume", "temperature", "energy", "time", "length")}
    finally:
        lib.ddcmi_setup_free(p)
    return s


def units_convert(value, frm=None, to=None):
    """value*[frm] in [to]; None = ddcMD internal units (units.c)."""
    lib = _lib.load_library()
    return lib.units_convert(float(value), frm.encode() if frm else None, to.encode() if to else None)


_ARRAY_FIELDS = ("h", "sigma", "eps", "shift", "mass", "charge", "ljtype", "moltype", "resitype", "atomoffset",
                 "mol_nspecies", "bpair_off", "bpairI", "bpairJ", "resi_natoms",
                 "bond_off", "bondI", "bondJ", "bond_kb", "bond_b0",
                 "angle_off", "angleI", "angleJ", "angleK", "angle_func", "angle_k", "angle_t0",
                 "tors_off", "torsI", "torsJ", "torsK", "torsL", "tors_func", "tors_n", "tors_k", "tors_delta",
                 "group_type", "group_interval", "group_Teq", "group_tau",
                 "rx", "ry", "rz", "vx", "vy", "vz", "gid", "species", "group")


def setup_to_dict(s):
    """Setup -> dict of numpy arrays (for np.savez fixtures)."""
    d = {"setup_" + k: np.asarray(getattr(s, k)) for k in _ARRAY_FIELDS}
    for k in Setup.scalar_fields:
        d["setup_" + k] = np.asarray(getattr(s, k))
    d["setup_species_name"] = np.array(s.species_name)
    d["setup_group_name"] = np.array(s.group_name)
    return d


def setup_from_dict(d):
    s = Setup()
    for k in _ARRAY_FIELDS:
        setattr(s, k, np.array(d["setup_" + k]))
    for k in Setup.scalar_fields:
        v = d["setup_" + k]
        setattr(s, k, v.item())
    s.species_name = [str(x) for x in d["setup_species_name"]]
    s.group_name = [str(x) for x in d["setup_group_name"]]
    return s
