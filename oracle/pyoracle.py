"""ctypes binding of the CPU oracle (libddc_oracle.so).

TEST INFRASTRUCTURE: import this only from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  PARITY UNPINNED (see ddc_oracle.h).
"""
import ctypes
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "libddc_oracle.so")

dp = ctypes.POINTER(ctypes.c_double)
ip = ctypes.POINTER(ctypes.c_int)
up = ctypes.POINTER(ctypes.c_uint64)

E_NAMES = ("lj", "ele", "bond", "angle", "tors", "impr", "total", "restraint")


class OrcParams(ctypes.Structure):
    _fields_ = [
        ("hxx", ctypes.c_double), ("hyy", ctypes.c_double), ("hzz", ctypes.c_double), ("pbc", ctypes.c_int),
        ("deltaR", ctypes.c_double),
        ("rmax", ctypes.c_double), ("krf", ctypes.c_double), ("crf", ctypes.c_double), ("keR", ctypes.c_double),
        ("nlj", ctypes.c_int), ("sigma", dp), ("eps", dp), ("shift", dp),
        ("nspecies", ctypes.c_int), ("mass", dp), ("charge", dp), ("ljtype", ip), ("moltype", ip), ("resitype", ip),
        ("nmoltype", ctypes.c_int), ("mol_nspecies", ip), ("bpair_off", ip), ("bpairI", ip), ("bpairJ", ip),
        ("nresi", ctypes.c_int), ("resi_natoms", ip),
        ("bond_off", ip), ("bondI", ip), ("bondJ", ip), ("bond_kb", dp), ("bond_b0", dp),
        ("angle_off", ip), ("angleI", ip), ("angleJ", ip), ("angleK", ip), ("angle_func", ip), ("angle_k", dp), ("angle_t0", dp),
        ("tors_off", ip), ("torsI", ip), ("torsJ", ip), ("torsK", ip), ("torsL", ip), ("tors_func", ip), ("tors_n", ip),
        ("tors_k", dp), ("tors_delta", dp),
        ("excludePotentialTerm", ctypes.c_int),
        ("nrest", ctypes.c_int), ("rest_gid", up), ("rest_fc", ip), ("rest_r0", dp), ("rest_kb", dp), ("rest_origin", ctypes.c_int),
        ("cons_off", ip), ("consI", ip), ("consJ", ip), ("cons_grp", ip), ("cons_r0", dp),
        ("baro_isotropic", ctypes.c_int),
    ]


class OrcGroup(ctypes.Structure):
    _fields_ = [("type", ctypes.c_int), ("Teq", ctypes.c_double), ("tau", ctypes.c_double), ("interval", ctypes.c_int),
                ("lambda_", ctypes.c_double), ("Tsum", ctypes.c_double), ("nT", ctypes.c_int), ("doScaling", ctypes.c_int),
                ("temperature", ctypes.c_double), ("seed", ctypes.c_ulonglong), ("lcg", ctypes.c_void_p), ("vcm", ctypes.c_double * 3)]


# LCG64_PARM (lcg64.h:8-12) as a numpy record
LCG64 = np.dtype([("state", "<u8"), ("multID", "<u4"), ("prime", "<u4")])


def build(native=False, out=None):
    """Compile the oracle with gcc.  native=True adds -march=native (CPU-baseline timing)."""
    out = out or LIB
    flags = ["-O3", "-std=gnu99", "-fPIC", "-ffp-contract=off"]
    if native:
        flags.append("-march=native")
    subprocess.check_call(["gcc"] + flags + ["-shared", "-o", out, os.path.join(_HERE, "ddc_oracle.c"), "-lm"])
    return out


_libs = {}


def lib(path=None):
    path = path or LIB
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        build(out=path)
    L = ctypes.CDLL(path)
    L.orc_nbr_build.restype = ctypes.c_void_p
    L.orc_nbr_build.argtypes = [ctypes.POINTER(OrcParams), ctypes.c_int, dp, dp, dp, up, ip]
    L.orc_nbr_free.argtypes = [ctypes.c_void_p]
    L.orc_nbr_npairs.restype = ctypes.c_long
    L.orc_nbr_npairs.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.orc_nbr_csr.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ip), ctypes.POINTER(ip)]
    L.orc_forces.argtypes = [ctypes.POINTER(OrcParams), ctypes.c_void_p, ctypes.c_int, dp, dp, dp, up, ip, dp, dp, dp, dp, dp]
    L.orc_bonded.argtypes = [ctypes.POINTER(OrcParams), ctypes.c_int, dp, dp, dp, up, ip, dp, dp, dp, dp, dp]
    L.orc_brute_force.argtypes = [ctypes.POINTER(OrcParams), ctypes.c_int, dp, dp, dp, up, ip, dp, dp, dp, dp, dp, dp,
                                  ctypes.POINTER(ctypes.c_long)]
    L.orc_kinetic.argtypes = [ctypes.POINTER(OrcParams), ctypes.c_int, dp, dp, dp, ip, dp, dp]
    L.orc_kinetic_detail.argtypes = [ctypes.POINTER(OrcParams), ctypes.c_int, dp, dp, dp, ip, ip, ctypes.c_int, ctypes.c_int, dp]
    L.orc_kinetic_detail.restype = None
    L.orc_energyinfo.argtypes = [ctypes.POINTER(OrcParams), ctypes.c_double, ctypes.c_int, ctypes.c_double, ctypes.c_double, dp, dp, dp]
    L.orc_nglf_step.argtypes = [ctypes.POINTER(OrcParams), ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_double,
                                ctypes.POINTER(ctypes.c_long), dp, ctypes.c_int,
                                dp, dp, dp, dp, dp, dp, dp, dp, dp, up, ip, ip,
                                ctypes.c_int, ctypes.POINTER(OrcGroup), dp, dp, dp, dp]
    L.orc_group_temperature.argtypes = [ctypes.POINTER(OrcParams), ctypes.c_int, dp, dp, dp, ip, ip, ctypes.c_int, ctypes.POINTER(OrcGroup)]
    L.orc_back_in_box.argtypes = [ctypes.POINTER(OrcParams), ctypes.c_int, dp, dp, dp]
    L.orc_velocity_constraint.argtypes = [ctypes.POINTER(OrcParams), ctypes.c_int, ctypes.c_double, ctypes.c_int, dp, dp, dp, dp, dp, dp, up, ip]
    L.orc_velocity_constraint.restype = ctypes.c_int
    L.orc_barostat_mol.argtypes = [ctypes.POINTER(OrcParams), ctypes.c_int, dp, dp, dp, dp, dp, dp, up, ip, dp] + [ctypes.c_double] * 5 + [dp]
    L.orc_lcg64.restype = ctypes.c_double
    L.orc_lcg64.argtypes = [ctypes.c_void_p]
    L.orc_gasdev3d.argtypes = [ctypes.c_void_p, dp]
    L.orc_is_prime1.argtypes = [ctypes.c_ulonglong]
    L.orc_lcg64_default.argtypes = [ctypes.c_int, up, ctypes.c_uint, ctypes.c_uint, ctypes.c_void_p]
    _libs[path] = L
    return L


def _d(a):
    return a.ctypes.data_as(dp)


def _i(a):
    return a.ctypes.data_as(ip)


def lcg64_default(labels, task=0, ntasks=1, libpath=None):
    """the random state particles get when their file carries none (collection.c:95-109, lcg64.c:98-110, primes.c), in file order"""
    labels = np.ascontiguousarray(labels, dtype=np.uint64)
    out = np.zeros(len(labels), dtype=LCG64)
    lib(libpath).orc_lcg64_default(len(labels), labels.ctypes.data_as(up), int(task), int(ntasks), out.ctypes.data)
    return out


def gasdev3d(parm, libpath=None):
    """three unit normals from one particle's stream (random.c:135-160); parm: a 1-element LCG64 array, advanced in place"""
    g = np.zeros(3)
    lib(libpath).orc_gasdev3d(parm.ctypes.data, _d(g))
    return g


class Oracle(object):
    """The reference CPU path on one rank, driven from a ddcmd_amd.Setup."""

    def __init__(self, setup, libpath=None, constraints=False, lcg64="deck"):
        self.L = lib(libpath)
        s = self.s = setup
        self._keep = {}
        p = self.p = OrcParams()
        p.hxx, p.hyy, p.hzz, p.pbc = s.h[0], s.h[4], s.h[8], s.pbc
        assert abs(s.h[1]) + abs(s.h[2]) + abs(s.h[3]) + abs(s.h[5]) + abs(s.h[6]) + abs(s.h[7]) < 1e-12, "orthorhombic only"
        p.deltaR, p.rmax, p.krf, p.crf, p.keR = s.deltaR, s.rmax, s.krf, s.crf, s.keR
        p.nlj, p.nspecies, p.nmoltype, p.nresi = s.nlj, s.nspecies, s.nmoltype, s.nresi
        p.excludePotentialTerm = s.excludePotentialTerm
        # RESTRAINT potential (optional Setup fields)
        nrest = int(getattr(s, "nrest", 0))
        p.nrest = nrest
        p.rest_origin = int(getattr(s, "rest_origin", 0))
        self._keep["rest_gid"] = np.ascontiguousarray(getattr(s, "rest_gid", np.zeros(1)), dtype=np.uint64) if nrest else np.zeros(1, np.uint64)
        self._keep["rest_fc"] = np.ascontiguousarray(getattr(s, "rest_fc", np.zeros(3)), dtype=np.int32).ravel() if nrest else np.zeros(3, np.int32)
        self._keep["rest_r0"] = np.ascontiguousarray(getattr(s, "rest_r0", np.zeros(3)), dtype=np.float64).ravel() if nrest else np.zeros(3)
        self._keep["rest_kb"] = np.ascontiguousarray(getattr(s, "rest_kb", np.zeros(1)), dtype=np.float64) if nrest else np.zeros(1)
        p.rest_gid = self._keep["rest_gid"].ctypes.data_as(up)
        p.rest_fc = _i(self._keep["rest_fc"])
        p.rest_r0 = _d(self._keep["rest_r0"])
        p.rest_kb = _d(self._keep["rest_kb"])
        for k in ("sigma", "eps", "shift", "mass", "charge", "bond_kb", "bond_b0", "angle_k", "angle_t0", "tors_k", "tors_delta"):
            a = np.ascontiguousarray(getattr(s, k), dtype=np.float64)
            if a.size == 0:
                a = np.zeros(1)
            self._keep[k] = a
            setattr(p, k, _d(a))
        for k in ("ljtype", "moltype", "resitype", "mol_nspecies", "bpair_off", "bpairI", "bpairJ", "resi_natoms",
                  "bond_off", "bondI", "bondJ", "angle_off", "angleI", "angleJ", "angleK", "angle_func",
                  "tors_off", "torsI", "torsJ", "torsK", "torsL", "tors_func", "tors_n"):
            a = np.ascontiguousarray(getattr(s, k), dtype=np.int32)
            if a.size == 0:
                a = np.zeros(1, np.int32)
            self._keep[k] = a
            setattr(p, k, _i(a))
        p.baro_isotropic = int(getattr(s, "npt_isotropic", 0))
        # distance constraints (nglfconstraint); off unless constraints=True
        self.constraints = bool(constraints) and int(np.asarray(getattr(s, "cons_off", [0]))[-1]) > 0
        if self.constraints:
            for k in ("cons_off", "consI", "consJ", "cons_grp"):
                self._keep[k] = np.ascontiguousarray(getattr(s, k), dtype=np.int32)
                setattr(p, k, _i(self._keep[k]))
            self._keep["cons_r0"] = np.ascontiguousarray(s.cons_r0, dtype=np.float64)
            p.cons_r0 = _d(self._keep["cons_r0"])
        n = self.n = s.natoms
        self.rx, self.ry, self.rz = (np.array(getattr(s, k), dtype=np.float64) for k in ("rx", "ry", "rz"))
        self.vx, self.vy, self.vz = (np.array(getattr(s, k), dtype=np.float64) for k in ("vx", "vy", "vz"))
        self.fx, self.fy, self.fz = np.zeros(n), np.zeros(n), np.zeros(n)
        self.gid = np.ascontiguousarray(s.gid, dtype=np.uint64)
        self.species = np.ascontiguousarray(s.species, dtype=np.int32)
        self.group = np.ascontiguousarray(s.group, dtype=np.int32)
        self.groups = (OrcGroup * max(1, s.ngroup))()
        for g in range(s.ngroup):
            self.groups[g].type = int(s.group_type[g]) if int(s.group_type[g]) in (1, 2) else 0
            self.groups[g].seed = int(getattr(s, "rng_seed", 0))
            self.groups[g].Teq = s.group_Teq[g]
            self.groups[g].tau = s.group_tau[g]
            self.groups[g].interval = max(1, int(s.group_interval[g]))
            self.groups[g].lambda_ = 1.0
            vcm = getattr(s, "group_vcm", None)
            if vcm is not None and len(np.ravel(vcm)) >= 3 * (g + 1):
                for k in range(3):
                    self.groups[g].vcm[k] = float(np.ravel(vcm)[3 * g + k])
        self.lcg = None
        if getattr(s, "lcg64", None) is not None and lcg64 == "deck" and np.any(np.asarray(s.group_type) == 2):
            self.set_lcg64(s.lcg64)       # RANDOM type=LCG64 in the deck (one task: system.c:135, langevin.c:95-96)
        self.nbr = None
        self.loop = ctypes.c_long(int(s.loop))
        self.time = ctypes.c_double(float(s.time))
        self.e = np.zeros(8)
        self.virial = np.zeros(6)
        self.rk = ctypes.c_double(0.0)
        self.tion = np.zeros(6)

    def __del__(self):
        try:
            if self.nbr:
                self.L.orc_nbr_free(self.nbr)
        except Exception:
            pass

    def build_list(self):
        if self.nbr:
            self.L.orc_nbr_free(self.nbr)
        self.nbr = self.L.orc_nbr_build(ctypes.byref(self.p), self.n, _d(self.rx), _d(self.ry), _d(self.rz),
                                        self.gid.ctypes.data_as(up), _i(self.species))
        return self.L.orc_nbr_npairs(self.nbr, 0), self.L.orc_nbr_npairs(self.nbr, 1)

    def forces(self):
        if not self.nbr:
            self.build_list()
        self.L.orc_forces(ctypes.byref(self.p), self.nbr, self.n, _d(self.rx), _d(self.ry), _d(self.rz),
                          self.gid.ctypes.data_as(up), _i(self.species), _d(self.fx), _d(self.fy), _d(self.fz),
                          _d(self.e), _d(self.virial))
        return dict(zip(E_NAMES, self.e.tolist())), self.virial.copy()

    def brute_force(self):
        fx, fy, fz = np.zeros(self.n), np.zeros(self.n), np.zeros(self.n)
        vlj, vele = ctypes.c_double(0), ctypes.c_double(0)
        vir = np.zeros(6)
        npair = ctypes.c_long(0)
        self.L.orc_brute_force(ctypes.byref(self.p), self.n, _d(self.rx), _d(self.ry), _d(self.rz),
                               self.gid.ctypes.data_as(up), _i(self.species), _d(fx), _d(fy), _d(fz),
                               ctypes.byref(vlj), ctypes.byref(vele), _d(vir), ctypes.byref(npair))
        return fx, fy, fz, vlj.value, vele.value, vir, npair.value

    def bonded_only(self):
        fx, fy, fz = np.zeros(self.n), np.zeros(self.n), np.zeros(self.n)
        e4, vir = np.zeros(4), np.zeros(6)
        self.L.orc_bonded(ctypes.byref(self.p), self.n, _d(self.rx), _d(self.ry), _d(self.rz),
                          self.gid.ctypes.data_as(up), _i(self.species), _d(fx), _d(fy), _d(fz), _d(e4), _d(vir))
        return fx, fy, fz, e4, vir

    def kinetic(self):
        rk = ctypes.c_double(0)
        tion = np.zeros(6)
        self.L.orc_kinetic(ctypes.byref(self.p), self.n, _d(self.vx), _d(self.vy), _d(self.vz), _i(self.species),
                           ctypes.byref(rk), _d(tion))
        return rk.value, tion

    def kinetic_detail(self, by_species):
        ncl = int(self.s.nspecies if by_species else max(1, self.s.ngroup))
        out = np.zeros((ncl, 12))
        self.L.orc_kinetic_detail(ctypes.byref(self.p), self.n, _d(self.vx), _d(self.vy), _d(self.vz), _i(self.species), _i(self.group),
                                  int(bool(by_species)), ncl, _d(out))
        return out

    def energy_info(self, eion, rk, virial, tion):
        out = np.zeros(9)
        self.L.orc_energyinfo(ctypes.byref(self.p), float(self.n), int(self.s.nConstraints), float(eion), float(rk),
                              _d(np.ascontiguousarray(virial)), _d(np.ascontiguousarray(tion)), _d(out))
        return {"temperature": out[0], "pressure": out[1], "sion": out[2:8].copy(), "energy": out[8]}

    def group_temperature(self):
        self.L.orc_group_temperature(ctypes.byref(self.p), self.n, _d(self.vx), _d(self.vy), _d(self.vz),
                                     _i(self.species), _i(self.group), self.s.ngroup, self.groups)

    def step_npt(self, nsteps, T, P0, beta, tau, dt=None, molecular=False):
        """nglfconstraint without constraints: barostat (from the last virial) + nglf, per step"""
        dt = self.s.dt if dt is None else dt
        out = None
        for _ in range(nsteps):
            if molecular:
                pm = np.zeros(3)
                self.L.orc_barostat_mol(ctypes.byref(self.p), self.n, _d(self.rx), _d(self.ry), _d(self.rz),
                                        _d(self.fx), _d(self.fy), _d(self.fz), self.gid.ctypes.data_as(up), _i(self.species), _d(self.virial),
                                        ctypes.c_double(T), ctypes.c_double(P0), ctypes.c_double(beta), ctypes.c_double(tau), ctypes.c_double(dt), _d(pm))
                self.pmol = pm
            else:
                self.L.orc_barostat(ctypes.byref(self.p), self.n, _d(self.rx), _d(self.ry), _d(self.rz), _d(self.virial),
                                    ctypes.c_double(T), ctypes.c_double(P0), ctypes.c_double(beta), ctypes.c_double(tau), ctypes.c_double(dt))
            out = self.step(1, dt)
        return out

    def constraint_sweep(self, location, dt=None):
        """one velocityConstraintOld call on the current state; returns the largest sweep count"""
        dt = self.s.dt if dt is None else dt
        return self.L.orc_velocity_constraint(ctypes.byref(self.p), self.n, ctypes.c_double(dt), int(location), _d(self.rx), _d(self.ry), _d(self.rz),
                                              _d(self.vx), _d(self.vy), _d(self.vz), self.gid.ctypes.data_as(up), _i(self.species))

    @property
    def box(self):
        return np.array([self.p.hxx, self.p.hyy, self.p.hzz])

    def set_lcg64(self, parms):
        """LANGEVIN groups draw from the reference's per-particle LCG64 streams (parms: LCG64 records in particle order, advanced in place)"""
        self.lcg = np.ascontiguousarray(parms, dtype=LCG64).copy()
        assert len(self.lcg) == self.n
        for g in range(self.s.ngroup):
            self.groups[g].lcg = self.lcg.ctypes.data

    def step(self, nsteps=1, dt=None):
        """nglf steps; forces()/first energy call must have run once (firstEnergyCall, masters.c:579)."""
        dt = self.s.dt if dt is None else dt
        nbr = ctypes.c_void_p(self.nbr)
        for _ in range(nsteps):
            self.L.orc_nglf_step(ctypes.byref(self.p), ctypes.byref(nbr), int(self.s.updateRate), float(dt),
                                 ctypes.byref(self.loop), ctypes.byref(self.time), self.n,
                                 _d(self.rx), _d(self.ry), _d(self.rz), _d(self.vx), _d(self.vy), _d(self.vz),
                                 _d(self.fx), _d(self.fy), _d(self.fz),
                                 self.gid.ctypes.data_as(up), _i(self.species), _i(self.group),
                                 self.s.ngroup, self.groups, _d(self.e), _d(self.virial), ctypes.byref(self.rk), _d(self.tion))
        self.nbr = nbr.value
        return dict(zip(E_NAMES, self.e.tolist())), self.virial.copy(), self.rk.value, self.tion.copy()
