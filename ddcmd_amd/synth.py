"""Synthetic Martini water boxes: the bench / parity-test workload.

FCC lattice of 4*n^3 sites at the example deck's density (6173 beads /
93.858^3 A^3; nearest-neighbour distance 5.74 A), uniform jitter +-0.2 A per axis,
10 % antifreeze beads (BP4, species WFxWF) chosen by splitmix64(seed ^ i) % 10 == 0,
gid = i << 32 (one bead per molecule, like examples/waterbox), Maxwell-Boltzmann
velocities with the centre-of-mass velocity removed.  The initial temperature is
50 K: relaxing the jittered mixed-bead lattice releases ~260 K, so the run settles
at the deck's 310 K within ~100 steps.  Force-field numbers are those of examples/waterbox/martini.data:45-47.
Everything is generated from splitmix64 hashes so the same (n, seed) gives
bit-identical inputs on every machine.

SURVEY 8(d) proposed a simple-cubic lattice (a = 5.12 A, jitter +-0.5 A).  That
start puts P4-BP4 pairs (sigma = 5.7 A) at 4.1-5.1 A, i.e. tens of kJ/mol up the
repulsive wall, and the box explodes within ~6 steps at dt = 20 fs -- in the CPU
oracle and on the GPU alike (lattice="sc" keeps it available for that
demonstration).  FCC at the same density is the nearest stable equivalent:
same bead count per volume, hence the same neighbour-list sizes and traffic.
"""
import numpy as np
from .deck import Setup, units_convert, GROUP_FREE, GROUP_BERENDSEN

SEED = 20261002
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x):
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def _uniform(seed, idx, stream):
    """uniform in [0,1) from hash(seed, stream, idx)"""
    with np.errstate(over="ignore"):
        key = splitmix64(np.uint64(seed) ^ (np.uint64(stream) * np.uint64(0xD1342543DE82EF95))) ^ idx.astype(np.uint64)
    bits = splitmix64(key) >> np.uint64(11)
    return bits.astype(np.float64) * (1.0 / 9007199254740992.0)


def lj_shift(sigma, eps, rcut):
    s6 = (sigma / rcut) ** 6
    return -4.0 * eps * (s6 * s6 - s6)


def water_forcefield(s, rcut_A=12.0, skin_A=4.0, dt_fs=20.0, update_rate=20):
    """Fill the force-field/run-control part of a Setup like examples/waterbox does."""
    s.dt = units_convert(dt_fs, "fs")
    s.deltaR = units_convert(skin_A, "Angstrom")
    s.updateRate = update_rate
    s.rmax = units_convert(rcut_A, "Angstrom")
    s.rcoulomb = s.rmax
    s.epsilon_r, s.epsilon_rf = 15.0, -1.0
    irc = 1.0 / s.rcoulomb
    s.krf, s.crf = 0.5 * irc ** 3, 1.5 * irc
    from . import _lib
    s.keR = _lib.load_library().units_ke() / s.epsilon_r
    s.nlj = 2                      # 0 = BP4, 1 = P4 (martini.data:8-9)
    sig = np.zeros((2, 2))
    eps = np.zeros((2, 2))
    nm, kj = units_convert(1.0, "nm"), units_convert(1.0, "kJ*mol^-1")
    sig[1, 1], eps[1, 1] = 0.47 * nm, 5.0 * kj
    sig[0, 1] = sig[1, 0] = 0.57 * nm
    eps[0, 1] = eps[1, 0] = 5.6 * kj
    sig[0, 0], eps[0, 0] = 0.47 * nm, 5.0 * kj
    s.sigma, s.eps = sig.ravel().copy(), eps.ravel().copy()
    s.shift = lj_shift(s.sigma, s.eps, s.rmax)
    s.nspecies = 2
    s.species_name = ["WxW", "WFxWF"]
    mass = units_convert(72.0, "M_p")
    s.mass = np.array([mass, mass])
    s.charge = np.zeros(2)
    s.ljtype = np.array([1, 0], np.int32)
    s.moltype = np.array([0, 1], np.int32)
    s.resitype = np.array([0, 1], np.int32)
    s.atomoffset = np.zeros(2, np.int32)
    s.nmoltype = 2
    s.mol_nspecies = np.array([1, 1], np.int32)
    s.bpair_off = np.zeros(3, np.int32)
    s.nresi = 2
    s.resi_natoms = np.array([1, 1], np.int32)
    s.bond_off = np.zeros(3, np.int32)
    s.angle_off = np.zeros(3, np.int32)
    s.tors_off = np.zeros(3, np.int32)
    s.ngroup = 1
    s.group_name = ["group"]
    s.group_type = np.array([GROUP_FREE], np.int32)
    s.group_Teq = np.zeros(1)
    s.group_tau = np.zeros(1)
    s.group_interval = np.ones(1, np.int32)
    return s


def make_water_setup(n, seed=SEED, temperature_K=50.0, rcut_A=12.0, skin_A=4.0, dt_fs=20.0,
                     update_rate=20, thermostat=None, lattice="fcc", jitter_A=None, density_scale=1.0):
    """Martini water box: 4*n^3 beads on FCC (n=25: 62.5k, n=64: 1.05M, n=100: 4.0M);
    lattice="sc" gives SURVEY's n^3 simple-cubic start (unstable at 20 fs)."""
    s = Setup()
    water_forcefield(s, rcut_A, skin_A, dt_fs, update_rate)
    ang = units_convert(1.0, "Angstrom")
    vol_per_bead_A3 = 93.858 ** 3 / 6173.0 / float(density_scale)      # (density_scale != 1: kernel-tuning experiments only, tools/)
    if lattice == "fcc":
        N = 4 * n * n * n
        a_A = (4.0 * vol_per_bead_A3) ** (1.0 / 3.0)
        basis = np.array([[0.0, 0.0, 0.0], [0.5, 0.5, 0.0], [0.5, 0.0, 0.5], [0.0, 0.5, 0.5]])
        jit_A = 0.2 if jitter_A is None else jitter_A
        nb = 4
    elif lattice == "sc":
        N = n * n * n
        a_A = vol_per_bead_A3 ** (1.0 / 3.0)
        basis = np.array([[0.25, 0.25, 0.25]])
        jit_A = 0.5 if jitter_A is None else jitter_A
        nb = 1
    else:
        raise ValueError("lattice must be 'fcc' or 'sc'")
    L = n * a_A * ang
    s.h = np.array([L, 0, 0, 0, L, 0, 0, 0, L], dtype=np.float64)
    s.pbc = 7
    idx = np.arange(N, dtype=np.uint64)
    b = (idx % np.uint64(nb)).astype(np.int64)
    cell = idx // np.uint64(nb)
    ix = (cell % np.uint64(n)).astype(np.float64)
    iy = ((cell // np.uint64(n)) % np.uint64(n)).astype(np.float64)
    iz = (cell // np.uint64(n * n)).astype(np.float64)
    a = a_A * ang
    jit = jit_A * ang
    s.rx = (ix + basis[b, 0] + 0.25) * a - 0.5 * L + jit * (2.0 * _uniform(seed, idx, 1) - 1.0)
    s.ry = (iy + basis[b, 1] + 0.25) * a - 0.5 * L + jit * (2.0 * _uniform(seed, idx, 2) - 1.0)
    s.rz = (iz + basis[b, 2] + 0.25) * a - 0.5 * L + jit * (2.0 * _uniform(seed, idx, 3) - 1.0)
    is_bp4 = (splitmix64(np.uint64(seed) ^ idx) % np.uint64(10)) == np.uint64(0)
    s.species = is_bp4.astype(np.int32)          # 0 = WxW (P4), 1 = WFxWF (BP4)
    s.group = np.zeros(N, np.int32)
    s.gid = idx << np.uint64(32)
    kT = units_convert(temperature_K, "K")       # kB = 1 in internal units
    sig_v = np.sqrt(kT / s.mass[s.species])
    vel = []
    for c in range(3):
        u1 = 1.0 - _uniform(seed + 1, idx, 10 + 2 * c)      # (0,1]
        u2 = _uniform(seed + 1, idx, 11 + 2 * c)
        vel.append(sig_v * np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2))
    m = s.mass[s.species]
    for c in range(3):
        vel[c] -= np.sum(m * vel[c]) / np.sum(m)
    s.vx, s.vy, s.vz = vel
    s.natoms = N
    if thermostat == "berendsen":
        s.group_type = np.array([GROUP_BERENDSEN], np.int32)
        s.group_Teq = np.array([units_convert(310.0, "K")])
        s.group_tau = np.array([units_convert(1.0, "ps")])
    return s


def replicate_setup(s, reps):
    """Tile a periodic Setup reps=(a,b,c) times along x,y,z: the lipid workload at any size.

    A periodic box repeated is the same system, so every copy of a bead feels the
    forces of the original -- the size-independent parity property the full-size
    tests use.  Molecule ids (gid bits 63:32) of copy k are offset by k * (number of
    molecule ids of the original), everything below the molecule id is kept."""
    import copy
    a, b, c = (int(x) for x in reps)
    out = copy.copy(s)
    L = np.array([s.h[0], s.h[4], s.h[8]])
    lo = -0.5 * L
    ncopy = a * b * c
    n = s.natoms
    molid = (np.asarray(s.gid, dtype=np.uint64) >> np.uint64(32)).astype(np.int64)
    nmol = int(molid.max()) + 1
    low = np.asarray(s.gid, dtype=np.uint64) & np.uint64(0xFFFFFFFF)
    Lnew = L * np.array([a, b, c])
    rx, ry, rz, gid = [], [], [], []
    k = 0
    # Molecules are made whole first (every bead moved to the periodic image nearest its
    # molecule's first bead): a molecule the original box wraps across its boundary would
    # otherwise be torn between two copies, L apart in a box where L is no longer a period.
    r = np.stack([np.asarray(s.rx, dtype=np.float64), np.asarray(s.ry, dtype=np.float64), np.asarray(s.rz, dtype=np.float64)], 1)
    first = np.zeros(nmol, np.int64)
    first[molid[::-1]] = np.arange(n - 1, -1, -1)            # index of the first bead of each molecule
    ref = r[first[molid]]
    r = r - L * np.rint((r - ref) / L)
    # positions measured from the original box corner, placed into a box centred on 0 again
    fx = r[:, 0] - lo[0]; fy = r[:, 1] - lo[1]; fz = r[:, 2] - lo[2]
    for iz in range(c):
        for iy in range(b):
            for ix in range(a):
                rx.append(fx + ix * L[0] - 0.5 * Lnew[0])
                ry.append(fy + iy * L[1] - 0.5 * Lnew[1])
                rz.append(fz + iz * L[2] - 0.5 * Lnew[2])
                gid.append(((molid + k * nmol).astype(np.uint64) << np.uint64(32)) | low)
                k += 1
    out.rx, out.ry, out.rz = np.concatenate(rx), np.concatenate(ry), np.concatenate(rz)
    out.gid = np.concatenate(gid)
    for f in ("vx", "vy", "vz", "species", "group"):
        setattr(out, f, np.tile(np.asarray(getattr(s, f)), ncopy))
    out.h = np.array(s.h, dtype=np.float64).copy()
    out.h[0], out.h[4], out.h[8] = Lnew
    out.natoms = n * ncopy
    return out


def relabel_types(s, ntypes, seed=SEED):
    """The same system under MORE bead types: every LJ type of the Setup is split into copies with identical parameters until there
    are `ntypes` of them (bioMartini.c:868-950 builds an nspecies^2 table: a real Martini deck has ~40 types where the decks of this
    repository have 2-6), every species gets one copy per LJ-type copy, and every bead draws its copy at random (splitmix64 of
    seed ^ bead index).  Physics -- forces, energies, trajectories -- is that of the original, bead for bead; what changes is
    the size of the (type, charge) class table the pair kernel has to keep."""
    import copy
    out = copy.copy(s)
    nlj0 = int(s.nlj)
    if ntypes < nlj0:
        raise ValueError("relabel_types: %d types asked for, the Setup has %d" % (ntypes, nlj0))
    ncopy = np.full(nlj0, ntypes // nlj0, np.int64)
    ncopy[: ntypes - int(ncopy.sum())] += 1
    first = np.concatenate(([0], np.cumsum(ncopy)))            # new LJ type of copy 0 of each old type
    base = np.repeat(np.arange(nlj0), ncopy)                    # old type of each new type
    sig0, eps0, sh0 = (np.asarray(getattr(s, k), dtype=np.float64).reshape(nlj0, nlj0) for k in ("sigma", "eps", "shift"))
    out.nlj = int(ntypes)
    out.sigma, out.eps, out.shift = (np.ascontiguousarray(a[np.ix_(base, base)]).ravel() for a in (sig0, eps0, sh0))
    # species: copy c of species sp has LJ type first[ljtype[sp]] + c
    lj_sp = np.asarray(s.ljtype, dtype=np.int64)
    kmax = int(ncopy.max())
    nsp0 = int(s.nspecies)
    sp_first = np.concatenate(([0], np.cumsum(ncopy[lj_sp])))   # first new species of each old species
    old_of_new = np.repeat(np.arange(nsp0), ncopy[lj_sp])
    copy_of_new = np.arange(int(sp_first[-1])) - sp_first[old_of_new]
    out.nspecies = int(sp_first[-1])
    for k in ("mass", "charge"):
        setattr(out, k, np.asarray(getattr(s, k), dtype=np.float64)[old_of_new].copy())
    for k in ("moltype", "resitype", "atomoffset"):
        a = np.asarray(getattr(s, k))
        if a.size == nsp0:
            setattr(out, k, a[old_of_new].astype(np.int32))
    out.ljtype = (first[lj_sp[old_of_new]] + copy_of_new).astype(np.int32)
    out.species_name = ["%s" % s.species_name[o] for o in old_of_new] if len(getattr(s, "species_name", [])) == nsp0 else []
    idx = np.arange(int(s.natoms), dtype=np.uint64)
    sp_old = np.asarray(s.species, dtype=np.int64)
    pick = (splitmix64(np.uint64(seed) ^ idx) % np.uint64(kmax)).astype(np.int64) % ncopy[lj_sp[sp_old]]
    out.species = (sp_first[sp_old] + pick).astype(np.int32)
    return out
