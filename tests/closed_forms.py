"""Energies of the Martini terms written straight from the reference's formulas, in numpy, for a handful of beads --
numbers that come neither from the oracle nor from the device code (every other golden value of the
suite is the oracle's).  Forces follow by central differences of these energies.  Conventions cited per function.
This file is test infrastructure only."""
import numpy as np


def bond_E(ri, rj, kb, b0):
    """resBondSorted (bioCharmmCovalentEnergiesSorted.c:55): kb (b - b0)^2"""
    return kb * (np.linalg.norm(ri - rj) - b0) ** 2


def _cos_angle(ri, rj, rk):
    u, w = ri - rj, rk - rj
    return float(np.dot(u, w) / (np.linalg.norm(u) * np.linalg.norm(w)))


def angle_E(ri, rj, rk, func, k, t0):
    """func 1 resAngleSorted (:171): k (theta - theta0)^2; func 2 resAngleCosineSorted (:295): k (cos theta - c0)^2;
    func 10 resAngleRestrainSorted (:417): k (cos theta - c0)^2 / sin^2 theta.  theta at the middle bead."""
    c = _cos_angle(ri, rj, rk)
    if func == 1:
        return k * (np.arccos(np.clip(c, -1.0, 1.0)) - t0) ** 2
    if func == 2:
        return k * (c - t0) ** 2
    if func == 10:
        return k * (c - t0) ** 2 / (1.0 - c * c)
    raise ValueError(func)


def dihedral_angle(ri, rj, rk, rl):
    """the reference's dihedral (bioDihedralFast, bioCharmmCovalentEnergies.c:266-351): built from v_ij = r_i - r_j ...,
    sign from v_jk . ((v_ij x v_jk) x (v_jk x v_kl)) -- "Flip the conventional in Bekker 1995", i.e. MINUS the IUPAC
    angle.  Computed here with atan2 from the bond vectors, no acos, no regularisers."""
    b1, b2, b3 = rj - ri, rk - rj, rl - rk
    n1, n2 = np.cross(b1, b2), np.cross(b2, b3)
    iupac = np.arctan2(np.linalg.norm(b2) * np.dot(b1, n2), np.dot(n1, n2))
    return -float(iupac)


def torsion_E(ri, rj, rk, rl, func, n, k, delta):
    """func 1 resTorsionSorted (:634): kchi (1 + cos(n phi - delta)); func 2 resImproperSorted (:776):
    kpsi (psi - psi0)^2 with the difference wrapped into (-pi, pi]"""
    phi = dihedral_angle(ri, rj, rk, rl)
    if func == 1:
        return k * (1.0 + np.cos(n * phi - delta))
    d = phi - delta
    if d < -np.pi:
        d += 2 * np.pi
    elif d > np.pi:
        d -= 2 * np.pi
    return k * d * d


def lj_rf_pair_E(r, sigma, eps, shift, kq, krf, crf, rcut):
    """martiniNonBond (bioMartini.c:1063-1090): 4 eps ((sigma/r)^12 - (sigma/r)^6) + shift + kq (1/r + krf r^2 - crf), r < rcut"""
    if r >= rcut:
        return 0.0
    s6 = (sigma / r) ** 6
    return 4 * eps * (s6 * s6 - s6) + shift + kq * (1.0 / r + krf * r * r - crf)


def lj_rf_pair_F(r, sigma, eps, kq, krf, rcut):
    """magnitude of the pair force along r_i - r_j (positive = repulsive): -dV/dr"""
    if r >= rcut:
        return 0.0
    s6 = (sigma / r) ** 6
    return 24 * eps * (2 * s6 * s6 - s6) / r + kq * (1.0 / (r * r) - 2 * krf * r)


def molecule_terms_E(s, atoms_xyz, rtype):
    """bond / angle / torsion / improper energies of ONE residue of type rtype whose beads (residue order) sit at
    atoms_xyz [natoms, 3], from the deck's residue-relative term tables"""
    e = {"bond": 0.0, "angle": 0.0, "tors": 0.0, "impr": 0.0}
    x = np.asarray(atoms_xyz, dtype=np.float64)
    for b in range(s.bond_off[rtype], s.bond_off[rtype + 1]):
        e["bond"] += bond_E(x[s.bondI[b]], x[s.bondJ[b]], s.bond_kb[b], s.bond_b0[b])
    for a in range(s.angle_off[rtype], s.angle_off[rtype + 1]):
        e["angle"] += angle_E(x[s.angleI[a]], x[s.angleJ[a]], x[s.angleK[a]], int(s.angle_func[a]), s.angle_k[a], s.angle_t0[a])
    for t in range(s.tors_off[rtype], s.tors_off[rtype + 1]):
        v = torsion_E(x[s.torsI[t]], x[s.torsJ[t]], x[s.torsK[t]], x[s.torsL[t]], int(s.tors_func[t]), int(s.tors_n[t]), s.tors_k[t], s.tors_delta[t])
        e["tors" if int(s.tors_func[t]) == 1 else "impr"] += v
    return e


def fd_forces(energy_of_xyz, xyz, h=1e-5):
    """-dE/dx by central differences"""
    x = np.array(xyz, dtype=np.float64)
    f = np.zeros_like(x)
    for i in range(x.shape[0]):
        for c in range(3):
            keep = x[i, c]
            x[i, c] = keep + h
            ep = energy_of_xyz(x)
            x[i, c] = keep - h
            em = energy_of_xyz(x)
            x[i, c] = keep
            f[i, c] = -(ep - em) / (2 * h)
    return f


def planar_chain(n_at, chord, alpha_deg, trans, centre, tilt=0.0):
    """n_at beads in a plane: all-cis (points on a circle: every dihedral 0) or all-trans (zigzag: every dihedral pi)"""
    a = np.radians(alpha_deg)
    pts = [np.zeros(2)]
    heading = 0.0
    for k in range(1, n_at):
        pts.append(pts[-1] + chord * np.array([np.cos(heading), np.sin(heading)]))
        heading += (a if not trans else (a if k % 2 else -a))
    p = np.array(pts)
    p -= p.mean(axis=0)
    x = np.zeros((n_at, 3))
    x[:, 0], x[:, 1] = p[:, 0], p[:, 1]
    if tilt:
        c, sn = np.cos(tilt), np.sin(tilt)
        x = x @ np.array([[1, 0, 0], [0, c, -sn], [0, sn, c]]).T
    return x + centre
