"""CPU tests: the oracle against the golden vectors and against its independent
checks (O(N^2), finite differences, Newton's third law, NVE), the host deck
loader, units.  No GPU needed."""
import os
import numpy as np
import pytest

import pyoracle
import ddcmd_amd
from ddcmd_amd.deck import units_convert, load_deck
from ddcmd_amd.synth import make_water_setup

REF_DECK = "/root/reference/examples/waterbox/object.data"


def test_units_ddcmd_internal():
    # ddcMD.c:71-73: bohr, fs, Rydberg; kB = 1; ke = e^2/(4 pi eps0) = 2 Ry*bohr
    assert abs(units_convert(1.0, "Angstrom") - 1.0 / 0.52917721067) < 1e-12
    assert abs(units_convert(1.0, "nm") - 10.0 / 0.52917721067) < 1e-11
    assert abs(units_convert(1.0, "Ry") - 1.0) < 1e-14
    assert abs(units_convert(2625.499638, "kJ*mol^-1") - 2.0) < 1e-6     # 1 Hartree
    assert abs(units_convert(1.0, "ps") - 1000.0) < 1e-12
    lib = ddcmd_amd.load_library()
    assert abs(lib.units_ke() - 2.0) < 1e-8
    assert abs(lib.units_kB() - 1.0) < 1e-7
    # temperature unit: Ry/kB kelvin
    assert abs(units_convert(157887.5, "K") - 1.0) < 1e-5
    # "i*t" charge in e, "l" external length = Angstrom
    assert abs(units_convert(1.0, "i*t") - 1.0) < 1e-14
    assert abs(units_convert(1.0, "l") - units_convert(1.0, "Angstrom")) < 1e-15
    # energy = mass * length^2 / time^2 closes in internal units
    m = units_convert(1.0, "amu")
    assert abs(m * units_convert(1.0, "Angstrom") ** 2 / units_convert(1.0, "fs") ** 2 - units_convert(1.0, "amu*Angstrom^2*fs^-2")) < 1e-12


def test_malformed_unit_expressions_return_nan_and_do_not_hang():
    """found by tools/fuzz_decks.py (round 6): "kJ.mol^-1" -- a lone '.' where '*' was meant -- made the expression reader
    spin forever because strtod consumed nothing.  Run in a child with a deadline so a regression is a failure, not a stuck suite."""
    import subprocess, sys, os
    code = ("import math; from ddcmd_amd.deck import units_convert\n"
            "for u in ('kJ.mol^-1', '.', 'kJ*.', 'nm^', 'nm^x', '@', 'kJ*mol^-1*.nm', 'nosuchunit'):\n"
            "    assert math.isnan(units_convert(1.0, u)), u\n"
            "assert abs(units_convert(1.0, '0.5*nm') - 0.5*units_convert(1.0, 'nm')) < 1e-14\n"
            "assert abs(units_convert(1.0, '.5*nm') - 0.5*units_convert(1.0, 'nm')) < 1e-14\n"
            "print('ok')")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout + r.stderr


def test_golden_waterbox_oracle(waterbox):
    """oracle reproduces the committed golden vectors bit-for-bit-ish (same code, same inputs)"""
    s, g = waterbox
    o = pyoracle.Oracle(s)
    npairs = o.build_list()
    assert npairs[0] == int(g["gold_npairs_list"])
    e, vir = o.forces()
    assert np.allclose([e[k] for k in pyoracle.E_NAMES[:7]], g["gold_e"], rtol=1e-13, atol=0)
    assert np.allclose(vir, g["gold_virial"], rtol=1e-12)
    for k, a in (("gold_fx", o.fx), ("gold_fy", o.fy), ("gold_fz", o.fz)):
        assert np.abs(a - g[k]).max() <= 1e-15
    # Newton's third law: total force vanishes
    assert abs(o.fx.sum()) < 1e-13 and abs(o.fy.sum()) < 1e-13 and abs(o.fz.sum()) < 1e-13


def test_oracle_vs_brute_force(waterbox):
    """cell list + half list == O(N^2) with rint minimum image (independent path)"""
    s, g = waterbox
    o = pyoracle.Oracle(s)
    e, vir = o.forces()
    fx, fy, fz, vlj, vele, bvir, nin = o.brute_force()
    assert nin == int(g["gold_npairs_cut"])
    fmax = np.abs(o.fx).max()
    assert max(np.abs(fx - o.fx).max(), np.abs(fy - o.fy).max(), np.abs(fz - o.fz).max()) < 1e-13 * fmax
    assert abs(vlj - e["lj"]) < 1e-12 * abs(vlj)
    assert np.allclose(bvir, vir, rtol=1e-11)


def test_oracle_finite_difference_forces():
    """forcetest.c:10-88 method: central differences of eion against analytic forces"""
    s = make_water_setup(4)          # 256 beads, box 32.5 A: needs rlist < L/2
    s.rmax = units_convert(9.0, "Angstrom")
    s.deltaR = units_convert(2.0, "Angstrom")
    s.shift = ddcmd_amd.synth.lj_shift(s.sigma, s.eps, s.rmax)
    o = pyoracle.Oracle(s)
    o.forces()
    f0 = np.stack([o.fx.copy(), o.fy.copy(), o.fz.copy()])
    rng = np.random.RandomState(1)
    delta = 1e-4
    worst = 0.0
    for i in rng.choice(s.natoms, 10, replace=False):
        for c, arr in enumerate((o.rx, o.ry, o.rz)):
            x0 = arr[i]
            arr[i] = x0 + delta
            o.build_list()
            ep = o.forces()[0]["total"]
            arr[i] = x0 - delta
            o.build_list()
            em = o.forces()[0]["total"]
            arr[i] = x0
            fd = -(ep - em) / (2 * delta)
            worst = max(worst, abs(fd - f0[c, i]) / np.abs(f0).max())
    assert worst < 1e-6


def test_oracle_virial_vs_volume_derivative():
    """testPressure method (masters.c:134-202): tr(virial) = -3V dE/dV under uniform scaling"""
    s = make_water_setup(4)
    s.rmax = units_convert(9.0, "Angstrom")
    s.deltaR = units_convert(2.0, "Angstrom")
    s.shift = 0.0 * s.shift          # un-shifted so E depends on V only through r
    o = pyoracle.Oracle(s)
    e0, vir = o.forces()

    def energy_scaled(lam):
        s2 = make_water_setup(4)
        s2.rmax, s2.deltaR, s2.shift = s.rmax, s.deltaR, s.shift
        s2.h = s.h * lam
        s2.rx, s2.ry, s2.rz = s.rx * lam, s.ry * lam, s.rz * lam
        o2 = pyoracle.Oracle(s2)
        return o2.forces()[0]["total"]
    d = 1e-5
    dEdlam = (energy_scaled(1 + d) - energy_scaled(1 - d)) / (2 * d)
    # E(lam r): dE/dlam = sum_pairs dV/dr * r = -sum f.d = -tr(virial); truncation at rmax
    # adds a surface term for pairs crossing the cutoff, small for d -> 0
    assert abs(dEdlam + (vir[0] + vir[1] + vir[2])) < 2e-3 * abs(vir[0] + vir[1] + vir[2])


def test_oracle_nve_energy_conservation():
    """velocity-Verlet: the energy error is O(dt^2) (shadow Hamiltonian) and does not drift"""
    errs = {}
    for dt, nsteps in ((2.0, 60), (1.0, 120)):
        s = make_water_setup(5, dt_fs=dt)
        s.rmax = units_convert(11.0, "Angstrom")
        s.deltaR = units_convert(4.0, "Angstrom")
        s.shift = ddcmd_amd.synth.lj_shift(s.sigma, s.eps, s.rmax)
        o = pyoracle.Oracle(s)
        e, _ = o.forces()
        rk, _ = o.kinetic()
        e0 = e["total"] + rk
        tr = []
        for _ in range(nsteps // 20):
            e, _, rk, _ = o.step(20)
            tr.append(e["total"] + rk - e0)
        errs[dt] = tr
        assert abs(tr[-1] - tr[-2]) < 0.1 * abs(tr[-1])        # plateau, no drift
        assert abs(tr[-1]) < 2e-3 * rk
    ratio = errs[2.0][-1] / errs[1.0][-1]
    assert 3.0 < ratio < 5.0


def test_synth_generator_is_deterministic():
    a, b = make_water_setup(3), make_water_setup(3)
    assert np.array_equal(a.rx, b.rx) and np.array_equal(a.vz, b.vz) and np.array_equal(a.species, b.species)
    assert a.natoms == 108
    assert abs(np.sum(a.mass[a.species] * a.vx)) < 1e-12
    frac = a.species.mean()
    big = make_water_setup(13)
    assert 0.08 < big.species.mean() < 0.12
    # initial kinetic temperature = the requested 50 K (kB = 1)
    T = np.sum(big.mass[big.species] * (big.vx ** 2 + big.vy ** 2 + big.vz ** 2)) / (3 * big.natoms)
    assert abs(T / units_convert(50.0, "K") - 1) < 0.03
    # gid convention of the deck: one bead per molecule, gid = i << 32
    assert int(big.gid[3]) == 3 << 32


@pytest.mark.skipif(not os.path.exists(REF_DECK), reason="reference deck only exists in the build container")
def test_deck_loader_on_reference_waterbox(waterbox):
    """host C loader on the real deck == the committed fixture"""
    s0, g = waterbox
    extra = "nglf INTEGRATOR {type = NGLF;}\n group GROUP { type = FREE; }\n free GROUP { type = FREE; }\n"
    s = load_deck(REF_DECK, None, extra)
    assert s.natoms == 6173 and s.nspecies == 2 and s.species_name == ["WxW", "WFxWF"]
    assert np.array_equal(s.gid, s0.gid) and np.array_equal(s.species, s0.species)
    assert np.array_equal(s.rx, s0.rx) and np.allclose(s.sigma, s0.sigma, rtol=0, atol=0)
    assert s.integrator_type == "NGLF" and s.updateRate == 20 and s.maxloop == 10
    assert abs(s.rmax - units_convert(11.0, "Angstrom")) < 1e-13
    assert abs(s.h[0] - units_convert(93.858, "Angstrom")) < 1e-12
    assert abs(s.dt - 20.0) < 1e-15
    # the shipped selection (no override) is NGLFCONSTRAINT + LANGEVIN groups
    raw = load_deck(REF_DECK)
    assert raw.integrator_type == "NGLFCONSTRAINT" and list(raw.group_type) == [2, 2]


LIPID_DECK = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lipid_deck", "object.data")


def test_lipid_deck_loader():
    """host C loader on our own ddcMD-format lipid deck: MMFF tree, exclusion lists, units"""
    s = load_deck(LIPID_DECK)
    assert s.natoms == 2363 and s.nspecies == 19 and s.nlj == 6 and s.nmoltype == 4 and s.nresi == 4
    assert list(s.mol_nspecies) == [1, 1, 12, 5]
    # genMartiniBondPair: func-1 bonds + valid exclusions (bioMartini.c:135-282)
    assert list(s.bpair_off) == [0, 0, 0, 11, 16]
    assert (int(s.bpairI[15]), int(s.bpairJ[15])) == (0, 2)
    assert list(s.resi_natoms) == [1, 1, 12, 5]
    assert list(s.bond_off) == [0, 0, 0, 11, 15] and list(s.angle_off) == [0, 0, 0, 8, 11] and list(s.tors_off) == [0, 0, 0, 0, 3]
    assert sorted(set(s.angle_func.tolist())) == [1, 2, 10] and sorted(set(s.tors_func.tolist())) == [1, 2]
    assert abs(s.bond_kb[0] - units_convert(1250.0, "kJ*mol^-1*nm^-2")) < 1e-15
    assert abs(s.bond_b0[2] - units_convert(0.37, "nm")) < 1e-13
    assert abs(s.charge[s.species_name.index("DPPCxNC3")] - 1.0) < 1e-15
    assert abs(s.charge[s.species_name.index("DPPCxPO4")] + 1.0) < 1e-15
    assert s.species_name[s.species[0]] == "DPPCxNC3" and int(s.gid[13]) == (1 << 32) | 1
    # every species maps to its residue/atom slot (getCGLJindexbySpecie)
    assert s.ljtype[s.species_name.index("DPPCxC3B")] == 5 and s.atomoffset[s.species_name.index("DPPCxC3B")] == 10
    assert s.updateRate == 10 and abs(s.dt - 10.0) < 1e-15


def test_lipid_oracle_vs_brute_force_and_third_law():
    """nonbonded part incl. excluded-pair reaction field vs the O(N^2) path; bonded forces sum to zero"""
    s = load_deck(LIPID_DECK)
    o = pyoracle.Oracle(s)
    npairs = o.build_list()
    assert npairs[1] == 120 * 11 + 8 * 5          # every bonded/excluded pair sits in list 1 (reOrgPairs)
    e, vir = o.forces()
    fx, fy, fz, vlj, vele, bvir, nin = o.brute_force()
    bfx, bfy, bfz, e4, bondvir = o.bonded_only()
    fmax = np.abs(o.fx).max()
    assert np.abs((o.fx - bfx) - fx).max() < 1e-12 * fmax
    assert abs(vlj - e["lj"]) < 1e-12 * abs(vlj) and abs(vele - e["ele"]) < 1e-11 * abs(vele)
    assert np.allclose(bvir + bondvir, vir, rtol=1e-10, atol=1e-12)
    for f in (bfx, bfy, bfz):
        assert abs(f.sum()) < 1e-12 * np.abs(f).max() * 50
    assert abs(e["total"] - (e["lj"] + e["ele"] + e4.sum())) < 1e-12 * abs(e["total"])


def test_lipid_oracle_finite_difference_all_terms():
    """forcetest.c method on the full potential: bonds, 3 angle kinds, torsion, improper, LJ, RF Coulomb"""
    s = load_deck(LIPID_DECK)
    o = pyoracle.Oracle(s)
    o.forces()
    f0 = np.stack([o.fx.copy(), o.fy.copy(), o.fz.copy()])
    # pick beads from the test molecules (all term kinds) and a few lipid beads
    tst = np.flatnonzero(np.array([s.species_name[k].startswith("TSTM") for k in s.species]))[:10]
    lip = np.array([0, 1, 2, 3, 7, 11])
    delta = 1e-4
    worst = 0.0
    for i in np.concatenate((tst, lip)):
        for c, arr in enumerate((o.rx, o.ry, o.rz)):
            x0 = arr[i]
            arr[i] = x0 + delta
            o.build_list()
            ep = o.forces()[0]["total"]
            arr[i] = x0 - delta
            o.build_list()
            em = o.forces()[0]["total"]
            arr[i] = x0
            fd = -(ep - em) / (2 * delta)
            worst = max(worst, abs(fd - f0[c, i]) / np.abs(f0).max())
    assert worst < 2e-7


def test_atoms_reader_follows_field_names(tmp_path):
    """restart files written by collection_writeBLOCK lead with a checksum column and are fixed-record:
    the reader takes the columns from the header's field_names (collection_write.c:57-186 layout)"""
    import os
    from ddcmd_amd.deck import load_deck, units_convert
    deck = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lipid_deck")
    s0 = load_deck(os.path.join(deck, "object.data"))
    A = units_convert(1.0, None, "Angstrom")
    snap = tmp_path / "snapshot.000000000040"
    snap.mkdir()
    with open(snap / "atoms#000000", "w") as f:
        f.write("particle FILEHEADER {type=MULTILINE; datatype=FIXRECORDASCII; checksum=CRC32; create_time=x; run_id=0x00000000;\n"
                "loop=40; time=400.000000 fs;\nnfiles=1; nrecord=%d; lrec=232; nfields=11; endian_key=875770417;\n"
                "field_names=checksum id class type group rx ry rz vx vy vz;\nfield_types=u u s s s f f f f f f;\n"
                "field_units=1 1 1 1 1 Ang Ang Ang Ang/fs Ang/fs Ang/fs;\n"
                "h=%f 0 0\n  0 %f 0\n  0 0 %f Ang;\nrandom = NONE;\nrandomFieldSize = 0;\ngroups = group;\ntypes = ATOM;\n}\n\n"
                % (s0.natoms, s0.h[0] * A, s0.h[4] * A, s0.h[8] * A))
        import ctypes
        lib = ddcmd_amd.load_library()
        lib.ddcmi_crc32.restype = ctypes.c_uint32
        lib.ddcmi_crc32.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
        for i in range(s0.natoms):
            body = (" %12.12d ATOM %s group %21.13e %21.13e %21.13e %21.13e %21.13e %21.13e" % (
                int(s0.gid[i]), s0.species_name[int(s0.species[i])], s0.rx[i] * A, s0.ry[i] * A, s0.rz[i] * A,
                s0.vx[i] * A, s0.vy[i] * A, s0.vz[i] * A)).ljust(223) + "\n"
            f.write("%08x" % lib.ddcmi_crc32(body.encode(), len(body)) + body)      # the reader verifies the record checksums
    with open(tmp_path / "restart", "w") as f:
        f.write("simulate SIMULATE { run_id=0x0; loop=40; time=400.000000 fs;}\nbox BOX {\n h  = %.14e 0 0\n 0 %.14e 0\n 0 0 %.14e;\n}\n"
                "collection COLLECTION { size=%d; files=snapshot.000000000040/atoms#;}\n" % (s0.h[0] * A, s0.h[4] * A, s0.h[8] * A, s0.natoms))
    s = load_deck(os.path.join(deck, "object.data"), restart_file=str(tmp_path / "restart"))
    assert s.natoms == s0.natoms and s.loop == 40
    assert np.array_equal(s.gid, s0.gid) and np.array_equal(s.species, s0.species)
    for a, b in ((s.rx, s0.rx), (s.ry, s0.ry), (s.rz, s0.rz), (s.vx, s0.vx), (s.vz, s0.vz)):
        assert np.abs(a - b).max() <= 1e-12 * max(np.abs(b).max(), 1e-300)


RESTRAINT_X = ("system SYSTEM { potential = martini restraintPot; } "
               "restraintPot POTENTIAL { type = RESTRAINT; parmfile = restraint.data; }")


def test_restraint_potential_deck_and_finite_differences():
    """POTENTIAL type=RESTRAINT (restraint.c:259-361): deck loading (RESTRAINTLIST / RESTRAINTPARMS, kb units,
    fractional x0) and the oracle's restatement checked by finite differences of the restraint energy"""
    import os
    from ddcmd_amd.deck import load_deck, units_convert
    deck = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lipid_deck", "object.data")
    s = load_deck(deck, extra_objects=RESTRAINT_X)
    assert s.nrest == 6 and s.rest_origin == 0
    assert list(s.rest_fc[0]) == [0, 0, 1] and list(s.rest_fc[4]) == [1, 1, 1]
    assert abs(s.rest_kb[0] - units_convert(200.0, "kJ*mol^-1*nm^-2")) < 1e-15
    assert abs(s.rest_r0[3][1] - 0.98) < 1e-15
    o = pyoracle.Oracle(s)
    e, vir = o.forces()
    s0 = load_deck(deck)
    o0 = pyoracle.Oracle(s0)
    e0, vir0 = o0.forces()
    assert e["restraint"] > 0 and abs(e["total"] - e0["total"] - e["restraint"]) < 1e-12 * abs(e["total"])
    # the restraint force = difference to the unrestrained run; check it against -dE_restraint/dr
    h = 1e-5
    for r in range(s.nrest):
        i = int(np.flatnonzero(np.asarray(s.gid) == s.rest_gid[r])[0])
        for c, (arr, f, f0) in enumerate(((o.rx, o.fx, o0.fx), (o.ry, o.fy, o0.fy), (o.rz, o.fz, o0.fz))):
            keep = arr[i]
            arr[i] = keep + h
            ep = o.forces()[0]["restraint"]
            arr[i] = keep - h
            em = o.forces()[0]["restraint"]
            arr[i] = keep
            fd = -(ep - em) / (2 * h)
            o.forces()
            assert abs((f[i] - f0[i]) - fd) < 1e-7 * max(abs(fd), 1e-6), (r, c)


# constraint lists added to the lipid deck's TSTM (a triangle 0-1-2 and the pair 3-4) and DPPC (the glycerol pair)
# residues; the parmfile's own keys stay, the new constraintList key is added to them
CONSTRAINT_X = ("TSTM RESIPARMS { constraintList = TSTM_cl0 TSTM_cl1; } "
                "TSTM_cl0 CONSLISTPARMS { constraintSubList = TSTM_c0 TSTM_c1 TSTM_c2; } "
                "TSTM_cl1 CONSLISTPARMS { constraintSubList = TSTM_c3; } "
                "TSTM_c0 CONSPARMS { atomI=0; atomJ=1; func=1; r0=0.40 nm; } "
                "TSTM_c1 CONSPARMS { atomI=1; atomJ=2; func=1; r0=0.40 nm; } "
                "TSTM_c2 CONSPARMS { atomI=0; atomJ=2; func=1; r0=0.655 nm; } "
                "TSTM_c3 CONSPARMS { atomI=3; atomJ=4; func=1; r0=0.40 nm; } "
                "DPPC RESIPARMS { constraintList = DPPC_cl0; } "
                "DPPC_cl0 CONSLISTPARMS { constraintSubList = DPPC_c0; } "
                "DPPC_c0 CONSPARMS { atomI=2; atomJ=3; func=1; r0=0.37 nm; } ")


def _constraint_lengths(s, rx, ry, rz, box):
    from ddcmd_amd.martini import expand_constraints
    po, pi, pj, dd = expand_constraints(s)
    d = np.stack((rx[pi] - rx[pj], ry[pi] - ry[pj], rz[pi] - rz[pj]), axis=1)
    d -= box * np.rint(d / box)
    return np.sqrt((d * d).sum(axis=1)), dd, (po, pi, pj)


def test_velocity_constraints_deck_and_solver():
    """nglfconstraint's velocity constraints (nglfconstraint.c:180-264): the deck's CONSLISTPARMS/CONSPARMS
    become one group per list and residue instance; after a step every constrained pair has its length r0
    (FRONT: |r + dt v| = d) and no relative velocity along the pair (BACK: r.v = 0)"""
    import os
    from ddcmd_amd.deck import load_deck, units_convert
    from ddcmd_amd.martini import expand_constraints
    deck = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lipid_deck")
    # the 310 K restart: bond lengths there are near the constrained lengths (the lattice start is not)
    s = load_deck(os.path.join(deck, "object_nvt.data"), restart_file=os.path.join(deck, "relaxed", "restart"), extra_objects=CONSTRAINT_X)
    s0 = load_deck(os.path.join(deck, "object_nvt.data"), restart_file=os.path.join(deck, "relaxed", "restart"))
    nd = int(np.sum(s.resitype[s.species] == list(s.resi_natoms).index(12)) // 12)
    nt = int(np.sum(s.resi_natoms[s.resitype[s.species]] == 5) // 5)
    assert nd > 0 and nt > 0
    assert s.nresicons == 5 and int(s.cons_off[-1]) == 5
    assert abs(s.cons_r0.max() - units_convert(0.655, "nm")) < 1e-12
    po, pi, pj, dd = expand_constraints(s)
    assert po.size - 1 == 2 * nt + nd and pi.size == 4 * nt + nd
    # constrained pairs that are not func-1 bonds join the exclusion (bpair) lists (genMartiniBondPair)
    assert s.bpair_off[-1] == s0.bpair_off[-1]          # 0-2 is already an exclusion, the rest are bonds
    o = pyoracle.Oracle(s, constraints=True)
    o.forces()
    box = s.box
    for step in range(3):
        o.step(1)
        L, d0, (po, pi, pj) = _constraint_lengths(s, o.rx, o.ry, o.rz, box)
        assert np.abs(L / d0 - 1.0).max() < 1e-10, step
        d = np.stack((o.rx[pi] - o.rx[pj], o.ry[pi] - o.ry[pj], o.rz[pi] - o.rz[pj]), axis=1)
        d -= box * np.rint(d / box)
        w = np.stack((o.vx[pi] - o.vx[pj], o.vy[pi] - o.vy[pj], o.vz[pi] - o.vz[pj]), axis=1)
        assert np.abs((d * w).sum(axis=1) * s.dt / d0 ** 2).max() < 1e-10, step
    # unconstrained beads are untouched by a sweep; a second BACK sweep is a no-op (one Gauss-Seidel pass)
    v0 = o.vx.copy()
    assert o.constraint_sweep(1) == 1
    assert np.abs(o.vx - v0).max() < 1e-10 * np.abs(v0).max()


def test_molecule_lists_of_the_lipid_deck():
    """molecule_lists: molecules = runs of equal gid >> 32; the barostat's molecular virial runs over those of two or
    more beads, N kB T counts all of them (molecularPressure.c:57-67)"""
    import os
    from ddcmd_amd.deck import load_deck
    from ddcmd_amd.martini import molecule_lists
    deck = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lipid_deck", "object.data")
    s = load_deck(deck)
    nmol, off, atoms = molecule_lists(s)
    assert nmol == 120 + 8 + 883
    assert off.size - 1 == 128 and atoms.size == 120 * 12 + 8 * 5
    sizes = np.diff(off)
    assert sorted(set(sizes.tolist())) == [5, 12]
    for m in (0, 57, 127):          # every listed molecule is one gid >> 32 group, complete
        g = np.asarray(s.gid)[atoms[off[m]:off[m + 1]]] >> np.uint64(32)
        assert np.all(g == g[0]) and int(np.sum((np.asarray(s.gid) >> np.uint64(32)) == g[0])) == sizes[m]


def test_crc32_known_answer_and_corrupted_snapshots(tmp_path):
    """the record checksum is the reference's checksum_crc32 (crc32.c:46-84): its own check string "123456789" gives
    0xcbf43926.  A snapshot with a flipped byte, or fewer records than its header announces, is refused
    (collection_read.c:274-286 verifies on read)"""
    import ctypes
    lib = ddcmd_amd.load_library()
    lib.ddcmi_crc32.restype = ctypes.c_uint32
    lib.ddcmi_crc32.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
    assert lib.ddcmi_crc32(b"123456789", 9) == 0xCBF43926
    assert lib.ddcmi_crc32(b"", 0) == 0
    deck = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lipid_deck")
    s0 = load_deck(os.path.join(deck, "object.data"))
    A = units_convert(1.0, None, "Angstrom")
    lrec = 232

    def write(snapdir, nrecord_header, flip=None, drop=0):
        os.makedirs(snapdir, exist_ok=True)
        with open(os.path.join(snapdir, "atoms#000000"), "wb") as f:
            f.write(("particle FILEHEADER {type=MULTILINE; datatype=FIXRECORDASCII; checksum=CRC32;\nloop=0; time=0.000000 fs;\n"
                     "nfiles=1; nrecord=%d; lrec=%d; nfields=11; endian_key=875770417;\n"
                     "field_names=checksum id class type group rx ry rz vx vy vz;\nfield_types=u u s s s f f f f f f;\n"
                     "h=%f 0 0\n  0 %f 0\n  0 0 %f Ang;\ngroups = group;\ntypes = ATOM;\n}\n\n"
                     % (nrecord_header, lrec, s0.h[0] * A, s0.h[4] * A, s0.h[8] * A)).encode())
            for i in range(s0.natoms - drop):
                body = (" %12.12d ATOM %s group %21.13e %21.13e %21.13e %21.13e %21.13e %21.13e" % (
                    int(s0.gid[i]), s0.species_name[int(s0.species[i])], s0.rx[i] * A, s0.ry[i] * A, s0.rz[i] * A,
                    s0.vx[i] * A, s0.vy[i] * A, s0.vz[i] * A)).ljust(lrec - 9).encode() + b"\n"
                crc = lib.ddcmi_crc32(body, len(body))
                rec = bytearray(("%08x" % crc).encode() + body)
                if flip is not None and i == flip:
                    rec[60] = ord("7") if rec[60] != ord("7") else ord("3")
                f.write(bytes(rec))
        with open(os.path.join(os.path.dirname(snapdir), "restart"), "w") as f:
            f.write("simulate SIMULATE { run_id=0x0; loop=0; time=0.000000 fs;}\nbox BOX {\n h  = %.14e 0 0\n 0 %.14e 0\n 0 0 %.14e;\n}\n"
                    "collection COLLECTION { size=%d; files=%s/atoms#;}\n" % (s0.h[0] * A, s0.h[4] * A, s0.h[8] * A, s0.natoms, snapdir))
        return os.path.join(os.path.dirname(snapdir), "restart")
    good = write(str(tmp_path / "a" / "snap"), s0.natoms)
    s = load_deck(os.path.join(deck, "object.data"), restart_file=good)
    assert s.natoms == s0.natoms and np.abs(s.rx - s0.rx).max() < 1e-10
    bad = write(str(tmp_path / "b" / "snap"), s0.natoms, flip=17)
    with pytest.raises(Exception) as ei:
        load_deck(os.path.join(deck, "object.data"), restart_file=bad)
    assert "CRC32" in str(ei.value)
    short = write(str(tmp_path / "c" / "snap"), s0.natoms, drop=5)
    with pytest.raises(Exception) as ei:
        load_deck(os.path.join(deck, "object.data"), restart_file=short)
    assert "nrecord" in str(ei.value)


def test_decks_the_mutation_fuzz_broke_the_loader_with_are_refused_with_a_message(tmp_path):
    """tools/fuzz_decks.py (round 6, host layer under ASan/UBSan) found three decks the loader did not survive: a key that is present and
    empty ("type = ;": strcmp on NULL), a MOLECULE that names no species (its ownership species indexed past the species tables), and a unit
    written with '.' for '*' (the expression reader never advanced).  Each is a message now; each runs in a child with a deadline."""
    import shutil, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    deck = os.path.join(root, "tests", "golden", "lipid_deck")
    cases = {
        "empty_type": ("object.data", "group GROUP { type = FREE; }", "group GROUP { type = ; }", None),      # loads: a group of no known type is GROUP_OTHER
        "no_species": ("object.data", "Wx MOLECULE {ownershipSpecies = WxW; species = WxW;}", "Wx MOLECULE {ownershipSpecies = WxW; species = ;}", "names no species"),
        "dot_unit": ("restart", None, None, "units: cannot parse"),      # (it used to spin; then it loaded with time = NaN; now the deck is refused)
    }
    for name, (victim, old, new, msg) in cases.items():
        work = str(tmp_path / name)
        shutil.copytree(deck, work)
        path = os.path.join(work, victim)
        text = open(path).read()
        if name == "dot_unit":
            text2 = text.replace("Angstrom", "kJ.mol^-1", 1) if "Angstrom" in text else text.replace(";", " kJ.mol^-1;", 2)
        else:
            assert old in text, name
            text2 = text.replace(old, new, 1)
        assert text2 != text
        open(path, "w").write(text2)
        code = ("from ddcmd_amd.deck import load_deck\n"
                "try:\n    s = load_deck(%r); print('LOADED', s.natoms)\n"
                "except RuntimeError as e:\n    print('REFUSED', e)\n" % os.path.join(work, "object.data"))
        r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, (name, r.stdout, r.stderr)
        out = r.stdout.strip()
        if msg:
            assert out.startswith("REFUSED") and msg in out, (name, out)
        else:
            assert out.startswith("REFUSED") or out.startswith("LOADED"), (name, out)


def _is_prime(n):
    if n < 2 or n % 2 == 0:
        return n == 2
    i = 3
    while i * i <= n:
        if n % i == 0:
            return False
        i += 2
    return True


def test_lcg64_streams_known_answers():
    """RANDOM type LCG64 as the oracle restates it (lcg64.c:127-146, random.c:135-160, primes.c:35-155, collection.c:95-109),
    against arithmetic done here with Python integers: the recurrence, the polar normals, the default states."""
    L = pyoracle.lib()
    MULT = [0x27bb2ee687b0b0fd, 0x2c6fe96ee78b6955, 0x369dea0f31a53f85]
    for mult_id, prime, state in ((0, 2147453653, 0x2bc6ffff8cfe166d), (1, 2147483659, 1), (2, 4294967291, 0xffffffffffffffff)):
        q = np.zeros(1, dtype=pyoracle.LCG64)
        q["state"], q["multID"], q["prime"] = state, mult_id, prime
        st = state
        for _ in range(50):
            u = L.orc_lcg64(q.ctypes.data)
            st = (MULT[mult_id] * st + prime) % (1 << 64)
            assert int(q["state"][0]) == st
            assert u == float(st) * 2.0 ** -64 and 0.0 <= u <= 1.0
        # gasdev3d: two polar draws over pairs of uniforms; x, y of the first accepted pair and x of the second
        ref = []
        for _ in range(200):
            g = []
            for draw in range(2):
                while True:
                    st = (MULT[mult_id] * st + prime) % (1 << 64); x = 2.0 * (float(st) * 2.0 ** -64) - 1.0
                    st = (MULT[mult_id] * st + prime) % (1 << 64); y = 2.0 * (float(st) * 2.0 ** -64) - 1.0
                    rsq = x * x + y * y
                    if 0.0 < rsq < 1.0:
                        break
                fac = np.sqrt(-2.0 * np.log(rsq) / rsq)
                g += [x * fac, y * fac] if draw == 0 else [x * fac]
            ref.append(g)
        got = np.array([pyoracle.gasdev3d(q) for _ in range(200)])
        assert int(q["state"][0]) == st                      # the same number of uniforms was consumed
        assert np.abs(got - np.array(ref)).max() < 1e-14
    # unit normals: moments over 3 x 100 000 draws
    q = np.zeros(1, dtype=pyoracle.LCG64)
    q["state"], q["multID"], q["prime"] = 0x2bc6ffff8cfe166d ^ (77 << 32), 1, 2147453657
    g = np.array([pyoracle.gasdev3d(q) for _ in range(100000)])
    assert np.abs(g.mean(axis=0)).max() < 0.02 and np.abs(g.var(axis=0) - 1.0).max() < 0.02
    assert abs(np.mean(g ** 4) - 3.0) < 0.1 and np.abs(np.corrcoef(g.T) - np.eye(3)).max() < 0.02
    # default states (file order on one task): INIT_SEED ^ label, multID 0,1,2,..., one prime per three particles --
    # the odd primes of the task's blocks of 30000 numbers, the first block ending at 2^31 + 1
    labels = (np.arange(1, 601, dtype=np.uint64) << np.uint64(32)) | np.uint64(5)
    for task, ntasks in ((0, 1), (3, 8)):
        d = pyoracle.lcg64_default(labels, task, ntasks)
        assert (d["state"] == (np.uint64(0x2bc6ffff8cfe166d) ^ labels)).all()
        assert (d["multID"] == np.arange(600) % 3).all()
        assert (d["prime"][0::3] == d["prime"][1::3]).all() and (d["prime"][0::3] == d["prime"][2::3]).all()
        want, block = [], 0
        while len(want) < 200:
            hi = (block * ntasks + task) * 30000 + (1 << 31) + 1
            lo = hi - 30000
            hi -= (hi % 2 == 0)
            lo += (lo % 2 == 0)
            want += [x for x in range(lo, hi, 2) if _is_prime(x)]
            block += 1
        assert [int(x) for x in d["prime"][0::3]] == want[:200]
    # isPrime1 itself on Carmichael numbers, strong pseudoprimes to base 2 and prime squares
    for n, expect in ((561, 0), (2047, 0), (3215031751, 0), (4294967291, 1), (4294967297, 0), (2147483647, 1), (46349 * 46349, 0), (2147483659, 1)):
        assert L.orc_is_prime1(n) == expect, n


def test_atoms_random_field_and_default_streams(tmp_path):
    """RANDOM type=LCG64 in the deck (system.c:135, random.c:44-71): the atoms reader takes "state multID prime" behind the
    velocities of every record (collection_read.c:160-166, lcg64_parse lcg64.c:89-95); a file written with random = NONE, or a
    single record without the field, puts every particle on lcg64_default's values (collection.c:95-109) -- which the
    loader (Miller-Rabin) and the oracle (the reference's own primality test, restated) derive independently"""
    import os
    import ctypes
    from ddcmd_amd.deck import load_deck, units_convert
    deck = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lipid_deck")
    s0 = load_deck(os.path.join(deck, "object.data"))
    assert s0.lcg64 is not None and s0.lcg_from_file == 0 and len(s0.lcg64) == s0.natoms
    want = pyoracle.lcg64_default(s0.gid)
    assert (s0.lcg64 == want).all()
    # the 6173-bead water deck: 2058 primes further up the same sequence
    sw = load_deck(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "water_deck", "object.data"))
    assert (sw.lcg64 == pyoracle.lcg64_default(sw.gid)).all() and len(set(sw.lcg64["prime"].tolist())) == (sw.natoms + 2) // 3
    # a deck without a RANDOM object carries no streams
    s_none = load_deck(os.path.join(deck, "object.data"), extra_objects="system SYSTEM { random = NONE; }")
    assert s_none.lcg64 is None
    A = units_convert(1.0, None, "Angstrom")
    lib = ddcmd_amd.load_library()
    lib.ddcmi_crc32.restype = ctypes.c_uint32
    lib.ddcmi_crc32.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
    rng = np.random.default_rng(5)
    mine = np.zeros(s0.natoms, dtype=pyoracle.LCG64)
    mine["state"] = rng.integers(1, 2 ** 63, s0.natoms, dtype=np.uint64) * 2 + 1
    mine["multID"] = rng.integers(0, 3, s0.natoms)
    mine["prime"] = want["prime"][::-1]

    def write(dirname, header_random, skip_record=None):
        snap = tmp_path / dirname / "snapshot.000000000040"
        snap.mkdir(parents=True)
        with open(snap / "atoms#000000", "w") as f:
            f.write("particle FILEHEADER {type=MULTILINE; datatype=FIXRECORDASCII; checksum=CRC32; create_time=x; run_id=0x00000000;\n"
                    "loop=40; time=400.000000 fs;\nnfiles=1; nrecord=%d; lrec=264; nfields=11; endian_key=875770417;\n"
                    "field_names=checksum id class type group rx ry rz vx vy vz;\nfield_types=u u s s s f f f f f f;\n"
                    "field_units=1 1 1 1 1 Ang Ang Ang Ang/fs Ang/fs Ang/fs;\n"
                    "h=%f 0 0\n  0 %f 0\n  0 0 %f Ang;\n%sgroups = group;\ntypes = ATOM;\n}\n\n"
                    % (s0.natoms, s0.h[0] * A, s0.h[4] * A, s0.h[8] * A, header_random))
            for i in range(s0.natoms):
                body = " %12.12d ATOM %s group %21.13e %21.13e %21.13e %21.13e %21.13e %21.13e" % (
                    int(s0.gid[i]), s0.species_name[int(s0.species[i])], s0.rx[i] * A, s0.ry[i] * A, s0.rz[i] * A,
                    s0.vx[i] * A, s0.vy[i] * A, s0.vz[i] * A)
                if i != skip_record:
                    body += " %16.16x %1u %8.8x" % (int(mine["state"][i]), int(mine["multID"][i]), int(mine["prime"][i]))      # lcg64_write
                body = body.ljust(255) + "\n"
                f.write("%08x" % lib.ddcmi_crc32(body.encode(), len(body)) + body)
        with open(tmp_path / dirname / "restart", "w") as f:
            f.write("simulate SIMULATE { run_id=0x0; loop=40; time=400.000000 fs;}\nbox BOX {\n h  = %.14e 0 0\n 0 %.14e 0\n 0 0 %.14e;\n}\n"
                    "collection COLLECTION { size=%d; files=%s/atoms#;}\n" % (s0.h[0] * A, s0.h[4] * A, s0.h[8] * A, s0.natoms, snap))
        return load_deck(os.path.join(deck, "object.data"), restart_file=str(tmp_path / dirname / "restart"))

    s1 = write("a", "random = lcg64;\nrandomFieldSize = 27;\n")
    assert s1.lcg_from_file == 1 and (s1.lcg64 == mine).all()
    s2 = write("b", "")                                      # no `random` key in the header: the records are tried (collection_read.c:102-104)
    assert s2.lcg_from_file == 1 and (s2.lcg64 == mine).all()
    s3 = write("c", "random = NONE;\nrandomFieldSize = 0;\n")      # the header says there is none: defaults, whatever follows vz
    assert s3.lcg_from_file == 0 and (s3.lcg64 == want).all()
    s4 = write("d", "random = lcg64;\nrandomFieldSize = 27;\n", skip_record=17)      # one record without the field: defaults for all
    assert s4.lcg_from_file == 0 and (s4.lcg64 == want).all()


def test_langevin_drift_velocity_closed_form():
    """langevin.c:106-118 with `vcm`: on force-free beads without noise (Teq = 0) the FRONT and BACK half updates of one step are
    v <- vcm + a^2 (v - vcm), a = exp(-dt/(2 tau)): the oracle's restatement against that closed form, and the deck loader reads the
    key with its units and refuses a Teq that is an equation of time"""
    import ddcmd_amd
    from ddcmd_amd.deck import units_convert, load_deck
    s = ddcmd_amd.make_water_setup(4)
    s.group_type = np.array([2], np.int32)
    s.group_Teq = np.array([0.0])
    tau = units_convert(0.2, "ps")
    s.group_tau = np.array([tau])
    vc = np.array([1.0e-4, 2.0e-4, -3.0e-4])
    s.group_vcm = vc.copy()
    s.excludePotentialTerm = 128 | 1 | 2 | 4 | 8        # no forces at all
    o = pyoracle.Oracle(s)
    o.forces()
    v0 = np.stack([o.vx.copy(), o.vy.copy(), o.vz.copy()])
    o.step(7)
    a2 = np.exp(-float(s.dt) / tau)
    want = vc[:, None] + a2 ** 7 * (v0 - vc[:, None])
    got = np.stack([o.vx, o.vy, o.vz])
    assert np.abs(got - want).max() < 1e-13 * np.abs(want).max()
    deck = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "water_deck", "object.data")
    x = "group GROUP { type = LANGEVIN; Teq = 310 K; tau = 1 ps; vcm = 0.001 0 -0.002 Angstrom/fs; }"
    d = load_deck(deck, extra_objects=x)
    g = list(d.group_name).index("group")
    assert np.allclose(d.group_vcm[3 * g:3 * g + 3], np.array([0.001, 0.0, -0.002]) * units_convert(1.0, "Angstrom/fs"), rtol=1e-12)
    with pytest.raises(Exception, match="not a constant temperature"):
        load_deck(deck, extra_objects="group GROUP { type = LANGEVIN; Teq = 300+10*t; tau = 1 ps; }")
    # tails made of unit characters only (`300-2*t`, `300*t`, `300 t`) passed as "a number with a unit"
    for eq in ("300-2*t", "300*t", "300 t", "300 K*t", "2*K"):
        with pytest.raises(Exception, match="not a constant temperature"):
            load_deck(deck, extra_objects="group GROUP { type = LANGEVIN; Teq = %s; tau = 1 ps; }" % eq)
    for const in ("310 K", "310", "0.001 eV/kB" if False else "310.5 K"):
        d2 = load_deck(deck, extra_objects="group GROUP { type = LANGEVIN; Teq = %s; tau = 1 ps; }" % const)
        assert d2.group_Teq[list(d2.group_name).index("group")] > 0.0
    # langevin.c:71-79: Teq_dynamics = GLOBAL_ENERGY is not a constant temperature either
    with pytest.raises(Exception, match="Teq_dynamics"):
        load_deck(deck, extra_objects="group GROUP { type = LANGEVIN; Teq = 310 K; tau = 1 ps; Teq_dynamics = GLOBAL_ENERGY; }")
    load_deck(deck, extra_objects="group GROUP { type = LANGEVIN; Teq = 310 K; tau = 1 ps; Teq_dynamics = EXPLICIT_TIME; }")


def test_relabelled_bead_types_leave_the_physics_alone():
    """ddcmd_amd.synth.relabel_types (the workload of the type-count bench rows and of the 40-type GPU test): every LJ type split into
    copies of itself, every species copied with it, every bead's copy drawn at random -- forces, energies, virial and a short
    trajectory of the oracle are those of the original deck in every bit, while the (type, charge) class table grows from 6 to 48"""
    from ddcmd_amd.deck import load_deck
    from ddcmd_amd.synth import relabel_types
    deck = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lipid_deck")
    s = load_deck(os.path.join(deck, "object_nvt.data"), restart_file=os.path.join(deck, "relaxed", "restart"))
    s2 = relabel_types(s, 40)
    assert s2.nlj == 40 and s2.natoms == s.natoms and s2.nspecies > s.nspecies
    classes = lambda t: len(set(zip(t.ljtype[t.species].tolist(), t.charge[t.species].tolist())))
    assert classes(s) <= 8 and classes(s2) >= 44
    assert np.array_equal(s2.mass[s2.species], s.mass[s.species]) and np.array_equal(s2.charge[s2.species], s.charge[s.species])
    assert np.array_equal(s2.resitype[s2.species], s.resitype[s.species]) and np.array_equal(s2.moltype[s2.species], s.moltype[s.species])
    a, b = pyoracle.Oracle(s), pyoracle.Oracle(s2)
    ea, va = a.forces(); eb, vb = b.forces()
    assert ea == eb and np.array_equal(va, vb)
    assert np.array_equal(a.fx, b.fx) and np.array_equal(a.fy, b.fy) and np.array_equal(a.fz, b.fz)
    a.group_temperature(); b.group_temperature()
    ra, rb = a.step(12), b.step(12)
    assert ra[0] == rb[0] and ra[2] == rb[2] and np.array_equal(a.rx, b.rx) and np.array_equal(a.vz, b.vz)
    with pytest.raises(ValueError):
        relabel_types(s, 3)
