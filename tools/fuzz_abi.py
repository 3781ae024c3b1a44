#!/usr/bin/env python3
"""Hostile-caller fuzz of the C-ABI (include/ddcmi.h) on a GPU: a valid system (the lipid deck: charges, bonded terms, four molecule types;
or a small water box) with ONE argument made wrong -- NaN / infinite / far-away positions, beads collapsed into a cell, species / group / LJ
type / term indices out of range, a zero, negative, NaN or too-small box, negative or NaN cut-offs and masses, a thermostat interval of 0,
a time step that blows the system up -- is set up, evaluated and stepped across a rebuild.  Every case must end in an error code with a
message or in a result; the process must not die (a GPU memory fault, SIGSEGV, SIGFPE) and must not hang.

The parent runs the cases in child processes (a GPU fault takes the process with it) and prints one line per case:
   python3 tools/fuzz_abi.py [ncases] [seed] [first]  # parent (cases first .. ncases-1); cases that killed or hung their child are listed at the end, exit 1 if any
"""
import copy, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BATCH = 16
STEPS = 25      # (the decks rebuild every 20 steps)


def mutations():
    import numpy as np
    M = []

    def m(name):
        def deco(f):
            M.append((name, f)); return f
        return deco

    def pick(s, rnd, frac=None):
        if frac is None:
            return rnd.randrange(s.natoms)
        return np.array(rnd.sample(range(s.natoms), max(1, int(frac * s.natoms))))
    for val, tag in ((float("nan"), "nan"), (float("inf"), "inf"), (1e300, "1e300"), (-1e30, "m1e30")):
        for arr in ("rx", "ry", "rz"):
            m("pos_%s_%s" % (arr, tag))(lambda s, rnd, t, arr=arr, val=val: getattr(s, arr).__setitem__(pick(s, rnd), val))
        m("vel_%s" % tag)(lambda s, rnd, t, val=val: s.vx.__setitem__(pick(s, rnd), val))
    m("pos_far")(lambda s, rnd, t: s.rx.__setitem__(pick(s, rnd), s.rx[0] + 1.0e6 * s.h[0]))
    m("pos_many_far")(lambda s, rnd, t: s.rz.__setitem__(pick(s, rnd, 0.1), 37.5 * s.h[8]))
    m("pos_all_nan")(lambda s, rnd, t: s.ry.fill(float("nan")))

    @m("pos_collapse_all")
    def _(s, rnd, t):
        s.rx[:] = s.rx[0]; s.ry[:] = s.ry[0]; s.rz[:] = s.rz[0]

    @m("pos_cluster_half")
    def _(s, rnd, t):
        k = pick(s, rnd, 0.5)
        s.rx[k] = s.rx[0] + 1e-3 * np.arange(k.size); s.ry[k] = s.ry[0]; s.rz[k] = s.rz[0]

    @m("pos_overlap_pair")
    def _(s, rnd, t):
        i, j = pick(s, rnd), pick(s, rnd)
        s.rx[j], s.ry[j], s.rz[j] = s.rx[i], s.ry[i], s.rz[i]
    m("vel_huge")(lambda s, rnd, t: s.vx.__setitem__(pick(s, rnd), 1.0e3))      # thousands of bohr per fs: across the box in a step
    m("vel_all_huge")(lambda s, rnd, t: s.vz.__imul__(1.0e6))
    for v, tag in ((None, "n"), (-1, "m1"), (1 << 30, "big"), (-(1 << 31), "min")):
        m("species_" + tag)(lambda s, rnd, t, v=v: s.species.__setitem__(pick(s, rnd), s.nspecies if v is None else v))
        m("group_" + tag)(lambda s, rnd, t, v=v: s.group.__setitem__(pick(s, rnd), s.ngroup if v is None else v))
        m("ljtype_" + tag)(lambda s, rnd, t, v=v: s.ljtype.__setitem__(rnd.randrange(s.nspecies), s.nlj if v is None else v))
        m("moltype_" + tag)(lambda s, rnd, t, v=v: s.moltype.__setitem__(rnd.randrange(s.nspecies), s.nmoltype if v is None else v))
    m("gid_dup")(lambda s, rnd, t: s.gid.__setitem__(pick(s, rnd), s.gid[0]))
    m("gid_all_equal")(lambda s, rnd, t: s.gid.fill(s.gid[0]))
    m("gid_top_bit")(lambda s, rnd, t: s.gid.__setitem__(pick(s, rnd), np.uint64(1) << np.uint64(63)))
    for k, tag in ((0, "xx"), (4, "yy"), (8, "zz")):
        for v, vt in ((0.0, "zero"), (-1.0, "neg"), (float("nan"), "nan"), (float("inf"), "inf"), (1e-300, "tiny"), (1e300, "huge")):
            m("box_%s_%s" % (tag, vt))(lambda s, rnd, t, k=k, v=v: s.h.__setitem__(k, v if v != -1.0 else -s.h[k]))
    m("box_offdiag")(lambda s, rnd, t: s.h.__setitem__(1, 0.3 * s.h[0]))
    m("box_under_2rlist")(lambda s, rnd, t: s.h.__setitem__(rnd.choice([0, 4, 8]), 1.7 * (s.rmax + s.deltaR)))
    m("box_one_cell")(lambda s, rnd, t: s.h.__setitem__(rnd.choice([0, 4, 8]), 0.5 * s.rmax))
    for v in (0, 1, 2, 3, 4, 5, 6, 8, -1, 255):
        m("pbc_%d" % v)(lambda s, rnd, t, v=v: setattr(s, "pbc", v))
    for fld in ("rmax", "deltaR", "krf", "crf", "keR", "dt"):
        for v, vt in ((0.0, "zero"), (-1.0, "neg"), (float("nan"), "nan"), (float("inf"), "inf"), (1e6, "1e6"), (1e-12, "tiny")):
            m("%s_%s" % (fld, vt))(lambda s, rnd, t, fld=fld, v=v: setattr(s, fld, v))
    for v in (-5, 1, 1 << 30):
        m("updateRate_%d" % v)(lambda s, rnd, t, v=v: setattr(s, "updateRate", v))
    m("updateRate_0_displacement")(lambda s, rnd, t: setattr(s, "updateRate", 0))
    for fld in ("mass", "charge", "sigma", "eps", "shift"):
        for v, vt in ((0.0, "zero"), (-1.0, "neg"), (float("nan"), "nan"), (float("inf"), "inf"), (1e30, "1e30")):
            m("%s_%s" % (fld, vt))(lambda s, rnd, t, fld=fld, v=v: getattr(s, fld).__setitem__(rnd.randrange(getattr(s, fld).size), v))
    m("nlj_more_than_table")(lambda s, rnd, t: setattr(s, "nlj", s.nlj + 3))
    m("nlj_zero")(lambda s, rnd, t: setattr(s, "nlj", 0))
    m("nlj_neg")(lambda s, rnd, t: setattr(s, "nlj", -2))
    m("nlj_65")(lambda s, rnd, t: (setattr(s, "nlj", 65), setattr(s, "sigma", np.ones(65 * 65)), setattr(s, "eps", np.ones(65 * 65) * 1e-3), setattr(s, "shift", np.zeros(65 * 65))))
    m("nspecies_zero")(lambda s, rnd, t: setattr(s, "nspecies", 0))
    m("nspecies_neg")(lambda s, rnd, t: setattr(s, "nspecies", -1))
    m("ngroup_zero")(lambda s, rnd, t: setattr(s, "ngroup", 0))
    m("ngroup_40")(lambda s, rnd, t: (setattr(s, "ngroup", 40), setattr(s, "group_type", np.ones(40, np.int32)), setattr(s, "group_Teq", np.ones(40) * 1e-3),
                                      setattr(s, "group_tau", np.ones(40) * 100.0), setattr(s, "group_interval", np.ones(40, np.int32))))
    m("natoms_zero")(lambda s, rnd, t: setattr(s, "natoms", 0))
    m("natoms_neg")(lambda s, rnd, t: setattr(s, "natoms", -7))
    m("natoms_one")(lambda s, rnd, t: setattr(s, "natoms", 1))

    def thermo(s, kind):
        s.group_type = np.full(s.ngroup, kind, np.int32)
        s.group_Teq = np.full(s.ngroup, 1.0e-3); s.group_tau = np.full(s.ngroup, 100.0); s.group_interval = np.ones(s.ngroup, np.int32)
    for kind, kt in ((1, "berendsen"), (2, "langevin")):
        m(kt + "_interval_0")(lambda s, rnd, t, kind=kind: (thermo(s, kind), s.group_interval.fill(0)))
        m(kt + "_interval_neg")(lambda s, rnd, t, kind=kind: (thermo(s, kind), s.group_interval.fill(-3)))
        m(kt + "_tau_0")(lambda s, rnd, t, kind=kind: (thermo(s, kind), s.group_tau.fill(0.0)))
        m(kt + "_tau_neg")(lambda s, rnd, t, kind=kind: (thermo(s, kind), s.group_tau.fill(-5.0)))
        m(kt + "_tau_nan")(lambda s, rnd, t, kind=kind: (thermo(s, kind), s.group_tau.fill(float("nan"))))
        m(kt + "_Teq_neg")(lambda s, rnd, t, kind=kind: (thermo(s, kind), s.group_Teq.fill(-1.0)))
        m(kt + "_Teq_nan")(lambda s, rnd, t, kind=kind: (thermo(s, kind), s.group_Teq.fill(float("nan"))))
    m("group_type_7")(lambda s, rnd, t: setattr(s, "group_type", np.full(s.ngroup, 7, np.int32)))
    # the expanded term lists (what ddcmi_set_bonded receives): t is the dict of martini.expand_bonded_terms
    for key, width in (("bond_ij", 2), ("angle_ijk", 3), ("tors_ijkl", 4)):
        for v, vt in ((None, "n"), (-1, "m1"), (1 << 30, "big")):
            def f(s, rnd, t, key=key, v=v):
                if t[key].size:
                    t[key][rnd.randrange(t[key].size)] = s.natoms if v is None else v
            m("%s_%s" % (key, vt))(f)

        def g(s, rnd, t, key=key, width=width):
            if t[key].size:
                r = rnd.randrange(t[key].size // width)
                t[key][r * width:(r + 1) * width] = t[key][r * width]      # a term of one bead with itself
        m("%s_self" % key)(g)
    for key in ("bond_kb", "bond_b0", "angle_k", "angle_t0", "tors_k", "tors_delta"):
        for v, vt in ((float("nan"), "nan"), (1e30, "1e30"), (-1.0, "neg")):
            m("%s_%s" % (key, vt))(lambda s, rnd, t, key=key, v=v: t[key].size and t[key].__setitem__(rnd.randrange(t[key].size), v))
    for key in ("angle_func", "tors_func", "tors_n"):
        for v in (0, -1, 99, 1 << 30):
            m("%s_%d" % (key, v))(lambda s, rnd, t, key=key, v=v: t[key].size and t[key].__setitem__(rnd.randrange(t[key].size), v))
    m("mol_nspecies_neg")(lambda s, rnd, t: s.nmoltype and s.mol_nspecies.__setitem__(rnd.randrange(s.nmoltype), -4))
    m("mol_nspecies_big")(lambda s, rnd, t: s.nmoltype and s.mol_nspecies.__setitem__(rnd.randrange(s.nmoltype), 1 << 20))
    # (not the last offset: that one IS the length of bpairI / bpairJ the caller promises -- a count larger than the arrays is nothing a C-ABI can see)
    m("bpair_off_unsorted")(lambda s, rnd, t: s.nmoltype and s.bpair_off.__setitem__(rnd.randrange(s.nmoltype), 1 << 20))
    m("bpair_off_neg")(lambda s, rnd, t: s.nmoltype and s.bpair_off.__setitem__(rnd.randrange(s.nmoltype + 1), -3))
    m("bpairI_big")(lambda s, rnd, t: s.bpairI.size and s.bpairI.__setitem__(rnd.randrange(s.bpairI.size), 1 << 20))
    m("bpairJ_neg")(lambda s, rnd, t: s.bpairJ.size and s.bpairJ.__setitem__(rnd.randrange(s.bpairJ.size), -9))
    m("exclude_all_terms")(lambda s, rnd, t: setattr(s, "excludePotentialTerm", 255))
    m("exclude_garbage")(lambda s, rnd, t: setattr(s, "excludePotentialTerm", -1))
    m("control")(lambda s, rnd, t: None)
    # ---- family "npt": the relaxed lipid deck with constraint groups, the barostat on the molecular pressure and restraints
    # (t["cons"] = [pair_off, pairI, pairJ, dist] of martini.expand_constraints, t["mols"] = [nmol_total, mol_off, mol_atoms] of martini.molecule_lists)
    m("npt_control")(lambda s, rnd, t: None)
    m("npt_cons_pair_off_unsorted")(lambda s, rnd, t: t["cons"][0].__setitem__(rnd.randrange(1, t["cons"][0].size - 1), 1 << 20))
    m("npt_cons_pair_off_neg")(lambda s, rnd, t: t["cons"][0].__setitem__(rnd.randrange(1, t["cons"][0].size), -5))
    m("npt_cons_pair_off_first")(lambda s, rnd, t: t["cons"][0].__setitem__(0, 3))
    for v, vt in ((None, "n"), (-1, "m1"), (1 << 30, "big")):
        m("npt_cons_pairI_" + vt)(lambda s, rnd, t, v=v: t["cons"][1].__setitem__(rnd.randrange(t["cons"][1].size), s.natoms if v is None else v))
        m("npt_cons_pairJ_" + vt)(lambda s, rnd, t, v=v: t["cons"][2].__setitem__(rnd.randrange(t["cons"][2].size), s.natoms if v is None else v))
        m("npt_mol_atoms_" + vt)(lambda s, rnd, t, v=v: t["mols"][2].__setitem__(rnd.randrange(t["mols"][2].size), s.natoms if v is None else v))
    m("npt_cons_pair_self")(lambda s, rnd, t: t["cons"][2].__setitem__(5, t["cons"][1][5]))
    m("npt_cons_far_partner")(lambda s, rnd, t: t["cons"][2].__setitem__(0, int(np.argmax((s.rx - s.rx[t["cons"][1][0]]) ** 2))))
    for v, vt in ((0.0, "zero"), (-1.0, "neg"), (float("nan"), "nan"), (float("inf"), "inf"), (1e4, "1e4"), (1e-9, "tiny")):
        m("npt_cons_dist_" + vt)(lambda s, rnd, t, v=v: t["cons"][3].__setitem__(rnd.randrange(t["cons"][3].size), v))
    m("npt_mol_off_unsorted")(lambda s, rnd, t: t["mols"][1].__setitem__(rnd.randrange(1, t["mols"][1].size - 1), 1 << 20))
    m("npt_mol_off_neg")(lambda s, rnd, t: t["mols"][1].__setitem__(rnd.randrange(1, t["mols"][1].size), -2))
    m("npt_mol_off_first")(lambda s, rnd, t: t["mols"][1].__setitem__(0, 2))
    m("npt_nmol_neg")(lambda s, rnd, t: t["mols"].__setitem__(0, -3))
    m("npt_nmol_zero")(lambda s, rnd, t: t["mols"].__setitem__(0, 0))
    m("npt_mol_dup_atom")(lambda s, rnd, t: t["mols"][2].__setitem__(1, t["mols"][2][0]))
    for fld in ("npt_T", "npt_P0", "npt_beta", "npt_tau"):
        for v, vt in ((0.0, "zero"), (-1.0, "neg"), (float("nan"), "nan"), (float("inf"), "inf"), (1e30, "1e30")):
            m("%s_%s" % (fld, vt))(lambda s, rnd, t, fld=fld, v=v: setattr(s, fld, v))
    m("npt_rest_gid_unknown")(lambda s, rnd, t: s.rest_gid.__setitem__(rnd.randrange(s.nrest), np.uint64(0x7fffffff) << np.uint64(32)))
    m("npt_rest_gid_dup")(lambda s, rnd, t: s.rest_gid.__setitem__(1, s.rest_gid[0]))
    for v, vt in ((float("nan"), "nan"), (float("inf"), "inf"), (-1.0, "neg"), (1e30, "1e30")):
        m("npt_rest_kb_" + vt)(lambda s, rnd, t, v=v: s.rest_kb.__setitem__(rnd.randrange(s.nrest), v))
        m("npt_rest_r0_" + vt)(lambda s, rnd, t, v=v: s.rest_r0.reshape(-1).__setitem__(rnd.randrange(s.rest_r0.size), v))
    for v in (-1, 2, 1 << 30):
        m("npt_rest_fc_%d" % v)(lambda s, rnd, t, v=v: s.rest_fc.reshape(-1).__setitem__(rnd.randrange(s.rest_fc.size), v))
        m("npt_rest_origin_%d" % v)(lambda s, rnd, t, v=v: setattr(s, "rest_origin", v))
    m("npt_nrest_neg")(lambda s, rnd, t: setattr(s, "nrest", -1))
    m("npt_pos_nan")(lambda s, rnd, t: s.rx.__setitem__(pick(s, rnd), float("nan")))
    m("npt_vel_huge")(lambda s, rnd, t: s.vx.__setitem__(pick(s, rnd), 1.0e3))
    m("npt_dt_1e6")(lambda s, rnd, t: setattr(s, "dt", 1e6))
    # ---- family "dec": the lipid deck on 2x2x2 bricks (an in-process group: migration, halo selection, terms named by gid, the gid -> slot tables)
    by_name = dict(M)
    for n in ("control", "pos_rx_nan", "pos_rz_inf", "pos_far", "pos_many_far", "pos_collapse_all", "pos_cluster_half", "pos_overlap_pair", "vel_huge", "vel_all_huge", "vel_nan",
              "species_big", "group_big", "moltype_big", "ljtype_big", "gid_dup", "gid_all_equal", "gid_top_bit", "box_xx_huge", "box_zz_inf", "box_under_2rlist", "pbc_0", "pbc_3", "pbc_5",
              "dt_1e6", "dt_nan", "charge_nan", "bond_ij_self", "tors_ijkl_self", "bond_kb_1e30", "bond_b0_1e30", "angle_t0_1e30", "rmax_tiny", "rmax_1e6", "deltaR_zero", "deltaR_1e6",
              "updateRate_0_displacement", "updateRate_1", "berendsen_tau_0", "berendsen_interval_0", "langevin_interval_0", "langevin_Teq_neg", "eps_1e30", "sigma_1e30",
              "mol_nspecies_big", "bpairI_big", "exclude_all_terms", "natoms_one"):
        m("dec_" + n)(by_name[n])

    # ---- family "lb": the same through the RCCL loopback (one rank whose periodic neighbours are reached through a 1-rank RCCL communicator: the count
    # rounds through the mailbox, grouped ncclSend / ncclRecv, the halo staged from the receive buffer, terms by gid)
    for n in ("control", "pos_rx_nan", "pos_rz_inf", "pos_far", "pos_many_far", "pos_collapse_all", "pos_cluster_half", "vel_huge", "vel_all_huge", "vel_nan",
              "moltype_big", "gid_dup", "gid_all_equal", "box_xx_huge", "box_under_2rlist", "pbc_0", "pbc_3", "dt_1e6", "dt_nan", "bond_ij_self", "bond_kb_1e30",
              "rmax_tiny", "deltaR_zero", "updateRate_0_displacement", "updateRate_1", "berendsen_tau_0", "langevin_interval_0", "eps_1e30", "sigma_1e30", "exclude_all_terms"):
        m("lb_" + n)(by_name[n])

    @m("dec_bond_partner_across_the_box")
    def _(s, rnd, t):
        j = int(t["bond_ij"][1])
        s.rz[j] += 0.45 * s.h[8]      # a bond longer than the halo is wide: its partner is on no rank that holds the other bead

    @m("dec_everything_in_one_brick")
    def _(s, rnd, t):
        s.rx[:] = 0.25 * s.h[0] + 0.2 * (s.rx - s.rx.min()) / max(np.ptp(s.rx), 1e-9) * s.h[0]
    return M


def base_setup(which):
    import numpy as np
    from ddcmd_amd.deck import load_deck
    from ddcmd_amd.synth import make_water_setup
    if which == "lipid":
        s = load_deck(os.path.join(ROOT, "tests", "golden", "lipid_deck", "object.data"))
    elif which == "npt":
        from ddcmd_amd.deck import units_convert
        # the constraint lists tests/test_oracle.py adds to the deck's residues (a triangle and a pair in TSTM, the glycerol pair in DPPC)
        CONSTRAINT_X = ("TSTM RESIPARMS { constraintList = TSTM_cl0 TSTM_cl1; } TSTM_cl0 CONSLISTPARMS { constraintSubList = TSTM_c0 TSTM_c1 TSTM_c2; } "
                        "TSTM_cl1 CONSLISTPARMS { constraintSubList = TSTM_c3; } TSTM_c0 CONSPARMS { atomI=0; atomJ=1; func=1; r0=0.40 nm; } "
                        "TSTM_c1 CONSPARMS { atomI=1; atomJ=2; func=1; r0=0.40 nm; } TSTM_c2 CONSPARMS { atomI=0; atomJ=2; func=1; r0=0.655 nm; } "
                        "TSTM_c3 CONSPARMS { atomI=3; atomJ=4; func=1; r0=0.40 nm; } DPPC RESIPARMS { constraintList = DPPC_cl0; } "
                        "DPPC_cl0 CONSLISTPARMS { constraintSubList = DPPC_c0; } DPPC_c0 CONSPARMS { atomI=2; atomJ=3; func=1; r0=0.37 nm; } ")
        deck = os.path.join(ROOT, "tests", "golden", "lipid_deck")
        extra = CONSTRAINT_X + " system SYSTEM { potential = martini restraintPot; } restraintPot POTENTIAL { type = RESTRAINT; parmfile = restraint.data; }"
        s = load_deck(os.path.join(deck, "object_nvt.data"), restart_file=os.path.join(deck, "relaxed", "restart"), extra_objects=extra)
        s.npt_T, s.npt_P0 = units_convert(310.0, "K"), units_convert(1.0, "bar")
        s.npt_beta, s.npt_tau = units_convert(3.0e-4, "1/bar") * 20.0, units_convert(1.0, "ps")
        s.rest_gid = np.array(s.rest_gid, dtype=np.uint64); s.rest_kb = np.array(s.rest_kb, dtype=np.float64)
        s.rest_r0 = np.ascontiguousarray(s.rest_r0, dtype=np.float64); s.rest_fc = np.ascontiguousarray(s.rest_fc, dtype=np.int32)
    else:
        s = make_water_setup(7, temperature_K=300.0)
    for a in ("rx", "ry", "rz", "vx", "vy", "vz", "h", "mass", "charge", "sigma", "eps", "shift", "group_Teq", "group_tau"):
        setattr(s, a, np.array(getattr(s, a), dtype=np.float64))
    for a in ("species", "group", "ljtype", "moltype", "mol_nspecies", "bpair_off", "bpairI", "bpairJ", "group_type", "group_interval"):
        setattr(s, a, np.array(getattr(s, a), dtype=np.int32))
    s.gid = np.array(s.gid, dtype=np.uint64)
    return s


def loopback_rank(martini, s):
    import ctypes
    import numpy as np
    os.environ["DDCMI_RCCL_LOOPBACK"] = "1"
    try:
        md = martini.MartiniRank(s, np.arange(s.natoms))
        try:
            martini._declare_domains(md.lib)
            buf = ctypes.create_string_buffer(128)
            if md.lib.ddcmi_comm_unique_id(buf) != 0:
                raise martini.DdcmiError("ddcmi_comm_unique_id failed")
            md.comm_init(0, 1, buf.raw, (1, 1, 1))
            md.upload_local()
        except Exception:
            md.close()
            raise
    finally:
        del os.environ["DDCMI_RCCL_LOOPBACK"]
    return md


def free_device_bytes():
    import ctypes
    hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")
    free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
    return free.value if hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0 else 0


def child(seed, lo, hi):
    import random
    import numpy as np
    import ddcmd_amd.martini as martini
    M = mutations()
    bases = {w: base_setup(w) for w in ("lipid", "water", "npt")}
    free0 = None
    for case in range(lo, hi):
        if case == lo + 2:
            if any(M[c % len(M)][0].startswith("lb_") for c in range(lo, hi)):
                try:      # (RCCL keeps ~360 MiB of its own for the life of the process from its first exchange on: not a context's)
                    w = loopback_rank(martini, copy.deepcopy(bases["water"]))
                    w.eval_forces(); w.step(2); w.close()
                except Exception:
                    pass
            free0 = free_device_bytes()      # (the first cases warm the runtime's own pools)
        rnd = random.Random(seed * 1000003 + case)
        name, f = M[case % len(M)]
        which = rnd.choice(["lipid", "lipid", "water"])
        if name.startswith("npt_"):
            which = "npt"
        elif name.startswith("dec_") or name.startswith("lb_"):
            which = "lipid"
        elif name.split("_")[0] in ("bond", "angle", "tors", "mol", "bpair", "bpairI", "bpairJ", "moltype", "charge", "krf", "crf", "keR"):
            which = "lipid"      # (the water box has no terms, no molecule tables and no charges: the mutation would change nothing)
        s = copy.deepcopy(bases[which])
        terms0 = {k: np.array(v) for k, v in martini.expand_bonded_terms(bases[which]).items()}
        if which == "npt":
            terms0["cons"] = [np.array(a) for a in martini.expand_constraints(bases[which])]
            terms0["mols"] = list(martini.molecule_lists(bases[which]))
            terms0["mols"][1:] = [np.array(a) for a in terms0["mols"][1:]]
        print("case %d %s %s ..." % (case, which, name), flush=True)
        out = "OK"
        try:
            f(s, rnd, terms0)
            orig = (martini.expand_bonded_terms, martini.expand_constraints, martini.molecule_lists)
            martini.expand_bonded_terms = lambda _s, t=terms0: t
            if which == "npt":
                martini.expand_constraints = lambda _s, t=terms0: tuple(t["cons"])
                martini.molecule_lists = lambda _s, t=terms0: tuple(t["mols"])
            try:
                if name.startswith("lb_"):
                    md = loopback_rank(martini, s)
                else:
                    md = martini.MartiniGroup(s, (2, 2, 2)) if name.startswith("dec_") else martini.MartiniHIP(s, constraints=(which == "npt"))
            finally:
                martini.expand_bonded_terms, martini.expand_constraints, martini.molecule_lists = orig
            try:
                e, _ = md.eval_forces()
                if os.environ.get("FUZZ_STEPWISE"):      # (reproducing a death: which step was it?  FUZZ_STEPWISE=1 python3 tools/fuzz_abi.py --child seed case case+1)
                    for k in range(STEPS):
                        md.step(1); md.energies(); print("   step %d done" % (k + 1), flush=True)
                else:
                    md.step(STEPS)
                e2, _, rk, _ = md.energies()
                md.gather() if name.startswith("dec_") else md.download_particles() if name.startswith("lb_") else md.download()
                out = "OK e_lj %.6g -> %.6g, total %.6g -> %.6g, kinetic %.6g" % (e["lj"], e2["lj"], e["total"], e2["total"], rk)
            finally:
                md.close()
        except martini.DdcmiError as ex:
            out = "REFUSED " + str(ex)[:160].replace("\n", " ")
        except Exception as ex:      # the Python driver's own checks (shapes, index errors before the library is reached)
            out = "PYTHON %s %s" % (type(ex).__name__, str(ex)[:120].replace("\n", " "))
        print("case %d %s %s -> %s" % (case, which, name, out), flush=True)
        if os.environ.get("FUZZ_MEM") and free0 is not None:
            import gc
            gc.collect()
            print("      free device memory: %.1f MiB below the mark" % ((free0 - free_device_bytes()) / 1048576.0), flush=True)
    if free0 is not None:
        import gc
        gc.collect()
        print("device memory after the cases %d..%d: %.1f MiB less free than after the first two" % (lo, hi - 1, (free0 - free_device_bytes()) / 1048576.0), flush=True)


def parent(ncases, seed, first=0):
    died, leaked = [], []
    lo = first
    while lo < ncases:
        hi = min(lo + BATCH, ncases)
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(seed), str(lo), str(hi)], cwd=ROOT, capture_output=True, text=True, timeout=600)
            out, rc, hung = r.stdout, r.returncode, False
            err = r.stderr
        except subprocess.TimeoutExpired as ex:
            out, rc, hung, err = (ex.stdout or b"").decode(errors="replace") if isinstance(ex.stdout, bytes) else (ex.stdout or ""), -1, True, ""
        done = [l for l in out.splitlines() if " -> " in l]
        for l in done:
            print(l, flush=True)
        for l in out.splitlines():
            if l.startswith("device memory after"):
                mib = float(l.split(":")[1].split()[0])
                leaked.append(mib)
                if mib > 64.0:
                    print(l + "   <-- a context's buffers did not come back", flush=True)
        started = [l for l in out.splitlines() if l.endswith(" ...")]
        if rc != 0 or hung:
            last = started[-1] if started else "case %d (before its first case)" % lo
            k = int(last.split()[1]) if started else lo
            tail = " | ".join(err.strip().splitlines()[-3:])[:400]
            print("%s -> %s rc=%d %s" % (last[:-4], "HUNG" if hung else "DIED", rc, tail), flush=True)
            died.append(last[:-4])
            lo = k + 1
        else:
            lo = hi
    print("%d cases: %d ended in a result or a message, %d killed or hung their process; device memory not returned by a batch of %d cases: %.1f MiB at worst"
          % (ncases - first, ncases - first - len(died), len(died), BATCH, max(leaked) if leaked else 0.0))
    for d in died:
        print("   " + d)
    if os.environ.get("FUZZ_MEM"):
        print("   MiB not returned per batch:", " ".join("%.0f" % x for x in leaked))
    return 1 if died or (leaked and max(leaked) > 64.0) else 0


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
    else:
        if len(sys.argv) > 1 and sys.argv[1] == "--list":
            for k, (n, _) in enumerate(mutations()):
                print(k, n)
            sys.exit(0)
        sys.exit(parent(int(sys.argv[1]) if len(sys.argv) > 1 else len(mutations()), int(sys.argv[2]) if len(sys.argv) > 2 else 1, int(sys.argv[3]) if len(sys.argv) > 3 else 0))
