"""A caller that hands the C-ABI wrong arguments gets an error code and a message -- not a GPU memory fault, a SIGSEGV or a silent NaN run.
The cases come from tools/fuzz_abi.py (round 6), whose first pass killed its process four times (a species' molecule type outside the
molecule tables; a bonded term naming bead 2^30; molecule tables with offsets out of order) and met eleven refusals without a message."""
import copy
import os
import subprocess
import sys

import numpy as np
import pytest

import ddcmd_amd.martini as martini
from ddcmd_amd.deck import load_deck

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lipid():
    return load_deck(os.path.join(ROOT, "tests", "golden", "lipid_deck", "object.data"))


def _run(s, terms=None, steps=3):
    orig = martini.expand_bonded_terms
    if terms is not None:
        martini.expand_bonded_terms = lambda _s: terms
    try:
        md = martini.MartiniHIP(s)
    finally:
        martini.expand_bonded_terms = orig
    try:
        md.eval_forces()
        md.step(steps)
    finally:
        md.close()


def _mut(lipid, f):
    s = copy.deepcopy(lipid)
    for a in ("moltype", "mol_nspecies", "bpair_off", "group_interval", "group_type"):
        setattr(s, a, np.array(getattr(s, a), dtype=np.int32))
    for a in ("h", "sigma", "charge", "group_tau", "group_Teq"):
        setattr(s, a, np.array(getattr(s, a), dtype=np.float64))
    t = {k: np.array(v) for k, v in martini.expand_bonded_terms(lipid).items()}
    f(s, t)
    return s, t


CASES = [
    # (what is wrong, the mutation, a piece of the message)
    ("moltype past the molecule tables", lambda s, t: s.moltype.__setitem__(3, 1 << 30), "molecule type"),
    ("moltype negative", lambda s, t: s.moltype.__setitem__(0, -(1 << 31)), "molecule type"),
    ("angle names bead 2^30", lambda s, t: t["angle_ijk"].__setitem__(7, 1 << 30), "names bead"),
    ("bond names bead natoms", lambda s, t: t["bond_ij"].__setitem__(5, s.natoms), "names bead"),
    ("dihedral names a negative bead", lambda s, t: t["tors_ijkl"].__setitem__(2, -1), "negative atom index"),
    ("bpair_off out of order", lambda s, t: s.bpair_off.__setitem__(0, 1 << 20), "bpair_off"),
    ("molecule type of no species", lambda s, t: s.mol_nspecies.__setitem__(1, -4), "has -4 species"),
    ("rmax = 0", lambda s, t: setattr(s, "rmax", 0.0), "rmax"),
    ("rmax NaN", lambda s, t: setattr(s, "rmax", float("nan")), "rmax"),
    ("deltaR NaN", lambda s, t: setattr(s, "deltaR", float("nan")), "deltaR"),
    ("deltaR negative", lambda s, t: setattr(s, "deltaR", -1.0), "deltaR"),
    ("krf NaN", lambda s, t: setattr(s, "krf", float("nan")), "krf"),
    ("no LJ types", lambda s, t: setattr(s, "nlj", 0), "LJ types"),
    ("no species", lambda s, t: setattr(s, "nspecies", 0), "species"),
    ("no groups", lambda s, t: setattr(s, "ngroup", 0), "groups"),
    ("negative bead count", lambda s, t: setattr(s, "natoms", -7), "beads"),
    ("box length infinite", lambda s, t: s.h.__setitem__(4, float("inf")), "finite"),
    ("pbc = 8", lambda s, t: setattr(s, "pbc", 8), "pbc"),
    ("sigma NaN", lambda s, t: s.sigma.__setitem__(1, float("nan")), "LJ table"),
    ("charge infinite", lambda s, t: s.charge.__setitem__(2, float("inf")), "charge"),
    ("bond constant NaN", lambda s, t: t["bond_kb"].__setitem__(0, float("nan")), "bond 0"),
    ("BERENDSEN tau negative", lambda s, t: (setattr(s, "group_type", np.ones(s.ngroup, np.int32)), s.group_tau.fill(-5.0)), "BERENDSEN"),
    ("LANGEVIN Teq NaN", lambda s, t: (setattr(s, "group_type", np.full(s.ngroup, 2, np.int32)), s.group_tau.fill(100.0), s.group_Teq.fill(float("nan"))), "LANGEVIN"),
    ("a NaN coordinate", lambda s, t: s.rx.__setitem__(11, float("nan")), "non-finite"),
]


def test_wrong_constraint_and_molecule_tables_are_refused(lipid):
    """the index-named tables of the nglfconstraint path: offsets out of order, beads the state does not hold"""
    md = martini.MartiniHIP(lipid)
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
    P, D = martini.ctypes.POINTER(martini.ctypes.c_int), martini.ctypes.POINTER(martini.ctypes.c_double)
    ip = lambda a: a.ctypes.data_as(P)
    try:
        dist = np.array([7.0, 7.0, 7.0])
        for off, pi, pj, msg in (([0, 2, 1], [0, 1, 2], [1, 2, 3], b"pair_off decreases"), ([1, 2, 3], [0, 1, 2], [1, 2, 3], b"pair_off[0]"),
                                 ([0, 1, 3], [0, 1, lipid.natoms], [1, 2, 3], b"names bead"), ([0, 1, 3], [0, -1, 2], [1, 2, 3], b"negative bead")):
            o, a, b = i32(off), i32(pi), i32(pj)
            rc = md.lib.ddcmi_set_constraints(md.ctx, 2, ip(o), ip(a), ip(b), dist.ctypes.data_as(D))
            assert rc != 0 and msg in md.lib.ddcmi_last_error(md.ctx), md.lib.ddcmi_last_error(md.ctx)
        for off, at, msg in (([0, 3, 2], [0, 1, 2], b"mol_off decreases"), ([0, 2, 3], [0, 1, 1 << 30], b"names bead"), ([0, 2, 3], [0, -4, 2], b"negative bead")):
            o, a = i32(off), i32(at)
            rc = md.lib.ddcmi_set_molecule_lists(md.ctx, 100, 2, ip(o), ip(a))
            assert rc != 0 and msg in md.lib.ddcmi_last_error(md.ctx), md.lib.ddcmi_last_error(md.ctx)
        rc = md.lib.ddcmi_set_barostat(md.ctx, -1.0, 0.0, 1.0, 1000.0)
        assert rc != 0 and b"ddcmi_set_barostat" in md.lib.ddcmi_last_error(md.ctx)
        md.eval_forces()      # the context still runs
    finally:
        md.close()


@pytest.mark.parametrize("what,f,msg", CASES, ids=[c[0].replace(" ", "_") for c in CASES])
def test_wrong_argument_is_refused_with_a_message(lipid, what, f, msg):
    s, t = _mut(lipid, f)
    with pytest.raises(martini.DdcmiError) as ei:
        _run(s, t)
    text = str(ei.value)
    assert msg in text and len(text.split(":", 1)[1].strip()) > 10, text


def test_a_refused_call_leaves_the_context_usable(lipid):
    """the refusal is an answer, not the end of the context: the same context takes the right arguments next and runs"""
    md = martini.MartiniHIP(lipid)
    try:
        e0, _ = md.eval_forces()
        h_bad = np.array(lipid.h, dtype=np.float64)
        h_bad[0] = float("nan")
        rc = md.lib.ddcmi_set_box(md.ctx, h_bad.ctypes.data_as(martini.ctypes.POINTER(martini.ctypes.c_double)), 7)
        assert rc != 0 and b"box lengths" in md.lib.ddcmi_last_error(md.ctx)
        rc = md.lib.ddcmi_set_neighbor(md.ctx, float("nan"), 20)
        assert rc != 0 and b"deltaR" in md.lib.ddcmi_last_error(md.ctx)
        rc = md.lib.ddcmi_step_nglf(md.ctx, float("nan"), 1)
        assert rc != 0 and b"time step" in md.lib.ddcmi_last_error(md.ctx)
        e1, _ = md.eval_forces()
        assert e1 == e0
        md.step(3)
    finally:
        md.close()


def test_one_pass_of_the_hostile_caller_fuzz():
    """tools/fuzz_abi.py: 363 systems with one wrong argument each (216 on the lipid deck / a water box, 67 on the relaxed deck with constraint
    groups, the barostat and restraints, 50 on 2x2x2 bricks of an in-process group, 30 through the RCCL loopback), set up, evaluated and stepped across
    a rebuild in child processes: none of them may kill or hang its process"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_abi.py"), "363", "7"], cwd=ROOT, capture_output=True, text=True, timeout=1500)
    tail = "\n".join(r.stdout.strip().splitlines()[-8:])
    assert r.returncode == 0 and "0 killed or hung" in tail, tail + r.stderr[-2000:]
    assert r.stdout.count(" -> REFUSED ") > 180 and r.stdout.count(" -> OK ") > 60, tail
