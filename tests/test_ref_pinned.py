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


# ---- the constraint solve against the reference's own linear solver (solve.c) -----------------------------------------------
def _ref_solve_batch(systems):
    """solve.c's solve() for a list of (M, rhs): ONE child process for the batch"""
    import json
    import subprocess
    import sys
    if not systems:
        return []
    payload = json.dumps([[int(len(b)), np.asarray(a, float).ravel().tolist(), np.asarray(b, float).tolist()] for a, b in systems])
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "ref_probe.py"), "solve"], input=payload, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    return [np.array(x) for x in json.loads(out.stdout.splitlines()[-1])]


def constraint_matrix(location, rab, pairs, rmass, dt, dist, V):
    """solveConstraintMatrix's assembly, nglfconstraint.c:139-160, as written there:
    M[ab][uv] = (r_ab . r_uv) (((u==a)-(v==a)) / m_a - ((u==b)-(v==b)) / m_b),  rhs[ab] = -func(dt, d^2, r_ab, V_a - V_b)
    with frontFunc = ((r_ab + dt v_ab)^2 - d^2) / (2 dt) (:121-130), backFunc = r_ab . v_ab (:133-137)"""
    n = len(pairs)
    M, rhs = np.zeros((n, n)), np.zeros(n)
    for ab, (a, b) in enumerate(pairs):
        for uv, (u, v) in enumerate(pairs):
            M[ab, uv] = np.dot(rab[ab], rab[uv]) * ((int(u == a) - int(v == a)) * rmass[a] - (int(u == b) - int(v == b)) * rmass[b])
        vab = V[a] - V[b]
        if location == 0:
            p = rab[ab] + dt * vab
            rhs[ab] = -(np.dot(p, p) - dist[ab] ** 2) / (2 * dt)
        else:
            rhs[ab] = -np.dot(rab[ab], vab)
    return M, rhs


def constraint_apply(lam, rab, pairs, rmass, V):
    """nglfconstraint.c:162-171: V_a += sum_uv ((u==a)-(v==a)) / m_a lambda_uv r_uv"""
    for a in range(len(V)):
        for uv, (u, v) in enumerate(pairs):
            V[a] += ((int(u == a) - int(v == a)) * rmass[a] * lam[uv]) * rab[uv]


def res_move_cons(location, groups, dt, tol=1.0e-12, maxit=60):
    """resMoveCons (nglfconstraint.c:266-312) for a list of constraint groups {rab, pairs, rmass, dist, V}: solveConstraintMatrix
    until err = |lambda| <= tol (BACK: err = 0 after the one linear solve, :172-173), the linear systems by the reference's solve()"""
    todo = list(range(len(groups)))
    for it in range(maxit):
        sysm = [constraint_matrix(location, g["rab"], g["pairs"], g["rmass"], dt, g["dist"], g["V"]) for g in (groups[k] for k in todo)]
        lams = _ref_solve_batch(sysm)
        nxt = []
        for k, lam in zip(todo, lams):
            g = groups[k]
            constraint_apply(lam, g["rab"], g["pairs"], g["rmass"], g["V"])
            if location == 0 and np.sqrt((lam * lam).sum()) > tol:
                nxt.append(k)
        todo = nxt
        if not todo:
            return it + 1
    raise AssertionError("the reference's matrix iteration did not converge")


def constraint_topologies():
    """(atoms, pairs) with 1..12 pairs: chains of n pairs, rings of n = 3, 4, 6, 12, a triangle with a tail, a branched star"""
    tops = [(n + 1, [(k, k + 1) for k in range(n)]) for n in range(1, 13)]
    tops += [(n, [(k, (k + 1) % n) for k in range(n)]) for n in (3, 4, 6, 12)]
    tops.append((5, [(0, 1), (1, 2), (0, 2), (2, 3), (3, 4)]))
    tops.append((7, [(0, k) for k in range(1, 7)]))
    return tops


def make_constraint_system(seed, box=400.0, copies=3, wrap=True):
    """molecules = rigidly rotated copies of one random template per topology (so one r0 per constraint holds for all of them),
    masses per atom, random velocities; atoms wrapped into the periodic box when wrap (a molecule may straddle a face)"""
    rng = np.random.default_rng(seed)
    tops = constraint_topologies()
    sp_off = np.concatenate(([0], np.cumsum([na for na, _ in tops])))
    mass = rng.uniform(30.0, 130.0, sp_off[-1])
    rx, v, gid, species, r0s, cons = [], [], [], [], [], []
    mol = 0
    for t, (na, pairs) in enumerate(tops):
        # a self-avoiding random template: steps of 3.5..5.5 length units
        tpl = np.zeros((na, 3))
        for a in range(1, na):
            d = rng.normal(size=3)
            tpl[a] = tpl[a - 1] + d / np.linalg.norm(d) * rng.uniform(3.5, 5.5)
        if pairs[0] == (0, 1) and len(pairs) == na and na in (3, 4, 6, 12) and pairs[-1] == (na - 1, 0):
            ang = 2 * np.pi * np.arange(na) / na
            tpl = np.stack((4.2 * np.cos(ang), 4.2 * np.sin(ang), rng.normal(scale=0.8, size=na)), axis=1)      # a puckered ring
        r0 = np.array([np.linalg.norm(tpl[a] - tpl[b]) for a, b in pairs])
        r0s.append(r0)
        cons.append(pairs)
        for c in range(copies):
            Q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
            pos = tpl @ Q.T + rng.uniform(-0.5 * box, 0.5 * box, 3)
            if wrap:
                pos -= box * np.rint(pos / box)
            rx.append(pos)
            v.append(rng.normal(scale=2.0e-3, size=(na, 3)))
            gid += [(mol << 32) | (t + 1) << 16 | a for a in range(na)]
            species += [int(sp_off[t]) + a for a in range(na)]
            mol += 1
    return dict(tops=tops, sp_off=sp_off, mass=mass, r=np.concatenate(rx), v=np.concatenate(v), gid=np.array(gid, np.uint64),
                species=np.array(species, np.int32), r0=r0s, box=box, copies=copies)


def groups_of(sysd, r, v):
    """the constraint groups of the system in the reference's terms (one CONSTRAINT per residue), with nearest-image pair vectors"""
    out, first = [], 0
    for t, (na, pairs) in enumerate(sysd["tops"]):
        for c in range(sysd["copies"]):
            idx = np.arange(first, first + na)
            rab = np.array([r[idx[a]] - r[idx[b]] for a, b in pairs])
            rab -= sysd["box"] * np.rint(rab / sysd["box"])
            out.append(dict(idx=idx, pairs=pairs, rab=rab, rmass=1.0 / sysd["mass"][sysd["species"][idx]], dist=sysd["r0"][t], V=v[idx].copy()))
            first += na
    return out


def oracle_constraint_params(sysd):
    """orc_params for orc_velocity_constraint: one residue type per topology, one constraint list each"""
    tops = sysd["tops"]
    p = pyoracle.OrcParams()
    keep = {}
    p.hxx = p.hyy = p.hzz = sysd["box"]
    p.pbc = 7
    p.nresi = len(tops)
    p.nspecies = int(sysd["sp_off"][-1])
    keep["mass"] = np.ascontiguousarray(sysd["mass"])
    keep["resitype"] = np.concatenate([np.full(na, t, np.int32) for t, (na, _) in enumerate(tops)])
    keep["cons_off"] = np.concatenate(([0], np.cumsum([len(pr) for _, pr in tops]))).astype(np.int32)
    keep["consI"] = np.array([a for _, pr in tops for a, _b in pr], np.int32)
    keep["consJ"] = np.array([b for _, pr in tops for _a, b in pr], np.int32)
    keep["cons_grp"] = np.zeros(keep["consI"].size, np.int32)
    keep["cons_r0"] = np.concatenate(sysd["r0"])
    p.mass = keep["mass"].ctypes.data_as(pyoracle.dp)
    p.cons_r0 = keep["cons_r0"].ctypes.data_as(pyoracle.dp)
    for k in ("resitype", "cons_off", "consI", "consJ", "cons_grp"):
        setattr(p, k, keep[k].ctypes.data_as(pyoracle.ip))
    return p, keep


@pytest.mark.parametrize("seed", [1, 2])
def test_constraint_solve_against_the_references_linear_solver(seed):
    """The velocity-constraint solve the oracle and the device restate is resMoveConsOld's Gauss-Seidel sweep
    (nglfconstraint.c:180-264).  The reference holds a second, direct form of the same solve -- resMoveCons (:266-312):
    solveConstraintMatrix (:139-175) assembles M lambda = rhs and hands it to solve() (solve.c:3-28, scaled partial pivoting) --
    and solve.c compiles by itself.  Groups of 1..12 pairs (chains, rings, a triangle with a tail, a star), random masses and
    velocities, molecules across the periodic faces: the oracle's converged sweeps must land on the velocities the reference's
    own solver gives, FRONT (|r + dt v| = d) and BACK (r . v = 0)."""
    sysd = make_constraint_system(seed)
    p, keep = oracle_constraint_params(sysd)
    L = pyoracle.lib()
    dt = 20.0
    n = len(sysd["gid"])
    for location in (0, 1):
        r = np.ascontiguousarray(sysd["r"].T)
        v = np.ascontiguousarray(sysd["v"].T.copy())
        sweeps = L.orc_velocity_constraint(ctypes.byref(p), n, ctypes.c_double(dt), location, pyoracle._d(r[0]), pyoracle._d(r[1]), pyoracle._d(r[2]),
                                           pyoracle._d(v[0]), pyoracle._d(v[1]), pyoracle._d(v[2]), sysd["gid"].ctypes.data_as(pyoracle.up), pyoracle._i(sysd["species"]))
        assert 1 < sweeps < 500
        groups = groups_of(sysd, sysd["r"], sysd["v"])
        its = res_move_cons(location, groups, dt)
        assert its == 1 if location == 1 else 2 <= its <= 30
        worst = 0.0
        for g in groups:
            vo = v[:, g["idx"]].T
            worst = max(worst, np.abs(vo - g["V"]).max())
            # and both satisfy the constraint itself
            for ab, (a, b) in enumerate(g["pairs"]):
                w = g["V"][a] - g["V"][b]
                if location == 0:
                    q = g["rab"][ab] + dt * w
                    assert abs(np.sqrt(np.dot(q, q)) / g["dist"][ab] - 1.0) < 1e-11
                else:
                    assert abs(np.dot(g["rab"][ab], w)) * dt / g["dist"][ab] ** 2 < 1e-11
        assert worst < 2e-10 * np.abs(sysd["v"]).max(), (location, worst)
