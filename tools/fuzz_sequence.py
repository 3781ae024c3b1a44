#!/usr/bin/env python3
"""Call-sequence fuzz of the C-ABI on a GPU: a long-lived context is driven through a random sequence of VALID calls -- steps, force evaluations,
explicit rebuilds, new neighbour settings, a new cut-off, new charges and masses, terms dropped and restored, molecule tables switched off and on,
thermostats switched, the state uploaded again (as it is, jittered, or a different system altogether), the box rescaled -- and after every few calls its
force evaluation must equal that of a FRESH context created from the same parameters and the same state (forces and sums, 1e-10): what a setter
invalidates (lists, tables, tags, slots, sums) must really be rebuilt, whatever came before.
   python3 tools/fuzz_sequence.py [nsequences] [seed] [ops per sequence]"""
import copy, ctypes, os, random, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ddcmd_amd.martini as martini
from ddcmd_amd.martini import MartiniHIP, _d, _i
from ddcmd_amd.deck import load_deck
from ddcmd_amd.synth import make_water_setup

f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
VERBOSE = bool(os.environ.get("FUZZ_VERBOSE"))
REPLAYS = []      # the trajectory replays of the checks (Live.check): their differences


def base(which):
    if which == "lipid":
        deck = os.path.join(ROOT, "tests", "golden", "lipid_deck")
        s = load_deck(os.path.join(deck, "object_nvt.data"), restart_file=os.path.join(deck, "relaxed", "restart"))
    elif which == "npt":      # the relaxed deck with constraint lists in its residues and six restrained beads: the nglfconstraint step's tables come and go
        from ddcmd_amd.deck import units_convert
        CONSTRAINT_X = ("TSTM RESIPARMS { constraintList = TSTM_cl0 TSTM_cl1; } TSTM_cl0 CONSLISTPARMS { constraintSubList = TSTM_c0 TSTM_c1 TSTM_c2; } "
                        "TSTM_cl1 CONSLISTPARMS { constraintSubList = TSTM_c3; } TSTM_c0 CONSPARMS { atomI=0; atomJ=1; func=1; r0=0.40 nm; } "
                        "TSTM_c1 CONSPARMS { atomI=1; atomJ=2; func=1; r0=0.40 nm; } TSTM_c2 CONSPARMS { atomI=0; atomJ=2; func=1; r0=0.655 nm; } "
                        "TSTM_c3 CONSPARMS { atomI=3; atomJ=4; func=1; r0=0.40 nm; } DPPC RESIPARMS { constraintList = DPPC_cl0; } "
                        "DPPC_cl0 CONSLISTPARMS { constraintSubList = DPPC_c0; } DPPC_c0 CONSPARMS { atomI=2; atomJ=3; func=1; r0=0.37 nm; } ")
        deck = os.path.join(ROOT, "tests", "golden", "lipid_deck")
        extra = CONSTRAINT_X + " system SYSTEM { potential = martini restraintPot; } restraintPot POTENTIAL { type = RESTRAINT; parmfile = restraint.data; }"
        s = load_deck(os.path.join(deck, "object_nvt.data"), restart_file=os.path.join(deck, "relaxed", "restart"), extra_objects=extra)
        s.baro = (units_convert(310.0, "K"), units_convert(1.0, "bar"), units_convert(3.0e-4, "1/bar") * 20.0, units_convert(1.0, "ps"))
        s.rest_gid = np.array(s.rest_gid, dtype=np.uint64); s.rest_kb = np.array(s.rest_kb, dtype=np.float64)
        s.rest_r0 = np.ascontiguousarray(s.rest_r0, dtype=np.float64); s.rest_fc = np.ascontiguousarray(s.rest_fc, dtype=np.int32)
        s.nrest_all = int(s.nrest)
    else:
        s = make_water_setup(7, temperature_K=150.0)
    for a in ("rx", "ry", "rz", "vx", "vy", "vz", "h", "mass", "charge", "sigma", "eps", "shift", "group_Teq", "group_tau"):
        setattr(s, a, np.array(getattr(s, a), dtype=np.float64))
    for a in ("group_type", "group_interval"):
        setattr(s, a, np.array(getattr(s, a), dtype=np.int32))
    return s


class GroupMD(object):
    """the lipid deck on 2x2x2 bricks of an in-process group behind the calls the sequences make (the state comes back gathered, in gid = caller order)"""
    def __init__(self, s):
        self.g = martini.MartiniGroup(s, (2, 2, 2))
        self.lib, self.ctxs, self.s = self.g.lib, [r.ctx for r in self.g.ranks], s
        self.step, self.eval_forces, self.energies, self.close = self.g.step, self.g.eval_forces, self.g.energies, self.g.close
        assert np.array_equal(np.sort(np.asarray(s.gid)), np.asarray(s.gid))

    def download(self):
        st = self.g.gather()
        return {"r": st["r"], "v": st["v"], "f": st["f"]}

    def build_list(self):      # (a group rebuilds when a list is not valid: the neighbour settings again, unchanged, say so)
        for r in self.g.ranks:
            r._chk(self.lib.ddcmi_set_neighbor(r.ctx, r.s.deltaR, int(r.s.updateRate)))

    def _chk(self, rc):
        self.g._chk(rc)


class LoopbackMD(object):
    """the lipid deck on ONE rank whose periodic neighbours are reached through a 1-rank RCCL communicator (DDCMI_RCCL_LOOPBACK=1): the transport path --
    count rounds, grouped ncclSend / ncclRecv, terms by gid, the halo staged from the receive buffer -- behind the same calls"""
    def __init__(self, s):
        os.environ["DDCMI_RCCL_LOOPBACK"] = "1"
        try:
            self.m = m = martini.MartiniRank(s, np.arange(s.natoms))
            martini._declare_domains(m.lib)
            buf = ctypes.create_string_buffer(128)
            m._chk(m.lib.ddcmi_comm_unique_id(buf))
            m.comm_init(0, 1, buf.raw, (1, 1, 1))
            m.upload_local()
        finally:
            del os.environ["DDCMI_RCCL_LOOPBACK"]
        self.lib, self.ctx, self.s = m.lib, m.ctx, s
        self.step, self.eval_forces, self.energies, self.close, self._chk, self.build_list = m.step, m.eval_forces, m.energies, m.close, m._chk, m.build_list
        assert np.array_equal(np.sort(np.asarray(s.gid)), np.asarray(s.gid))

    def download(self):
        p = self.m.download_particles()
        o = np.argsort(p["gid"], kind="stable")
        return {k: [a[o] for a in p[k]] for k in ("r", "v", "f")}


class Live(object):
    def __init__(self, s, rnd, group=False):
        self.group = group
        self.s, self.rnd = copy.deepcopy(s), rnd
        self.terms = {k: np.array(v) for k, v in martini.expand_bonded_terms(s).items()}
        self.molecules_on = s.nmoltype > 0
        self.npt = hasattr(s, "baro")
        self.cons_on = self.baro_on = False
        self.md = (LoopbackMD(self.s) if group == "loopback" else GroupMD(self.s)) if group else self.make(self.s, self.terms, self.molecules_on)
        self.log = []

    @staticmethod
    def make(s, terms, molecules_on):
        orig = martini.expand_bonded_terms
        martini.expand_bonded_terms = lambda _s: terms
        s2 = copy.copy(s)
        if not molecules_on:
            s2.nmoltype = 0
        try:
            return MartiniHIP(s2)
        finally:
            martini.expand_bonded_terms = orig

    def chk(self, rc):
        self.md._chk(rc)

    def each(self, call):
        """a setter on the context -- on every domain's context of an in-process group"""
        for ctx in (self.md.ctxs if hasattr(self.md, "ctxs") else [self.md.ctx]):
            rc = call(self.md.lib, ctx)
            if rc != 0:
                raise martini.DdcmiError("ddcmi error %d: %s" % (rc, self.md.lib.ddcmi_last_error(ctx).decode()))

    def state(self):
        d = self.md.download()
        return [np.array(a) for a in d["r"]], [np.array(a) for a in d["v"]]

    # ---- the calls ----
    def op_step(self):
        k = self.rnd.choice([1, 2, 3, 7, 20, 25])
        self.md.step(k)
        if self.baro_on:
            b = self.md.box(); self.s.h[0], self.s.h[4], self.s.h[8] = b[0], b[1], b[2]      # (the barostat moves the box: the fresh context starts from the current one)
        return "step %d" % k

    def op_eval(self):
        self.md.eval_forces(); return "eval_forces"

    def op_build(self):
        self.md.build_list(); return "build_list"

    def op_neighbor(self):
        s = self.s
        s.deltaR = float(s.deltaR * self.rnd.choice([0.5, 1.0, 1.25]))
        s.deltaR = min(max(s.deltaR, 2.0), 0.45 * min(s.h[0], s.h[4], s.h[8]) - s.rmax)
        s.updateRate = self.rnd.choice([0, 1, 5, 10, 20])
        self.each(lambda lib_, ctx_: lib_.ddcmi_set_neighbor(ctx_, s.deltaR, int(s.updateRate)))
        for r in getattr(getattr(self.md, "g", None), "ranks", []):
            r.s.deltaR, r.s.updateRate = s.deltaR, s.updateRate
        return "set_neighbor deltaR %.3f updateRate %d" % (s.deltaR, s.updateRate)

    def op_rmax(self):
        s = self.s
        s.rmax = float(min(s.rmax * self.rnd.choice([0.85, 1.0, 1.1]), 0.45 * min(s.h[0], s.h[4], s.h[8]) - s.deltaR))
        self.each(lambda lib_, ctx_: lib_.ddcmi_set_nonbonded(ctx_, s.nlj, _d(f64(s.sigma)), _d(f64(s.eps)), _d(f64(s.shift)), s.rmax, s.keR, s.krf, s.crf))
        return "set_nonbonded rmax %.3f" % s.rmax

    def op_table(self):
        s = self.s
        k = self.rnd.randrange(s.eps.size)
        s.eps[k] *= self.rnd.choice([0.5, 1.0, 2.0])
        s.eps = 0.5 * (s.eps.reshape(s.nlj, s.nlj) + s.eps.reshape(s.nlj, s.nlj).T).ravel()      # (the table is symmetric)
        self.each(lambda lib_, ctx_: lib_.ddcmi_set_nonbonded(ctx_, s.nlj, _d(f64(s.sigma)), _d(f64(s.eps)), _d(f64(s.shift)), s.rmax, s.keR, s.krf, s.crf))
        return "set_nonbonded eps[%d]" % k

    def op_species(self):
        s = self.s
        what = self.rnd.choice(["charges_off", "charges_half", "charges_back", "mass"])
        if what == "charges_off": s.charge[:] = 0.0
        elif what == "charges_half": s.charge[:] = 0.5 * self.charge0
        elif what == "charges_back": s.charge[:] = self.charge0
        else: s.mass[self.rnd.randrange(s.nspecies)] *= self.rnd.choice([0.5, 2.0])
        self.each(lambda lib_, ctx_: lib_.ddcmi_set_species(ctx_, s.nspecies, _d(f64(s.mass)), _d(f64(s.charge)), _i(i32(s.ljtype)), _i(i32(s.moltype))))
        return "set_species " + what

    def op_terms(self):
        t0, t = self.terms0, {}
        what = self.rnd.choice(["all", "half_bonds", "no_angles", "no_dihedrals", "none"])
        for k, v in t0.items():
            t[k] = np.array(v)
        def cut(prefix, width, keep):
            n = t[prefix + "_k" if prefix != "bond" else "bond_kb"].size
            sel = np.arange(n)[:keep(n)]
            for k in list(t):
                if k.startswith(prefix + "_"):
                    w = width if k.endswith(("_ij", "_ijk", "_ijkl")) else 1
                    t[k] = np.ascontiguousarray(t[k].reshape(n, w)[sel].ravel()) if n else t[k]
        if what == "half_bonds": cut("bond", 2, lambda n: n // 2)
        if what == "no_angles": cut("angle", 3, lambda n: 0)
        if what == "no_dihedrals": cut("tors", 4, lambda n: 0)
        if what == "none":
            cut("bond", 2, lambda n: 0); cut("angle", 3, lambda n: 0); cut("tors", 4, lambda n: 0)
        self.terms = t
        nb, na, nt = t["bond_kb"].size, t["angle_k"].size, t["tors_k"].size
        z = np.zeros(4, np.int32)
        self.each(lambda lib_, ctx_: lib_.ddcmi_set_bonded(ctx_, nb, _i(t["bond_ij"] if nb else z), _d(t["bond_kb"]), _d(t["bond_b0"]),
                                              na, _i(t["angle_ijk"] if na else z), _i(t["angle_func"] if na else z), _d(t["angle_k"]), _d(t["angle_t0"]),
                                              nt, _i(t["tors_ijkl"] if nt else z), _i(t["tors_func"] if nt else z), _i(t["tors_n"] if nt else z), _d(t["tors_k"]), _d(t["tors_delta"]),
                                              int(self.s.excludePotentialTerm)))
        return "set_bonded " + what

    def op_molecules(self):
        s = self.s
        self.molecules_on = not self.molecules_on if s.nmoltype > 0 else False
        if self.molecules_on:
            bi, bj = i32(s.bpairI), i32(s.bpairJ)
            if bi.size == 0:
                bi = bj = np.zeros(1, np.int32)
            self.each(lambda lib_, ctx_: lib_.ddcmi_set_molecules(ctx_, s.nmoltype, _i(i32(s.mol_nspecies)), _i(i32(s.bpair_off)), _i(bi), _i(bj)))
        else:
            self.each(lambda lib_, ctx_: lib_.ddcmi_set_molecules(ctx_, 0, None, None, None, None))
        return "set_molecules %s" % ("on" if self.molecules_on else "off")

    def op_groups(self):
        s = self.s
        kind = self.rnd.choice([0, 1, 2])
        s.group_type = np.full(s.ngroup, kind, np.int32)
        s.group_Teq = np.full(s.ngroup, 1.0e-3); s.group_tau = np.full(s.ngroup, 500.0); s.group_interval = np.ones(s.ngroup, np.int32)
        self.each(lambda lib_, ctx_: lib_.ddcmi_set_groups(ctx_, s.ngroup, _i(i32(s.group_type)), _d(f64(s.group_Teq)), _d(f64(s.group_tau)), _i(i32(s.group_interval))))
        return "set_groups %d" % kind

    def op_reupload(self):
        r, v = self.state()
        how = self.rnd.choice(["same", "jitter", "shuffle_box"])
        if how == "jitter":
            r = [a + 0.05 * np.array([self.rnd.uniform(-1, 1) for _ in range(a.size)]) for a in r]
        if how == "shuffle_box":      # every bead by a whole box length along some axis: the same system
            r[0] = r[0] + self.s.h[0] * np.array([self.rnd.choice([-1, 0, 1]) for _ in range(r[0].size)])
        self.md.upload(r[0], r[1], r[2], v[0], v[1], v[2])
        return "upload_state " + how

    def op_positions(self):
        r, v = self.state()
        r = [a + 0.02 * np.array([self.rnd.uniform(-1, 1) for _ in range(a.size)]) for a in r]
        self.md.upload_positions(r, v)
        return "upload_positions jitter"

    def op_box(self):
        s = self.s
        f = self.rnd.choice([0.99, 1.01, 1.02])
        r, v = self.state()
        s.h = s.h * f
        self.each(lambda lib_, ctx_: lib_.ddcmi_set_box(ctx_, _d(f64(s.h)), int(s.pbc)))
        self.md.upload(r[0] * f, r[1] * f, r[2] * f, v[0], v[1], v[2])
        return "set_box x %.2f + upload_state" % f

    def op_restraints(self):
        s = self.s
        s.nrest = 0 if s.nrest else s.nrest_all
        up = ctypes.POINTER(ctypes.c_uint64)
        self.each(lambda lib_, ctx_: lib_.ddcmi_set_restraints(ctx_, int(s.nrest), s.rest_gid.ctypes.data_as(up), _i(i32(np.asarray(s.rest_fc).ravel())), _d(f64(np.asarray(s.rest_r0).ravel())),
                                                   _d(f64(s.rest_kb)), int(getattr(s, "rest_origin", 0))))
        return "set_restraints %d" % s.nrest

    def op_constraints(self):
        self.cons_on = not self.cons_on
        if self.cons_on:
            po, pi, pj, dd = self._cons = martini.expand_constraints(self.s)
            self.each(lambda lib_, ctx_: lib_.ddcmi_set_constraints(ctx_, int(po.size - 1), _i(po), _i(pi), _i(pj), _d(dd)))
        else:
            self.each(lambda lib_, ctx_: lib_.ddcmi_set_constraints(ctx_, 0, None, None, None, None))
        return "set_constraints %s" % ("on" if self.cons_on else "off")

    def op_barostat(self):
        self.baro_on = not self.baro_on
        if self.baro_on:
            self.md.set_barostat(*self.s.baro)
        else:
            self.each(lambda lib_, ctx_: lib_.ddcmi_set_barostat(ctx_, 0.0, 0.0, 0.0, 1.0))
        return "set_barostat %s" % ("on" if self.baro_on else "off")

    def op_other_system(self):
        """water: a box of another size into the same context (more or fewer beads, another box): every array sized by the bead count, the image count hint,
        the rows of pending lean steps"""
        n = self.rnd.choice([6, 7, 8])
        new = copy.deepcopy(self.water_sizes[n])
        for a in ("rmax", "deltaR", "updateRate", "eps", "sigma", "shift", "mass", "charge", "group_type", "group_Teq", "group_tau", "group_interval", "nlj", "keR", "krf", "crf"):
            setattr(new, a, copy.deepcopy(getattr(self.s, a)))
        if 0.45 * new.h[0] < new.rmax + new.deltaR:
            return "another system: skipped (its box is under 2 (rmax + deltaR))"
        self.s = new
        self.md.s, self.md.n = new, new.natoms
        self.each(lambda lib_, ctx_: lib_.ddcmi_set_box(ctx_, _d(f64(new.h)), int(new.pbc)))
        self.md.upload(new.rx, new.ry, new.rz, new.vx, new.vy, new.vz)
        return "another system: %d beads" % new.natoms

    def op_misc(self):
        what = self.rnd.choice(["clock", "timing_on", "timing_off", "stats", "get_list", "kinetic"])
        if what == "clock": self.each(lambda lib_, ctx_: lib_.ddcmi_set_clock(ctx_, self.rnd.randrange(0, 1000), 0.0))
        elif what == "timing_on": self.md.timing(True)
        elif what == "timing_off": self.md.timing(False)
        elif what == "stats": self.md.list_stats()
        elif what == "get_list":
            try: self.md.get_list(0)
            except martini.DdcmiError: pass      # (no valid list at the moment: an answer)
        else: self.md.energies()
        return what

    # ---- the check ----
    def check(self, rebuild_first=False):
        if rebuild_first:
            self.md.build_list()
        e1, v1 = self.md.eval_forces()
        d = self.md.download()
        s2 = copy.deepcopy(self.s)
        s2.rx, s2.ry, s2.rz = [np.array(a) for a in d["r"]]
        s2.vx, s2.vy, s2.vz = [np.array(a) for a in d["v"]]
        fresh = self.make(s2, self.terms, self.molecules_on)
        try:
            e2, v2 = fresh.eval_forces()
            d2 = fresh.download()
        finally:
            fresh.close()
        fmax = max(np.abs(d2["f"][c]).max() for c in range(3)) + 1e-300
        ferr = max(np.abs(d["f"][c] - d2["f"][c]).max() for c in range(3)) / fmax
        scale = max(abs(e2["total"]), abs(e2["lj"]), 1e-12)
        eerr = max(abs(e1[k] - e2[k]) for k in e2) / scale
        verr = np.abs(v1 - v2).max() / max(np.abs(v2).max(), 1e-300)
        err = max(ferr, eerr, verr)
        # ... and the next steps: a fresh context given the same state, clock and parameters must MOVE like the long-lived one (the integrator's side: drift and
        # kick bookkeeping, lean steps and their rings, the Langevin noise keyed by gid and loop count).  Not for BERENDSEN groups -- their factor carries the
        # temperatures published so far, which a fresh context has not seen -- nor while constraint groups or the barostat are on, nor on decomposed contexts.
        kind = int(np.asarray(self.s.group_type).ravel()[0]) if np.asarray(self.s.group_type).size else 0
        if err < 1e-9 and not rebuild_first and not self.group and not self.cons_on and not self.baro_on and kind in (0, 2):
            k = self.rnd.choice([3, 11, 23])
            s2.loop, s2.time = self.md.clock()
            s2.lcg64 = None      # (the long-lived context draws from the counter-based stream: its groups were not LANGEVIN when the deck's LCG64 records could have been set)
            fresh = self.make(s2, self.terms, self.molecules_on)
            try:
                fresh.eval_forces()
                try:
                    fresh.step(k); self.md.step(k)
                except martini.DdcmiError as ex:
                    if "unstable" in str(ex) or "non-finite" in str(ex):
                        raise      # (both or either: the sequence ends as one the parameters made unstable)
                    raise
                da, db = self.md.download(), fresh.download()
            finally:
                fresh.close()
            if self.baro_on is False:
                Lbox = np.array([self.s.h[0], self.s.h[4], self.s.h[8]])
                rerr = max(np.abs((da["r"][c] - db["r"][c]) - Lbox[c] * np.rint((da["r"][c] - db["r"][c]) / Lbox[c])).max() for c in range(3)) / Lbox.min()
                vmax = max(np.abs(db["v"][c]).max() for c in range(3)) + 1e-300
                terr = max(rerr, max(np.abs(da["v"][c] - db["v"][c]).max() for c in range(3)) / vmax)
                self.log.append("   + %d steps beside a fresh context: %.1e" % (k, terr))
                REPLAYS.append(terr)
                if VERBOSE: print("   ", self.log[-1], flush=True)
                err = max(err, terr * 1e-2)      # (1e-7 of the box / of the largest speed after up to 23 steps counts as a difference)
        return err, (ferr, eerr, verr), (e1, e2)


def main():
    nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    nops = int(sys.argv[3]) if len(sys.argv) > 3 else 24
    bases = {w: base(w) for w in ("lipid", "water", "npt")}
    water_sizes = {}
    for n in (6, 7, 8):
        w = make_water_setup(n, temperature_K=150.0)
        for a in ("rx", "ry", "rz", "vx", "vy", "vz", "h"):
            setattr(w, a, np.array(getattr(w, a), dtype=np.float64))
        water_sizes[n] = w
    bad = stale = 0
    worst = 0.0
    for q in range(nseq):
        rnd = random.Random(seed * 7919 + q)
        which = rnd.choice(["lipid", "lipid", "water", "npt", "bricks", "loopback"])
        which = os.environ.get("FUZZ_ONLY", which)      # (FUZZ_ONLY=bricks: only that family)
        L = Live(bases["lipid" if which in ("bricks", "loopback") else which], rnd, group=(which if which in ("bricks", "loopback") else False))
        L.charge0 = np.array(L.s.charge); L.terms0 = {k: np.array(v) for k, v in L.terms.items()}
        ops = [L.op_step, L.op_step, L.op_eval, L.op_build, L.op_neighbor, L.op_rmax, L.op_table, L.op_species, L.op_groups, L.op_reupload, L.op_positions, L.op_box, L.op_misc]
        if which in ("bricks", "loopback"):      # the calls a decomposed run can take in mid-run (terms by gid and the beads stay as they were given)
            ops = [L.op_step, L.op_step, L.op_eval, L.op_neighbor, L.op_rmax, L.op_table, L.op_species, L.op_species, L.op_groups, L.op_molecules] + ([L.op_build] if which == "loopback" else [])
        if which in ("lipid", "npt"):
            ops += [L.op_terms, L.op_terms, L.op_molecules, L.op_species]
        if which == "water":
            L.water_sizes = water_sizes
            ops += [L.op_other_system, L.op_other_system]
        if which == "npt":
            ops += [L.op_restraints, L.op_restraints, L.op_constraints, L.op_constraints, L.op_barostat, L.op_barostat]
            ops.remove(L.op_box); ops.remove(L.op_molecules)      # (the barostat owns the box; it and the constraint groups need the molecule tables)
        L.md.eval_forces()
        try:
            for k in range(nops):
                op = rnd.choice(ops)
                try:
                    L.log.append(op())
                except martini.DdcmiError as ex:
                    L.log.append("%s -> REFUSED %s" % (op.__name__, str(ex)[:100]))
                    if "unstable" in str(ex) or "non-finite" in str(ex):
                        raise
                    L.md.eval_forces() if "needs forces" in str(ex) else None
                if VERBOSE: print("   ", L.log[-1], flush=True)
                if k % 4 == 3 or k == nops - 1:
                    err, parts, es = L.check()
                    worst = max(worst, err) if err < 1e-9 else worst
                    L.log.append("check %.1e" % err)
                    if VERBOSE: print("   ", L.log[-1], flush=True)
                    if not err < 1e-9:
                        # a list on a fixed cadence (updateRate > 0) goes stale when beads outrun the skin between two rebuilds -- the reference's lists do too;
                        # a context that agrees with the fresh one after an explicit rebuild had such a list, not a setter that forgot to invalidate something
                        err2 = L.check(rebuild_first=True)[0]
                        if err2 < 1e-9 and L.s.updateRate > 0:
                            stale += 1
                            print("sequence %d (%s): a list gone stale within its cadence (updateRate %d, deltaR %.2f): %.1e before, %.1e after an explicit rebuild; after: %s"
                                  % (q, which, L.s.updateRate, L.s.deltaR, err, err2, " | ".join(L.log[-6:])), flush=True)
                            continue
                        bad += 1
                        print("sequence %d (%s): the context and a fresh one differ by %.2e (forces %.1e, sums %.1e, virial %.1e) after:" % ((q, which, err) + parts))
                        for l in L.log[-12:]: print("      " + l)
                        print("      context:", {k: "%.10g" % v for k, v in es[0].items()})
                        print("      fresh  :", {k: "%.10g" % v for k, v in es[1].items()})
                        break
            else:
                print("sequence %d (%s): %d calls, %d checks, worst difference to a fresh context %.1e" % (q, which, nops, sum(l.startswith("check") for l in L.log), max(float(l.split()[1]) for l in L.log if l.startswith("check"))), flush=True)
        except martini.DdcmiError as ex:
            print("sequence %d (%s): ended by the library: %s   after: %s" % (q, which, str(ex)[:160], " | ".join(L.log[-5:])), flush=True)
        finally:
            L.md.close()
    print("%d sequences of %d calls: %d differ from a fresh context (+ %d checks that met a list gone stale within its fixed cadence); worst difference of the others %.1e; %d replays of 3-23 steps beside a fresh context, worst %.1e"
          % (nseq, nops, bad, stale, worst, len(REPLAYS), max(REPLAYS) if REPLAYS else 0.0))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
