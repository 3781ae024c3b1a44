"""Python driver of the C-ABI (include/ddcmi.h) used by tests and bench.py.

Mirrors the call order of ddcMD's own plugin glue: accelerator_init ->
martini_parms (set_* calls) -> sendGPUState (upload) -> firstEnergyCall
(eval_forces) -> eval_integrator (step_nglf).  No computation happens in Python.
"""
import ctypes
import numpy as np
from . import _lib

E_NAMES = ("lj", "ele", "bond", "angle", "tors", "impr", "total", "restraint")
POS, VEL, FORCE = 1, 2, 4
_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)
_up = ctypes.POINTER(ctypes.c_uint64)
_MOLRES = np.uint64(0xFFFFFFFFFFFF0000)


class DdcmiError(RuntimeError):
    pass


def _d(a):
    return a.ctypes.data_as(_dp) if a is not None else None


def _i(a):
    return a.ctypes.data_as(_ip) if a is not None else None


def _declare(lib):
    if getattr(lib, "_ddcmi_declared", False):
        return
    vp = ctypes.c_void_p
    lib.ddcmi_create.argtypes = [ctypes.POINTER(vp), ctypes.c_int]
    lib.ddcmi_destroy.argtypes = [vp]
    lib.ddcmi_destroy.restype = None
    lib.ddcmi_last_error.argtypes = [vp]
    lib.ddcmi_last_error.restype = ctypes.c_char_p
    lib.ddcmi_version.restype = ctypes.c_char_p
    lib.ddcmi_set_box.argtypes = [vp, _dp, ctypes.c_int]
    lib.ddcmi_set_species.argtypes = [vp, ctypes.c_int, _dp, _dp, _ip, _ip]
    lib.ddcmi_set_nonbonded.argtypes = [vp, ctypes.c_int, _dp, _dp, _dp, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double]
    lib.ddcmi_set_molecules.argtypes = [vp, ctypes.c_int, _ip, _ip, _ip, _ip]
    lib.ddcmi_set_bonded.argtypes = [vp, ctypes.c_int, _ip, _dp, _dp, ctypes.c_int, _ip, _ip, _dp, _dp,
                                     ctypes.c_int, _ip, _ip, _ip, _dp, _dp, ctypes.c_int]
    lib.ddcmi_set_bonded_gid.argtypes = [vp, ctypes.c_int, _up, _dp, _dp, ctypes.c_int, _up, _ip, _dp, _dp,
                                         ctypes.c_int, _up, _ip, _ip, _dp, _dp, ctypes.c_int]
    lib.ddcmi_set_neighbor.argtypes = [vp, ctypes.c_double, ctypes.c_int]
    lib.ddcmi_set_groups.argtypes = [vp, ctypes.c_int, _ip, _dp, _dp, _ip]
    lib.ddcmi_set_group_vcm.argtypes = [vp, ctypes.c_int, _dp]
    lib.ddcmi_set_group_temperature.argtypes = [vp, ctypes.c_int, ctypes.c_double]
    lib.ddcmi_set_clock.argtypes = [vp, ctypes.c_int64, ctypes.c_double]
    lib.ddcmi_set_random.argtypes = [vp, ctypes.c_uint64]
    lib.ddcmi_set_barostat.argtypes = [vp, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double]
    lib.ddcmi_get_box.argtypes = [vp, _dp]
    lib.ddcmi_get_barostat_pressure.argtypes = [vp, _dp]
    lib.ddcmi_set_barostat_isotropic.argtypes = [vp, ctypes.c_int]
    lib.ddcmi_set_molecule_lists.argtypes = [vp, ctypes.c_long, ctypes.c_int, _ip, _ip]
    lib.ddcmi_set_constraints.argtypes = [vp, ctypes.c_int, _ip, _ip, _ip, _dp]
    lib.ddcmi_set_constraints_gid.argtypes = [vp, ctypes.c_int, _ip, _up, _up, _dp]
    lib.ddcmi_set_molecule_lists_gid.argtypes = [vp, ctypes.c_long, ctypes.c_int, _ip, _up, _dp]
    lib.ddcmi_constraint_stats.argtypes = [vp, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.c_int]
    lib.ddcmi_set_restraints.argtypes = [vp, ctypes.c_int, _up, _ip, _dp, _dp, ctypes.c_int]
    lib.ddcmi_get_clock.argtypes = [vp, ctypes.POINTER(ctypes.c_int64), _dp]
    lib.ddcmi_upload_state.argtypes = [vp, ctypes.c_int, _dp, _dp, _dp, _dp, _dp, _dp, _up, _ip, _ip]
    lib.ddcmi_download_state.argtypes = [vp, ctypes.c_int] + [_dp] * 9
    lib.ddcmi_nlocal.argtypes = [vp]
    lib.ddcmi_build_list.argtypes = [vp]
    lib.ddcmi_eval_forces.argtypes = [vp, _dp, _dp]
    lib.ddcmi_step_nglf.argtypes = [vp, ctypes.c_double, ctypes.c_int]
    lib.ddcmi_get_energies.argtypes = [vp, _dp, _dp, _dp, _dp]
    lib.ddcmi_kinetic.argtypes = [vp, _dp, _dp]
    lib.ddcmi_group_temperatures.argtypes = [vp, _dp]
    lib.ddcmi_sync.argtypes = [vp]
    lib.ddcmi_list_stats.argtypes = [vp, ctypes.POINTER(ctypes.c_int64)]
    lib.ddcmi_get_list.argtypes = [vp, ctypes.c_int, _ip, _ip, ctypes.POINTER(ctypes.c_int64)]
    lib.ddcmi_timing_enable.argtypes = [vp, ctypes.c_int]
    lib.ddcmi_timing_read.argtypes = [vp, ctypes.POINTER(ctypes.c_int64), _dp, ctypes.c_int]
    lib.ddcmi_stream.argtypes = [vp]
    lib.ddcmi_stream.restype = vp
    lib.ddcmi_comm_unique_id.argtypes = [ctypes.c_char_p]
    lib.ddcmi_comm_init.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.ddcmi_comm_allreduce_sum.argtypes = [vp, _dp, ctypes.c_int]
    lib.ddcmi_comm_init_host.argtypes = [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    # process rendezvous (host/rdzv.c)
    szt = ctypes.c_size_t
    lib.ddcmi_rdzv_create.argtypes = [ctypes.POINTER(vp), ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_double]
    lib.ddcmi_rdzv_destroy.argtypes = [vp]
    lib.ddcmi_rdzv_destroy.restype = None
    lib.ddcmi_rdzv_last_error.argtypes = [vp]
    lib.ddcmi_rdzv_last_error.restype = ctypes.c_char_p
    lib.ddcmi_rdzv_rank.argtypes = [vp]
    lib.ddcmi_rdzv_world.argtypes = [vp]
    lib.ddcmi_rdzv_bcast.argtypes = [vp, vp, szt, ctypes.c_int]
    lib.ddcmi_rdzv_barrier.argtypes = [vp]
    lib.ddcmi_rdzv_allreduce_f64.argtypes = [vp, _dp, ctypes.c_int, ctypes.c_int]
    lib.ddcmi_rdzv_allgather.argtypes = [vp, vp, vp, szt]
    lib.ddcmi_rdzv_exchange.argtypes = [vp, ctypes.c_int, _ip, ctypes.POINTER(vp), ctypes.POINTER(szt), ctypes.c_int, _ip, ctypes.POINTER(vp), ctypes.POINTER(szt)]
    lib.ddcmi_domain_bounds.argtypes = [vp, _dp, _dp]
    lib.ddcmi_download_particles.argtypes = [vp, ctypes.c_int, _ip, _up, _ip] + [_dp] * 9
    lib._ddcmi_declared = True


def default_port_file():
    """Where rank 0 publishes its port when the launcher keeps MASTER_PORT for itself
    (torch.distributed.run's store listens there): one name per launch -- all workers of a
    launch share the launcher as parent."""
    import os
    import tempfile
    key = "%s_%s_%s_%d" % (os.environ.get("MASTER_ADDR", "127.0.0.1"), os.environ.get("MASTER_PORT", "0"),
                           os.environ.get("TORCHELASTIC_RUN_ID", "none"), os.getppid())
    return os.environ.get("DDCMI_RDZV_FILE") or os.path.join(tempfile.gettempdir(), "ddcmi_rdzv_" + key.replace("/", "_"))


class Rendezvous(object):
    """Ranks of one launch meeting over TCP (host/rdzv.c): what ddcMD takes from MPI
    (rank/size, Bcast, Barrier, Allreduce) for launches that only provide
    RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT.  No torch, no MPI."""

    def __init__(self, rank, world, addr="127.0.0.1", port=0, port_file=None, timeout=300.0):
        self.lib = _lib.load_library()
        _declare(self.lib)
        self.h = ctypes.c_void_p()
        pf = port_file.encode() if port_file else None
        rc = self.lib.ddcmi_rdzv_create(ctypes.byref(self.h), int(rank), int(world), addr.encode(), int(port), pf, float(timeout))
        if rc != 0:
            raise DdcmiError("ddcmi_rdzv_create failed (%d): %s" % (rc, self.lib.ddcmi_rdzv_last_error(None).decode()))
        self.rank, self.world = int(rank), int(world)

    @classmethod
    def from_env(cls, timeout=300.0):
        """RANK / WORLD_SIZE / MASTER_ADDR from the environment; the port through a file in the temporary
        directory (MASTER_PORT itself belongs to the launcher) unless DDCMI_RDZV_PORT names a free one"""
        import os
        rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
        addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
        port = int(os.environ.get("DDCMI_RDZV_PORT", "0"))
        return cls(rank, world, addr, port, None if port > 0 else default_port_file(), timeout)

    def _chk(self, rc):
        if rc != 0:
            raise DdcmiError("rendezvous error %d: %s" % (rc, self.lib.ddcmi_rdzv_last_error(self.h).decode()))

    def close(self):
        if self.h:
            self.lib.ddcmi_rdzv_destroy(self.h)
            self.h = ctypes.c_void_p()

    def bcast(self, data, root=0):
        """bytes in (significant on root), bytes out; the length must agree on all ranks"""
        buf = ctypes.create_string_buffer(bytes(data), len(data))
        self._chk(self.lib.ddcmi_rdzv_bcast(self.h, ctypes.cast(buf, ctypes.c_void_p), len(data), int(root)))
        return buf.raw

    def barrier(self):
        self._chk(self.lib.ddcmi_rdzv_barrier(self.h))

    def allreduce(self, values, op="sum"):
        v = np.ascontiguousarray(values, dtype=np.float64).copy()
        self._chk(self.lib.ddcmi_rdzv_allreduce_f64(self.h, _d(v), v.size, 0 if op == "sum" else 1))
        return v

    def allgather(self, arr):
        a = np.ascontiguousarray(arr)
        out = np.zeros((self.world,) + a.shape, a.dtype)
        self._chk(self.lib.ddcmi_rdzv_allgather(self.h, a.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), a.nbytes))
        return out

    def exchange(self, sends, recvs):
        """sends: [(peer, ndarray)], recvs: [(peer, ndarray to fill)]; matched per peer in order"""
        vp, szt = ctypes.c_void_p, ctypes.c_size_t
        sends = [(p, np.ascontiguousarray(a)) for p, a in sends]
        ns, nr = len(sends), len(recvs)
        sp = (ctypes.c_int * max(ns, 1))(*[p for p, _ in sends])
        sb = (vp * max(ns, 1))(*[a.ctypes.data for _, a in sends])
        sz = (szt * max(ns, 1))(*[a.nbytes for _, a in sends])
        rp = (ctypes.c_int * max(nr, 1))(*[p for p, _ in recvs])
        rb = (vp * max(nr, 1))(*[a.ctypes.data for _, a in recvs])
        rz = (szt * max(nr, 1))(*[a.nbytes for _, a in recvs])
        self._chk(self.lib.ddcmi_rdzv_exchange(self.h, ns, sp, sb, sz, nr, rp, rb, rz))


def plan_recv_counts(grid, rank, pbc, all_counts, loopback=False):
    """recv_cnt[27] from the all-gathered send counts [nranks, 27] (host logic of libddcmi; test API: libddcmi_test.so)"""
    lib = _lib.load_test_library()
    _declare(lib)
    _declare_domains(lib)
    ac = np.ascontiguousarray(all_counts, dtype=np.int32)
    out = np.zeros(27, np.int32)
    rc = lib.ddcmi_plan_recv_counts(grid[0], grid[1], grid[2], int(rank), int(pbc), int(loopback), _i(ac), _i(out))
    if rc != 0:
        raise DdcmiError("ddcmi_plan_recv_counts failed: %d" % rc)
    return out


def plan_halo_layout(grid, rank, pbc, send_cnt, recv_cnt, loopback=False):
    """(send_off[28], recv_off[28], send messages [(peer, off, cnt)], receive messages) of the per-step halo
    exchange: libddcmi's own peer-major layout (ddcmi_multigpu.inl plan_halo_layout; test API: libddcmi_test.so)"""
    lib = _lib.load_test_library()
    _declare(lib)
    _declare_domains(lib)
    sc, rcn = np.ascontiguousarray(send_cnt, dtype=np.int32), np.ascontiguousarray(recv_cnt, dtype=np.int32)
    so, ro = np.zeros(28, np.int32), np.zeros(28, np.int32)
    ms, mr = np.zeros(1 + 3 * 27, np.int32), np.zeros(1 + 3 * 27, np.int32)
    rc = lib.ddcmi_plan_halo_layout(grid[0], grid[1], grid[2], int(rank), int(pbc), int(loopback), _i(sc), _i(rcn), _i(so), _i(ro), _i(ms), _i(mr))
    if rc != 0:
        raise DdcmiError("ddcmi_plan_halo_layout failed: %d" % rc)
    unpack = lambda m: [tuple(int(x) for x in m[1 + 3 * k:4 + 3 * k]) for k in range(int(m[0]))]
    return so, ro, unpack(ms), unpack(mr)


def expand_bonded_terms(s):
    """Residue-relative bonded tables -> term lists over caller-order atom indices.

    Follows charmmResidues (bioCharmmCovalent.c:48-93): sort atoms by gid, cut
    residue runs at changes of gid & molResMask; term atoms are offsets into the run.
    """
    n = s.natoms
    empty_i = np.zeros(0, np.int32)
    out = {"bond_ij": empty_i, "bond_kb": np.zeros(0), "bond_b0": np.zeros(0),
           "angle_ijk": empty_i, "angle_func": empty_i, "angle_k": np.zeros(0), "angle_t0": np.zeros(0),
           "tors_ijkl": empty_i, "tors_func": empty_i, "tors_n": empty_i, "tors_k": np.zeros(0), "tors_delta": np.zeros(0)}
    if s.nresi == 0 or (s.bond_off[-1] == 0 and s.angle_off[-1] == 0 and s.tors_off[-1] == 0):
        return out
    order = np.argsort(s.gid, kind="stable")
    key = s.gid[order] & _MOLRES
    first = np.flatnonzero(np.concatenate(([True], key[1:] != key[:-1])))
    cnt = np.diff(np.concatenate((first, [n])))
    rt = s.resitype[s.species[order[first]]]
    if np.any(cnt != s.resi_natoms[rt]):
        raise DdcmiError("incomplete residue in the particle set")
    bi, bk, b0 = [], [], []
    ai, af, ak, a0 = [], [], [], []
    ti, tf, tn, tk, td = [], [], [], [], []
    for r in range(s.nresi):
        starts = first[rt == r]
        if starts.size == 0:
            continue
        b = slice(s.bond_off[r], s.bond_off[r + 1])
        if b.stop > b.start:
            I = order[starts[:, None] + s.bondI[b][None, :]]
            J = order[starts[:, None] + s.bondJ[b][None, :]]
            bi.append(np.stack((I, J), axis=-1).reshape(-1, 2))
            bk.append(np.tile(s.bond_kb[b], starts.size))
            b0.append(np.tile(s.bond_b0[b], starts.size))
        a = slice(s.angle_off[r], s.angle_off[r + 1])
        if a.stop > a.start:
            idx = [order[starts[:, None] + getattr(s, k)[a][None, :]] for k in ("angleI", "angleJ", "angleK")]
            ai.append(np.stack(idx, axis=-1).reshape(-1, 3))
            af.append(np.tile(s.angle_func[a], starts.size))
            ak.append(np.tile(s.angle_k[a], starts.size))
            a0.append(np.tile(s.angle_t0[a], starts.size))
        t = slice(s.tors_off[r], s.tors_off[r + 1])
        if t.stop > t.start:
            idx = [order[starts[:, None] + getattr(s, k)[t][None, :]] for k in ("torsI", "torsJ", "torsK", "torsL")]
            ti.append(np.stack(idx, axis=-1).reshape(-1, 4))
            tf.append(np.tile(s.tors_func[t], starts.size))
            tn.append(np.tile(s.tors_n[t], starts.size))
            tk.append(np.tile(s.tors_k[t], starts.size))
            td.append(np.tile(s.tors_delta[t], starts.size))

    def cat(lst, dt):
        return np.ascontiguousarray(np.concatenate(lst), dtype=dt) if lst else np.zeros(0, dt)

    out.update(bond_ij=cat(bi, np.int32).ravel(), bond_kb=cat(bk, np.float64), bond_b0=cat(b0, np.float64),
               angle_ijk=cat(ai, np.int32).ravel(), angle_func=cat(af, np.int32), angle_k=cat(ak, np.float64), angle_t0=cat(a0, np.float64),
               tors_ijkl=cat(ti, np.int32).ravel(), tors_func=cat(tf, np.int32), tors_n=cat(tn, np.int32),
               tors_k=cat(tk, np.float64), tors_delta=cat(td, np.float64))
    return out


def expand_constraints(s):
    """Residue-relative constraint lists -> constraint groups over caller-order atom indices.

    One group per CONSLISTPARMS of every residue instance (genConstraint, bioMartini.c:300-445),
    pairs in deck order.  Returns (pair_off, pairI, pairJ, dist)."""
    n = s.natoms
    cons_off = np.asarray(getattr(s, "cons_off", np.zeros(1, np.int32)))
    if s.nresi == 0 or cons_off[-1] == 0:
        return np.zeros(1, np.int32), np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0)
    order = np.argsort(s.gid, kind="stable")
    key = s.gid[order] & _MOLRES
    first = np.flatnonzero(np.concatenate(([True], key[1:] != key[:-1])))
    cnt = np.diff(np.concatenate((first, [n])))
    rt = s.resitype[s.species[order[first]]]
    if np.any(cnt != s.resi_natoms[rt]):
        raise DdcmiError("incomplete residue in the particle set")
    counts, pi, pj, dd = [], [], [], []
    for r in range(s.nresi):
        starts = first[rt == r]
        if starts.size == 0 or cons_off[r + 1] == cons_off[r]:
            continue
        sl = np.arange(cons_off[r], cons_off[r + 1])
        grp = s.cons_grp[sl]
        for c in dict.fromkeys(grp.tolist()):
            k = sl[grp == c]
            I = order[starts[:, None] + s.consI[k][None, :]]
            J = order[starts[:, None] + s.consJ[k][None, :]]
            pi.append(I.ravel()); pj.append(J.ravel())
            dd.append(np.tile(s.cons_r0[k], starts.size))
            counts.append(np.full(starts.size, k.size, np.int64))
    counts = np.concatenate(counts)
    pair_off = np.concatenate(([0], np.cumsum(counts))).astype(np.int32)
    return (pair_off, np.ascontiguousarray(np.concatenate(pi), dtype=np.int32), np.ascontiguousarray(np.concatenate(pj), dtype=np.int32),
            np.ascontiguousarray(np.concatenate(dd), dtype=np.float64))


def molecule_lists(s):
    """Molecules = runs of equal gid & molMask.  Returns (nmol_total, mol_off, mol_atoms) with only the
    molecules of two or more beads listed (caller-order atom indices)."""
    n = s.natoms
    order = np.argsort(s.gid, kind="stable")
    key = np.asarray(s.gid, dtype=np.uint64)[order] >> np.uint64(32)
    first = np.flatnonzero(np.concatenate(([True], key[1:] != key[:-1])))
    cnt = np.diff(np.concatenate((first, [n])))
    multi = cnt >= 2
    sel = np.repeat(multi, cnt)
    mol_off = np.concatenate(([0], np.cumsum(cnt[multi]))).astype(np.int32)
    return int(first.size), mol_off, np.ascontiguousarray(order[sel], dtype=np.int32)


class MartiniHIP(object):
    """One device context running the Martini hot path for a Setup."""

    def __init__(self, setup, device=0, upload=True, bonded_by_gid=False, constraints=False, test_api=False):
        # test_api: the context lives in libddcmi_test.so (same objects + the entry points of include/ddcmi_test.h)
        self.lib = _lib.load_test_library() if test_api else _lib.load_library()
        _declare(self.lib)
        self.s = setup
        self.ctx = ctypes.c_void_p()
        rc = self.lib.ddcmi_create(ctypes.byref(self.ctx), int(device))
        if rc != 0:
            raise DdcmiError("ddcmi_create failed (%d): %s" % (rc, self.lib.ddcmi_last_error(None).decode()))
        s = setup
        f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        self._chk(self.lib.ddcmi_set_box(self.ctx, _d(f64(s.h)), int(s.pbc)))
        self._chk(self.lib.ddcmi_set_species(self.ctx, s.nspecies, _d(f64(s.mass)), _d(f64(s.charge)), _i(i32(s.ljtype)), _i(i32(s.moltype))))
        self._chk(self.lib.ddcmi_set_nonbonded(self.ctx, s.nlj, _d(f64(s.sigma)), _d(f64(s.eps)), _d(f64(s.shift)),
                                               s.rmax, s.keR, s.krf, s.crf))
        if s.nmoltype > 0:
            bi, bj = i32(s.bpairI), i32(s.bpairJ)
            if bi.size == 0:
                bi = bj = np.zeros(1, np.int32)
            self._chk(self.lib.ddcmi_set_molecules(self.ctx, s.nmoltype, _i(i32(s.mol_nspecies)), _i(i32(s.bpair_off)), _i(bi), _i(bj)))
        else:
            self._chk(self.lib.ddcmi_set_molecules(self.ctx, 0, None, None, None, None))
        t = self.terms = expand_bonded_terms(s)
        nb, na, nt = t["bond_kb"].size, t["angle_k"].size, t["tors_k"].size
        if bonded_by_gid:
            # decomposed runs: every rank holds the global term list, atoms named by gid
            gids = np.asarray(s.gid, dtype=np.uint64)
            g = lambda idx: np.ascontiguousarray(gids[np.asarray(idx, dtype=np.int64)], dtype=np.uint64)
            self._term_gids = (g(t["bond_ij"]), g(t["angle_ijk"]), g(t["tors_ijkl"]))
            u = lambda a: a.ctypes.data_as(_up)
            self._chk(self.lib.ddcmi_set_bonded_gid(self.ctx, nb, u(self._term_gids[0]), _d(t["bond_kb"]), _d(t["bond_b0"]),
                                                    na, u(self._term_gids[1]), _i(t["angle_func"]), _d(t["angle_k"]), _d(t["angle_t0"]),
                                                    nt, u(self._term_gids[2]), _i(t["tors_func"]), _i(t["tors_n"]), _d(t["tors_k"]), _d(t["tors_delta"]),
                                                    int(s.excludePotentialTerm)))
        else:
            self._chk(self.lib.ddcmi_set_bonded(self.ctx, nb, _i(t["bond_ij"]), _d(t["bond_kb"]), _d(t["bond_b0"]),
                                                na, _i(t["angle_ijk"]), _i(t["angle_func"]), _d(t["angle_k"]), _d(t["angle_t0"]),
                                                nt, _i(t["tors_ijkl"]), _i(t["tors_func"]), _i(t["tors_n"]), _d(t["tors_k"]), _d(t["tors_delta"]),
                                                int(s.excludePotentialTerm)))
        self._chk(self.lib.ddcmi_set_neighbor(self.ctx, s.deltaR, int(s.updateRate)))
        gt = i32(np.where(np.isin(np.asarray(s.group_type), (1, 2)), np.asarray(s.group_type), 0))     # FREE / BERENDSEN / LANGEVIN
        self._chk(self.lib.ddcmi_set_groups(self.ctx, s.ngroup, _i(gt), _d(f64(s.group_Teq)), _d(f64(s.group_tau)), _i(i32(s.group_interval))))
        vcm = getattr(s, "group_vcm", None)
        if vcm is not None and np.any(np.asarray(vcm) != 0.0):      # LANGEVIN groups: the velocity the friction relaxes towards (langevin.c:167)
            self._vcm = f64(np.ravel(vcm))
            self._chk(self.lib.ddcmi_set_group_vcm(self.ctx, s.ngroup, _d(self._vcm)))
        self._chk(self.lib.ddcmi_set_random(self.ctx, int(getattr(s, "rng_seed", 0))))
        if float(getattr(s, "npt_beta", 0.0)) > 0.0:      # INTEGRATOR type=NGLFCONSTRAINT: barostat on the molecular pressure
            self.set_barostat(float(s.npt_T), float(s.npt_P0), float(s.npt_beta), float(s.npt_tau), isotropic=bool(getattr(s, "npt_isotropic", 0)),
                              by_gid=bonded_by_gid)
        if constraints:                                   # INTEGRATOR type=NGLFCONSTRAINT: velocity constraints
            self._cons = expand_constraints(s)
            po, pi, pj, dd = self._cons
            if bonded_by_gid:                             # decomposed runs: the groups' atoms named by gid
                gids = np.asarray(s.gid, dtype=np.uint64)
                self._cons_gid = (np.ascontiguousarray(gids[pi]), np.ascontiguousarray(gids[pj]))
                self._chk(self.lib.ddcmi_set_constraints_gid(self.ctx, int(po.size - 1), _i(po), self._cons_gid[0].ctypes.data_as(_up),
                                                             self._cons_gid[1].ctypes.data_as(_up), _d(dd)))
            else:
                self._chk(self.lib.ddcmi_set_constraints(self.ctx, int(po.size - 1), _i(po), _i(pi), _i(pj), _d(dd)))
        nrest = int(getattr(s, "nrest", 0))
        if nrest > 0:     # RESTRAINT potential
            self._rest = (np.ascontiguousarray(s.rest_gid, dtype=np.uint64), i32(np.asarray(s.rest_fc).ravel()),
                          f64(np.asarray(s.rest_r0).ravel()), f64(s.rest_kb))
            self._chk(self.lib.ddcmi_set_restraints(self.ctx, nrest, self._rest[0].ctypes.data_as(_up), _i(self._rest[1]), _d(self._rest[2]),
                                                    _d(self._rest[3]), int(getattr(s, "rest_origin", 0))))
        self._chk(self.lib.ddcmi_set_clock(self.ctx, int(s.loop), float(s.time)))
        self.n = s.natoms
        if upload:
            self.upload(s.rx, s.ry, s.rz, s.vx, s.vy, s.vz)

    def _chk(self, rc):
        if rc != 0:
            raise DdcmiError("ddcmi error %d: %s" % (rc, self.lib.ddcmi_last_error(self.ctx).decode()))

    def close(self):
        if self.ctx:
            self.lib.ddcmi_destroy(self.ctx)
            self.ctx = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, rx, rz_or_ry, rz=None, vx=None, vy=None, vz=None):
        s = self.s
        ry = rz_or_ry
        f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        a = [f64(x) for x in (rx, ry, rz)]
        v = [f64(x) for x in (vx, vy, vz)] if vx is not None else [None, None, None]
        gid = np.ascontiguousarray(s.gid, dtype=np.uint64)
        sp = np.ascontiguousarray(s.species, dtype=np.int32)
        gr = np.ascontiguousarray(s.group, dtype=np.int32)
        self._chk(self.lib.ddcmi_upload_state(self.ctx, self.n, _d(a[0]), _d(a[1]), _d(a[2]), _d(v[0]), _d(v[1]), _d(v[2]),
                                              gid.ctypes.data_as(_up), _i(sp), _i(gr)))
        lcg = getattr(s, "lcg64", None)
        if lcg is not None and type(self) is MartiniHIP and len(lcg) == self.n and np.any(np.asarray(s.group_type) == 2):
            self.set_random_lcg64(lcg)      # RANDOM type=LCG64 in the deck: the Langevin groups of a one-domain run draw from the particles' own streams

    def build_list(self):
        self._chk(self.lib.ddcmi_build_list(self.ctx))

    def eval_forces(self):
        e = np.zeros(8)
        v = np.zeros(6)
        self._chk(self.lib.ddcmi_eval_forces(self.ctx, _d(e), _d(v)))
        return dict(zip(E_NAMES, e.tolist())), v

    def step(self, nsteps=1, dt=None):
        self._chk(self.lib.ddcmi_step_nglf(self.ctx, float(self.s.dt if dt is None else dt), int(nsteps)))

    def energies(self):
        e, v, t = np.zeros(8), np.zeros(6), np.zeros(6)
        rk = ctypes.c_double(0)
        self._chk(self.lib.ddcmi_get_energies(self.ctx, _d(e), _d(v), ctypes.byref(rk), _d(t)))
        return dict(zip(E_NAMES, e.tolist())), v, rk.value, t

    def lean_history(self):
        """(test_api) sums of the lean steps formed by the last batch launch: [nsteps, 32] = pair sums {lj, ele, vir6} as the full list counts
        them (twice), kinetic sums {rk, tion6}, 0, bonded sums {bond, angle, tors, impr, vir6}, zeros -- include/ddcmi_test.h"""
        out = np.zeros((32, 32))
        nst = ctypes.c_int(0)
        self.lib.ddcmi_debug_lean_history.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), _dp]
        self._chk(self.lib.ddcmi_debug_lean_history(self.ctx, ctypes.byref(nst), _d(out)))
        return out[:nst.value].copy()

    def kinetic(self):
        t = np.zeros(6)
        rk = ctypes.c_double(0)
        self._chk(self.lib.ddcmi_kinetic(self.ctx, ctypes.byref(rk), _d(t)))
        return rk.value, t

    def set_random_lcg64(self, parms):
        """RANDOM type LCG64: the particles' own streams, records {state, multID, prime} (lcg64.h:8-12) in upload order; None clears them"""
        u32p = ctypes.POINTER(ctypes.c_uint32)
        self.lib.ddcmi_set_random_lcg64.argtypes = [ctypes.c_void_p, ctypes.c_int, _up, u32p, u32p]
        if parms is None:
            self._lcg_set = False
            return self._chk(self.lib.ddcmi_set_random_lcg64(self.ctx, 0, None, None, None))
        st = np.ascontiguousarray(parms["state"], dtype=np.uint64)
        mu = np.ascontiguousarray(parms["multID"], dtype=np.uint32)
        pr = np.ascontiguousarray(parms["prime"], dtype=np.uint32)
        self._chk(self.lib.ddcmi_set_random_lcg64(self.ctx, int(st.size), st.ctypes.data_as(_up), mu.ctypes.data_as(u32p), pr.ctypes.data_as(u32p)))
        self._lcg_set = True

    def get_random_lcg64(self):
        u32p = ctypes.POINTER(ctypes.c_uint32)
        self.lib.ddcmi_get_random_lcg64.argtypes = [ctypes.c_void_p, ctypes.c_int, _up, u32p, u32p]
        n = self.n
        out = np.zeros(n, dtype=[("state", "<u8"), ("multID", "<u4"), ("prime", "<u4")])
        st, mu, pr = np.zeros(n, np.uint64), np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        self._chk(self.lib.ddcmi_get_random_lcg64(self.ctx, n, st.ctypes.data_as(_up), mu.ctypes.data_as(u32p), pr.ctypes.data_as(u32p)))
        out["state"], out["multID"], out["prime"] = st, mu, pr
        return out

    def kinetic_detail(self, by_species):
        """per-group / per-species {rk, tion[6], mass, number, J[3]} (energy.c:104-147) of this rank's beads"""
        ncl = int(self.s.nspecies if by_species else max(1, self.s.ngroup))
        out = np.zeros((ncl, 12))
        self.lib.ddcmi_kinetic_detail.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, _dp]
        self._chk(self.lib.ddcmi_kinetic_detail(self.ctx, int(bool(by_species)), ncl, _d(out)))
        return out

    def set_barostat(self, T, P0, beta, tau, isotropic=False, by_gid=False):
        """nglfconstraint's Berendsen barostat (isotropic: NGLFGPULANGEVIN's); the molecule lists feed its molecular virial
        (by_gid: atoms named by gid + the molecules' masses, for decomposed runs)"""
        nmol, off, atoms = self._mols = molecule_lists(self.s)
        if by_gid:
            gids = np.ascontiguousarray(np.asarray(self.s.gid, dtype=np.uint64)[atoms]) if atoms.size else np.zeros(1, np.uint64)
            m_at = np.asarray(self.s.mass, dtype=np.float64)[np.asarray(self.s.species)[atoms]] if atoms.size else np.zeros(0)
            mtot = np.ascontiguousarray(np.add.reduceat(m_at, off[:-1])) if off.size > 1 else np.zeros(1)
            self._mol_gid = (gids, mtot)
            self._chk(self.lib.ddcmi_set_molecule_lists_gid(self.ctx, nmol, int(off.size - 1), _i(off), gids.ctypes.data_as(_up), _d(mtot)))
        else:
            self._chk(self.lib.ddcmi_set_molecule_lists(self.ctx, nmol, int(off.size - 1), _i(off), _i(atoms if atoms.size else np.zeros(1, np.int32))))
        self._chk(self.lib.ddcmi_set_barostat(self.ctx, float(T), float(P0), float(beta), float(tau)))
        self._chk(self.lib.ddcmi_set_barostat_isotropic(self.ctx, int(bool(isotropic))))

    def constraint_stats(self, reset=True):
        a, b = ctypes.c_int(0), ctypes.c_int(0)
        self._chk(self.lib.ddcmi_constraint_stats(self.ctx, ctypes.byref(a), ctypes.byref(b), int(reset)))
        return a.value, b.value

    def barostat_pressure(self):
        p = np.zeros(3)
        self._chk(self.lib.ddcmi_get_barostat_pressure(self.ctx, _d(p)))
        return p

    def box(self):
        h = np.zeros(9)
        self._chk(self.lib.ddcmi_get_box(self.ctx, _d(h)))
        return h[[0, 4, 8]]

    def group_temperatures(self):
        T = np.zeros(max(1, self.s.ngroup))
        self._chk(self.lib.ddcmi_group_temperatures(self.ctx, _d(T)))
        return T

    def download(self, mask=POS | VEL | FORCE):
        n = self.n
        out = [np.zeros(n) for _ in range(9)]
        self._chk(self.lib.ddcmi_download_state(self.ctx, int(mask), *[_d(a) for a in out]))
        return {"r": out[0:3], "v": out[3:6], "f": out[6:9]}

    def upload_positions(self, r, v=None):
        """new positions (and velocities) of the same beads in caller order: sendGPUState of the host-integrator mode"""
        self.lib.ddcmi_upload_positions.argtypes = [ctypes.c_void_p] + [_dp] * 6
        vv = v if v is not None else (None, None, None)
        self._chk(self.lib.ddcmi_upload_positions(self.ctx, _d(r[0]), _d(r[1]), _d(r[2]), _d(vv[0]), _d(vv[1]), _d(vv[2])))

    def sync(self):
        self._chk(self.lib.ddcmi_sync(self.ctx))

    def clock(self):
        loop = ctypes.c_int64(0)
        t = ctypes.c_double(0)
        self._chk(self.lib.ddcmi_get_clock(self.ctx, ctypes.byref(loop), ctypes.byref(t)))
        return loop.value, t.value

    def list_stats(self):
        st = (ctypes.c_int64 * 8)()
        self._chk(self.lib.ddcmi_list_stats(self.ctx, st))
        return {"entries": st[0], "excluded": st[1], "ell_width": st[2], "images": st[3], "cells": st[4], "rebuilds": st[5], "npad": st[6]}

    def comm_stats(self):
        st = (ctypes.c_int64 * 8)()
        self.lib.ddcmi_comm_stats.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64)]
        self._chk(self.lib.ddcmi_comm_stats(self.ctx, st))
        v = int(st[4])
        return {"send_beads": int(st[0]), "recv_beads": int(st[1]), "send_msgs": int(st[2]), "recv_msgs": int(st[3]),
                "rccl_version": "%d.%d.%d" % (v // 10000, (v // 100) % 100, v % 100) if v else None,
                "transport": ("none", "rccl", "host", "rccl-loopback")[int(st[5])], "ranks": int(st[6]), "rank": int(st[7])}

    def comm_peers(self):
        """[(peer rank, beads sent per step, beads received per step)] of the per-step halo exchange as laid out at the last rebuild"""
        i64p = ctypes.POINTER(ctypes.c_int64)
        self.lib.ddcmi_comm_peer_stats.argtypes = [ctypes.c_void_p, ctypes.c_int, _ip, i64p, i64p]
        peer = np.zeros(27, np.int32)
        sb, rb = np.zeros(27, np.int64), np.zeros(27, np.int64)
        n = int(self.lib.ddcmi_comm_peer_stats(self.ctx, 27, _i(peer), sb.ctypes.data_as(i64p), rb.ctypes.data_as(i64p)))
        return [(int(peer[k]), int(sb[k]), int(rb[k])) for k in range(max(n, 0))]

    def get_list(self, which=0):
        n = self.n
        tot = ctypes.c_int64(0)
        start = np.zeros(n + 1, np.int32)
        self._chk(self.lib.ddcmi_get_list(self.ctx, which, _i(start), None, ctypes.byref(tot)))
        j = np.zeros(max(1, tot.value), np.int32)
        self._chk(self.lib.ddcmi_get_list(self.ctx, which, _i(start), _i(j), ctypes.byref(tot)))
        return start, j[:tot.value]

    def timing(self, on=True):
        self._chk(self.lib.ddcmi_timing_enable(self.ctx, 1 if on else 0))

    def timing_fused(self):
        """of the last timing_read: (launches, ms) of the pair kernel with the integrator's pass as its epilogue"""
        n, ms = ctypes.c_int64(0), ctypes.c_double(0.0)
        self.lib.ddcmi_timing_fused.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64), _dp]
        self._chk(self.lib.ddcmi_timing_fused(self.ctx, ctypes.byref(n), ctypes.byref(ms)))
        return n.value, ms.value

    def timing_read(self, reset=True):
        n = ctypes.c_int64(0)
        ms = ctypes.c_double(0)
        self._chk(self.lib.ddcmi_timing_read(self.ctx, ctypes.byref(n), ctypes.byref(ms), 1 if reset else 0))
        return n.value, ms.value


def _declare_domains(lib):
    """the test-only entry points (include/ddcmi_test.h): lib is the handle of libddcmi_test.so"""
    if getattr(lib, "_ddcmi_dom_declared", False):
        return
    if not hasattr(lib, "ddcmi_group_create"):
        return      # (a handle of the product library: it has none of them -- nothing to declare)
    vp = ctypes.c_void_p
    lib.ddcmi_plan_directions.argtypes = [ctypes.c_int] * 5 + [_ip, _ip]
    lib.ddcmi_plan_recv_counts.argtypes = [ctypes.c_int] * 6 + [_ip, _ip]
    lib.ddcmi_plan_halo_layout.argtypes = [ctypes.c_int] * 6 + [_ip, _ip, _ip, _ip, _ip, _ip]
    lib.ddcmi_group_create.argtypes = [ctypes.POINTER(vp), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.ddcmi_group_destroy.argtypes = [ctypes.POINTER(vp), ctypes.c_int]
    lib.ddcmi_group_eval_forces.argtypes = [ctypes.POINTER(vp), ctypes.c_int]
    lib.ddcmi_group_step_nglf.argtypes = [ctypes.POINTER(vp), ctypes.c_int, ctypes.c_double, ctypes.c_int]
    lib.ddcmi_group_temperatures_all.argtypes = [ctypes.POINTER(vp), ctypes.c_int, _dp]
    lib._ddcmi_dom_declared = True


def plan_directions(px, py, pz, rank, pbc=7):
    """(dest[27], shift[27,3]) of the 26 neighbour directions (host logic, no GPU needed; test API: libddcmi_test.so)."""
    lib = _lib.load_test_library()
    _declare(lib)
    _declare_domains(lib)
    dest = np.zeros(27, np.int32)
    shift = np.zeros(81, np.int32)
    rc = lib.ddcmi_plan_directions(px, py, pz, rank, pbc, _i(dest), _i(shift))
    if rc != 0:
        raise DdcmiError("ddcmi_plan_directions failed: %d" % rc)
    return dest, shift.reshape(27, 3)


def domain_of(setup, grid):
    """owner rank of every bead for a px*py*pz brick decomposition (box centred on the origin)"""
    px, py, pz = grid
    L = setup.box
    out = np.zeros(setup.natoms, np.int64)
    mult = 1
    for a, (r, P) in enumerate(zip((setup.rx, setup.ry, setup.rz), (px, py, pz))):
        x = r - L[a] * np.rint(r / L[a])
        b = np.clip(np.floor((x + 0.5 * L[a]) / (L[a] / P)).astype(np.int64), 0, P - 1)
        out += mult * b
        mult *= P
    return out


def select_rank(setup, owner, rank):
    """(index array) of the beads rank owns"""
    return np.flatnonzero(owner == rank)


class DomainMixin(object):
    """gid-addressed download for decomposed runs"""

    def download_particles(self):
        cap = int(self.lib.ddcmi_nlocal(self.ctx)) + 16
        n = ctypes.c_int(0)
        gid = np.zeros(cap, np.uint64)
        sp = np.zeros(cap, np.int32)
        arr = [np.zeros(cap) for _ in range(9)]
        self._chk(self.lib.ddcmi_download_particles(self.ctx, cap, ctypes.byref(n), gid.ctypes.data_as(_up), _i(sp), *[_d(a) for a in arr]))
        k = n.value
        out = {"gid": gid[:k], "species": sp[:k], "r": [a[:k] for a in arr[0:3]], "v": [a[:k] for a in arr[3:6]], "f": [a[:k] for a in arr[6:9]]}
        if getattr(self, "_lcg_set", False) and k > 0:
            self.n = k
            out["lcg64"] = self.get_random_lcg64()      # same order: the beads this rank owns now
        return out


class MartiniRank(MartiniHIP, DomainMixin):
    """One rank of a decomposed run: uploads only the beads `index` selects."""

    def __init__(self, setup, index, device=0, constraints=False, test_api=False):
        MartiniHIP.__init__(self, setup, device=device, upload=False, bonded_by_gid=True, constraints=constraints, test_api=test_api)
        self.index = np.asarray(index)

    def upload_local(self):
        s, ix = self.s, self.index
        f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        gid = np.ascontiguousarray(s.gid[ix], dtype=np.uint64)
        sp = np.ascontiguousarray(s.species[ix], dtype=np.int32)
        gr = np.ascontiguousarray(s.group[ix], dtype=np.int32)
        a = [f64(x[ix]) for x in (s.rx, s.ry, s.rz, s.vx, s.vy, s.vz)]
        self.n = len(ix)
        self._chk(self.lib.ddcmi_upload_state(self.ctx, self.n, _d(a[0]), _d(a[1]), _d(a[2]), _d(a[3]), _d(a[4]), _d(a[5]),
                                              gid.ctypes.data_as(_up), _i(sp), _i(gr)))
        lcg = getattr(s, "lcg64", None)
        if lcg is not None and self.n > 0 and np.any(np.asarray(s.group_type) == 2):
            self.set_random_lcg64(lcg[ix])      # the records migrate with their beads from here on

    def comm_init(self, rank, nranks, uid, grid):
        self._chk(self.lib.ddcmi_comm_init(self.ctx, rank, nranks, uid, grid[0], grid[1], grid[2]))

    def comm_init_host(self, rdzv, grid):
        """decomposition over the host transport (TCP streams of the rendezvous) instead of RCCL"""
        self._rdzv = rdzv      # must outlive the context
        self._chk(self.lib.ddcmi_comm_init_host(self.ctx, rdzv.h, grid[0], grid[1], grid[2]))

    def preflight(self, timeout=60.0):
        """ddcmi_comm_preflight: one grouped exchange of a known pattern with every peer of the brick plan, one 24-double all-reduce,
        one int all-gather, verified; raises DdcmiError naming the stage / peer / direction.  Returns the report as a dict."""
        self.lib.ddcmi_comm_preflight.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.POINTER(ctypes.c_int64)]
        rep = (ctypes.c_int64 * 16)()
        rc = self.lib.ddcmi_comm_preflight(self.ctx, float(timeout), rep)
        out = {"peers": [int(rep[8 + k]) for k in range(min(int(rep[0]), 8))], "directions": int(rep[1]), "bytes_per_direction": int(rep[2]),
               "stages_verified": int(rep[6]), "elapsed_us": int(rep[7])}
        if rep[3]:
            out.update(failed_peer=int(rep[4]), failed_direction_code=int(rep[5]))
        self.preflight_report = out
        if rc != 0:
            raise DdcmiError("ddcmi error %d: %s" % (rc, self.lib.ddcmi_last_error(self.ctx).decode()))
        return out

    def allreduce(self, values):
        v = np.ascontiguousarray(values, dtype=np.float64)
        self._chk(self.lib.ddcmi_comm_allreduce_sum(self.ctx, _d(v), v.size))
        return v


class MartiniGroup(object):
    """px*py*pz domains emulated inside one process on one GPU (device copies
    instead of RCCL): exercises migration, halo tables and per-step halo refresh."""

    def __init__(self, setup, grid, device=0, constraints=False):
        self.s = setup
        self.grid = tuple(grid)
        self.n = grid[0] * grid[1] * grid[2]
        owner = domain_of(setup, grid)
        self.ranks = [MartiniRank(setup, select_rank(setup, owner, r), device=device, constraints=constraints, test_api=True) for r in range(self.n)]
        self.lib = self.ranks[0].lib
        _declare_domains(self.lib)
        self.arr = (ctypes.c_void_p * self.n)(*[r.ctx for r in self.ranks])
        rc = self.lib.ddcmi_group_create(self.arr, self.n, grid[0], grid[1], grid[2])
        if rc != 0:
            msgs = [self.lib.ddcmi_last_error(r.ctx).decode() for r in self.ranks]
            raise DdcmiError("ddcmi_group_create failed: %d %s" % (rc, "; ".join(sorted(set(m for m in msgs if m)))))
        for r in self.ranks:
            r.upload_local()

    def _chk(self, rc):
        if rc != 0:
            msgs = [self.lib.ddcmi_last_error(r.ctx).decode() for r in self.ranks]
            raise DdcmiError("ddcmi group error %d: %s" % (rc, "; ".join(m for m in msgs if m)))

    def eval_forces(self):
        self._chk(self.lib.ddcmi_group_eval_forces(self.arr, self.n))
        return self.energies()[:2]

    def step(self, nsteps=1, dt=None):
        self._chk(self.lib.ddcmi_group_step_nglf(self.arr, self.n, float(self.s.dt if dt is None else dt), int(nsteps)))

    def group_temperatures(self):
        T = np.zeros(max(1, self.s.ngroup))
        self._chk(self.lib.ddcmi_group_temperatures_all(self.arr, self.n, _d(T)))
        return T

    def energies(self):
        """sum over ranks = energyInfo.c allreduce()"""
        tot_e, tot_v, tot_rk, tot_t = None, None, 0.0, None
        for r in self.ranks:
            e, v, rk, t = r.energies()
            if tot_e is None:
                tot_e, tot_v, tot_t = dict(e), v.copy(), t.copy()
            else:
                for k in e:
                    tot_e[k] += e[k]
                tot_v += v
                tot_t += t
            tot_rk += rk
        return tot_e, tot_v, tot_rk, tot_t

    def gather(self):
        """all beads ordered by gid"""
        parts = [r.download_particles() for r in self.ranks]
        gid = np.concatenate([p["gid"] for p in parts])
        order = np.argsort(gid, kind="stable")
        out = {"gid": gid[order], "nlocal": [len(p["gid"]) for p in parts]}
        for k in ("r", "v", "f"):
            out[k] = [np.concatenate([p[k][c] for p in parts])[order] for c in range(3)]
        if all("lcg64" in p or len(p["gid"]) == 0 for p in parts) and any("lcg64" in p for p in parts):
            out["lcg64"] = np.concatenate([p["lcg64"] for p in parts if "lcg64" in p])[order]
        return out

    def close(self):
        try:
            self.lib.ddcmi_group_destroy(self.arr, self.n)
        except Exception:
            pass
        for r in self.ranks:
            r.close()
