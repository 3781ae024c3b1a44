"""CPU, world_size=2: the host side of the spatial decomposition, between two real processes.

What runs on the device in ddcmi_multigpu.inl (k_halo_select: which owned beads lie within
rmax+deltaR of a face; k_pack_halo: the sender applies the periodic shift) is re-stated in numpy here;
everything the HOST decides is libddcmi's own code, called through the C-ABI:
  * ddcmi_plan_directions   -- destination rank and shift of the 26 directions (domain.c:61-208),
  * ddcmi_plan_recv_counts  -- what a rank receives, from the all-gathered per-direction counts,
  * ddcmi_plan_halo_layout  -- the peer-major send/receive buffer layout and the ONE message per peer
                               (sseg/rseg of mg_layout_halo; ddcSendRecvTables, ddcSendRecv.c:126-225).
Transport: gloo (torch.distributed) or libddcmi's own TCP rendezvous (host/rdzv.c, what bench.py uses).
Check: owned + received halo beads give every rank its complete neighbourhood -- summed energies and
per-bead forces equal the single-rank oracle -- and every received segment lies on the side of the
domain its direction code says."""
import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _pair_terms(s, ri, ti, rj, tj, same):
    """LJ energy/forces of beads i against beads j (no periodic arithmetic: halo beads carry the shift)"""
    d = ri[:, None, :] - rj[None, :, :]
    r2 = np.sum(d * d, axis=2)
    mask = r2 < s.rmax ** 2
    if same:
        np.fill_diagonal(mask, False)
    sij = tj[None, :] + s.nlj * ti[:, None]
    sig, eps, sh = s.sigma[sij], s.eps[sij], s.shift[sij]
    r2s = np.where(mask, r2, 1.0)
    s6 = (sig * sig / r2s) ** 3
    e = np.where(mask, 4 * eps * (s6 * s6 - s6) + sh, 0.0)
    dvdr = np.where(mask, 24 * eps * (s6 - 2 * s6 * s6) / r2s, 0.0)
    f = -(dvdr[:, :, None] * d).sum(axis=1)
    return 0.5 * e.sum(), f


class _Gloo(object):
    def __init__(self, rank, world, port):
        import torch
        import torch.distributed as dist
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        self.torch, self.dist, self.world = torch, dist, world

    def allgather(self, a):
        out = [self.torch.zeros(a.shape, dtype=self.torch.int32) for _ in range(self.world)]
        self.dist.all_gather(out, self.torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)))
        return np.stack([o.numpy() for o in out])

    def exchange(self, sends, recvs):
        reqs = [self.dist.isend(self.torch.from_numpy(np.ascontiguousarray(a)), p) for p, a in sends]
        tens = [self.torch.zeros(a.shape, dtype=self.torch.float64) for _, a in recvs]
        reqs += [self.dist.irecv(t, p) for (p, _), t in zip(recvs, tens)]
        for rq in reqs:
            rq.wait()
        for (_, a), t in zip(recvs, tens):
            a[...] = t.numpy()

    def allreduce(self, v):
        t = self.torch.tensor(v, dtype=self.torch.float64)
        self.dist.all_reduce(t)
        return t.numpy()

    def close(self):
        self.dist.barrier()
        self.dist.destroy_process_group()


class _Tcp(object):
    def __init__(self, rank, world, port):
        from ddcmd_amd.martini import Rendezvous
        self.r = Rendezvous(rank, world, "127.0.0.1", port, None, timeout=120.0)

    def allgather(self, a):
        return self.r.allgather(np.ascontiguousarray(a, dtype=np.int32))

    def exchange(self, sends, recvs):
        self.r.exchange(sends, recvs)

    def allreduce(self, v):
        return self.r.allreduce(v)

    def close(self):
        self.r.barrier()
        self.r.close()


def _worker(rank, world, port, grid, transport, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    tr = _Gloo(rank, world, port) if transport == "gloo" else _Tcp(rank, world, port)
    import ddcmd_amd
    from ddcmd_amd.martini import plan_directions, domain_of, plan_recv_counts, plan_halo_layout
    s = ddcmd_amd.make_water_setup(6)            # 864 beads, box 48.7 A; 2 bricks of 24.4 A >= rlist 16 A
    L = s.box
    rlist = s.rmax + s.deltaR
    dest, shift = plan_directions(grid[0], grid[1], grid[2], rank, s.pbc)
    owner = domain_of(s, grid)
    mine = np.flatnonzero(owner == rank)
    r = np.stack([s.rx[mine], s.ry[mine], s.rz[mine]], axis=1)
    typ = s.ljtype[s.species[mine]]
    pc = (rank % grid[0], (rank // grid[0]) % grid[1], rank // (grid[0] * grid[1]))
    lo = np.array([-0.5 * L[a] + pc[a] * L[a] / grid[a] for a in range(3)])
    W = np.array([L[a] / grid[a] for a in range(3)])
    near_lo = r < lo + rlist
    near_hi = r >= lo + W - rlist
    halo_r, halo_t = [], []
    per_dir = {}
    hs_cnt = np.zeros(27, np.int32)
    # selection per direction (k_halo_select) + sender-side shift (k_pack_halo)
    for code in range(27):
        if code == 13 or dest[code] < 0:
            continue
        d = (code % 3 - 1, (code // 3) % 3 - 1, code // 9 - 1)
        sel = np.ones(len(mine), bool)
        for a in range(3):
            if d[a] < 0:
                sel &= near_lo[:, a]
            elif d[a] > 0:
                sel &= near_hi[:, a]
        out = np.concatenate([r[sel] + shift[code] * L, typ[sel, None].astype(np.float64)], axis=1)
        if dest[code] == rank:
            halo_r.append(out[:, :3]); halo_t.append(out[:, 3].astype(np.int64))      # local periodic image
        else:
            per_dir[code] = out
            hs_cnt[code] = len(out)
    # mg_xchg_counts: one all-gather of the 27 counts, then libddcmi's lookup
    allc = tr.allgather(hs_cnt)
    rcnt = plan_recv_counts(grid, rank, s.pbc, allc)
    # mg_layout_halo: libddcmi's peer-major layout and message list
    so, ro, msgs, msgr = plan_halo_layout(grid, rank, s.pbc, hs_cnt, rcnt)
    sendbuf = np.zeros((int(so[27]), 4))
    for code, out in per_dir.items():
        sendbuf[so[code]:so[code] + len(out)] = out
    recvbuf = np.full((int(ro[27]), 4), np.nan)
    npeers = len({int(d_) for d_ in dest if d_ >= 0 and d_ != rank})
    assert len(msgs) == npeers and len(msgr) == npeers, "one message per peer and direction of travel"
    tr.exchange([(p, sendbuf[o:o + c]) for p, o, c in msgs], [(p, recvbuf[o:o + c]) for p, o, c in msgr])
    assert not np.isnan(recvbuf).any()
    # every received segment lies where its (sender's) direction code says: sent towards +a => it is my -a side
    side_ok = True
    for code in range(27):
        seg = recvbuf[ro[code]:ro[code] + rcnt[code], :3]
        d = (code % 3 - 1, (code // 3) % 3 - 1, code // 9 - 1)
        for a in range(3):
            if d[a] > 0:
                side_ok &= bool(np.all(seg[:, a] < lo[a]) and np.all(seg[:, a] >= lo[a] - rlist))
            elif d[a] < 0:
                side_ok &= bool(np.all(seg[:, a] >= lo[a] + W[a]) and np.all(seg[:, a] < lo[a] + W[a] + rlist))
            else:
                side_ok &= bool(np.all(seg[:, a] >= lo[a]) and np.all(seg[:, a] < lo[a] + W[a]))
    halo_r.append(recvbuf[:, :3]); halo_t.append(recvbuf[:, 3].astype(np.int64))
    hr = np.concatenate(halo_r) if halo_r else np.zeros((0, 3))
    ht = np.concatenate(halo_t) if halo_t else np.zeros(0, np.int64)
    e1, f1 = _pair_terms(s, r, typ, r, typ, True)
    e2, f2 = _pair_terms(s, r, typ, hr, ht, False)
    etot = tr.allreduce([e1 + e2, float(len(mine)), float(len(hr))])      # energyInfo.c allreduce()
    import pyoracle
    o = pyoracle.Oracle(s)
    e0, _ = o.forces()
    if rank == 0:
        q.put(("energy", float(etot[0]), e0["lj"], int(etot[1]), s.natoms))
    ref = np.stack([o.fx[mine], o.fy[mine], o.fz[mine]], axis=1)
    q.put(("force", rank, float(np.abs(f1 + f2 - ref).max() / np.abs(ref).max()), len(hr), side_ok, len(msgs)))
    tr.close()


@pytest.mark.parametrize("grid,transport", [((2, 1, 1), "gloo"), ((1, 2, 1), "gloo"), ((2, 1, 1), "tcp"), ((1, 1, 2), "tcp")])
def test_world2_halo_protocol(built, grid, transport):
    import multiprocessing as mp          # plain spawn: the parent never loads torch (its bundled ROCm libs
    ctx = mp.get_context("spawn")         # must not meet the system ones already loaded through libddcmi.so)
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 300) + 7 * (grid[1] + 2 * grid[2]) + (1000 if transport == "tcp" else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, grid, transport, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(3)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for item in res:
        if item[0] == "energy":
            _, e, e0, ntot, n = item
            assert ntot == n
            assert abs(e - e0) < 1e-10 * abs(e0)
        else:
            _, rank, err, nh, side_ok, nmsg = item
            assert err < 1e-10, (rank, err)
            assert nh > 0 and side_ok
            assert nmsg == 1          # both periodic sides of the split axis lead to the same peer: ONE message


def test_halo_layout_is_consistent_across_ranks(built):
    """ddcmi_plan_halo_layout on every rank of 2x2x2, 2x2x1, 3x2x1 and the loopback: what A sends to B as its
    k-th message is what B expects from A as its k-th message, segment by segment"""
    from ddcmd_amd.martini import plan_directions, plan_recv_counts, plan_halo_layout
    rng = np.random.default_rng(5)
    for grid, loop in [((2, 2, 2), False), ((2, 2, 1), False), ((3, 2, 1), False), ((2, 1, 1), False), ((1, 1, 1), True)]:
        n = grid[0] * grid[1] * grid[2]
        dests = [plan_directions(grid[0], grid[1], grid[2], r, 7)[0] for r in range(n)]
        scnt = np.zeros((n, 27), np.int32)
        for r in range(n):
            for c in range(27):
                if c != 13 and dests[r][c] >= 0 and (dests[r][c] != r or loop):
                    scnt[r, c] = rng.integers(0, 50)
        plans = []
        for r in range(n):
            rc = plan_recv_counts(grid, r, 7, scnt, loopback=loop)
            for c in range(27):      # what the rank in my direction opp(c) sends along ITS direction c
                src = dests[r][26 - c]
                assert rc[c] == (scnt[src, c] if (c != 13 and src >= 0 and (src != r or loop)) else 0)
            plans.append((rc,) + plan_halo_layout(grid, r, 7, scnt[r], rc, loopback=loop))
        for a in range(n):
            rc_a, so_a, ro_a, ms_a, mr_a = plans[a]
            assert sum(c for _, _, c in ms_a) == scnt[a].sum() == so_a[27]
            assert sum(c for _, _, c in mr_a) == rc_a.sum() == ro_a[27]
            assert len({p for p, _, _ in ms_a}) == len(ms_a)          # one message per peer
            if grid == (2, 2, 2):
                assert len(ms_a) <= 7 and len(mr_a) <= 7               # the 7 xGMI peers
            for peer, off, cnt in ms_a:
                rc_b, so_b, ro_b, ms_b, mr_b = plans[peer]
                back = [m for m in mr_b if m[0] == a]
                assert len(back) == 1 and back[0][2] == cnt
                # segment by segment: my direction codes towards `peer`, ascending, are the receiver's segments
                # for sender codes ascending
                codes = [c for c in range(27) if c != 13 and dests[a][c] == peer and (peer != a or loop)]
                o_s, o_r = off, back[0][1]
                for c in codes:
                    assert so_a[c] == o_s and ro_b[c] == o_r
                    o_s += scnt[a, c]; o_r += scnt[a, c]


def test_plan_directions_topology(built):
    """ddcmi_plan_directions: destinations are mutual, shifts opposite, 2x2x2 has exactly 7 distinct peers"""
    from ddcmd_amd.martini import plan_directions
    for grid in [(2, 2, 2), (2, 2, 1), (2, 1, 1), (3, 2, 1), (1, 1, 1)]:
        n = grid[0] * grid[1] * grid[2]
        plans = [plan_directions(grid[0], grid[1], grid[2], r, 7) for r in range(n)]
        for r in range(n):
            dest, shift = plans[r]
            assert dest[13] == -1
            for code in range(27):
                if code == 13:
                    continue
                b = dest[code]
                assert 0 <= b < n
                # the neighbour's opposite direction leads back here with the opposite shift
                assert plans[b][0][26 - code] == r
                assert np.array_equal(plans[b][1][26 - code], -shift[code])
            if grid == (2, 2, 2):
                assert len(set(dest[dest >= 0].tolist()) - {r}) == 7
        # open boundaries: no neighbour across a non-periodic face
        dest, shift = plan_directions(grid[0], grid[1], grid[2], 0, 0)
        d = np.array([(c % 3 - 1, (c // 3) % 3 - 1, c // 9 - 1) for c in range(27)])
        assert all(dest[c] == -1 for c in range(27) if (d[c] < 0).any())
