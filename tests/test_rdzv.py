"""CPU: the process rendezvous of libddcmi (host/rdzv.c) between real processes --
what bench.py and the multi-rank driver use in place of MPI / torch.distributed."""
import os
import sys
import subprocess
import tempfile
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, %(root)r)
from ddcmd_amd.martini import Rendezvous
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
r = Rendezvous.from_env(timeout=60.0)
out = {"rank": rank}
# MPI_Bcast of an opaque id
msg = bytes(range(128)) if rank == 0 else bytes(128)
out["bcast"] = list(r.bcast(msg, 0)) == list(range(128))
msg = bytes([7] * 16) if rank == world - 1 else bytes(16)
out["bcast_last"] = list(r.bcast(msg, world - 1)) == [7] * 16
r.barrier()
v = r.allreduce([float(rank + 1), 1.0, -float(rank)])
out["sum"] = v.tolist()
out["max"] = r.allreduce([float(rank), -float(rank)], "max").tolist()
g = r.allgather(np.arange(3, dtype=np.int32) + 10 * rank)
out["gather"] = g.tolist()
# grouped exchange: two messages to every other rank (sizes differ per pair, the large ones exceed any
# socket buffer so both directions must progress together), one message to self
rng = lambda a, b, k, n: (np.arange(n, dtype=np.float64) * 1e-3 + 1000.0 * a + 10.0 * b + k)
sends, recvs, expect = [], [], []
for p in range(world):
    for k in range(2):
        n = 700000 + 1000 * p + 10 * rank + k if k == 0 else 5 + p + rank
        if p == rank:
            n = 17 + k
        sends.append((p, rng(rank, p, k, n)))
for p in range(world):
    for k in range(2):
        n = 700000 + 1000 * rank + 10 * p + k if k == 0 else 5 + p + rank
        if p == rank:
            n = 17 + k
        buf = np.zeros(n)
        recvs.append((p, buf))
        expect.append(rng(p, rank, k, n))
r.exchange(sends, recvs)
out["exchange"] = all(np.array_equal(b, e) for (_, b), e in zip(recvs, expect))
r.barrier()
r.close()
print("RESULT " + json.dumps(out), flush=True)
'''


def _launch(world, env_extra):
    procs = []
    for rank in range(world):
        env = dict(os.environ)
        env.update({"RANK": str(rank), "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1"})
        env.update(env_extra)
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER % {"root": ROOT}], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    res = []
    for p in procs:
        o, e = p.communicate(timeout=180)
        assert p.returncode == 0, e[-2000:]
        import json
        res.append(json.loads([l for l in o.splitlines() if l.startswith("RESULT ")][-1][7:]))
    return sorted(res, key=lambda d: d["rank"])


@pytest.mark.parametrize("world,mode", [(2, "file"), (4, "file"), (3, "port")])
def test_rendezvous_collectives(built, world, mode):
    if mode == "file":
        # the launcher keeps MASTER_PORT (torch.distributed.run): rank 0 publishes an ephemeral port in a file
        with tempfile.TemporaryDirectory() as d:
            res = _launch(world, {"DDCMI_RDZV_FILE": os.path.join(d, "port"), "MASTER_PORT": "1"})
    else:
        port = 30000 + os.getpid() % 20000
        res = _launch(world, {"DDCMI_RDZV_PORT": str(port)})
    for r, d in enumerate(res):
        assert d["bcast"] and d["bcast_last"] and d["exchange"], d
        assert d["sum"] == [world * (world + 1) / 2.0, float(world), -world * (world - 1) / 2.0]
        assert d["max"] == [float(world - 1), 0.0]
        assert d["gather"] == [[10 * q, 10 * q + 1, 10 * q + 2] for q in range(world)]


def test_rendezvous_times_out_instead_of_hanging(built):
    """a rank whose peers never arrive gets DDCMI_ECOMM with a message, not a hang"""
    sys.path.insert(0, ROOT)
    from ddcmd_amd.martini import Rendezvous, DdcmiError
    with tempfile.TemporaryDirectory() as d:
        with pytest.raises(DdcmiError) as ei:
            Rendezvous(1, 2, "127.0.0.1", 0, os.path.join(d, "nobody"), timeout=1.0)
        assert "rank 0 did not answer" in str(ei.value)
        with pytest.raises(DdcmiError) as ei:
            Rendezvous(0, 2, "127.0.0.1", 0, os.path.join(d, "alone"), timeout=1.0)
        assert "arrived" in str(ei.value)
