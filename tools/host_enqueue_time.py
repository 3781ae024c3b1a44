#!/usr/bin/env python3
"""How long does the HOST take to queue a step of a decomposed rank, against the device's time for it?  (a rank whose host needs longer than its GPU is
host-bound: the GPU-side step time stops mattering).  The 530 k brick through the RCCL loopback; DDCMI_DEBUG_HOOKS=1 DDCMI_DEBUG_SPLIT_MSGS=k hands RCCL k
send/recv pairs per step instead of one (a 2x2x2 rank has seven peers).   python3 tools/host_enqueue_time.py [lattice]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ddcmd_amd
from ddcmd_amd.martini import MartiniRank
n = int(sys.argv[1]) if len(sys.argv) > 1 else 51
os.environ["DDCMI_RCCL_LOOPBACK"] = "1"
s = ddcmd_amd.make_water_setup(n)
m = MartiniRank(s, np.arange(s.natoms))
buf = ctypes.create_string_buffer(128)
assert m.lib.ddcmi_comm_unique_id(buf) == 0
m.comm_init(0, 1, buf.raw, (1, 1, 1)); m.upload_local()
m.eval_forces(); m.step(240); m.sync()
host, total = [], []
for _ in range(10):
    # the step that rebuilds (loop % 20 == 0) waits for the device inside the rebuild; the 18 steps behind it have no host wait at all:
    # their queueing time is the host's own cost per step
    m.step(1); m.sync()
    t0 = time.perf_counter(); m.step(18); t1 = time.perf_counter(); m.sync(); t2 = time.perf_counter()
    host.append((t1 - t0) / 18); total.append((t2 - t0) / 18)
    m.step(1); m.sync()
print("split %s, %d beads: the host queues a step between rebuilds in %.1f us (median of 10 x 18 steps), the device finishes it in %.1f us" % (os.environ.get("DDCMI_DEBUG_SPLIT_MSGS", "1"), s.natoms, 1e6 * sorted(host)[5], 1e6 * sorted(total)[5]))
m.close()
