#!/usr/bin/env python3
"""How loose is the displacement bound that ends the rows of the pair kernel early?  D = sum over the steps since the rebuild of dt * (largest |v| of any bead)
against the largest displacement any bead really has -- over the whole box, and over the beads of one tile's neighbourhood (what a tile-local bound could use).
   python3 tools/disp_bound_r06.py [n (FCC cells per edge, 64 = 1.05 M beads)]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ddcmd_amd.synth import make_water_setup
from ddcmd_amd.martini import MartiniHIP
from ddcmd_amd.deck import units_convert
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
s = make_water_setup(n)
m = MartiniHIP(s, test_api=True)
m.eval_forces()
m.step(400)      # the lattice start melts; the run settles near 300 K
period = int(s.updateRate)
while m.clock()[0] % period != 0:
    m.step(1)
A = units_convert(1.0, None, "Angstrom")
L = np.array([s.h[0], s.h[4], s.h[8]])
m.lib.ddcmi_debug_disp.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)]
r0 = None
print("T = %.0f K, skin %.2f A, cut-off %.2f A, rebuild every %d steps of %g fs" % (2.0 * m.energies()[2] / (3.0 * s.natoms) / units_convert(1.0, "K"), s.deltaR * A, s.rmax * A, period, s.dt))
print("step  bound D (A)   largest real displacement (A)   95th percentile over 4000 'tiles' of ~3000 neighbouring beads (A)   walk radius by bound / by box maximum / by tile maximum (A)")
rng = np.random.RandomState(1)
for k in range(period + 1):
    d = m.download()
    r = np.stack(d["r"], 1)
    if k == 0:
        r0 = r.copy()
        # 'tiles': beads binned into boxes of about a tile neighbourhood's size (12 x 8 x 8 cells of r_list / 2)
        cell = 0.5 * (s.rmax + s.deltaR)
        nb = np.maximum((L / (np.array([12, 8, 8]) * cell)).astype(int), 1)
        key = (np.floor((r0 / L + 0.5) * nb).astype(int) % nb) @ np.array([1, nb[0], nb[0] * nb[1]])
    else:
        dr = r - r0
        dr -= L * np.rint(dr / L)
        disp = np.sqrt((dr ** 2).sum(1))
        tilemax = np.zeros(key.max() + 1); np.maximum.at(tilemax, key, disp)
        D = ctypes.c_double(0); ring = (ctypes.c_float * 64)(); nr = ctypes.c_int(0)
        m.lib.ddcmi_debug_disp(m.ctx, ctypes.byref(D), ring, ctypes.byref(nr))
        # the lean steps' share: the decaying maximum scan over the ring (LEAN_C = 0.81 on |v|^2), as the kernel forms it
        e, bound = 0.0, D.value
        for q in range(nr.value):
            e = max(ring[q], e * 0.81)
            bound += s.dt * np.sqrt(e)
        t95 = np.percentile(tilemax, 95)
        print("%4d  %9.3f   %9.3f   %9.3f      %.2f / %.2f / %.2f" % (k, bound * A, disp.max() * A, t95 * A, min((s.rmax + 2 * bound) * A, (s.rmax + s.deltaR) * A), min((s.rmax + 2 * disp.max()) * A, (s.rmax + s.deltaR) * A),
                                                                 min((s.rmax + 2 * t95) * A, (s.rmax + s.deltaR) * A)), flush=True)
    if k < period:
        m.step(1)
