#!/usr/bin/env python3
"""Write tests/golden/water_deck: a 2048-bead Martini water box in ddcMD's deck format, set up the way
the reference's shipped example is -- INTEGRATOR NGLFCONSTRAINT (barostat, no constraints), two LANGEVIN
groups, printMolecularPressure -- but with our own numbers (FCC lattice from ddcmd_amd.synth, fixed RNG
seed).  Nothing is copied from the reference.  Run from the repo root."""
import os
import sys
import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from ddcmd_amd.synth import make_water_setup  # noqa: E402
from ddcmd_amd.deck import units_convert  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "water_deck")

OBJECT = """simulate SIMULATE
{
   type = MD; system = system; integrator = nglf;
   deltaloop = 10; maxloop = 40; dt = 20; printrate = 10; snapshotrate = 100000; checkpointrate = 100000;
   printinfo = printinfo; ddc = ddc; accelerator = accelerator;
}
accelerator ACCELERATOR { type = HIP; }
ddc DDC { updateRate = 20; }
printinfo PRINTINFO { PRESSURE = bar; VOLUME = Ang^3; TEMPERATURE = K; ENERGY = kJ/mol; TIME = ns; printStress = 0; printMolecularPressure = 1; }
martini POTENTIAL
{
   type = MARTINI; excludePotentialTerm = 0; cutoff = 12.0 Angstrom; rcoulomb = 12.0 Angstrom; epsilon_r = 15; epsilon_rf = -1;
   function = lennardjones; parmfile = martini.data;
}
nglf INTEGRATOR { type = NGLFCONSTRAINT; T = 310 K; P0 = 1.0 bar; beta = 3.0e-4 1/bar; tauBarostat = 1.0 ps; }
system SYSTEM
{
   type = NORMAL; potential = martini; neighbor = nbr; groups = group free; random = lcg64; box = box;
   collection = collection; moleculeClass = moleculeClass; nConstraints = 0;
}
box BOX { type = ORTHORHOMBIC; pbc = 7; }
nbr NEIGHBOR { type = NORMAL; deltaR = 4.0; minBoxSide = 6; }
group GROUP { type = LANGEVIN; Teq = 310 K; tau = 1 ps; }
free GROUP { type = LANGEVIN; Teq = 310 K; tau = 1 ps; }
lcg64 RANDOM { type = LCG64; seed = 20261002; randomizeSeed = 0; }
moleculeClass MOLECULECLASS { molecules = Wx WFx; }
Wx MOLECULE { ownershipSpecies = WxW; species = WxW; }
WFx MOLECULE { ownershipSpecies = WFxWF; species = WFxWF; }
WxW SPECIES { type = ATOM; charge = 0.0; id = 1; mass = 72.0 M_p; }
WFxWF SPECIES { type = ATOM; charge = 0.0; id = 0; mass = 72.0 M_p; }
"""


def main():
    os.makedirs(os.path.join(OUT, "snapshot.mem"), exist_ok=True)
    s = make_water_setup(8)           # 2048 beads, box 65 A
    A = units_convert(1.0, None, "Angstrom")
    L = s.h[0] * A
    # force field file: copy the water part of the lipid deck's own martini.data conventions
    src = open(os.path.join(ROOT, "tests", "golden", "lipid_deck", "martini.data")).read()
    open(os.path.join(OUT, "martini.data"), "w").write(src)
    open(os.path.join(OUT, "object.data"), "w").write(OBJECT)
    h = "h=     %.6f 0.0 0.0\n       0.0 %.6f 0.0\n       0.0 0.0 %.6f ;" % (L, L, L)
    n = s.natoms
    open(os.path.join(OUT, "restart"), "w").write(
        "simulate SIMULATE { loop=0; time=0.000000 ;}\nbox BOX {\n%s\n}\ncollection COLLECTION { mode=VARRECORDASCII; size=%d; files=snapshot.mem/atoms#;}\n" % (h, n))
    names = ["WxW", "WFxWF"]
    with open(os.path.join(OUT, "snapshot.mem", "atoms#000000"), "w") as f:
        f.write("particle FILEHEADER {type=MULTILINE; datatype=VARRECORDASCII; checksum=NONE;\nloop=0; time=0.000000;\n"
                "nfiles=1; nrecord=%d; nfields=10;\nfield_names=id class type group rx ry rz vx vy vz;\n"
                "field_types=u s s s f f f f f f;\n%s\ngroups = group free ;\ntypes = ATOM ;\n} \n\n" % (n, h))
        for i in range(n):
            f.write("%14d ATOM %10s %s %21.13e %21.13e %21.13e %21.13e %21.13e %21.13e\n" % (
                int(s.gid[i]), names[int(s.species[i])], "group" if i % 2 == 0 else "free",
                s.rx[i] * A, s.ry[i] * A, s.rz[i] * A, s.vx[i] * A, s.vy[i] * A, s.vz[i] * A))
    print("wrote", n, "beads to", OUT)


if __name__ == "__main__":
    main()
