"""Author a small Martini lipid + water deck in ddcMD's own input formats
(object.data / martini.data / restart / atoms#000000) under tests/golden/lipid_deck/.

The reference ships water only (examples/waterbox), so the deck that exercises
bonds, cosine/harmonic/ReB angles, proper/improper dihedrals, exclusions and
charges has to be written by us.  Topology: 12-bead Martini-2 style
DPPC (NC3 +1, PO4 -1, 11 bonds, cosine angles) plus a 5-bead test molecule TSTM
carrying the term kinds DPPC lacks (func-1 and func-10 angles, a proper and an
improper dihedral, an explicit exclusion).  Numbers are Martini-like, not a
validated force field.  Deterministic; run:  python tests/golden/make_lipid_deck.py
"""
import os
import numpy as np

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lipid_deck")
LX, LY, LZ = 60.0, 64.0, 96.0        # Angstrom

TYPES = ["BP4", "P4", "Q0", "Qa", "Na", "C1"]
# (eps kJ/mol, sigma nm), Martini-2-like interaction levels
LJ = {("P4", "P4"): (5.0, 0.47), ("P4", "BP4"): (5.6, 0.57), ("BP4", "BP4"): (5.0, 0.47),
      ("Q0", "Q0"): (3.5, 0.47), ("Q0", "Qa"): (4.5, 0.47), ("Q0", "Na"): (4.0, 0.47), ("Q0", "C1"): (2.0, 0.62),
      ("Q0", "P4"): (5.6, 0.47), ("Q0", "BP4"): (5.6, 0.47),
      ("Qa", "Qa"): (5.0, 0.47), ("Qa", "Na"): (4.0, 0.47), ("Qa", "C1"): (2.0, 0.62), ("Qa", "P4"): (5.6, 0.47), ("Qa", "BP4"): (5.6, 0.47),
      ("Na", "Na"): (4.0, 0.47), ("Na", "C1"): (2.7, 0.47), ("Na", "P4"): (4.0, 0.47), ("Na", "BP4"): (4.0, 0.47),
      ("C1", "C1"): (3.5, 0.47), ("C1", "P4"): (2.0, 0.47), ("C1", "BP4"): (2.0, 0.47)}

DPPC_ATOMS = [("NC3", "Q0", 1.0), ("PO4", "Qa", -1.0), ("GL1", "Na", 0.0), ("GL2", "Na", 0.0),
              ("C1A", "C1", 0.0), ("C2A", "C1", 0.0), ("C3A", "C1", 0.0), ("C4A", "C1", 0.0),
              ("C1B", "C1", 0.0), ("C2B", "C1", 0.0), ("C3B", "C1", 0.0), ("C4B", "C1", 0.0)]
DPPC_BONDS = [(0, 1, 0.47, 1250.0), (1, 2, 0.47, 1250.0), (2, 3, 0.37, 1250.0), (2, 4, 0.47, 1250.0), (4, 5, 0.47, 1250.0),
              (5, 6, 0.47, 1250.0), (6, 7, 0.47, 1250.0), (3, 8, 0.47, 1250.0), (8, 9, 0.47, 1250.0), (9, 10, 0.47, 1250.0), (10, 11, 0.47, 1250.0)]
# func 2: k (cos(theta) - cos0)^2 ; theta0 holds cos(theta0)
DPPC_ANGLES = [(1, 2, 3, 2, 25.0, np.cos(np.radians(120.0))), (1, 2, 4, 2, 25.0, -1.0), (2, 4, 5, 2, 25.0, -1.0), (4, 5, 6, 2, 25.0, -1.0),
               (5, 6, 7, 2, 25.0, -1.0), (3, 8, 9, 2, 25.0, -1.0), (8, 9, 10, 2, 25.0, -1.0), (9, 10, 11, 2, 25.0, -1.0)]
TST_ATOMS = [("T1", "Na", 0.5), ("T2", "C1", 0.0), ("T3", "Na", -0.5), ("T4", "C1", 0.0), ("T5", "P4", 0.0)]
TST_BONDS = [(0, 1, 0.40, 5000.0), (1, 2, 0.40, 5000.0), (2, 3, 0.40, 5000.0), (3, 4, 0.40, 5000.0)]
# The angles that enter a dihedral are stiff enough to stay away from 180 degrees, where bioDihedralFast is singular
# (a 75 k-bead tiling of an earlier, softer version -- 40 and 35 kJ/mol, and a dihedral over the unbonded 2-4 --
# produced a runaway test molecule after ~3000 steps at 310 K)
TST_ANGLES = [(0, 1, 2, 1, 120.0, np.radians(110.0)), (1, 2, 3, 10, 30.0, np.cos(np.radians(120.0))), (2, 3, 4, 2, 350.0, np.cos(np.radians(130.0)))]
TST_TORS = [(0, 1, 2, 3, 1, 2, 4.0, 0.6), (1, 2, 3, 4, 2, 1, 20.0, 0.5), (0, 1, 2, 3, 1, 3, 1.5, 0.0)]   # (I,J,K,L,func,n,k,delta)
TST_EXCL = [(0, 2)]


def lj(a, b):
    return LJ[(a, b)] if (a, b) in LJ else LJ[(b, a)]


def martini_data():
    o = []
    o.append("martini MMFF\n{\nresiParms=W WF DPPC TSTM ;\natomTypeList=%s ;\nljParms=%s ;\n}\n" % (
        " ".join(TYPES), " ".join("%s_%s" % (a, b) for i, a in enumerate(TYPES) for b in TYPES[i:])))
    for i, t in enumerate(TYPES):
        o.append("%s MASSPARMS { atomType=%s; atomTypeID=%d; mass=72.0M_p ; }\n" % (t, t, i))
    for name, typ, rid in (("W", "P4", 1), ("WF", "BP4", 2)):
        o.append("%s RESIPARMS\n{\n  resID=%d;\n  resType=0;\n  resName=%s;\n  charge=0.0;\n  groupList=%s_g0;\n  centerAtom=0;\n}\n" % (name, rid, name, name))
        o.append("%s_g0 GROUPPARMS{\n  groupID=0;\n  atomList=%s_%s ;\n}\n" % (name, name, name))
        o.append("%s_%s ATOMPARMS{atomID=0; atomName=%s; atomType=%s; atomTypeID=%d; charge=0.0; mass=72.0 M_p ; }\n" % (name, name, name, typ, TYPES.index(typ)))

    def resi(name, rid, atoms, bonds, angles, tors, excl):
        o.append("%s RESIPARMS\n{\n  resID=%d;\n  resType=0;\n  resName=%s;\n  charge=%g;\n  groupList=%s_g0;\n  centerAtom=1;\n" % (
            name, rid, name, sum(a[2] for a in atoms), name))
        if bonds:
            o.append("  bondList=%s ;\n" % " ".join("%s_b%d" % (name, i) for i in range(len(bonds))))
        if excl:
            o.append("  exclusionList=%s ;\n" % " ".join("%s_e%d" % (name, i) for i in range(len(excl))))
        if angles:
            o.append("  angleList=%s ;\n" % " ".join("%s_a%d" % (name, i) for i in range(len(angles))))
        if tors:
            o.append("  dihedralList=%s ;\n" % " ".join("%s_d%d" % (name, i) for i in range(len(tors))))
        o.append("}\n")
        o.append("%s_g0 GROUPPARMS{\n  groupID=0;\n  atomList=%s ;\n}\n" % (name, " ".join("%s_%s" % (name, a[0]) for a in atoms)))
        for i, (an, typ, q) in enumerate(atoms):
            o.append("%s_%s ATOMPARMS{atomID=%d; atomName=%s; atomType=%s; atomTypeID=%d; charge=%g; mass=72.0 M_p ; }\n" % (
                name, an, i, an, typ, TYPES.index(typ), q))
        for i, (a, b, b0, kb) in enumerate(bonds):
            o.append("%s_b%d BONDPARMS{atomI=%d; atomJ=%d; func=1; atomTypeI=%s; atomTypeJ=%s; kb=%g kJ*mol^-1*nm^-2; b0=%g nm;}\n" % (
                name, i, a, b, atoms[a][1], atoms[b][1], kb, b0))
        for i, (a, b) in enumerate(excl):
            o.append("%s_e%d EXCLUDEPARMS{atomI=%d; atomJ=%d; atomTypeI=%s; atomTypeJ=%s;}\n" % (name, i, a, b, atoms[a][1], atoms[b][1]))
        for i, (a, b, c, f, k, t0) in enumerate(angles):
            o.append("%s_a%d ANGLEPARMS{atomI=%d; atomJ=%d; atomK=%d; func=%d; ktheta=%g kJ*mol^-1; theta0=%.17g;}\n" % (name, i, a, b, c, f, k, t0))
        for i, (a, b, c, d, f, n, k, dl) in enumerate(tors):
            o.append("%s_d%d TORSPARMS{atomI=%d; atomJ=%d; atomK=%d; atomL=%d; func=%d; n=%d; kchi=%g kJ*mol^-1; delta=%.17g;}\n" % (name, i, a, b, c, d, f, n, k, dl))
    resi("DPPC", 3, DPPC_ATOMS, DPPC_BONDS, DPPC_ANGLES, [], [])
    resi("TSTM", 4, TST_ATOMS, TST_BONDS, TST_ANGLES, TST_TORS, TST_EXCL)
    for i, a in enumerate(TYPES):
        for b in TYPES[i:]:
            e, s = lj(a, b)
            o.append("%s_%s LJPARMS{atomtypeI=%s; indexI=%d; atomtypeJ=%s; indexJ=%d; sigma=%g nm; eps=%g kJ*mol^-1;}\n" % (
                a, b, a, TYPES.index(a), b, TYPES.index(b), s, e))
    return "".join(o)


def object_data(species):
    o = ["""simulate SIMULATE
{
   type = MD;
   system=system;
   integrator=nglf;
   deltaloop=20;
   maxloop =1000000;
   dt = 10;
   printrate=1;
   snapshotrate=100000;
   checkpointrate=100000;
   printinfo=printinfo;
   heap=heap;
   ddc = ddc;
}
energyInfo ENERGYINFO{}
heap HEAP { size = 1000 ;}
ddc DDC { updateRate=10; }
printinfo PRINTINFO { PRESSURE=bar; VOLUME = Ang^3; TEMPERATURE = K; ENERGY = kJ/mol; TIME = ns; printStress=0; }
martini  POTENTIAL
{
   type = MARTINI;
   excludePotentialTerm=0;
   cutoff=11.0 Angstrom;
   rcoulomb=11.0 Angstrom; epsilon_r=15; epsilon_rf=-1;
   function=lennardjones;
   parmfile=martini.data;
}
nglf INTEGRATOR {type = NGLF; }
system SYSTEM
{
   type = NORMAL;
   potential = martini ;
   neighbor=nbr;
   groups= group ;
   random = lcg64;
   box = box;
   collection=collection;
   moleculeClass = moleculeClass;
   nConstraints=0;
}
box BOX { type=ORTHORHOMBIC; pbc=7; }
nbr NEIGHBOR { type = NORMAL; deltaR=4.0000; minBoxSide=6; }
group GROUP { type = FREE; }
lcg64 RANDOM {type = LCG64;randomizeSeed=0;}
moleculeClass MOLECULECLASS { molecules =  Wx WFx DPPCx TSTMx; }
Wx MOLECULE {ownershipSpecies = WxW; species = WxW;}
WFx MOLECULE {ownershipSpecies = WFxWF; species = WFxWF;}
"""]
    o.append("DPPCx MOLECULE {ownershipSpecies = DPPCxPO4; species = %s;}\n" % " ".join("DPPCx" + a[0] for a in DPPC_ATOMS))
    o.append("TSTMx MOLECULE {ownershipSpecies = TSTMxT2; species = %s;}\n" % " ".join("TSTMx" + a[0] for a in TST_ATOMS))
    for name, q in species:
        o.append("%s SPECIES { type = ATOM ; charge =%g; mass =72.0 M_p ; }\n" % (name, q))
    return "".join(o)


def main():
    os.makedirs(os.path.join(OUT, "snapshot.mem"), exist_ok=True)
    rng = np.random.RandomState(20261002)
    recs = []          # (gid, species, x, y, z)
    mol = 0
    placed = []

    def add(gid, sp, r):
        recs.append((gid, sp, r[0], r[1], r[2]))
        placed.append(r)

    # bilayer: 6x10 lipids per leaflet (10 A x 6.4 A cells: tail columns 5 A apart), tails towards z=0
    zs = [31.2, 26.5, 21.8, 21.8, 17.1, 12.4, 7.7, 3.0, 17.1, 12.4, 7.7, 3.0]
    dx = [0.0, 0.0, -1.85, 1.85, -2.5, -2.5, -2.5, -2.5, 2.5, 2.5, 2.5, 2.5]
    for leaf in (1, -1):
        for ix in range(6):
            for iy in range(10):
                x0 = -LX / 2 + 5.0 + 10.0 * ix
                y0 = -LY / 2 + 3.2 + 6.4 * iy
                for a in range(12):
                    r = np.array([x0 + dx[a], y0, leaf * zs[a]]) + 0.3 * (rng.rand(3) - 0.5)
                    add((mol << 32) | a, "DPPCx" + DPPC_ATOMS[a][0], r)
                mol += 1
    # test molecules in the water slab: zig-zag chains
    for k in range(8):
        x0 = -LX / 2 + 5.0 + 7.0 * k
        for a in range(5):
            r = np.array([x0 + 3.3 * a * 0.5, -20.0 + 2.4 * (a % 2) + 5.0 * (k % 3), 41.0 + 1.2 * (a // 2) + 0.9 * ((a * k) % 2)]) + 0.4 * (rng.rand(3) - 0.5)
            add((mol << 32) | a, "TSTMx" + TST_ATOMS[a][0], r)
        mol += 1
    # water: 5 A lattice in |z| > 35.5 A, skipping sites within 4.3 A of anything placed
    P = np.array(placed)
    nw = 0
    for ix in range(12):
        for iy in range(13):
            for iz in range(20):
                r = np.array([-LX / 2 + 2.4 + 4.93 * ix, -LY / 2 + 2.4 + 4.93 * iy, -LZ / 2 + 2.4 + 4.8 * iz]) + 0.4 * (rng.rand(3) - 0.5)
                if abs(r[2]) < 35.5:
                    continue
                d = P - r
                d -= np.array([LX, LY, LZ]) * np.rint(d / np.array([LX, LY, LZ]))
                if (np.sum(d * d, axis=1) < 4.3 ** 2).any():
                    continue
                sp = "WFxWF" if nw % 10 == 9 else "WxW"
                recs.append((mol << 32, sp, r[0], r[1], r[2]))
                mol += 1
                nw += 1
    n = len(recs)
    species = [("WxW", 0.0), ("WFxWF", 0.0)] + [("DPPCx" + a[0], a[2]) for a in DPPC_ATOMS] + [("TSTMx" + a[0], a[2]) for a in TST_ATOMS]
    open(os.path.join(OUT, "object.data"), "w").write(object_data(species))
    open(os.path.join(OUT, "martini.data"), "w").write(martini_data())
    h = "h=     %.3f 0.0 0.0\n       0.0 %.3f 0.0\n       0.0 0.0 %.3f ;" % (LX, LY, LZ)
    open(os.path.join(OUT, "restart"), "w").write(
        "simulate SIMULATE { loop=0; time=0.000000 ;}\nbox BOX {\n%s\n}\ncollection COLLECTION { mode=VARRECORDASCII; size=%d; files=snapshot.mem/atoms#;}\n" % (h, n))
    with open(os.path.join(OUT, "snapshot.mem", "atoms#000000"), "w") as f:
        f.write("particle FILEHEADER {type=MULTILINE; datatype=VARRECORDASCII; checksum=NONE;\nloop=0; time=0.000000;\n"
                "nfiles=1; nrecord=%d; nfields=10;\nfield_names=id class type group rx ry rz vx vy vz;\n"
                "field_types=u s s s f f f f f f;\n%s\ngroups = group ;\ntypes = ATOM ;\n} \n\n" % (n, h))
        for gid, sp, x, y, z in recs:
            v = 2.0e-3 * rng.randn(3)       # A/fs, ~300 K for 72 amu
            f.write("%14d ATOM %10s group %21.13e %21.13e %21.13e %21.13e %21.13e %21.13e\n" % (gid, sp, x, y, z, v[0], v[1], v[2]))
    print("wrote %d beads (%d lipids, 8 test molecules, %d water) to %s" % (n, 120, nw, OUT))


if __name__ == "__main__":
    main()
