#!/usr/bin/env python3
"""Mutation fuzz of the ddc object-file / deck loader (host/object.c, host/deck.c, host/units.c): the lipid deck's object.data, martini.data and
atoms file with random bytes deleted, duplicated, flipped or truncated must load or be refused with a message -- never crash, never spin (120 s alarm per case).  Run it with the host
layer built under ASan/UBSan and the runtimes preloaded (profiles/r06_sanitizers.txt).   python3 tools/fuzz_decks.py [ncases] [seed]"""
import os, shutil, signal, sys, tempfile, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ddcmd_amd.deck import load_deck
RUN = "--run" in sys.argv      # on a GPU: every mutated deck also goes through the C driver (ddcmd_amd/bin/ddcmi_md: plugin.c's mirrors of ddcMD's objects, the
if RUN:                        # integrator plugins, the data / restart writers): it must end with exit code 0 or with a message and a non-zero code -- no signal, no hang
    sys.argv.remove("--run")
    import subprocess
    EXE = os.path.join(ROOT, "ddcmd_amd", "bin", "ddcmi_md")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
src = os.path.join(ROOT, "tests", "golden", "lipid_deck")
ok = refused = 0
run_ok = run_refused = 0
run_bad = []
with tempfile.TemporaryDirectory() as d:
    for case in range(n):
        work = os.path.join(d, "c%d" % case)
        shutil.copytree(src, work)
        victim = rnd.choice(["object.data", "martini.data", "restart", os.path.join("snapshot.mem", "atoms#000000")])
        p = os.path.join(work, victim)
        b = bytearray(open(p, "rb").read())
        for _ in range(rnd.choice([1, 1, 2, 5, 20])):
            if not b: break
            k = rnd.randrange(len(b)); op = rnd.randrange(5)
            if op == 0: del b[k:k + rnd.choice([1, 1, 3, 40])]
            elif op == 1: b[k:k] = b[k:k + rnd.choice([1, 8, 64])]
            elif op == 2: b[k] = rnd.randrange(256)
            elif op == 3: b[k:k] = rnd.choice([b"{", b"}", b";", b"=", b" 1e999 ", b" -1 ", b"\x00", b"\n\n", b"nm", b"kJ*mol^-1"])
            else: del b[k:]
        open(p, "wb").write(bytes(b))
        if os.environ.get("FUZZ_VERBOSE"): print("case", case, victim, flush=True); shutil.copy(p, "/tmp/fuzz_last_victim")
        signal.alarm(120)      # no handler on purpose: a loader that spins inside C is killed ("Alarm clock") and the case number is the last line
        try:
            s = load_deck(os.path.join(work, "object.data"))
            ok += 1
        except Exception as ex:
            refused += 1
        signal.alarm(0)
        if RUN:
            try:
                r = subprocess.run([EXE, "-o", os.path.join(work, "object.data"), "-d", os.path.join(work, "data")], cwd=work, capture_output=True, text=True, timeout=120, errors="replace")
                rc, tail = r.returncode, (r.stderr.strip().splitlines() or r.stdout.strip().splitlines() or [""])[-1][:160]
            except subprocess.TimeoutExpired:
                rc, tail = "HUNG", ""
            run_ok += rc == 0
            run_refused += (rc != 0 and rc != "HUNG" and rc > 0 and bool(tail))
            if rc == "HUNG" or rc < 0 or (rc != 0 and not tail):
                run_bad.append((case, victim, rc, tail))
                shutil.copy(p, os.path.join(ROOT, "gpurun_out", "fuzz_deck_case%d_%s" % (case, os.path.basename(victim)))) if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else None
            if os.environ.get("FUZZ_VERBOSE"): print("   driver rc", rc, tail, flush=True)
        shutil.rmtree(work)
print("%d mutated decks: %d loaded, %d refused with a message, 0 crashes" % (n, ok, refused))
if RUN:
    print("through the C driver: %d ran to the end, %d ended with a message and a non-zero exit code, %d died on a signal / hung / left without a word" % (run_ok, run_refused, len(run_bad)))
    for b in run_bad:
        print("   case %d %s rc %s %s" % b)
    sys.exit(1 if run_bad else 0)
