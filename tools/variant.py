#!/usr/bin/env python3
"""Tuning builds that never touch the product: copy csrc/hip to a scratch directory, apply text substitutions,
compile for gfx950 and link tuning/libddcmi_<name>.so (git-ignored; select it with DDCMI_LIB=...).

   python3 tools/variant.py <name> [-D FLAG ...] [--sub 'old' 'new' ...] [--hsub 'old' 'new' ...] [--patch file.py]      (--hsub: in ddcmi_internal.h)

--patch file.py: a python file defining edit(src: str) -> str, applied to ddcmi.hip."""
import os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ddcmd_amd", "csrc")


def main():
    name = sys.argv[1]
    flags, subs, patch, hsubs = [], [], None, []
    a = sys.argv[2:]
    while a:
        if a[0] == "-D": flags.append("-D" + a[1]); a = a[2:]
        elif a[0] == "--sub": subs.append((a[1], a[2])); a = a[3:]
        elif a[0] == "--patch": patch = a[1]; a = a[2:]
        elif a[0] == "--hsub": hsubs.append((a[1], a[2])); a = a[3:]
        else: raise SystemExit("unknown argument " + a[0])
    work = os.path.join("/tmp", "ddcmi_variant_" + name)
    shutil.rmtree(work, ignore_errors=True)
    shutil.copytree(os.path.join(CSRC, "hip"), os.path.join(work, "hip"))
    # --sub applies to whichever device source holds the text (ddcmi.hip or one of the .inl parts it includes); --patch to ddcmi.hip
    parts = sorted(f for f in os.listdir(os.path.join(work, "hip")) if f.endswith((".hip", ".inl")))
    for old, new in subs:
        for f in parts:
            fnp = os.path.join(work, "hip", f)
            src = open(fnp).read()
            if old in src:
                open(fnp, "w").write(src.replace(old, new))
                break
        else:
            raise SystemExit("substitution source not found: " + old[:60])
    fn = os.path.join(work, "hip", "ddcmi.hip")
    if patch:
        g = {}
        exec(open(patch).read(), g)
        open(fn, "w").write(g["edit"](open(fn).read()))
    if hsubs:
        hn = os.path.join(work, "hip", "ddcmi_internal.h")
        h = open(hn).read()
        for old, new in hsubs:
            if old not in h: raise SystemExit("header substitution source not found: " + old[:60])
            h = h.replace(old, new)
        open(hn, "w").write(h)
    subprocess.check_call(["make", "-s", "-C", CSRC])      # host objects
    objs = []
    procs = []
    for f in ("ddcmi", "scan", "bonded"):
        o = os.path.join(work, f + ".o")
        objs.append(o)
        procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=fast",
                                       "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(work, "hip")] + flags +
                                      ["-c", os.path.join(work, "hip", f + ".hip"), "-o", o]))
    if any(p.wait() for p in procs): raise SystemExit("compile failed")
    out = os.path.join(ROOT, "tuning", "libddcmi_%s.so" % name)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    host = [os.path.join(CSRC, "build", "host", "rdzv.o")]      # the device library = HIP objects + the rendezvous; the soname of the tree's, the WIDER export map (include/ddcmi.h + ddcmi_test.h): a tuning build serves every test
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-o", out] + host + objs +
                          ["-Wl,--version-script=" + os.path.join(CSRC, "build", "ddcmi_test.map"), "-Wl,-soname,libddcmi.so", "-L/opt/rocm/lib", "-lrccl", "-lm", "-Wl,-rpath,/opt/rocm/lib"])
    print("built", out)


if __name__ == "__main__":
    main()
