"""CPU tests of the drop-in boundary: libddcmi.so loads without a GPU, exports every
symbol include/ddcmi.h declares, and fails loudly (error code + message) when no
device is present -- there is no CPU fallback."""
import ctypes
import os
import re

import ddcmd_amd
from ddcmd_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions(header):
    text = open(header).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ddcmi_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(built):
    lib = ddcmd_amd.load_library()
    names = _declared_functions(os.path.join(ROOT, "include", "ddcmi.h"))
    assert len(names) >= 35
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_device_library_exports_nothing_but_the_declared_c_abi(built):
    """libddcmi.so used to export ddcMD's OWN names (nglf, ddcenergy, object_get, units_convert ...) with different
    signatures, plus every mangled internal: -lddcmi into ddcMD meant duplicate definitions / interposition.  Its dynamic symbol
    table is now exactly the functions include/ddcmi.h declares (version script generated from the header); the stand-alone host
    layer with the reference-named symbols lives in libddcmi_host.so."""
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = sorted(l.split()[-1] for l in out.splitlines() if l.strip())
    assert exported == _declared_functions(os.path.join(ROOT, "include", "ddcmi.h")), [n for n in exported if not n.startswith("ddcmi_")][:10]
    host = subprocess.check_output(["nm", "-D", "--defined-only", _lib.HOST_LIB_PATH], text=True)
    host = {l.split()[-1] for l in host.splitlines() if l.strip()}
    assert {"nglf", "ddcenergy", "object_get", "units_convert", "martiniHIP", "nglfHIP"} <= host and not (host & set(exported))
    deps = subprocess.check_output(["readelf", "-d", _lib.HOST_LIB_PATH], text=True)
    assert "libddcmi.so" in deps                      # the host layer reaches the device through the C-ABI like any other client


def test_test_only_entry_points_live_in_a_second_library(built):
    """ddcmi_group_* (in-process domain groups), ddcmi_plan_* (the halo planner's host logic) and
    ddcmi_debug_branch_census shipped in libddcmi.so's export table.  They are declared in include/ddcmi_test.h now and exported by
    libddcmi_test.so only -- a second link of the SAME objects with a wider version script -- so the drop-in library exports the
    plugin surface + comm and nothing else, while the tests that need the extra entry points still run the product's kernels."""
    import subprocess
    prod = _declared_functions(os.path.join(ROOT, "include", "ddcmi.h"))
    extra = _declared_functions(os.path.join(ROOT, "include", "ddcmi_test.h"))
    assert extra and not (set(prod) & set(extra))
    assert all(n.startswith(("ddcmi_group_", "ddcmi_plan_", "ddcmi_debug_")) for n in extra), extra
    assert not [n for n in prod if n.startswith(("ddcmi_plan_", "ddcmi_debug_")) or (n.startswith("ddcmi_group_") and n != "ddcmi_group_temperatures")]
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.TEST_LIB_PATH], text=True)
    exported = sorted(l.split()[-1] for l in out.splitlines() if l.strip())
    assert exported == sorted(prod + extra)
    # the same device code: both libraries are links of the same objects (the gfx950 code objects are byte for byte alike)
    def code_objects(path):
        import hashlib
        for line in subprocess.check_output(["readelf", "-S", "-W", path], text=True).splitlines():
            f = line.replace("[", " ").replace("]", " ").split()
            if len(f) > 5 and f[1] == ".hip_fatbin":
                off, size = int(f[4], 16), int(f[5], 16)
                with open(path, "rb") as fh:
                    fh.seek(off)
                    return size, hashlib.sha256(fh.read(size)).hexdigest()
        return None
    assert code_objects(_lib.LIB_PATH) is not None and code_objects(_lib.LIB_PATH)[0] > 100000
    assert code_objects(_lib.LIB_PATH) == code_objects(_lib.TEST_LIB_PATH)
    tl = _lib.load_test_library()
    assert all(hasattr(tl, n) for n in extra)


def test_links_beside_definitions_of_ddcmds_own_names(built, tmp_path):
    """a program that defines nglf, ddcenergy, kinetic_terms, object_get, units_convert ... with the reference's signatures links
    with -lddcmi alone, reaches its own definitions, and the library never calls them (tests/abi/link_with_ddcmd_names.c; the
    -m gpu twin runs forces and steps through it)"""
    import subprocess
    exe = str(tmp_path / "link_with_ddcmd_names")
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(["gcc", "-std=gnu99", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"), "-o", exe,
                           os.path.join(ROOT, "tests", "abi", "link_with_ddcmd_names.c"), "-L" + libdir, "-lddcmi",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    lib = ddcmd_amd.load_library()
    lib.ddcmi_device_count.restype = ctypes.c_int
    if lib.ddcmi_device_count() > 0:
        return          # (with a device the program wants its input file: tests/test_gpu_abi_link.py)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.startswith("nodevice stubs_called_by_library 0"), (out.stdout, out.stderr)


def test_integrator_struct_and_label_masks_against_the_references_own_headers(built, tmp_path):
    """integrator.h, bioGid.h and gid.h compile by themselves: where /root/reference exists they are INCLUDED (tests/abi/ref_headers.c),
    not transcribed; what they print is the committed fixture tests/golden/ref_headers.txt (made by that very program), and host/plugin.h +
    include/ddcmi.h must print the same (integrator.h:5-17, bioGid.h:13-22, gid.h:13)"""
    import subprocess
    abi = os.path.join(ROOT, "tests", "abi")
    want = open(os.path.join(ROOT, "tests", "golden", "ref_headers.txt")).read().splitlines()
    exe = str(tmp_path / "our_headers")
    subprocess.check_call(["gcc", "-std=gnu99", "-Wall", "-I" + os.path.join(ROOT, "ddcmd_amd", "csrc", "host"), "-I" + os.path.join(ROOT, "include"),
                           "-o", exe, os.path.join(abi, "our_headers.c")])
    assert subprocess.check_output([exe], text=True).splitlines() == want
    ref = "/root/reference/src"
    if os.path.exists(os.path.join(ref, "bioGid.h")):
        exe = str(tmp_path / "ref_headers")
        subprocess.check_call(["gcc", "-std=gnu99", "-Wall", "-I" + ref, "-o", exe, os.path.join(abi, "ref_headers.c")])
        assert subprocess.check_output([exe], text=True).splitlines() == want
    assert len(want) > 20


def test_host_layer_symbols(built):
    lib = ddcmd_amd.load_library()
    for n in ("object_compilefile", "object_get", "object_getv", "object_find", "units_convert", "units_internal", "units_external",
              "ddcmi_deck_load", "accelerator_init", "accelerator_getAccelerator", "potential_init", "integrator_init",
              "martiniHIP", "nglfHIP", "ddcenergy", "kinetic_terms", "eval_energyInfo", "simulate_init", "simulateMaster"):
        assert hasattr(lib, n), n


def test_create_without_gpu_fails_loudly(built):
    lib = ddcmd_amd.load_library()
    lib.ddcmi_device_count.restype = ctypes.c_int
    if lib.ddcmi_device_count() > 0:
        return          # on a GPU box the positive path is covered by the -m gpu tests
    ctx = ctypes.c_void_p()
    lib.ddcmi_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int]
    rc = lib.ddcmi_create(ctypes.byref(ctx), 0)
    assert rc == -1 and not ctx.value           # DDCMI_ENODEVICE
    lib.ddcmi_last_error.restype = ctypes.c_char_p
    lib.ddcmi_last_error.argtypes = [ctypes.c_void_p]
    assert b"no HIP device" in lib.ddcmi_last_error(None)
    from ddcmd_amd.martini import MartiniHIP, DdcmiError
    import pytest
    with pytest.raises(DdcmiError):
        MartiniHIP(ddcmd_amd.make_water_setup(3))


def test_product_does_not_import_the_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/"""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "ddcmd_amd")):
        for f in files:
            if f.endswith((".py", ".c", ".h", ".hip", ".inl")) and "build" not in dirpath:
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "pyoracle" not in text and "ddc_oracle" not in text, os.path.join(dirpath, f)


def test_plugin_structs_are_layout_compatible_with_the_reference(built, tmp_path):
    """POTENTIAL / INTEGRATOR / ACCELERATOR (+ RCUT_TYPE and the enumerators the glue tests) of host/plugin.h have the
    offsets, sizes and values of the reference's declarations (potential.h:42-58, integrator.h:5-17, accelerator.h:22-31,
    neighbor.h:42-50; transcribed in tests/abi/ref_structs.c)"""
    import subprocess
    abi = os.path.join(ROOT, "tests", "abi")
    outs = []
    for src, inc in (("ref_structs.c", []), ("our_structs.c", ["-I" + os.path.join(ROOT, "ddcmd_amd", "csrc", "host"), "-I" + os.path.join(ROOT, "include")])):
        exe = str(tmp_path / src.replace(".c", ""))
        subprocess.check_call(["gcc", "-std=gnu99", "-Wall", "-I" + abi] + inc + ["-o", exe, os.path.join(abi, src)])
        outs.append(subprocess.check_output([exe], text=True).splitlines())
    ref, ours = outs
    assert len(ref) > 40 and ref == ours, [(a, b) for a, b in zip(ref, ours) if a != b]


def test_potential_object_is_complete(built):
    """potential_init fills the members ddcMD's drivers read: itype, getCutoffs (charmmCutoff's answer), commMode"""
    text = open(os.path.join(ROOT, "ddcmd_amd", "csrc", "host", "plugin.c")).read()
    for needle in ("potential->getCutoffs =", "potential->itype = MARTINI", "RCUT_LOCAL", "in->itype ="):
        assert needle in text, needle


def test_driver_process_grid(built):
    """the multi-rank driver's process grid: the deck's ddc { lx ly lz } when it fits the launch, else WORLD_SIZE factored
    onto the axes with the widest bricks (bench.py's 2x1x1 / 2x2x1 / 2x2x2 for a cubic box)"""
    lib = ddcmd_amd.load_library()
    lib.plugin_plan_grid.argtypes = [ctypes.c_int] * 4 + [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)]
    lib.plugin_plan_grid.restype = None

    def plan(world, l, box):
        h = (ctypes.c_double * 9)(box[0], 0, 0, 0, box[1], 0, 0, 0, box[2])
        g = (ctypes.c_int * 3)()
        lib.plugin_plan_grid(world, l[0], l[1], l[2], h, g)
        return tuple(g)

    cube = (100.0, 100.0, 100.0)
    assert plan(1, (1, 1, 1), cube) == (1, 1, 1)
    assert plan(2, (1, 1, 1), cube) == (2, 1, 1) and plan(4, (1, 1, 1), cube) == (2, 2, 1) and plan(8, (1, 1, 1), cube) == (2, 2, 2)
    assert plan(8, (4, 2, 1), cube) == (4, 2, 1)                 # the DDC object wins when it multiplies to the rank count
    assert plan(8, (4, 4, 1), cube) == (2, 2, 2)                 # ... and is ignored when it does not
    assert plan(6, (0, 0, 0), cube) == (2, 3, 1)
    assert plan(4, (1, 1, 1), (200.0, 50.0, 50.0)) == (4, 1, 1)  # a slab is cut along its long axis
    assert plan(8, (1, 1, 1), (50.0, 50.0, 400.0)) == (1, 1, 8)


def test_driver_fails_loudly_without_a_gpu(built, tmp_path):
    """ddcmi_md (the C host layer) on a machine without a HIP device: a message and a non-zero exit, no CPU force path"""
    import subprocess
    lib = ddcmd_amd.load_library()
    lib.ddcmi_device_count.restype = ctypes.c_int
    if lib.ddcmi_device_count() > 0:
        return
    exe = os.path.join(ROOT, "ddcmd_amd", "bin", "ddcmi_md")
    deck = os.path.join(ROOT, "tests", "golden", "lipid_deck", "object.data")
    out = subprocess.run([exe, "-o", deck, "-d", str(tmp_path / "data")], capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "no HIP device" in out.stderr
    assert not os.path.exists(str(tmp_path / "data")) or os.path.getsize(str(tmp_path / "data")) == 0
