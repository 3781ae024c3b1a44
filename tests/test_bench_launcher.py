"""CPU: bench.py under the driver's own multi-GPU launcher.

The driver starts N>1 as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
--master-port P bench.py ...`.  libddcmi.so is built and validated against /opt/rocm's HIP and RCCL; torch bundles
its own copies under the same sonames, so a bench process that imports torch would bind libddcmi to a runtime it was
never tested on.  bench.py therefore keeps torch out of the process and meets its peers over libddcmi's own TCP
rendezvous (the launcher keeps MASTER_PORT for its store: the port travels through a file).  `--check-runtime`
stops before the first device call, so this runs without a GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_world2_bench_process_maps_only_system_rocm(built):
    port = 29700 + os.getpid() % 200
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--check-runtime"]
    env = dict(os.environ)
    env.pop("DDCMI_RDZV_FILE", None); env.pop("DDCMI_RDZV_PORT", None)
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    recs = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert sorted(r["rank"] for r in recs) == [0, 1]
    tokens = {r["token"] for r in recs}
    assert len(tokens) == 1 and tokens.pop().startswith("ddcmi-runtime-check-")      # rank 0's broadcast reached rank 1
    for r in recs:
        assert r["world"] == 2 and r["ranks_met"] == 2
        assert r["torch_loaded"] is False
        hip = [l for l in r["runtime_libs"] if "libamdhip64" in l]
        rccl = [l for l in r["runtime_libs"] if "librccl" in l]
        assert hip and rccl
        for l in r["runtime_libs"]:
            assert os.path.realpath(l).startswith(os.path.realpath("/opt/rocm") + os.sep), "foreign runtime mapped: %s" % l
            assert "torch" not in l


def test_bench_source_has_no_torch_import():
    import re
    pat = re.compile(r"^\s*(import|from)\s+torch\b", re.M)
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert not pat.search(src) and "torch.cuda" not in src
    for f in ("martini.py", "_lib.py", "deck.py", "synth.py", "__init__.py"):
        assert not pat.search(open(os.path.join(ROOT, "ddcmd_amd", f)).read())


def test_bench_rank_that_never_arrives_ends_the_launch(built):
    """a launch whose second rank never shows up: rank 0 gives up at the rendezvous deadline with a message and a non-zero exit
    (no hang, no re-exec) -- the first contact with a real multi-GPU node must not be able to wedge the driver"""
    import time
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        env = dict(os.environ)
        env.update({"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "1",
                    "DDCMI_RDZV_FILE": os.path.join(d, "port"), "DDCMI_TRANSPORT": "host", "DDCMI_RDZV_TIMEOUT": "3"})
        t0 = time.time()
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--lattice", "10"], cwd=ROOT, env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
        assert p.returncode == 2 and time.time() - t0 < 60
        assert "rendezvous failed" in p.stderr and "only 1 of 2 ranks arrived" in p.stderr, p.stderr[-800:]
        assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


def _no_launcher_env():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "DDCMI_RDZV_FILE", "DDCMI_RDZV_PORT"):
        env.pop(k, None)
    return env


def test_bench_gpus_n_without_a_launcher_starts_n_ranks(built):
    """`python3 bench.py --gpus N` as the driver starts N = 1 -- no launcher, no WORLD_SIZE -- used to run ONE rank
    and print n_gpus 1.  The command itself now starts its N ranks (fresh children, before any library is loaded)."""
    for n in (2, 8):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--check-runtime"], cwd=ROOT, env=_no_launcher_env(),
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-3000:]
        recs = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
        assert sorted(r["rank"] for r in recs) == list(range(n))
        assert all(r["world"] == n and r["ranks_met"] == n and r["torch_loaded"] is False for r in recs)
        assert len({r["token"] for r in recs}) == 1


def test_bench_gpus_must_agree_with_the_launchers_world(built):
    env = _no_launcher_env()
    env.update({"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--check-runtime"], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert p.returncode == 2 and "must agree" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_bench_self_launch_ends_with_the_failing_ranks_code(built):
    """one rank of a self-started launch fails (here: an unsupported GPU count is refused before any rank starts; a rank that dies
    takes the launch down with its exit code)"""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--check-runtime"], cwd=ROOT, env=_no_launcher_env(),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert p.returncode == 2 and "1, 2, 4 or 8" in p.stderr
    env = _no_launcher_env()
    env["DDCMI_BENCH_FAIL_RANK"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--check-runtime"], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    # (rank 0 may notice its peer's death -- a reset connection -- and fail first: either way the launch ends non-zero, at once, and says who)
    assert p.returncode in (1, 7) and "ending the other ranks" in p.stderr and not [l for l in p.stdout.splitlines() if l.startswith("{")]
