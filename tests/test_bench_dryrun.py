"""bench.py --gpus N without a GPU (FV3_BENCH_DRYRUN=1): the driver's own launch line for N = 2, 4, 8 (px x py blocks of a doubly periodic
domain, halo messages between the ranks) and N = 6 (one cubed-sphere face per rank, cube-edge messages through the library's exchange, whole
Jablonowski-Williamson steps), with the host logic harness in place of the HIP library and gloo in place of RCCL -- so that the first
SCALE run on hardware cannot die on plumbing: rank layout, communicators, barriers, the MAX over ranks, ONE JSON line from rank 0."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("n", [2, 4, 6, 8])
def test_bench_gpus_n_dry_run(n, tmp_path):
    if not os.path.exists(os.path.join(ROOT, "tests", "hostemu", "libfv3_hostemu.so")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "hostemu")], check=True, capture_output=True, timeout=1800)
    env = dict(os.environ, FV3_BENCH_DRYRUN="1", OMP_NUM_THREADS="1")
    env.pop("FV3_MI355X_SO", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(29600 + n), os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1",
           "--nx", "16", "--npz", "6", "--strong-domain", "32", "--strong-steps", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]                       # the contract: ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == n and out["steps"] == 2 and out["warmup"] == 1 and out["finite"] is True
    assert out["metric"] == "c_sw+d_sw cell-updates/s" and out["scaling"] == "weak" and "dry_run" in out
    cells = 16 * 16 * 6
    assert abs(out["value"] - n * cells * 2 / (out["ms_per_step"] * 2e-3)) <= 1e-6 * out["value"]   # whole-job units over the MAX-over-ranks time
    if n == 6:
        assert out["config"]["layout"] == "6 faces x 1x1" and out["sphere_six_gpus"].get("finite") is True, out["sphere_six_gpus"]
    else:
        assert out["config"]["halo"] == "RCCL send/recv" and out["config"]["layout"] in ("2x1", "1x2", "2x2", "4x2", "2x4")
        # BASELINE config 4 beside the headline: ONE 32 x 32 domain split over the same layout (16 x 32, 16 x 16, 8 x 16 blocks)
        ss = out["strong_scaling"]
        assert ss and "error" not in ss, ss
        px, py = (int(x) for x in out["config"]["layout"].split("x"))
        assert ss["scaling"] == "strong" and ss["finite"] is True and f"{32 // px}x{32 // py} per GPU" in ss["workload"]
        assert abs(ss["value"] - 32 * 32 * 6 / (ss["ms_per_step"] * 1e-3)) <= 1e-6 * ss["value"]   # the DOMAIN's cells, whatever N


@pytest.mark.parametrize("n", [2, 4, 8])
def test_bench_strong_scaling_domain_dry_run(n, tmp_path):
    """--domain D: `value` is the strong-scaling number of ONE D x D domain (BASELINE config 4's decomposition), the block shrinks with N"""
    env = dict(os.environ, FV3_BENCH_DRYRUN="1", OMP_NUM_THREADS="1")
    env.pop("FV3_MI355X_SO", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(29700 + n), os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "1",
           "--domain", "32", "--npz", "6"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == n and out["scaling"] == "strong" and out["finite"] is True and out["strong_scaling"] is None
    px, py = (int(x) for x in out["config"]["layout"].split("x"))
    assert f"split {px}x{py}: {32 // px}x{32 // py} per GPU" in out["config"]["workload"]
    assert abs(out["value"] - 32 * 32 * 6 / (out["ms_per_step"] * 1e-3)) <= 1e-6 * out["value"]
