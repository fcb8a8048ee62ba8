"""N > 1 on hardware.  One process per GPU over RCCL (backend "nccl") when the box has at least two GPUs -- skipped on the
one-GPU boxes of this pool --, and the SAME worker script with every rank on GPU 0 over gloo, which runs everywhere, so the
script's logic is known to be right before it first meets a second GPU (three rounds without an
N > 1 RCCL run).  Ranks are fresh child processes (never a re-exec of a process that holds the GPU), each under a
timeout that kills the whole group."""
import json
import os
import signal
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "multi_rank_worker.py")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _ngpus():
    import torch
    return torch.cuda.device_count()


def _run_ranks(world, backend, outdir, B, timeout=600):
    port = str(_free_port())
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, WORKER, str(r), str(world), port, backend, str(outdir), str(B)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, start_new_session=True)
             for r in range(world)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout)[0])
    except subprocess.TimeoutExpired:
        for p in procs:
            try:
                os.killpg(p.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
        pytest.fail("ranks did not finish in %d s" % timeout)
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d:\n%s" % (r, outs[r][-4000:])
        assert os.path.exists(os.path.join(str(outdir), "ok%d" % r))


@pytest.mark.parametrize("world,B", [(2, 37), (2, 64), (3, 10)])
def test_ranks_share_one_gpu_over_gloo(tmp_path, world, B):
    """the worker script with every rank on GPU 0 (exchanges staged through the host): ragged and equal shards"""
    _run_ranks(world, "gloo", tmp_path, B)


@pytest.mark.parametrize("B", [37, 64])
def test_two_gpus_over_rccl(tmp_path, B):
    """one GPU per rank, RCCL: torch.distributed collectives, the grouped point-to-point gather, and the library's own
    wx_comm_* / wx_allgather_out_* / wx_allreduce_moments_* entry points, against the single-rank results"""
    if _ngpus() < 2:
        pytest.skip("needs two GPUs (this pool's boxes have one)")
    _run_ranks(2, "nccl", tmp_path, B)
    assert "nccl rank 0/2 device 0" in open(os.path.join(str(tmp_path), "ok0")).read()
    assert "nccl rank 1/2 device 1" in open(os.path.join(str(tmp_path), "ok1")).read()


def test_all_gpus_over_rccl(tmp_path):
    """every GPU of the node (8 on the scaling box), ragged shards"""
    n = _ngpus()
    if n < 3:
        pytest.skip("needs more than two GPUs")
    _run_ranks(n, "nccl", tmp_path, 8 * n + 3)


def _bench(args, env_extra=None, timeout=900, expect_failure=False):
    env = dict(os.environ, **(env_extra or {}))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                       timeout=timeout)
    assert (r.returncode != 0) if expect_failure else (r.returncode == 0), (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


@pytest.mark.parametrize("mode", ["p2p", "collective", "auto"])
def test_bench_two_gpus_over_rccl(tmp_path, mode):
    """`bench.py --gpus 2 --workload cfg2 --batch 64` end to end over RCCL: both exchange schedules, the gathered
    reconstruction of two ranks bit-identical to one rank's, and the line says which backend, RCCL version and devices"""
    if _ngpus() < 2:
        pytest.skip("needs two GPUs (this pool's boxes have one)")
    one, two = str(tmp_path / "one.npy"), str(tmp_path / "two.npy")
    common = ["--workload", "cfg2", "--batch", "64", "--steps", "3", "--warmup", "1", "--no-cpu", "--watchdog", "300"]
    _bench(common + ["--dump", one])
    j = _bench(common + ["--gpus", "2", "--gather", mode, "--dump", two])
    assert j["n_gpus"] == 2 and j["ranks"]["backend"] == "nccl" and j["ranks"]["rccl_version"]
    assert j["ranks"]["devices"] == [0, 1]
    assert j["with_allgather"]["mode"] == ("collective" if mode == "collective" else "p2p")
    assert np.array_equal(np.load(one), np.load(two))


@pytest.mark.parametrize("mode", ["p2p", "collective"])
def test_bench_gather_modes_on_one_gpu(tmp_path, mode):
    """both exchange schedules of bench.py's all-gather loop, two ranks on this GPU over gloo; the line carries the build
    identity of the library and each rank's device"""
    one, two = str(tmp_path / "one.npy"), str(tmp_path / "two.npy")
    common = ["--workload", "cfg2", "--batch", "37", "--steps", "2", "--warmup", "1", "--no-cpu", "--watchdog", "600"]
    j1 = _bench(common + ["--dump", one])
    assert j1["build"].startswith("libwaveletsext_hip") and "src=" in j1["build"]
    j2 = _bench(common + ["--gpus", "2", "--gather", mode, "--chunks", "3", "--dump", two], {"WX_BENCH_BACKEND": "gloo"})
    assert j2["with_allgather"]["mode"] == mode and j2["with_allgather"]["p2p_error"] is None
    assert j2["ranks"]["devices"] == [0, 0] and j2["ranks"]["backend"] == "gloo"
    assert np.array_equal(np.load(one), np.load(two))


def test_bench_watchdog_ends_a_stuck_rank():
    """a rank that does not finish inside --watchdog seconds says where it is and exits 3 (so a hung collective cannot hold
    a machine until gpurun's own limit): a watchdog shorter than the start-up makes every run 'stuck'"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "cfg2", "--batch", "64", "--steps", "1",
                        "--warmup", "1", "--no-cpu", "--no-also", "--watchdog", "0.05"], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 3, (r.returncode, r.stderr[-800:])
    assert "watchdog: rank 0 of 1 not finished" in r.stderr


def test_bench_abandons_a_stuck_exchange_loop_and_keeps_the_headline():
    """the all-gather loop is reported next to `value`, never part of it: if it does not finish inside --gather-timeout rank 0 prints
    the line of the main loop (serialised before the loop started) with the failure noted and `exchange_abandoned`, and every rank
    leaves with exit code 4 -- so the launcher reports a NON-ZERO code (a hung collective is never a healthy run) while the headline
    number survives on stdout.  A timeout shorter than the loop's warm-up makes every run 'stuck'."""
    j = _bench(["--workload", "cfg2", "--batch", "37", "--steps", "2", "--warmup", "1", "--no-cpu", "--watchdog", "600", "--gpus", "2",
                "--gather-timeout", "0.0001"], {"WX_BENCH_BACKEND": "gloo"}, expect_failure=True)
    assert j["n_gpus"] == 2 and j["value"] > 0 and j["ms_per_step"] > 0
    assert "abandoned" in j["with_allgather"]["error"] and j["with_allgather_value"] is None
    assert j["exchange_abandoned"] is True
