"""bench.py --gpus N: the launcher contract.

Started plainly with --gpus N > 1 the script starts its N ranks itself (torch.distributed.run on 127.0.0.1); any
disagreement between --gpus, WORLD_SIZE and the communicator ends the run with a non-zero code and no JSON line, so a
line that says n_gpus = 1 can never come out of a run that was asked for 8.  The CPU tests cover the refusals (nothing
here touches a GPU); the GPU test rehearses the whole N = 2 path on one device (gloo group, host-staged tile gather)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None, timeout=600):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "HR_BENCH_ONE_DEVICE", "HR_BENCH_FORCE_EXCHANGE"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          timeout=timeout, cwd=ROOT)


def _json_lines(out):
    return [l for l in out.decode(errors="replace").splitlines() if l.startswith("{")]


def test_world_size_mismatch_is_refused_before_the_gpu_is_touched():
    r = _run(["--gpus", "3", "--quick"], env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"}, timeout=120)
    assert r.returncode != 0
    assert b"--gpus 3 but WORLD_SIZE=2" in r.stderr
    assert not _json_lines(r.stdout)
    # ... also when the environment says one rank and the command line eight
    r = _run(["--gpus", "8", "--quick"], env={"WORLD_SIZE": "1"}, timeout=120)
    assert r.returncode != 0 and not _json_lines(r.stdout)
    r = _run(["--gpus", "0", "--quick"], timeout=120)
    assert r.returncode != 0


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="CPU-only check: on a GPU box the launcher's children would render")
def test_plain_start_with_two_gpus_launches_ranks_and_fails_loudly_without_devices():
    # no GPU here: the launcher must start the ranks (torch.distributed.run), they must refuse, and the launcher must pass the failure on
    r = _run(["--gpus", "2", "--quick", "--steps", "2", "--workload", "c1"], timeout=400)
    assert r.returncode != 0
    assert not _json_lines(r.stdout)
    err = r.stderr.decode(errors="replace")
    assert "torch.distributed" in err or "ChildFailedError" in err or "device(s) visible" in err or "HIP" in err, err[-2000:]


@pytest.mark.gpu
def test_two_rank_rehearsal_on_one_device():
    args = ["--quick", "--parity-seconds", "30", "--steps", "70", "--warmup", "1", "--workload", "c2", "--width", "512", "--height", "288"]  # (two exchanges on the way + the final one)
    r = _run(["--gpus", "2"] + args, env={"HR_BENCH_ONE_DEVICE": "1"}, timeout=900)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-3000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout.decode(errors="replace")[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rehearsal_one_device"] is True and d["steps"] == 70
    assert "REHEARSAL" in d["metric"]
    assert d["value"] > 0 and d["extra"]["rays"] > 0
    # SURVEY 8e acceptance, checked on the line itself: the frame assembled from the two ranks' tiles equals the frame ONE rank renders
    # (same passes => same digest), and both equal the CPU oracle's render of those passes, every pixel
    r1 = _run(["--gpus", "1"] + args, timeout=900)
    assert r1.returncode == 0, r1.stderr.decode(errors="replace")[-3000:]
    d1 = json.loads(_json_lines(r1.stdout)[0])
    assert d1["n_gpus"] == 1 and d1["steps"] == 70
    assert isinstance(d["frame_sha256"], str) and len(d["frame_sha256"]) == 64
    assert d["frame_sha256"] == d1["frame_sha256"]
    for line in (d, d1):
        p = line["parity"]
        assert p["bit_exact"] is True and p["rel_l2"] == 0.0 and p["pixels"] == 512 * 288 and p["passes"] == 70
    ss = d1["ms_per_step_steady_state"]
    assert ss["macro_steps"] >= 3 and ss["k_trace_launch_ms"]["min"] > 0
    if ss["steady_state_steps"] >= 2:
        assert d1["ms_per_step_min"] <= d1["ms_per_step_median"] <= ss["max"]


@pytest.mark.gpu
def test_rccl_exchange_path_with_one_rank():
    # The real collective library on the real device, as far as one GPU allows: HR_BENCH_FORCE_EXCHANGE=1 runs the N > 1 code path — an
    # "nccl" (= RCCL) process group, hr_frame_pack_owned -> dist.gather on a side stream after every resolved batch, hr_frame_unpack,
    # the final assembly — in a single rank.  Same passes => the same digest as the plain run, and the oracle's frame.
    args = ["--quick", "--parity-seconds", "30", "--steps", "70", "--warmup", "1", "--workload", "c2", "--width", "512", "--height", "288"]
    r = _run(["--gpus", "1"] + args, env={"HR_BENCH_FORCE_EXCHANGE": "1"}, timeout=900)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-3000:]
    d = json.loads(_json_lines(r.stdout)[0])
    assert "RCCL gather" in d["config"]["sharding"]
    r1 = _run(["--gpus", "1"] + args, timeout=900)
    assert r1.returncode == 0, r1.stderr.decode(errors="replace")[-3000:]
    d1 = json.loads(_json_lines(r1.stdout)[0])
    assert d["frame_sha256"] == d1["frame_sha256"]
    assert d["parity"]["bit_exact"] is True and d["parity"]["pixels"] == 512 * 288
