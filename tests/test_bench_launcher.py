"""bench.py --gpus N must really start N ranks.  CPU-only: the ranks rendezvous over gloo and
time a stand-in step (--stub-step); what is checked is the launcher contract -- N processes, one JSON line from
rank 0 with n_gpus == N, a failing rank or a world-size mismatch ends the run with a non-zero exit code."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env=None, timeout=240):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, env=e, capture_output=True, text=True, timeout=timeout)


def test_launcher_spawns_two_ranks():
    r = _run(["--gpus", "2", "--stub-step", "--steps", "4", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["warmup"] == 1 and out["stub"] is True
    assert out["value"] > 0 and out["ms_per_step"] >= 2.0          # the stand-in step sleeps 2 ms


def test_single_rank_needs_no_launcher():
    r = _run(["--gpus", "1", "--stub-step", "--steps", "2", "--warmup", "0"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_world_size_mismatch_is_an_error():
    # started as ONE rank of a 1-rank world but told --gpus 2: refuse instead of measuring one GPU
    r = _run(["--gpus", "2", "--stub-step"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE=1" in r.stderr


def test_too_few_devices_is_an_error():
    import torch
    if torch.cuda.device_count() >= 2:
        return
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode == 2 and "visible" in r.stderr


def test_torchrun_style_env_is_honoured():
    # the driver's own launch: python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2
    port = "29611"
    procs = []
    for rank in range(2):
        e = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, BENCH, "--gpus", "2", "--stub-step", "--steps", "2", "--warmup", "0"],
                                      env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=240) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-500:] for o in outs]
    assert json.loads(outs[0][0].strip().splitlines()[-1])["n_gpus"] == 2
    assert outs[1][0].strip() == ""
