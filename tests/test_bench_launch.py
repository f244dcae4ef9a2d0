"""bench.py's own N-rank launch path: ``--gpus N`` without WORLD_SIZE must start N ranks itself, and a
rank count that does not match ``--gpus`` must fail instead of printing a mislabeled line.  ``--launch-check`` runs the
rendezvous + the padded all-gather of dist.all_gather_rows only, so this runs on the CPU (gloo)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["RIBCA_DIST_BACKEND"] = "gloo"
    return env


def _last_json(stdout: str) -> dict:
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert lines, stdout
    return json.loads(lines[-1])


def test_gpus2_launches_two_ranks():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], env=_env(), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = _last_json(r.stdout)
    assert out["n_gpus"] == 2 and out["gather_ok"] is True
    # what makes a future 8-GPU line checkable: the backend and the world size THE COLLECTIVE saw, its time and payload
    coll = out["collective"]
    assert coll["backend"] == "gloo" and coll["world_size_seen"] == 2
    assert coll["allgather_ms_per_step"] > 0 and coll["bytes_per_rank"] == 500 * 33 * 4      # rank 0 holds rows [0, 500) of 1001, 33 floats each


def test_parent_counts_gpus_without_hip(monkeypatch):
    """the launcher parent must not initialise the GPU: devices are counted from the environment or the kfd topology"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1,2")
    assert mod.visible_gpu_count() == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert mod.visible_gpu_count() == 0
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    monkeypatch.delenv("CUDA_VISIBLE_DEVICES", raising=False)
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES", raising=False)
    assert mod.visible_gpu_count() >= 0          # sysfs path: no exception without /sys/class/kfd
    src = open(os.path.join(ROOT, "bench.py")).read()
    body = src[src.index("def launch_ranks"):src.index("def launch_check")]
    assert "torch.cuda" not in body


def test_gpus1_stays_in_process():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--launch-check"], env=_env(), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _last_json(r.stdout)["n_gpus"] == 1


def test_rank_count_mismatch_is_an_error():
    env = _env()
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 2 and "WORLD_SIZE=1" in r.stderr


def test_rank0_only_passes_issue_no_collective():
    """The roofline block runs profiled passes on rank 0 ALONE, after the other ranks have printed nothing and left: a collective inside
    them hangs or dies ("Connection closed by peer": found by the 2-rank rehearsal of round 4 -- rounds 2 and 3 shipped it).  Static
    check: every one_pass(...) call of that block says gather=False, and one_pass honours it."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    start = src.index("if not args.no_roofline and rank == 0:")
    end = src.index("if not args.no_dropin and rank == 0", start)
    block = src[start:end]
    calls = [block[i:block.index(")", i) + 1] for i in range(len(block)) if block.startswith("one_pass(", i)]
    assert calls, "the roofline block no longer calls one_pass: update this test"
    assert all("gather=False" in c for c in calls), calls
    assert "if sharded and gather:" in src


def test_a_line_whose_collective_saw_another_world_size_is_refused():
    """bench.py itself compares the world size the all-gather's process group reports with --gpus and exits non-zero
    instead of printing a line labelled with a GPU count the collective did not run on.  Static check of the guard (it sits behind the timed
    region of a sharded GPU run, which this CPU suite cannot reach) + the record it reads."""
    import importlib.util
    src = open(os.path.join(ROOT, "bench.py")).read()
    i = src.index('out["collective"] = collective_record(')
    guard = src[i:i + 700]
    assert 'out["collective"]["world_size_seen"] != args.gpus' in guard and "sys.exit(3)" in guard
    assert src.index("sys.exit(3)") < src.index("print(json.dumps(out))")      # the guard runs before the line is printed
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    import torch
    rec = mod.collective_record(1, 0.0, 1, torch.zeros(0))
    assert rec["world_size_seen"] == 1 and rec["backend"] is None
