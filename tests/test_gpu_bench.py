"""bench.py end to end on the GPU in its N-rank form: two ranks sharing the one device, gloo carrying the all-gather
(RIBCA_DIST_BACKEND=gloo RIBCA_SHARE_GPU=1) -- the whole line, roofline block included (it runs on rank 0 alone and must not enter a
collective: the 2-rank rehearsal of round 4 found that it did)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_bench_line_on_a_shared_gpu():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(RIBCA_DIST_BACKEND="gloo", RIBCA_SHARE_GPU="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--cells", "3000", "--size", "768", "--steps", "1",
                        "--warmup", "1", "--no-cpu-baseline", "--no-dropin"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["value"] > 0
    assert out["collective"]["backend"] == "gloo" and out["collective"]["world_size_seen"] == 2
    assert out["roofline"]["achieved"] > 0 and out["roofline"]["launches"] > 0
    assert set(("normalise", "label_table", "crop", "vit", "all_gather", "vote", "d2h")) <= set(out["per_stage_ms"])
    assert out["replicated_preprocessing_ms"] > 0
