#!/bin/bash
# (GPU box) the W-panel tile walk of the MX kernel (RIBCA_MX_PANEL_KB): parity, time per kernel, and FETCH_SIZE per kernel, off vs on.
#   bash tools/ab_mx_panel.sh 2048   ->  gpurun_out/ab_mx_panel/
set -eo pipefail
KB=${1:-2048}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/ab_mx_panel
mkdir -p "$OUT"
cd "$ROOT"
export RIBCA_MX_PANEL_KB=$KB
timeout -k 10 400 python -m pytest tests/test_gpu_mx.py tests/test_gpu_kernels.py -x -q -k "mx or vit_forward" > "$OUT/tests_panel_on.log" 2>&1
tail -2 "$OUT/tests_panel_on.log"
for r in 1 2; do
  RIBCA_MX_PANEL_KB=0 timeout -k 10 200 python tools/bench_block.py immune_full 2>&1 | grep -v amdgpu.ids | sed "s/^/[panel off] /"
  RIBCA_MX_PANEL_KB=$KB timeout -k 10 200 python tools/bench_block.py immune_full 2>&1 | grep -v amdgpu.ids | sed "s/^/[panel $KB KB] /"
done | tee "$OUT/time.txt"
cd /tmp && export TMPDIR=/tmp
for v in 0 $KB; do
  export RIBCA_MX_PANEL_KB=$v
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch_$v" -o fetch -- python3 "$ROOT/tools/bench_block.py" immune_full > "$OUT/fetch_$v.log" 2>&1
  python3 - "$OUT/fetch_$v" "$v" <<'PY' | tee -a "$OUT/fetch.txt"
import csv, glob, sys, collections
d, v = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0.0, set()])
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] == "FETCH_SIZE":
            k = row["Kernel_Name"].split("(")[0].replace("void ribca::", "")
            acc[k][0] += float(row["Counter_Value"]); acc[k][1].add(row["Dispatch_Id"])
for k, (kb, ids) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:8]:
    print(f"[panel {v} KB] {k[:70]:70s} launches {len(ids):5d}  fetch {2.0 * kb * 1024 / max(len(ids), 1) / 1e6:8.1f} MB per launch (2 x FETCH_SIZE)")
PY
  find "$OUT/fetch_$v" -name '*.csv' -delete
done
