#!/bin/bash
# (GPU box) non-temporal output stores of the MX qkv / fc1 launches (RIBCA_MX_NT bit 0: q / k / v rows, bit 1: the MX3 planes of h): time per
# kernel and FETCH_SIZE per kernel, 0 vs 3.   bash tools/ab_mx_nt.sh -> gpurun_out/ab_mx_nt/
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/ab_mx_nt
mkdir -p "$OUT"
cd "$ROOT"
for r in 1 2; do
  for v in 0 3; do
    RIBCA_MX_NT=$v timeout -k 10 200 python tools/bench_block.py immune_full 2>&1 | grep -v amdgpu.ids | sed "s/^/[nt $v] /"
  done
done | tee "$OUT/time.txt"
cd /tmp && export TMPDIR=/tmp
for v in 0 3; do
  export RIBCA_MX_NT=$v
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
for k, (kb, ids) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:5]:
    print(f"[nt {v}] {k[:70]:70s} launches {len(ids):5d}  fetch {2.0 * kb * 1024 / max(len(ids), 1) / 1e6:8.1f} MB per launch (2 x FETCH_SIZE)")
PY
  find "$OUT/fetch_$v" -name '*.csv' -delete
done
