#!/bin/bash
# same-box A/B of three builds of the library: ab_prev (bf16 split, 4 B), ab_f16 (fp16 split, 4 B), working tree
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for d in ab_prev ab_f16 .; do
  extra="--no-dropin"; [ "$d" = "ab_prev" ] && extra=""
  echo "== $d (rep $rep)"
  (cd $d && timeout -k 10 200 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline $extra 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
done
done
