#!/bin/bash
# same-box A/B: a reference snapshot directory (default ab_f16) against the working tree, interleaved, 3 rounds
cd $GRAFT_REPO_ROOT
REF=${1:-ab_f16}
for rep in 1 2 3; do
for d in $REF .; do
  echo "== $d (rep $rep)"
  (cd $d && timeout -k 10 200 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-dropin 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
done
done
