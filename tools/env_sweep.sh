#!/bin/bash
# usage: tools/env_sweep.sh VAR v1 v2 ...   -- one short bench pass per value of an A/B environment switch (same box, back to back)
cd $GRAFT_REPO_ROOT
VAR=$1; shift
for v in "$@"; do
  echo "== $VAR=$v"
  env $VAR=$v timeout -k 10 200 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-dropin 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
