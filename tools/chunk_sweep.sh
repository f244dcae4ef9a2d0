#!/bin/bash
cd $GRAFT_REPO_ROOT
for cfg in "1024 3" "512 3" "256 3" "128 3" "64 3" "128 6" "256 6" "64 8" "2048 2"; do
  set -- $cfg
  echo "== chunk $1 streams $2"
  timeout -k 10 200 python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-dropin --chunk $1 --streams $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
