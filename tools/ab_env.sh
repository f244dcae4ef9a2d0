#!/bin/bash
# same-box A/B of one environment switch on the working tree, interleaved: tools/ab_env.sh VAR A_VALUE B_VALUE [rounds]
cd ${GRAFT_REPO_ROOT:-.}
VAR=$1; A=$2; B=$3; N=${4:-3}
for rep in $(seq 1 $N); do
for v in $A $B; do
  echo -n "== $VAR=$v (rep $rep): "
  env $VAR=$v timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-dropin 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
done
