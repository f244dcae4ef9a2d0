#!/bin/bash
# same-box A/B of two builds of the library through the default bench workload, interleaved:  tools/ab_lib.sh libribca_ab_<name>.so [rounds]
# (the second build comes from tools/build_ab_lib.py; prints cells/s, ms per step and the per-kernel milliseconds of every run)
cd ${GRAFT_REPO_ROOT:-.}
ALT=${1:?alternative library}
for rep in $(seq 1 ${2:-3}); do
  for lib in $ALT libribca_hip.so; do
    RIBCA_LIB=$lib timeout -k 10 250 python3 bench.py --steps 3 --no-cpu-baseline --no-dropin 2>/dev/null \
      | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['ms_per_step'], json.dumps(d['per_kernel_ms']))" || exit 1
  done
done
