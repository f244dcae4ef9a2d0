#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): SQ / LDS / MFMA-busy counters for every kernel of a reduced bench pass (full-size GEMM
# launches: 1024-cell chunks), in separate --pmc passes (8 SQ slots per pass), no trace domains besides --kernel-trace.
#   tools/collect_pmc_sq.sh <tag>   ->  gpurun_out/pmc_<tag>/sq_summary.{json,txt}   (copy into profiles/<tag>/)
set -eo pipefail
TAG=${1:-r2}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-dropin --cells 8000 --size 1280 --streams 1"
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
P2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE"
P3="SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F6F4 SQ_INSTS_VALU_MFMA_MOPS_F8 SQ_BUSY_CU_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_DATA_FIFO_FULL GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  echo "[pmc] pass $i: $P"; date
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d "$OUT/p$i" -o p$i -- python3 "$ROOT/bench.py" $ARGS > "$OUT/p$i.json" 2> "$OUT/p$i.log"
done
python3 "$ROOT/tools/summarize_sq.py" "$OUT"
find "$OUT" -name '*kernel_trace.csv' -delete
find "$OUT" -name '*counter_collection.csv' -delete
find "$OUT" -name '*agent_info.csv' -delete
du -sh "$OUT"
