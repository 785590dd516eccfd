#!/bin/bash
# PMC passes (rocprofv3 --pmc, one pass per run, counters only) for the headline kernels; run on the GPU box:
#   [PMC_TAG=r03] bash tools/profile_pmc.sh [workload ...]        -> gpurun_out/<tag>_pmc/<workload>_<pass>.csv
# Every pass is its own process under a hard time limit; the first failure ends the script.
set -e -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/${PMC_TAG:-r03}_pmc
mkdir -p $OUT
WORK=${@:-"fwd:256:a32 bwd:256:a32 fwdwide:32:a32 kmeans kmeans_prepared kmeans_shard nms64k nms64k_dense nms256 nms2000 pairwise"}
declare -A PASS
PASS[fetch]="FETCH_SIZE"
PASS[write]="WRITE_SIZE"
PASS[sq1]="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES"
PASS[sq2]="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
PASS[grbm]="GRBM_GUI_ACTIVE"
for w in $WORK; do
  IFS=: read -r name a1 a2 <<< "$w"
  for p in fetch write sq1 sq2 grbm; do
    d=$OUT/${name}_$p
    rm -rf $d
    echo "== $name $p" >&2
    timeout -s KILL 170 rocprofv3 --pmc ${PASS[$p]} -d $d -o p -- python3 tools/prof_roi.py $name $a1 $a2 > $d.log 2>&1 || { tail -5 $d.log >&2; exit 1; }
    python3 tools/rocpd_summary.py pmc $d/p_results.db > $OUT/${name}_$p.csv
    rm -rf $d
  done
done
echo done
