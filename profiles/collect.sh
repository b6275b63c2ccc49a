#!/bin/bash
# Collects the profiles of the bench path on one MI355X (run from the repository root on the GPU box):
#   bash profiles/collect.sh <tag> [batches in flight, default 8]   -> gpurun_out/prof_<tag>/*.csv  (copy the ones to keep into profiles/)
# One rocprofv3 run per counter set (the counter blocks cannot be shared between passes); every run under its own time limit.
set -u
tag=${1:-r2}
inflight=${2:-8}
root=$PWD
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
BENCH="python3 $root/bench.py --no-cpu-baseline --streams 1 --in-flight $inflight"
summ() {   # summ <stats|pmc> <dir> <csv>
    db=$(find "$2" -name '*results.db' | head -n 1)
    [ -n "$db" ] || { echo "no results.db under $2"; ls -R "$2" | head -20; return 1; }
    python3 "$root/profiles/summarize_rocpd.py" "$1" "$db" "$3"
}
cd /tmp
trap 'rm -rf "$out"/stats "$out"/stats_complete "$out"/fetch_size "$out"/write_size "$out"/sq "$out"/mfma "$out"/mfma_complete_table' EXIT
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d "$out/stats" -- $BENCH --steps $((4 * inflight)) --warmup $inflight > "$out/stats.log" 2>&1 || { echo "stats pass failed"; exit 1; }
summ stats "$out/stats" "$out/kernel_stats.csv"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d "$out/stats_complete" -- $BENCH --steps $((4 * inflight)) --warmup $inflight --complete-table > "$out/stats_complete.log" 2>&1 || { echo "stats (complete table) pass failed"; exit 1; }
summ stats "$out/stats_complete" "$out/kernel_stats_complete_table.csv"
pass() {   # pass <name> <counters...>   (BENCH_EXTRA: more bench.py arguments)
    name=$1; shift
    timeout -s KILL 240 rocprofv3 --pmc "$@" -d "$out/$name" -- $BENCH --steps $inflight --warmup 0 ${BENCH_EXTRA:-} > "$out/$name.log" 2>&1 || { echo "pmc pass $name failed"; return 1; }
    summ pmc "$out/$name" "$out/pmc_$name.csv"
}
pass fetch_size FETCH_SIZE || exit 1
pass write_size WRITE_SIZE || exit 1
pass sq SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS || exit 1
pass mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES || echo "mfma pass failed (counter names?)"
BENCH_EXTRA=--complete-table pass mfma_complete_table SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_BUSY_CYCLES || echo "mfma (complete table) pass failed"
rm -rf "$out"/stats "$out"/stats_complete "$out"/fetch_size "$out"/write_size "$out"/sq "$out"/mfma "$out"/mfma_complete_table
ls -la "$out"
