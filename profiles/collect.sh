#!/bin/bash
# Collects the profiles of the bench path on one MI355X (run from the repository root on the GPU box):
#   bash profiles/collect.sh <tag> [batches in flight, default 8] [what: all | headline | others]
#     -> gpurun_out/prof_<tag>/*.csv  (copy the ones to keep into profiles/)
# One rocprofv3 run per counter set (the counter blocks cannot be shared between passes); every run under its own time limit.
# headline: the timed path of bench.py (BASELINE configs[1]).  others: the other BASELINE configs, one run per config
# (bench.py --other-configs <name> with a one-step headline in front), kernel trace + FETCH_SIZE / WRITE_SIZE / MFMA passes.
set -u
tag=${1:-r3}
inflight=${2:-8}
what=${3:-all}
root=$PWD
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
BENCH="python3 $root/bench.py --no-cpu-baseline --streams 1 --in-flight $inflight --no-other-configs"
summ() {   # summ <stats|pmc> <dir> <csv>
    db=$(find "$2" -name '*results.db' | head -n 1)
    [ -n "$db" ] || { echo "no results.db under $2"; ls -R "$2" | head -20; return 1; }
    python3 "$root/profiles/summarize_rocpd.py" "$1" "$db" "$3"
}
cd /tmp
trap 'find "$out" -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +' EXIT
pass() {   # pass <name> <counters...>   (BENCH_EXTRA: more bench.py arguments)
    name=$1; shift
    timeout -s KILL 240 rocprofv3 --pmc "$@" -d "$out/$name" -- $BENCH --steps $inflight --warmup 0 ${BENCH_EXTRA:-} > "$out/$name.log" 2>&1 || { echo "pmc pass $name failed"; return 1; }
    summ pmc "$out/$name" "$out/pmc_$name.csv"
    rm -rf "$out/$name"
}
if [ "$what" = all ] || [ "$what" = headline ]; then
    timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d "$out/stats" -- $BENCH --steps $((4 * inflight)) --warmup $inflight > "$out/stats.log" 2>&1 || { echo "stats pass failed"; exit 1; }
    summ stats "$out/stats" "$out/kernel_stats.csv"; rm -rf "$out/stats"
    timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d "$out/stats_complete" -- $BENCH --steps $((4 * inflight)) --warmup $inflight --complete-table > "$out/stats_complete.log" 2>&1 || { echo "stats (complete table) pass failed"; exit 1; }
    summ stats "$out/stats_complete" "$out/kernel_stats_complete_table.csv"; rm -rf "$out/stats_complete"
    pass fetch_size FETCH_SIZE || exit 1
    pass write_size WRITE_SIZE || exit 1
    pass sq SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS || exit 1
    pass mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES || echo "mfma pass failed (counter names?)"
fi
if [ "$what" = all ] || [ "$what" = others ]; then
    OTHER="python3 $root/bench.py --no-cpu-baseline --streams 1 --in-flight 1 --steps 1 --warmup 0 --repeats 0"
    for cfg in k8 six_d classification clipped fast_inference; do
        timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d "$out/stats_$cfg" -- $OTHER --other-configs $cfg > "$out/stats_$cfg.log" 2>&1 || { echo "stats pass ($cfg) failed"; continue; }
        summ stats "$out/stats_$cfg" "$out/kernel_stats_$cfg.csv"; rm -rf "$out/stats_$cfg"
    done
    for cfg in six_d classification fast_inference; do
        for cs in "fetch_size FETCH_SIZE" "write_size WRITE_SIZE" "mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_BUSY_CYCLES"; do
            set -- $cs; name=$1; shift
            timeout -s KILL 300 rocprofv3 --pmc "$@" -d "$out/${name}_$cfg" -- $OTHER --other-configs $cfg > "$out/${name}_$cfg.log" 2>&1 || { echo "pmc pass $name ($cfg) failed"; continue; }
            summ pmc "$out/${name}_$cfg" "$out/pmc_${name}_$cfg.csv"; rm -rf "$out/${name}_$cfg"
        done
    done
fi
ls -la "$out"
