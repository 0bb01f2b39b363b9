#!/bin/bash
# pmc_run.sh OUTDIR -- python3 script args...   : kernel trace + three SQ passes + FETCH / WRITE passes, each its own
# rocprofv3 run (counters never combined with other trace domains), then the per-kernel summary.
# The profiler's preload initialises the GPU before the program starts, and bench.py's cpu_baseline forks worker
# processes: a bench.py target is therefore always run with --no-cpu (appended here when missing).
set -e
if [ "$#" -lt 3 ] || [ "$2" != "--" ]; then
    echo "usage: pmc_run.sh OUTDIR -- python3 script args..." >&2
    exit 2
fi
OUT=$1; shift; shift
case " $* " in
    *bench.py*) case " $* " in *" --no-cpu "*) ;; *) set -- "$@" --no-cpu ;; esac ;;
esac
export TMPDIR=/tmp
mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats -d "$OUT/trace" --output-format csv -- "$@" ${TRACE_EXTRA} > "$OUT/trace.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d "$OUT/sq1" --output-format csv -- "$@" > "$OUT/sq1.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d "$OUT/sq2" --output-format csv -- "$@" > "$OUT/sq2.log" 2>&1
rocprofv3 --pmc FETCH_SIZE -d "$OUT/fetch" --output-format csv -- "$@" > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE -d "$OUT/write" --output-format csv -- "$@" > "$OUT/write.log" 2>&1
python3 "$(dirname "$0")/pmc_summary.py" "$OUT/trace" "$OUT/sq1" "$OUT/sq2" "$OUT/fetch" "$OUT/write" ${PMC_MATCH:+--match "$PMC_MATCH"} --json "$OUT/summary.json" > "$OUT/summary.txt"
