#!/bin/bash
# Profiles behind DESIGN.md section 7 (run on the GPU box from the repo root):  bash benchmarks/profile.sh <outdir>
#   1. rocprofv3 --kernel-trace --stats of the command bench.py times (kernel durations)
#   2. separate --pmc passes (HBM bytes: FETCH_SIZE, WRITE_SIZE; issue statistics: SQ_*) over bare launches of the
#      headline kernel (benchmarks/qp_launch.py = the same call), one pass per run as the PMC slot limits require
set -u
OUT=${1:-gpurun_out/prof}
mkdir -p "$OUT"
export TMPDIR=/tmp
run() { echo "== $*"; timeout -s KILL 240 "$@" > "$OUT/last.log" 2>&1 || { echo "FAILED: $*"; tail -5 "$OUT/last.log"; exit 1; }; }
run rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_bench" -- python3 bench.py --steps 20 --warmup 3 --no-cpu
run rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_configs" -- python3 benchmarks/configs.py --only c1,c3,c4,c5
for s in 0.3 1.0; do
  run rocprofv3 --output-format csv --pmc FETCH_SIZE -d "$OUT/pmc_fetch_$s" -- python3 benchmarks/qp_launch.py $s 3
  run rocprofv3 --output-format csv --pmc WRITE_SIZE -d "$OUT/pmc_write_$s" -- python3 benchmarks/qp_launch.py $s 3
done
run rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY -d "$OUT/pmc_sq1" -- python3 benchmarks/qp_launch.py 0.3 3
run rocprofv3 --output-format csv --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d "$OUT/pmc_sq2" -- python3 benchmarks/qp_launch.py 0.3 3
python3 benchmarks/pmc_summary.py --steps 3 qp_solve_kernel "$OUT/pmc_fetch_0.3" "$OUT/pmc_write_0.3" "$OUT/pmc_sq1" "$OUT/pmc_sq2" > "$OUT/pmc_s03.json"
python3 benchmarks/pmc_summary.py --steps 3 qp_solve_kernel "$OUT/pmc_fetch_1.0" "$OUT/pmc_write_1.0" > "$OUT/pmc_s10.json"
python3 benchmarks/pmc_summary.py --steps 3 mpqp_screen_kernel "$OUT/pmc_fetch_0.3" "$OUT/pmc_write_0.3" "$OUT/pmc_sq1" "$OUT/pmc_sq2" > "$OUT/pmc_screen_s03.json"
python3 benchmarks/pmc_summary.py --steps 3 mpqp_gram_kernel "$OUT/pmc_fetch_0.3" "$OUT/pmc_write_0.3" "$OUT/pmc_sq1" "$OUT/pmc_sq2" > "$OUT/pmc_gram_s03.json"
python3 benchmarks/pmc_summary.py --steps 3 mpqp_gram_kernel "$OUT/pmc_fetch_1.0" "$OUT/pmc_write_1.0" > "$OUT/pmc_gram_s10.json"
python3 benchmarks/pmc_summary.py --steps 3 mpqp_screen_kernel "$OUT/pmc_fetch_1.0" "$OUT/pmc_write_1.0" > "$OUT/pmc_screen_s10.json"
find "$OUT/trace_bench" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats_bench.csv" \;
find "$OUT/trace_configs" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats_configs.csv" \;
# keep only the summaries (the raw traces are large)
find "$OUT" -name "*.csv" | head -20
rm -rf "$OUT"/trace_* "$OUT"/pmc_fetch_* "$OUT"/pmc_write_* "$OUT"/pmc_sq*
ls -la "$OUT"
