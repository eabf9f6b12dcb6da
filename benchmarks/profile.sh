#!/bin/bash
# Profiles behind DESIGN.md section 7 (run on the GPU box from the repo root):  bash benchmarks/profile.sh <outdir>
#   1. bench.py lines at s = 0.3 and s = 1.0
#   2. rocprofv3 --kernel-trace of benchmarks/qp_launch.py (the call bench.py times), summarised per STEP with the warm-up
#      steps left out (benchmarks/kernel_summary.py), plus the plain --stats table of the same run
#   3. separate --pmc passes (HBM bytes: FETCH_SIZE, WRITE_SIZE; issue statistics: SQ_*), one pass per run as the PMC
#      slot limits require
#   4. kernel + memory-copy trace of the host-to-host pipeline (benchmarks/pcie_pipeline.py) and its overlap summary
#   5. the other configurations (benchmarks/configs.py)
set -u
OUT=${1:-gpurun_out/prof}
mkdir -p "$OUT"
export TMPDIR=/tmp
run() { echo "== $*"; timeout -s KILL 240 "$@" > "$OUT/last.log" 2>&1 || { echo "FAILED: $*"; tail -5 "$OUT/last.log"; exit 1; }; }
runo() { out=$1; shift; echo "== $* > $out"; timeout -s KILL 300 "$@" > "$out" 2> "$OUT/last.log" || { echo "FAILED: $*"; tail -5 "$OUT/last.log"; exit 1; }; }
runo "$OUT/bench_s03.json" python3 bench.py --steps 20 --warmup 3
runo "$OUT/bench_s10.json" python3 bench.py --steps 20 --warmup 3 --scale 1.0
for s in 0.3 1.0; do
  tag=s${s/./}
  run rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$tag" -- python3 benchmarks/qp_launch.py $s 13
  python3 benchmarks/kernel_summary.py "$OUT/trace_$tag" --launches 13 --skip-launches 3 > "$OUT/kernel_summary_$tag.json"
  find "$OUT/trace_$tag" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats_qp_launch_$tag.csv" \;
  run rocprofv3 --output-format csv --pmc FETCH_SIZE -d "$OUT/pmc_fetch_$s" -- python3 benchmarks/qp_launch.py $s 6
  run rocprofv3 --output-format csv --pmc WRITE_SIZE -d "$OUT/pmc_write_$s" -- python3 benchmarks/qp_launch.py $s 6
done
run rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY -d "$OUT/pmc_sq1" -- python3 benchmarks/qp_launch.py 0.3 6
run rocprofv3 --output-format csv --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d "$OUT/pmc_sq2" -- python3 benchmarks/qp_launch.py 0.3 6
for k in qp_solve_kernel mpqp_screen_kernel mpqp_gram_kernel; do
  short=${k#mpqp_}; short=${short%_kernel}; short=${short#qp_}
  python3 benchmarks/pmc_summary.py --last 3 --calls 18 $k "$OUT/pmc_fetch_0.3" "$OUT/pmc_write_0.3" "$OUT/pmc_sq1" "$OUT/pmc_sq2" > "$OUT/pmc_${short}_s03.json"
  python3 benchmarks/pmc_summary.py --last 3 --calls 18 $k "$OUT/pmc_fetch_1.0" "$OUT/pmc_write_1.0" > "$OUT/pmc_${short}_s10.json"
done
python3 benchmarks/pmc_traffic.py "$OUT" > "$OUT/pmc_traffic.json"
run rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$OUT/trace_pcie" -- python3 benchmarks/pcie_pipeline.py --variant picked2c_depth3 --steps 12
python3 benchmarks/pcie_overlap.py "$OUT/trace_pcie" --steps 12 > "$OUT/pcie_pipeline.json"
run rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_configs" -- python3 benchmarks/configs.py --only c1,c3,c4,c5
find "$OUT/trace_configs" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats_configs.csv" \;
runo "$OUT/configs.jsonl" python3 benchmarks/configs.py
# keep only the summaries (the raw traces are large)
rm -rf "$OUT"/trace_* "$OUT"/pmc_fetch_* "$OUT"/pmc_write_* "$OUT"/pmc_sq* "$OUT/last.log"
ls -la "$OUT"
