#!/bin/bash
# Run ON THE GPU BOX (through gpurun): kernel-trace stats + separate PMC passes for the coupled bench.
# Usage: bash tools/profile_gpu.sh <tag> [extra bench.py flags]
# Outputs under gpurun_out/prof_<tag>/ ; tools/summarize_profile.py turns them into profiles/<tag>_*.{md,json}
set -u
TAG=${1:-r01}; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
# --full-config-samples 0: only the per-GPU shard launches (1.25e6 samples), so that every traced launch is the launch the
# roofline line is about (eight batches in rotation, no single-batch comparison run); the whole-config launch (1e7 samples)
# is traced separately below
# --spin-up-ms 0: every launch of the run is traced / counted, so the untimed clock spin-up of bench.py is left out here (the
# traced mean then covers the 30 warm-up + 300 timed + 50 event-timed launches, the first ~80 of them on ramping clocks)
BENCH="bench.py --steps 20 --warmup 3 --spin-up-ms 0 --campaign-samples 0 --no-cpu-baseline --full-config-samples 0 --no-single-batch $*"
# the kernel trace runs the bench with its DEFAULT step counts (300 timed + 30 warm-up launches), so that the traced
# average duration is the one bench.py itself reports; the counter passes below only need a few launches
echo "== kernel trace + stats" | tee "$OUT/log.txt"
rocprofv3 --kernel-trace --stats -f csv -d "$OUT/trace" -o bench -- python3 bench.py --spin-up-ms 0 --campaign-samples 0 --no-cpu-baseline --full-config-samples 0 --no-single-batch $* >> "$OUT/log.txt" 2>&1 || exit 1
echo "== kernel trace of the whole-config launch (1e7 samples)" | tee -a "$OUT/log.txt"
rocprofv3 --kernel-trace --stats -f csv -d "$OUT/trace_full" -o bench -- python3 bench.py --spin-up-ms 0 --campaign-samples 0 --no-cpu-baseline --no-single-batch --steps 3 --warmup 1 $* >> "$OUT/log.txt" 2>&1 || echo "full-config trace failed"
# PMC passes, one hardware block at a time (FETCH_SIZE and WRITE_SIZE do not fit one TCC pass)
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum"; do
  name=$(echo "$pass" | tr ' ' '_' | cut -c1-40)
  echo "== pmc $pass" | tee -a "$OUT/log.txt"
  rocprofv3 --pmc $pass -f csv -d "$OUT/pmc_$name" -o bench -- python3 $BENCH >> "$OUT/log.txt" 2>&1 || echo "pass failed: $pass" | tee -a "$OUT/log.txt"
done
ls -R "$OUT" | head -50
