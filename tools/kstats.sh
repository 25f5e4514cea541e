#!/bin/bash
# Run ON THE GPU BOX: per-kernel durations of one script.  Usage: bash tools/kstats.sh <tag> <script.py> [args...]
# -> gpurun_out/ks_<tag>/ (rocprofv3 csv) and gpurun_out/ks_<tag>.txt (per kernel: calls, median, max, total)
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/ks_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -f csv -d "$OUT" -o t -- python3 "$GRAFT_REPO_ROOT/$1" "${@:2}" > "$OUT/log.txt" 2>&1 || { tail -20 "$OUT/log.txt"; exit 1; }
grep -v "^W2026\|^E2026" "$OUT/log.txt"
python3 - "$OUT" <<'PY' | tee "$GRAFT_REPO_ROOT/gpurun_out/ks_$(basename $OUT | sed s/ks_//).txt"
import csv, sys, statistics
from collections import defaultdict
from pathlib import Path
d = defaultdict(list)
for f in Path(sys.argv[1]).rglob('*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        d[r['Kernel_Name']].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
print(f"{'kernel':70s} {'calls':>6s} {'median us':>10s} {'max us':>9s} {'total ms':>9s}")
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:40]:
    print(f"{k[:70]:70s} {len(v):6d} {statistics.median(v):10.1f} {max(v):9.1f} {sum(v) / 1e3:9.2f}")
PY
