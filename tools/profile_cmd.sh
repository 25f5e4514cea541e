#!/bin/bash
# Generic PMC profile of an arbitrary python command on the GPU box: tools/profile_cmd.sh <tag> <script.py> [args]
set -u
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d "$OUT/trace" -o run -- python3 "$@" > "$OUT/log.txt" 2>&1
for pass in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU" "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo "$pass" | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $pass -f csv -d "$OUT/pmc_$name" -o run -- python3 "$@" >> "$OUT/log.txt" 2>&1 || echo "pass failed: $pass"
done
python3 - "$OUT" <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
d = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/pmc_*/*counter_collection.csv'):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name']
        if 'at::native' in k or 'rocclr' in k: continue
        d[k[:60]][row['Counter_Name']].append(float(row['Counter_Value']))
for k, c in d.items():
    print('==', k)
    for name, v in sorted(c.items()):
        print(f'   {name:24s} {sum(v)/len(v):14.6g}  (n={len(v)})')
dur = collections.defaultdict(list)
for f in glob.glob(out + '/trace/*kernel_trace.csv'):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name']
        if 'at::native' in k or 'rocclr' in k: continue
        dur[k[:60]].append(int(row['End_Timestamp']) - int(row['Start_Timestamp']))
for k, v in dur.items():
    print(f'dur {k}: mean {sum(v)/len(v)/1e3:.1f} us n={len(v)}')
PY
