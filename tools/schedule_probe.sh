#!/bin/bash
# The N > 1 schedule and the headline-kernel experiments of round 3, on ONE box, interleaved:
#   * bench.py at N = 1 with the inputs as 15 arrays / tile-interleaved, launches dealt onto 1 / 2 / 4 streams;
#   * the 1-rank RCCL path (PEM_BENCH_FORCE_DIST=1) at the full 1.25e6-sample shard with 1 / 2 / 4 / 8 equal chunks on tile
#     boundaries and 4 round-aligned chunks, on 1 and 2 streams: `value` (with the all-gather) and `config.value_without_gather`.
# usage: tools/schedule_probe.sh OUTDIR [STEPS]
set -o pipefail
out=${1:-gpurun_out/schedule}
steps=${2:-200}
mkdir -p "$out"
common="--steps $steps --warmup 20 --no-cpu-baseline --full-config-samples 0 --no-single-batch"
run() {   # name, env assignments..., -- args
    name=$1; shift
    echo "== $name" >&2
    env "$@" > "$out/$name.json" 2> "$out/$name.err" || { echo "FAILED $name" >&2; tail -5 "$out/$name.err" >&2; return 1; }
}
for rep in 1 2; do
    run n1_soa_1s_r$rep     python bench.py $common --streams 1 || exit 1
    run n1_tile_1s_r$rep    python bench.py $common --streams 1 --layout tile || exit 1
    run n1_soa_4s_r$rep     python bench.py $common --streams 4 || exit 1
    run n1_soa_2s_r$rep     python bench.py $common || exit 1
    run n1_tile_2s_r$rep    python bench.py $common --layout tile || exit 1
    for st in 1 2; do
        for k in 1 2 4 8; do
            run dist_k${k}_${st}s_r$rep PEM_BENCH_FORCE_DIST=1 MASTER_PORT=$((29600 + k + 10 * rep + 100 * st)) python bench.py $common --chunks $k --streams $st || exit 1
        done
        run dist_k4_roundalign_${st}s_r$rep PEM_BENCH_FORCE_DIST=1 MASTER_PORT=$((29650 + rep + 100 * st)) python bench.py $common --chunks 4 --chunk-align round --streams $st || exit 1
    done
done
python - "$out" <<'PY'
import json, sys, glob, os
out = sys.argv[1]
print(f'{"run":28s} {"ms/step":>9s} {"evals/s":>11s} {"no-gather":>11s} {"kernel us":>10s} {"frac":>6s}  chunks')
for f in sorted(glob.glob(os.path.join(out, '*.json'))):
    try:
        line = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), 'unreadable', e)
        continue
    c = line['config']
    ng = c.get('value_without_gather')
    print(f'{os.path.basename(f)[:-5]:28s} {line["ms_per_step"]:9.4f} {line["value"]:11.4e} {ng if ng is None else format(ng, "11.4e")!s:>11s} '
          f'{1e3 * line["roofline"]["kernel_ms_mean"]:10.2f} {line["roofline"]["frac"]:6.3f}  {[b[1] for b in c["launch_rounds"]["chunks"]]}')
PY
