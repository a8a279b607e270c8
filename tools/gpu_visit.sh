#!/usr/bin/env bash
# ONE parameterised GPU-box visit (replaces round 5's 22 one-off gpu_r05[a-v].sh scripts, which are in the history at
# 55c447f).  Every argument is a step, run in order; the output of each goes to gpurun_out/<tag>_*.  Steps:
#   tag=<name>                      prefix of the files written (default: visit)
#   test=<pytest args>              python -m pytest -m gpu -q -x <args>      -> <tag>_pytest_<n>.log, last lines echoed
#   line=<name>:<bench args>        python bench.py <args> --no-cpu-baseline  -> <tag>_bench_<name>.json, one summary line
#   fullline=<name>:<bench args>    the same WITH the cpu baseline leg
#   env=<VAR=val,...>               environment for the steps that follow (env= alone clears it)
#   lib=<variant>                   LIPREADING_HIP_LIB = lipreading_amd/_lib/alt/<variant>.so for the steps that follow
#                                   (built beforehand by tools/build_variant.sh; lib= alone: the in-tree build)
#   kt=<name>:<bench args>          rocprofv3 --kernel-trace --stats of the bench configuration -> <tag>_<name>_kernel_stats.txt
#   tl=<name>:<anchor>:<bench args> kernel-trace timeline of one timed step (tools/rocpd_timeline.py; TL_STEP env) -> <tag>_<name>_step_timeline.txt
#   py=<script> [args]              python <script> args                      -> <tag>_<script>.log, echoed
#   sh=<command>                    bash -c <command>, echoed
# e.g.  gpurun --timeout 900 -- 'bash tools/gpu_visit.sh tag=r06a "test=tests/test_gpu_encoder.py -k cluster" \
#          "line=gru256_b64:--regime landmarks --model gru256 --batch 64" env=LIPREADING_X=1 "line=x:--regime pixels"'
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp
TAG=visit; NTEST=0; ENVS=()

summarise() {   # file, label
  python - "$1" "$2" <<'PY'
import json, sys
f, label = sys.argv[1:3]
try:
  d = json.loads(open(f).read().strip().splitlines()[-1])
  r = d.get("roofline") or {}
  p = d.get("parity") or {}
  print(label, d["ms_per_step"], "ms  min", (d.get("timing") or {}).get("ms_per_step_min"), " frames/s", d["value"],
        "| parity", p.get("abs_diff"), p.get("ok"), "| errs", d.get("pair_errors"), "| loss", d.get("final_loss"),
        "|", d["config"].get("launch_probe"), d["config"].get("decoder_recurrence"))
  k = r.get("avg_launch_us_by_kernel") or {}
  if k:
    print("   ", {n: round(v, 1) for n, v in k.items()}, "frac", r.get("frac"))
except Exception as e:
  print(label, "UNREADABLE", e)
  try:
    print(open(f.replace(".json", ".err")).read()[-2500:])
  except OSError:
    pass
PY
}

for step in "$@"; do
  kind=${step%%=*}; arg=${step#*=}
  case $kind in
    tag) TAG=$arg ;;
    env) ENVS=(); if [ -n "$arg" ]; then IFS=',' read -r -a ENVS <<< "$arg"; fi ;;
    lib) if [ -n "$arg" ]; then export LIPREADING_HIP_LIB=$R/lipreading_amd/_lib/alt/$arg.so; [ -f "$LIPREADING_HIP_LIB" ] || echo "lib=$arg: no such variant library"
         else unset LIPREADING_HIP_LIB; fi ;;
    test)
      NTEST=$((NTEST + 1)); log=$OUT/${TAG}_pytest_$NTEST.log
      # shellcheck disable=SC2086
      env ${ENVS[@]+"${ENVS[@]}"} timeout 1800 python -m pytest -m gpu -q -x $arg > "$log" 2>&1
      echo "test[$arg] exit $?: $(grep -E 'passed|failed|error' "$log" | tail -1)"; grep -E "^(FAILED|ERROR)|Warning: " "$log" | head -8 ;;
    line|fullline)
      name=${arg%%:*}; bargs=${arg#*:}; f=$OUT/${TAG}_bench_$name.json
      extra="--no-cpu-baseline"; [ "$kind" = fullline ] && extra=""
      # shellcheck disable=SC2086
      env ${ENVS[@]+"${ENVS[@]}"} timeout 900 python bench.py $bargs $extra 2> "${f%.json}.err" | tail -1 > "$f"
      summarise "$f" "line[$name ${ENVS[*]:-} ${LIPREADING_HIP_LIB:+lib=$(basename "$LIPREADING_HIP_LIB")}]" ;;
    kt)
      name=${arg%%:*}; bargs=${arg#*:}
      # shellcheck disable=SC2086
      (cd /tmp && env ${ENVS[@]+"${ENVS[@]}"} timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/kt_$name" -o kt -- \
        python "$R/bench.py" $bargs --no-graph --no-cpu-baseline --steps 10 --warmup 2 --repeats 1 > /dev/null 2>&1)
      python tools/rocpd_summary.py "$(find "$OUT/kt_$name" -name '*.db' | head -1)" 60 > "$OUT/${TAG}_${name}_kernel_stats.txt"
      rm -rf "$OUT/kt_$name"; head -30 "$OUT/${TAG}_${name}_kernel_stats.txt" | cut -c1-160 ;;
    tl)
      name=${arg%%:*}; rest=${arg#*:}; anchor=${rest%%:*}; bargs=${rest#*:}
      # shellcheck disable=SC2086
      (cd /tmp && env ${ENVS[@]+"${ENVS[@]}"} timeout 900 rocprofv3 --kernel-trace -d "$OUT/tl_$name" -o kt -- \
        python "$R/bench.py" $bargs --steps 8 --warmup 2 --repeats 1 --no-cpu-baseline > /dev/null 2>&1)
      DB=$(find "$OUT/tl_$name" -name '*.db' | head -1)
      python tools/rocpd_timeline.py "$DB" "$anchor" "${TL_STEP:--9}" > "$OUT/${TAG}_${name}_step_timeline.txt"
      python tools/rocpd_summary.py "$DB" 60 > "$OUT/${TAG}_${name}_tl_kernel_stats.txt"
      rm -rf "$OUT/tl_$name"; head -5 "$OUT/${TAG}_${name}_step_timeline.txt" ;;
    py)
      script=${arg%% *}; log=$OUT/${TAG}_$(basename "${script%.py}").log
      # shellcheck disable=SC2086
      env ${ENVS[@]+"${ENVS[@]}"} timeout 1500 python $arg > "$log" 2>&1; echo "py[$arg] exit $?"; tail -40 "$log" ;;
    sh) env ${ENVS[@]+"${ENVS[@]}"} timeout 1500 bash -c "$arg" ;;
    *) echo "unknown step: $step" ;;
  esac
done
