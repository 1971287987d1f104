set -euo pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/rs; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for mode in on off; do
  if [ $mode = off ]; then export NANOVLLM_ROWSCALE=0; fi
  rm -rf $O/t
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/t -o p -- python3 $R/tools/prefill_probe.py 1024 0 6 > $O/log_$mode.txt 2>&1
  python3 $R/tools/trace_breakdown.py "$(find $O/t -name '*kernel_trace.csv' | head -1)" > $O/chunk_$mode.csv
  echo "== $mode"; head -12 $O/chunk_$mode.csv; tail -3 $O/log_$mode.txt
done
rm -rf $O/t
