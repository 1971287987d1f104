for v in old new old new; do
  line=$(NANOVLLM_HIP_LIB=$PWD/lib_$v.so timeout -k 10 200 python bench.py --no-extras --steps 64 --warmup 8)
  echo "lib=$v $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["prefill"]["ms_per_chunk"])')"
done
