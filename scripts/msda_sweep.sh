for v in 12 12; do
  echo -n "vec=$v "
  OCCF_MSDA_VEC=$v timeout 200 python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],2), round(d['kernels']['msda3d']['total_ms'],3))"
done
