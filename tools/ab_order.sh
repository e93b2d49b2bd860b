export CAPF_LIB=$PWD/tools/ab/libcapf_diag.so
for o in 0123 0312 0321 3210 0213 0132 3012; do
  echo "== order $o: $(CAPF_WINO43_ORDER=$o python tools/bench_wino.py 2>&1 | tail -1 | sed 's/.*F(4,3) group//')  $(CAPF_WINO43_ORDER=$o python bench.py --no-cpu-baseline --profile-steps 1 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=64 fps', j['value'])")"
done
