# A/B of the F(4,3) tile variants (diagnosis build): bash tools/ab_wino43.sh
export CAPF_LIB=$PWD/tools/ab/libcapf_diag.so
for v in 0 1 2; do
  echo "== CAPF_WINO43_SHORT=$v"
  CAPF_WINO43_SHORT=$v python tools/bench_wino.py 2>&1 | tail -6
  CAPF_WINO43_SHORT=$v python bench.py --no-cpu-baseline --profile-steps 1 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=64 fps', j['value'], 'wino us', j['roofline']['avg_launch_us'])"
  CAPF_WINO43_SHORT=$v python bench.py --no-cpu-baseline --profile-steps 1 --batch 256 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=256 fps', j['value'], 'wino us', j['roofline']['avg_launch_us'])"
done
CAPF_WINO43_SHORT=2 python tools/wino_level_timeline.py
