# A/B of two builds of the library on ONE box (boxes of the pool differ by 2-4 %): bash tools/ab_libs.sh libA.so libB.so [bench args]
A=$1; B=$2; shift; shift
for rep in 1 2 3; do
  for L in $A $B; do
    echo "$(basename $L): $(CAPF_LIB=$L python bench.py --no-cpu-baseline --profile-steps 1 $* 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], 'fps', j['roofline']['kernel'], j['roofline']['avg_launch_us'], 'us')")"
  done
done
