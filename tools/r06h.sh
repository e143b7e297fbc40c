set -x
cd /root/repo
mkdir -p gpurun_out/r06h
O=gpurun_out/r06h
timeout 1200 python -m pytest tests/test_ingest.py -q -m gpu > $O/tests_ingest.log 2>&1; tail -3 $O/tests_ingest.log
for N in 100000 500000 2000000 5000000 10000000; do
  HW="--height 720 --width 1280"; if [ $N = 100000 ]; then HW="--height 260 --width 346"; fi
  EBOS_HIP_LIBRARY=$PWD/event_based_bos_amd/lib/libebos_oldplan.so python tools/bench_plan_build.py --events $N $HW 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('OLD', d['events'], {k: round(v,4) for k,v in d.items() if k.startswith('lean')})"
  python tools/bench_plan_build.py --events $N $HW 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('NEW', d['events'], {k: round(v,4) for k,v in d.items() if k.startswith('lean')})"
done
