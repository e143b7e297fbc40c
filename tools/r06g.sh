set -x
cd /root/repo
mkdir -p gpurun_out/r06g
O=gpurun_out/r06g
timeout 3000 python -m pytest tests -q -m gpu -x > $O/tests.log 2>&1; tail -6 $O/tests.log
python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-900 $O/bench.json
python tools/bench_plan_build.py > $O/plan_build.json 2> $O/plan_build.err; cat $O/plan_build.json
python tools/bench_plan_build.py --events 100000 --height 260 --width 346 > $O/plan_build_100k.json 2> $O/plan_build_100k.err; cat $O/plan_build_100k.json
python tools/bench_plan_build.py --events 2000000 > $O/plan_build_2m.json 2> $O/plan_build_2m.err; cat $O/plan_build_2m.json
