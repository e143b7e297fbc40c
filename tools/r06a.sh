# round 6, first GPU call: small-flow evidence on HEAD + a baseline default line and the plan-build / combine figures to beat
set -x
cd /root/repo
mkdir -p gpurun_out/r06a
O=gpurun_out/r06a
python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
for F in 2 6; do
  python bench.py --flow-max $F --halo auto --no-cpu-baseline --no-extras > $O/bench_config2_flow$F.json 2> $O/e2_$F.err
  python bench.py --flow-max $F --halo 32 --no-cpu-baseline --no-extras > $O/bench_config2_h32_flow$F.json 2> $O/e2h_$F.err
  python bench.py --config 4 --flow-max $F --halo auto > $O/bench_config4_flow$F.json 2> $O/e4_$F.err
done
python bench.py --config 4 > $O/bench_config4.json 2> $O/e4.err
python tools/bench_plan_build.py > $O/plan_build.json 2> $O/plan_build.err
for f in $O/*.json; do echo $f; cut -c1-600 $f; done
