# round 6: GPU tests of the ADVICE fixes + samplers + bench line format
set -x
cd /root/repo
mkdir -p gpurun_out/r06b
O=gpurun_out/r06b
timeout 1500 python -m pytest tests/test_ingest.py tests/test_sharding.py tests/test_abi.py -q -m gpu -x > $O/tests_a.log 2>&1; tail -5 $O/tests_a.log
timeout 1500 python -m pytest tests/test_solver.py -q -m gpu -x -k "random_sampler or tpe_sampler or integer_pixel_2dof or torn_resident or hot_plate1 or fractional" > $O/tests_b.log 2>&1; tail -15 $O/tests_b.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_steps20.json 2> $O/bench.err; cut -c1-1500 $O/bench_steps20.json
