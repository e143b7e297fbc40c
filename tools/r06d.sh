set -x
cd /root/repo
mkdir -p gpurun_out/r06d
O=gpurun_out/r06d
for S in f64 raw; do EBOS_HIP_LIBRARY=$PWD/event_based_bos_amd/lib/libebos_lean_stamps.so python tools/stamp_plan_lean.py --source $S > $O/stamps_$S.txt 2>&1; cat $O/stamps_$S.txt | grep -v amdgpu.ids; done
python bench.py --no-cpu-baseline > $O/bench_poll.json 2> $O/bench_poll.err; cut -c1-300 $O/bench_poll.json
EBOS_COMBINE_POLL=0 python bench.py --no-cpu-baseline --no-extras > $O/bench_counted.json 2> $O/bench_counted.err; cut -c1-300 $O/bench_counted.json
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_sharding.py -q -m gpu -x > $O/tests.log 2>&1; tail -5 $O/tests.log
