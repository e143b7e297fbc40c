cd /root/repo
mkdir -p gpurun_out/r06i
for N in 2000000 500000; do EBOS_HIP_LIBRARY=$PWD/event_based_bos_amd/lib/libebos_lean_stamps.so python tools/stamp_plan_lean.py --source raw --events $N 2>&1 | grep -v amdgpu.ids; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r06i/prof -- python3 /root/repo/tools/bench_plan_build.py --events 2000000 > /dev/null 2> /root/repo/gpurun_out/r06i/prof.err
EBOS_HIP_LIBRARY=/root/repo/event_based_bos_amd/lib/libebos_oldplan.so rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r06i/prof_old -- python3 /root/repo/tools/bench_plan_build.py --events 2000000 > /dev/null 2> /root/repo/gpurun_out/r06i/prof_old.err
cd /root/repo
for d in prof prof_old; do echo $d; find gpurun_out/r06i/$d -name "*kernel_stats.csv" | head -1 | xargs grep "lean_\|plan_parts" | cut -c1-70,150-400 | sed 's/(anonymous namespace):://g' | cut -c1-200; done
rm -rf gpurun_out/r06i/prof gpurun_out/r06i/prof_old
