set -x
mkdir -p gpurun_out/r05j
cd /root/repo
bash tools/build_stamps_lib.sh > gpurun_out/r05j/build.log 2>&1; tail -2 gpurun_out/r05j/build.log
export EBOS_HIP_LIBRARY=$PWD/event_based_bos_amd/lib/libebos_stamps.so
python tools/stamp_resident.py > gpurun_out/r05j/stamps_plain.txt 2>&1
python tools/stamp_resident.py --blur 1.0 > gpurun_out/r05j/stamps_blur.txt 2>&1
cat gpurun_out/r05j/stamps_plain.txt gpurun_out/r05j/stamps_blur.txt | grep -v amdgpu
