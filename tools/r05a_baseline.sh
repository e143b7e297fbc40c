set -x
mkdir -p gpurun_out/r05a
cd /root/repo
python tools/run_cmax.py --config_file tests/golden/config_hot_plate1.json --height 260 --width 346 > gpurun_out/r05a/run_cmax_ref_346x260.json 2> gpurun_out/r05a/run_cmax_ref_346x260.err
python tools/run_cmax.py --config_file tests/golden/config_hot_plate1.json > gpurun_out/r05a/run_cmax_ref_720x1280.json 2> gpurun_out/r05a/run_cmax_ref_720x1280.err
python tools/run_cmax.py --config_file configs/cmax_hot_plate1.yaml > gpurun_out/r05a/run_cmax_own.json 2> gpurun_out/r05a/run_cmax_own.err
python tools/run_cmax.py --config_file configs/cmax_hot_plate1.yaml --n-iter 600 > gpurun_out/r05a/run_cmax_own_600.json 2> gpurun_out/r05a/run_cmax_own_600.err
timeout 600 python tools/bench_skew_solver.py --out gpurun_out/r05a/skew_solver.json > gpurun_out/r05a/skew_solver.log 2>&1
tail -3 gpurun_out/r05a/*.json gpurun_out/r05a/*.err
