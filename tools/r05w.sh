set -x
mkdir -p gpurun_out/r05w
cd /root/repo
python tools/run_cmax.py --config_file tests/golden/config_hot_plate1.json --height 260 --width 346 > gpurun_out/r05w/run_cmax_ref_346x260.json 2> gpurun_out/r05w/err1.txt
python tools/run_cmax.py --config_file tests/golden/config_hot_plate1.json > gpurun_out/r05w/run_cmax_ref_720x1280.json 2> gpurun_out/r05w/err2.txt
python tools/run_cmax.py --config_file configs/cmax_hot_plate1.yaml --n-iter 600 > gpurun_out/r05w/run_cmax_own_600.json 2> gpurun_out/r05w/err3.txt
cat gpurun_out/r05w/*.json
