#!/bin/bash
# bench.py --config 4: batched launches vs per-window calls (graph replay / eager)
mkdir -p gpurun_out/c4
run() {
  python bench.py --config 4 "$@" 2> gpurun_out/c4/err.log | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*', '| Gev/s', round(d['value']/1e3,1), 'ms/pass', d['ms_per_step'], 'host enqueue', d['host_enqueue_ms_per_step'], 'graph', d['pass_is_a_replayed_hip_graph'], 'windows/launch', d['windows_per_launch'], d['contrast_first_windows'][:2])"
  grep -h "failed\|Error" gpurun_out/c4/err.log | head -3
}
run --batch 64
run --batch 64 --no-tail-stream
run --batch 32
run --batch 16 --no-tail-stream
run --batch 0 --streams 3
