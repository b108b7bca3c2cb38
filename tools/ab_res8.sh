export NUM_MELS=40
for i in 1 2; do
for lib in build/old/libhowl_old.so howl_amd/libhowl_hip.so; do
  HOWL_HIP_LIBRARY=$PWD/$lib timeout 300 python bench.py --config c3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']; o = r['other_kernels']
print('$lib', d['value'], d['ms_per_step'], 'pair', r['avg_launch_ms'], 'fwd', [v.get('avg_launch_ms') for k, v in o.items() if k.startswith('conv3x3')])"
done; done
