for rep in 1 2; do for l in ${LIBS:-libse_raux0.so libsceneego_hip.so}; do
  SCENEEGO_HIP_LIB=sceneego_amd/$l timeout 300 python bench.py --no-extras --no-cpu-baseline --no-repeats 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('$l', d['value'], d['single_stream_value'], d['step_ms']['median'], r['avg_launch_ms'], r['stage']['conv7_avg_ms'], d['parity']['pass'])"
done; done
