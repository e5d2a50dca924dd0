mkdir -p gpurun_out/r05
timeout 900 python bench.py > gpurun_out/r05/bench_default.json 2> gpurun_out/r05/bench_default.err
tail -1 gpurun_out/r05/bench_default.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value', d['value'], d['ms_per_step'], 'serial', d['serial_frame'], 'generic', d['generic_program'])
print('roofline', d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['roofline'].get('pmc_note'))
print('host', d['host_buffers'])
print('cpu', {k:v for k,v in d['cpu_baseline'].items() if k!='reference'})
print('f64', d['f64'], 'fwd', d['fwd_only'], 'unb', d['unbiased'])"
timeout 600 python bench.py --absorb 0.5 --min-bounces 1 --no-cpu-baseline > gpurun_out/r05/bench_roulette.json 2>/dev/null
tail -1 gpurun_out/r05/bench_roulette.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('roulette value', d['value'], d['ms_per_step'], 'serial', d['serial_frame'])"
