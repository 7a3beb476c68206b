"""Print the interesting parts of a bench.py JSON line (file argument): value, per-step median, clock, roofline, kernels[]."""
import json
import sys

for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        d = json.loads(l)
        t = d.get('timing') or {}
        print('value %.2f img/s  mean %.3f ms  median %s ms  clock %s' % (d['value'], d['ms_per_step'], t.get('ms_per_step_median'),
                                                                         (d.get('engine_clock') or {}).get('median_mhz')))
        r = d['roofline']
        print('roofline %s frac %.4f (%.1f TF, %.4f ms x %d); serial %s' % (r['kernel'], r['frac'], r['achieved'], r['avg_launch_ms'],
                                                                              r['launches_per_step'], (d.get('roofline_serial') or {}).get('frac')))
        tot = 0.0
        for e in d['kernels']:
            if not e['kernel'].startswith('weight gradient'):
                tot += e['ms_per_step']
            print('  %-92s %3d %7.3f  %s' % (e['kernel'][:92], e['launches'], e['ms_per_step'], e.get('frac')))
        print('  sum of bracketed serial kernels %.3f ms' % tot)
        if d.get('alt_modes'):
            print('alt', {k: v['images_per_s'] for k, v in d['alt_modes'].items()})
        if d.get('cpu_baseline'):
            print('cpu', d['cpu_baseline'])
