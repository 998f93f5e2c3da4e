"""print the headline fields of a bench.py JSON line"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('ms/step %.3f  value %.0f %s' % (d['ms_per_step'], d['value'], d['unit']))
r = d.get('roofline', {})
print('roofline: %s %.1f %s frac %.3f avg %.1f us traffic %s' % (r.get('kernel'), r.get('achieved', 0), r.get('unit'), r.get('frac', 0), r.get('avg_launch_us', 0), r.get('traffic')))
c = d.get('conv_stack', {})
print('conv_stack: %.1f TF frac %.3f' % (c.get('achieved', 0), c.get('frac', 0)))
print('cpu_baseline:', d.get('cpu_baseline'))
