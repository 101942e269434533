import json, sys
d=json.load(open(sys.argv[1]))
for k in ('value','ms_per_step','em_iter_ms'): print(k,d[k])
print('frac',d['roofline']['frac'],'dom_ms',d['roofline']['avg_launch_ms'],'estep_mfma_frac',d['roofline']['estep_mfma_frac'])
print(d['kernels_ms']); print(d['em_kernels_ms'])
o=d.get('other_models') or {}
for k in o:
    if k.endswith('_ms') or k=='error': print(k,o[k])
print(d['parity'])
