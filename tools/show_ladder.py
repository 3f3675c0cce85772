import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1], 'value', round(d['value'],3))
for r in d.get('ladder',[]): print('  ', {k:(round(v,4) if isinstance(v,float) else v) for k,v in r.items() if k in ('grid','jvps_per_s','forward_year_s','nlaunch','base_year_free_running_s')})
