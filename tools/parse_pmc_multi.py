import csv, glob, os, sys, json
from collections import defaultdict
d = sys.argv[1]
rows = defaultdict(lambda: defaultdict(list)); dur = defaultdict(list)
for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name']
        key = 'gvd_gemm' if 'gemm_nt_kernel' in name else 'rocblas_gemm' if name.startswith('Cijk') else 'flash_attn' if 'flash_attn' in name else None
        if key is None: continue
        rows[key][r['Counter_Name']].append(float(r['Counter_Value']))
        dur[key].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
out = {}
for k, c in rows.items():
    o = {n: sum(v) / len(v) for n, v in c.items()}
    o['avg_duration_us'] = sum(dur[k]) / len(dur[k]) / 1e3
    out[k] = o
print(json.dumps(out, indent=1))
json.dump(out, open(sys.argv[2], 'w'), indent=1)
