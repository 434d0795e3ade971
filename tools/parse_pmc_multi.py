"""Average PMC counters per kernel group from a rocprofv3 --pmc output directory.
   python tools/parse_pmc_multi.py <dir> <out.json> [name=substr ...]   (default groups below; a group may add @<grid> to
   select launches by Grid_Size_X)"""
import csv, glob, os, sys, json
from collections import defaultdict
d = sys.argv[1]
groups = [a.split('=', 1) for a in sys.argv[3:]] or [
    ['gemm_pipe_fc7', 'gemm_pipe_kernel@8192000'], ['gemm_pipe_qkv', 'gemm_pipe_kernel@12800000'],
    ['gemm64_logit', 'gemm_nt_kernel<64, 64, 2, 2, false@80896'], ['gemm64_attn_hid', 'gemm_nt_kernel<64, 64, 2, 2, false@16384'],
    ['rocblas_gemm', 'Cijk'], ['flash_attn_pad', 'flash_attn_pad'], ['flash_attn16', 'flash_attn16'],
    ['attn_partial', 'attn_partial_kernel'], ['attn_bwd_step', 'attn_bwd_step_kernel'], ['attn_bwd_pfeats', 'attn_bwd_pfeats_kernel']]
rows = defaultdict(lambda: defaultdict(list)); dur = defaultdict(list)
for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name']
        for key, pat in groups:
            sub, _, grid = pat.partition('@')
            if sub in name and (not grid or r.get('Grid_Size_X', r.get('Grid_Size', '')) == grid):
                rows[key][r['Counter_Name']].append(float(r['Counter_Value']))
                dur[key].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
                break
out = {}
for k, c in rows.items():
    o = {n: sum(v) / len(v) for n, v in c.items()}
    o['avg_duration_us'] = sum(dur[k]) / len(dur[k]) / 1e3
    o['launches'] = len(dur[k]) // max(len(c), 1)
    out[k] = o
print(json.dumps(out, indent=1))
json.dump(out, open(sys.argv[2], 'w'), indent=1)
