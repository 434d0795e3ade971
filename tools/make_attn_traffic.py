"""profiles/attn_traffic.json from the two rocprofv3 --pmc passes of tools/profile_attn.py (FETCH_SIZE, WRITE_SIZE), with
the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE counts half of wide coalesced reads).
    python tools/make_attn_traffic.py <fetch.json> <write.json> <out.json> [B] [Ft]"""
import json, sys
f = json.load(open(sys.argv[1]))['counters']['FETCH_SIZE']['mean']
w = json.load(open(sys.argv[2]))['counters']['WRITE_SIZE']['mean']
B = int(sys.argv[4]) if len(sys.argv) > 4 else 256
Ft = int(sys.argv[5]) if len(sys.argv) > 5 else 10
alg = B * (1000 + Ft) * (512 + 1024) * 4
out = {'kernel': 'attn_partial_kernel', 'batch': B, 't_attn': Ft, 'regions': 1000,
       'FETCH_SIZE_KB_mean': f, 'WRITE_SIZE_KB_mean': w,
       'correction': 'gfx950 rocprofv3 FETCH_SIZE reports 1/2 of wide (16 B/lane) coalesced reads (MI355X_MICROARCH.md HBM '
                     'section): read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE * 1024 taken as is (uncalibrated)',
       'hbm_bytes_per_launch': int(2 * f * 1024 + w * 1024), 'algorithmic_bytes_per_launch': alg,
       'traffic_over_algorithmic': round((2 * f * 1024 + w * 1024) / alg, 4),
       'source': 'separate rocprofv3 --pmc passes of tools/profile_attn.py %d %d 3' % (B, Ft),
       'note': 'the kernel does not fetch rows the attention mask removes (their softmax weight is exactly 0): with the 20 % '
               'masked proposals of the SURVEY 8d workload the HBM traffic is ~0.8 of the algorithmic bytes, which count '
               'every row'}
json.dump(out, open(sys.argv[3], 'w'), indent=1)
print(json.dumps(out, indent=1))
