"""profiles/attn_traffic.json from the rocprofv3 --pmc passes of tools/profile_attn.py (FETCH_SIZE and WRITE_SIZE in their own
passes, summarised by tools/pmc_summary.py), with the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE counts half of
wide coalesced reads).  The file records the git head and the source hash of the library the counters were taken on:
bench.py quotes `roofline.traffic` from it only when that hash equals the loaded library's.
    python tools/make_attn_traffic.py <pmc_summary.json> <out.json> <git head>
pmc_summary.json labels: greedy_fetch / greedy_write (profile_attn.py 256 10 3) and beam_fetch / beam_write
(profile_attn.py 64 10 3 2000 5)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pmc = json.load(open(sys.argv[1]))
with open(os.path.join(ROOT, 'grounded-video-description_amd', 'libgvd_hip.so.srchash')) as f:
    srchash = f.read().strip()
out = {'head': sys.argv[3] if len(sys.argv) > 3 else None, 'lib_srchash': srchash,
       'correction': 'gfx950 rocprofv3 FETCH_SIZE reports 1/2 of wide (16 B/lane) coalesced reads (MI355X_MICROARCH.md HBM '
                     'section): read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE * 1024 taken as is (uncalibrated)',
       'note': 'the kernels do not fetch rows the attention mask removes (their softmax weight is exactly 0): with the 20 % '
               'masked proposals of the SURVEY 8d workload the HBM traffic is below the algorithmic bytes, which count every row'}
for kind, kernel, B, Ft, R, cmd in (('greedy', 'attn_partial_kernel', 256, 10, 1000, 'tools/profile_attn.py 256 10 3'),
                                    ('beam', 'attn_partial_group_kernel<5>', 64, 10, 2000, 'tools/profile_attn.py 64 10 3 2000 5')):
    f, w = pmc.get(kind + '_fetch'), pmc.get(kind + '_write')
    if not f or not w or 'FETCH_SIZE' not in f or 'WRITE_SIZE' not in w:
        continue
    fk, wk = f['FETCH_SIZE'], w['WRITE_SIZE']
    alg = B * (R + Ft) * (512 + 1024) * 4
    out[kind] = {'kernel': kernel, 'batch': B, 't_attn': Ft, 'regions': R, 'FETCH_SIZE_KB_mean': fk, 'WRITE_SIZE_KB_mean': wk,
                 'hbm_bytes_per_launch': int(2 * fk * 1024 + wk * 1024), 'algorithmic_bytes_per_launch': alg,
                 'traffic_over_algorithmic': round((2 * fk * 1024 + wk * 1024) / alg, 4),
                 'avg_duration_us_under_pmc': f.get('avg_duration_us'),
                 'source': 'separate rocprofv3 --pmc passes of ' + cmd}
# the grounding stream of the training step (tools/stream_mm_bench.py 64 5 grounder_fwd under the same PMC passes): labels
# gfwd_fetch / gfwd_write; bench.py's configs2_train_b64.grounding_stream quotes it for the shape it times
f, w = pmc.get('gfwd_fetch'), pmc.get('gfwd_write')
if f and w and 'FETCH_SIZE' in f and 'WRITE_SIZE' in w:
    B, M, R, K = 64, 20, 1000, 2048
    alg = 4 * (B * R * K + B * M * K + 2 * B * M * R + B * M) + B * M * R
    out['grounder_fwd'] = {'kernel': 'grounder_fwd_kernel<8, 4>', 'shape': [B, M, R, K], 'FETCH_SIZE_KB_mean': f['FETCH_SIZE'],
                           'WRITE_SIZE_KB_mean': w['WRITE_SIZE'],
                           'hbm_bytes_per_launch': int(2 * f['FETCH_SIZE'] * 1024 + w['WRITE_SIZE'] * 1024),
                           'algorithmic_bytes_per_launch': alg,
                           'traffic_over_algorithmic': round((2 * f['FETCH_SIZE'] * 1024 + w['WRITE_SIZE'] * 1024) / alg, 4),
                           'avg_duration_us_under_pmc': f.get('avg_duration_us'),
                           'source': 'separate rocprofv3 --pmc passes of tools/stream_mm_bench.py 64 5 grounder_fwd'}
# the two backward streams of the grounder at the same shape (stream_mm_bench.py filters 'transposed S' / 'N=2048'): labels
# gdw_fetch / gdw_write (rows_contract_kernel: d words) and gdr_fetch / gdr_write (rank_update_kernel: d regions)
for key, lab, kernel, alg, flt in (
        ('grounder_d_words', 'gdw', 'rows_contract_kernel', 4 * (64 * 1000 * 2048 + 64 * 1000 * 32 + 64 * 20 * 2048), 'transposed'),
        ('grounder_d_regions', 'gdr', 'rank_update_kernel<3>', 4 * (64 * 1000 * 2048 + 64 * 20 * 1000 + 64 * 20 * 2048), 'N=2048')):
    f, w = pmc.get(lab + '_fetch'), pmc.get(lab + '_write')
    if f and w and 'FETCH_SIZE' in f and 'WRITE_SIZE' in w:
        hbm = int(2 * f['FETCH_SIZE'] * 1024 + w['WRITE_SIZE'] * 1024)
        out[key] = {'kernel': kernel, 'shape': [64, 20, 1000, 2048], 'FETCH_SIZE_KB_mean': f['FETCH_SIZE'],
                    'WRITE_SIZE_KB_mean': w['WRITE_SIZE'], 'hbm_bytes_per_launch': hbm, 'algorithmic_bytes_per_launch': alg,
                    'traffic_over_algorithmic': round(hbm / alg, 4), 'avg_duration_us_under_pmc': f.get('avg_duration_us'),
                    'source': 'separate rocprofv3 --pmc passes of tools/stream_mm_bench.py 64 5 ' + flt}
json.dump(out, open(sys.argv[2], 'w'), indent=1)
print(json.dumps(out, indent=1))
