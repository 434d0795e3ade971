"""Per-kernel averages of every counter found in one or more rocprofv3 --pmc output directories -> JSON (+ derived fractions).
   python tools/pmc_summary.py <out.json> <label>=<dir>[:<kernel substring>] ...
Derived (MI355X_MICROARCH.md, rocprofv3 PMC slots): GRBM_GUI_ACTIVE is summed over the 8 XCDs -> shader cycles = /8;
SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD (1024 SIMDs); SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_* count quad-cycles
per wave; FETCH_SIZE / WRITE_SIZE are KB (FETCH_SIZE reports half of wide coalesced reads on gfx950)."""
import csv, glob, json, os, sys
from collections import defaultdict

out = {}
for spec in sys.argv[2:]:
    label, _, rest = spec.partition('=')
    d, _, sub = rest.partition(':')
    vals, dur = defaultdict(list), []
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if sub and sub not in r['Kernel_Name']:
                continue
            vals[r['Counter_Name']].append(float(r['Counter_Value']))
            dur.append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
    if not dur:
        out[label] = None
        continue
    o = {k: sum(v) / len(v) for k, v in vals.items()}
    o['avg_duration_us'] = sum(dur) / len(dur) / 1e3
    o['launches'] = len(dur) // max(len(vals), 1)
    if 'GRBM_GUI_ACTIVE' in o:
        cyc = o['GRBM_GUI_ACTIVE'] / 8.0
        o['shader_cycles'] = cyc
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in o:
            o['mfma_busy_frac'] = o['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024)
    if 'SQ_WAVE_CYCLES' in o:
        for k in ('SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_VMEM', 'SQ_ACTIVE_INST_LDS'):
            if k in o:
                o[k + '_per_wave_cycle'] = o[k] / o['SQ_WAVE_CYCLES']
    out[label] = o
json.dump(out, open(sys.argv[1], 'w'), indent=1)
print(json.dumps(out, indent=1))
