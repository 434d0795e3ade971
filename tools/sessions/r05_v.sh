#!/bin/bash
# files -> captions section of bench.py alone, default reader threads (CPU-budget sized), both read modes
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
for how in mapped pread; do
  echo "== GVD_INGEST_READ=$how"
  GVD_INGEST_READ=$how timeout 600 python - <<'P' 2>&1 | grep -v "amdgpu.ids"
import json, torch, importlib.util
spec = importlib.util.spec_from_file_location('bench_mod', 'bench.py'); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
print(json.dumps(b.section_files_to_captions(torch.device('cuda', 0)), indent=1))
P
done | tee $O/r05v_files.txt
