#!/bin/bash
# recursion-size throughput lines only (tooling)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_prove_openings.py tests/test_parity.py -m gpu -q -k "many" 2>&1 | tail -3
timeout 600 python - <<'PY'
import json, torch, bench
from plonky2_amd import Engine
eng = Engine(0)
out = {}
bench.recursion_lines(eng, torch, out)
for k, v in out.items():
    print(k, json.dumps({a: b for a, b in v.items() if a != "workload"}))
PY
