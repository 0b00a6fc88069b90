#!/bin/bash
# the whole GPU suite + the configs tool (GPU box, repo root)
python -m pytest tests -m gpu -x -q > gpurun_out/r06_suite_1.log 2>&1; tail -4 gpurun_out/r06_suite_1.log
python tools/bench_configs.py > gpurun_out/r06_configs_local.json 2>/dev/null
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r06_configs_local.json"))
print({k:round(v["cull_ms"]*1e3,1) for k,v in d["config2"]["gpu"].items()}, d["config2"]["gpu_output_bit_identical"])
print({k:round(v["two_pass_frame_ms"]*1e3,1) for k,v in d["config3"]["gpu"].items()})
print(d["frame_configs_3_and_4"]["ms"])
PY
