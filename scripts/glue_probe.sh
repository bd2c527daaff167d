#!/bin/bash
# GPU box: the drop-in's rate per gpu.fft1_batch_n for configs[2] and configs[1] (bench.py --glue-only), into gpurun_out/
mkdir -p gpurun_out
python bench.py --glue-only "$@" > gpurun_out/glue.json 2> gpurun_out/glue.err
python - <<'PY'
import json
g = json.load(open("gpurun_out/glue.json"))["glue"]
for c in g:
    if not c: print("no harness"); continue
    print(c["config"], c["fft1_size"], c["fft2_size"])
    for r in c["runs"]:
        if "error" in r: print("  batch_n", r["fft1_batch_n"], "ERROR", r["error"]); continue
        sc = r["stage_calls"]
        print("  batch_n %d: %8.1f Msamples/s  %7.2f us/block | " % (r["fft1_batch_n"], r["value"], r["us_per_block"]) +
              " ".join("%s %.0fus x%d (%.0f%%)" % ({"finish_rx_read_hook":"ingest","fft1_b":"fft1_b","fft1_c":"fft1_c","make_timf2":"timf2","first_noise_blanker":"blank","make_fft2":"fft2","fft2_mix1_fixed":"mix1","fft3_mix2_host":"fft3h","dispatcher_wait_room":"d_room","dispatcher_wait_input":"d_in"}.get(k,k), v["us_per_call"], v["calls"], 100 * v["busy_frac"]) for k, v in sc.items() if v["calls"]))
PY
