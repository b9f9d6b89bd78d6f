#!/bin/bash
# the multi-rank flow of bench.py on one GPU (test knob XDEM_BENCH_SHARE_GPU=1: 2 ranks on GPU 0 over gloo) after this round's changes
O=gpurun_out/r04ac; mkdir -p $O
export PYTHONUNBUFFERED=1 XDEM_BENCH_SHARE_GPU=1
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 > $O/bench2.log 2> $O/bench2.err; echo "rc $?"
tail -3 $O/bench2.err | cut -c1-300
python - $O <<'P'
import json, sys
for l in open(sys.argv[1] + "/bench2.log"):
    if l.startswith("{"):
        d = json.loads(l); s = d.get("secondary", {})
        print("value", d["value"], "ms", d["ms_per_step"], "n_gpus", d["n_gpus"], d["config"]["partition"])
        print("c4", s.get("c4_terrain_row_blocks", {}).get("value"), s.get("c4_terrain_row_blocks", {}).get("ms_per_step"))
        for k in ("variogram", "variogram_c5a"):
            v = s.get(k, {}); print("  ", k, v.get("pairs"), v.get("matheron_pass_Gpairs_s"), v.get("dowd_exact_median_Gpairs_s"), v.get("n_gpus"))
        n = s.get("nuthkaab", {}); print("   nk", n.get("ms_per_iteration"), n.get("routes"), s.get("error"))
P
