#!/bin/bash
# round-5, session x: the state after the one-pass step on partitioned plans -- whole GPU suite, bench.py (one GPU), bench.py's multi-rank flow on
# ONE GPU (2 ranks over gloo: the partitioned Nuth-Kaab leg with its routes and reductions), dispatch sequence of the hooked step
TAG=${1:-r05x}
O=gpurun_out/$TAG; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 420 python -m pytest tests -q -m gpu -p no:cacheprovider -x > $O/pytest_all.log 2>&1
echo "pytest rc=$?" >> $O/pytest_all.log
tail -5 $O/pytest_all.log | cut -c1-300
timeout 300 python bench.py > $O/bench.log 2> $O/bench.err; echo "bench rc=$?"; python - $O <<'PY'
import json, sys
for l in open(sys.argv[1] + "/bench.log"):
    if l.startswith("{"):
        d = json.loads(l)
        r = d["roofline"]; nk = d["secondary"]["nuthkaab"]; v = d["secondary"]["variogram"]; va = d["secondary"]["variogram_c5a"]
        print("headline", d["ms_per_step"], r["kernel_ms"], r["frac"], "| caller planes", r.get("kernel_ms_caller_planes"), r.get("frac_caller_planes"))
        print("nk", nk["ms_per_iteration"], nk["ms_per_iteration_whole_fit"], nk["routes"], nk["roofline"]["frac"], nk["roofline"].get("frac_at_survey_bytes"))
        print("vario", v["matheron_pass_Gpairs_s"], v["dowd_exact_median_Gpairs_s"], va["matheron_pass_Gpairs_s"], va["dowd_exact_median_Gpairs_s"])
        for row in d["secondary"].get("terrain_sets", []):
            print("  set", row)
PY
XDEM_BENCH_SHARE_GPU=1 timeout 360 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 > $O/bench2.log 2> $O/bench2.err; echo "bench2 rc $?"
tail -3 $O/bench2.err | cut -c1-300
python - $O <<'P'
import json, sys
for l in open(sys.argv[1] + "/bench2.log"):
    if l.startswith("{"):
        d = json.loads(l); s = d.get("secondary", {})
        print("value", d["value"], "ms", d["ms_per_step"], "n_gpus", d["n_gpus"], d["config"]["partition"])
        n = s.get("nuthkaab", {}); print("   nk", n.get("ms_per_iteration"), n.get("routes"), n.get("partition"), s.get("error"))
P
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
NK_HOOKED=1 timeout 150 rocprofv3 --kernel-trace --output-format csv -d $O/trace_hooked -o hooked -- python -u tools/nk_trace.py 20000 2 > $O/trace_hooked.log 2>&1; echo "trace rc=$?"
python tools/trace_sequence.py $O/trace_hooked 70 > $O/sequence_hooked.txt 2>&1; tail -75 $O/sequence_hooked.txt | cut -c1-150
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete
