#!/bin/bash
# Rehearsal of bench.py's N > 1 path on ONE GPU (ranks share the device, host-staged collectives):
# iteration counts to OPTIMAL must equal the 1-GPU run's.  usage: tools_rehearse_ranks.sh <tag> <ranks> [bench args / env via ENV=...]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
tag=$1; shift
ranks=$1; shift
port=$((29600 + RANDOM % 300))
if [ "$ranks" = 1 ]; then
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 50 --warmup 5 "$@" > gpurun_out/reh_$tag.json 2> gpurun_out/reh_$tag.err
else
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $ranks --master-addr 127.0.0.1 --master-port $port bench.py --gpus $ranks --comm host --no-cpu-baseline --steps 50 --warmup 5 "$@" > gpurun_out/reh_$tag.json 2> gpurun_out/reh_$tag.err
fi
rc=$?
python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/reh_$tag.json").read().strip().splitlines()[-1])
    print("$tag", "rc=$rc", {k: d.get(k) for k in ("value", "init_s", "time_to_eps_s", "iters_to_eps", "state_at_eps")}, d["config"].get("comm", "")[:200])
except Exception as e:
    print("$tag", "rc=$rc", "ERR", e)
PY
exit 0
