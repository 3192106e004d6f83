cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { # tag ranks script args...
  tag=$1; ranks=$2; script=$3; shift 3
  port=$((29600 + RANDOM % 300))
  if [ "$ranks" = 1 ]; then
    timeout -k 10 400 python $script "$@" > gpurun_out/reh_$tag.json 2> gpurun_out/reh_$tag.err
  else
    timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $ranks --master-addr 127.0.0.1 --master-port $port $script --gpus $ranks --comm host "$@" > gpurun_out/reh_$tag.json 2> gpurun_out/reh_$tag.err
  fi
  echo "$tag rc=$? $(tail -1 gpurun_out/reh_$tag.json | cut -c1-700)"
}
run mnist1 1 bench_mnist.py
run mnist2 2 bench_mnist.py
run mnist4 4 bench_mnist.py
run cons1 1 bench_consensus.py --steps 30 --warmup 5
run cons2 2 bench_consensus.py --steps 30 --warmup 5
run rpca1 1 bench_rpca.py --size 4096
run rpca2 2 bench_rpca.py --size 4096
