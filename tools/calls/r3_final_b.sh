#!/bin/bash
# round-3 final evidence, part 2: TV profile (n = 1e8 with HBM counters; n = 1e5 timing), microbenchmarks,
# config-4 / config-5 benches, the reference's published problems, fp64 bench line
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
bash tools_profile_tv1d.sh > gpurun_out/final_tv.log 2>&1; echo "tv rc=$?"; head -14 gpurun_out/tv1d_profile.txt | cut -c1-200
for n in 100000 1000000 10000000; do
  it=30; timeout -k 10 200 python bench_tv1d.py --n $n --iters $it --cpu-n 1000 > gpurun_out/final_tv1d_n$n.json 2>/dev/null
  python -c "import json; a=json.load(open('gpurun_out/final_tv1d_n$n.json')); print('tv n=$n %.3f ms, %d levels' % (1e3*a['seconds'], a['levels']))"
done
python tools_microbench.py syrk gemm inverse > gpurun_out/final_micro.txt 2>&1; tail -12 gpurun_out/final_micro.txt
EPSILON_HIP_BENCH_RANDOM=1 python tools_microbench.py syrk > gpurun_out/final_micro_random.txt 2>&1; tail -2 gpurun_out/final_micro_random.txt
python tools_bench_nuclear_prox.py 10000 > gpurun_out/nuclear_prox.jsonl 2>/dev/null; echo "nuclear rc=$?"; cat gpurun_out/nuclear_prox.jsonl
timeout -k 10 400 python bench_rpca.py > gpurun_out/final_rpca_default.json 2> gpurun_out/final_rpca.err; echo "rpca rc=$?"
timeout -k 10 400 python bench_rpca.py --sweeps 5 > gpurun_out/final_rpca_5sweeps.json 2>> gpurun_out/final_rpca.err; echo "rpca5 rc=$?"
timeout -k 10 300 python bench_mnist.py > gpurun_out/final_mnist.json 2> gpurun_out/final_mnist.err; echo "mnist rc=$?"; cut -c1-300 gpurun_out/final_mnist.json
timeout -k 10 300 python bench_suite.py > gpurun_out/bench_suite.jsonl 2>/dev/null; echo "suite-bench rc=$?"; cut -c1-160 gpurun_out/bench_suite.jsonl
timeout -k 10 300 python bench.py --dtype f64 --no-cpu-baseline > gpurun_out/final_bench_f64.json 2>/dev/null; echo "f64 rc=$?"
python - <<'PY'
import json
for f in ("final_rpca_default","final_rpca_5sweeps","final_bench_f64"):
    try:
        d=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1])
        print(f, {k:d[k] for k in d if k in ("solve_s","sweeps","state","sweep_s","value","init_s","time_to_eps_s","iters_to_eps")})
    except Exception as e:
        print(f, "error", e)
PY
