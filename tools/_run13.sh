export HIP_FORCE_DEV_KERNARG=1
mkdir -p gpurun_out/r03
GFT_BENCH_BACKEND=gloo GFT_BENCH_EXCHANGE=torch timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus 2 --workload small --steps 3 --warmup 1 > gpurun_out/r03/bench_n2_gloo_smoke.json 2> gpurun_out/r03/bench_n2_gloo_smoke.err; echo rc=$?
tail -c 1800 gpurun_out/r03/bench_n2_gloo_smoke.json; tail -5 gpurun_out/r03/bench_n2_gloo_smoke.err
python -c "import __graft_entry__ as g; g.smoke()"
