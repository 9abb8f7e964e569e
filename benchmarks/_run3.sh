mkdir -p gpurun_out
./benchmarks/build_probe/issue_sweep_probe 2000 > gpurun_out/r6_issue_sweep2.jsonl 2> gpurun_out/r6_issue_sweep2.err
bash benchmarks/multi_ab.sh laplace base lapmom lapscalar > gpurun_out/r6_ab2.json 2>gpurun_out/r6_ab2.err
BENCH_ARGS="--precision f16x2" bash benchmarks/multi_ab.sh mcdropout base nopk > gpurun_out/r6_ab3.json 2>gpurun_out/r6_ab3.err
BENCH_ARGS="--precision f16x2" bash benchmarks/multi_ab.sh active base nopk > gpurun_out/r6_ab4.json 2>gpurun_out/r6_ab4.err
rm -f gpurun_out/parity_report.jsonl
timeout 1500 python -m pytest tests/test_gpu_concurrency.py tests/test_gpu_trained_like.py tests/test_gpu_fullsize_parity.py tests/test_gpu_splat.py tests/test_gpu_nerf_e2e.py -m gpu -q -k "concurr or threads or count_intersects or trained_like or full_size or reference_precision" > gpurun_out/r6_gputest_2.txt 2>&1
tail -8 gpurun_out/r6_gputest_2.txt
cp gpurun_out/parity_report.jsonl gpurun_out/r6_parity_report_2.jsonl
python bench.py --method ensemble --steps 2 --warmup 1 > gpurun_out/r6_ensemble.json 2> gpurun_out/r6_ensemble.err; tail -c 1500 gpurun_out/r6_ensemble.json
cat gpurun_out/r6_ab2.json gpurun_out/r6_ab3.json gpurun_out/r6_ab4.json
