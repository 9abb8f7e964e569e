mkdir -p gpurun_out
rm -f gpurun_out/parity_report.jsonl
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r6_gputest_3.txt 2>&1
tail -6 gpurun_out/r6_gputest_3.txt
cp gpurun_out/parity_report.jsonl gpurun_out/r6_parity_report_3.jsonl
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6_smoke.txt 2>&1; tail -3 gpurun_out/r6_smoke.txt
bash benchmarks/collect_profiles.sh r6_02 > gpurun_out/r6_collect.log 2>&1
tail -5 gpurun_out/r6_collect.log
tail -c 600 gpurun_out/profiles_r6_02/r6_02_default_bench.json
