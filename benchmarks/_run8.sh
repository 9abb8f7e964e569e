mkdir -p gpurun_out
rm -f gpurun_out/parity_report.jsonl
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r6_gputest_5.txt 2>&1
tail -4 gpurun_out/r6_gputest_5.txt
cp gpurun_out/parity_report.jsonl gpurun_out/r6_parity_report_5.jsonl
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6_smoke.txt 2>&1; tail -2 gpurun_out/r6_smoke.txt
bash benchmarks/collect_profiles.sh r6_03 > gpurun_out/r6_collect.log 2>&1
tail -3 gpurun_out/r6_collect.log
