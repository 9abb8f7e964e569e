mkdir -p gpurun_out
bash benchmarks/multi_ab.sh laplace base base2 lapnf > gpurun_out/r6_ab5.json 2>gpurun_out/r6_ab5.err
UNERF_LIB=$PWD/benchmarks/build_probe/libunerf_lapnf.so timeout 900 python -m pytest tests/test_gpu_nerf_kernels.py tests/test_gpu_nerf_e2e.py tests/test_gpu_repeatability.py -m gpu -x -q -k "laplace" > gpurun_out/r6_lapnf_tests.txt 2>&1
tail -3 gpurun_out/r6_lapnf_tests.txt
timeout 900 python -m pytest tests/test_gpu_nerf_kernels.py tests/test_gpu_models.py tests/test_gpu_empty_inputs.py -m gpu -q -k "generate_rays or camera or empty" > gpurun_out/r6_cam_tests.txt 2>&1
tail -5 gpurun_out/r6_cam_tests.txt
python bench.py > gpurun_out/r6_01_bench.json 2> gpurun_out/r6_01_bench.err
tail -c 400 gpurun_out/r6_01_bench.json
cat gpurun_out/r6_ab5.json
