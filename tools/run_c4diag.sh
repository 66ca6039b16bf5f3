set -u
cd /root/repo
o=gpurun_out
./tools/areg_fma_probe > $o/r5_c4_areg_fma_probe.txt 2>&1
OEMGPU_LIB=oem_amd/liboemgpu_diag.so python3 tools/symcoop_diag.py 4096 100 65536 2>&1 | grep -v "amdgpu.ids\|warn" > $o/r5_symcoop_stamped.txt
OEMGPU_LIB=oem_amd/liboemgpu_diag.so python3 tools/symcoop_diag.py 4096 100 65536 0.3 2>&1 | grep -v "amdgpu.ids\|warn" >> $o/r5_symcoop_stamped.txt
python3 tools/c4_time.py "" 2>&1 | grep -v amdgpu.ids > $o/r5_c4_time.txt
python -m pytest tests/test_gpu_distributed.py tests/test_gpu_host.py -q -m gpu -x -k "eight_ranks or not_tried_again or interrupt" 2>&1 | tail -5 > $o/r5_tests_a.txt
python bench.py > $o/bench_b.json 2> $o/bench_b.err
cat $o/r5_c4_areg_fma_probe.txt $o/r5_symcoop_stamped.txt $o/r5_c4_time.txt $o/r5_tests_a.txt
