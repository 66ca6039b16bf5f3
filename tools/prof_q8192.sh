cd /root/repo; export TMPDIR=/tmp; o=gpurun_out; tag=r6
rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_q8192_trace -o ${tag} -- python3 tools/run_q8192.py > $o/${tag}_q8192_trace.log 2>&1
cp "$(find $o/${tag}_q8192_trace -name '*kernel_stats.csv' | head -1)" $o/${tag}_q8192_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_q8192g_trace -o ${tag} -- python3 tools/run_q8192.py 8192 grp.lasso > $o/${tag}_q8192g_trace.log 2>&1
cp "$(find $o/${tag}_q8192g_trace -name '*kernel_stats.csv' | head -1)" $o/${tag}_q8192_grp_kernel_stats.csv
# the FULL product (a dense vector: no block skipped) -- the kernel bench.py's q8192_ms roofline prices: its average duration, then its bytes
rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_q8192f_trace -o ${tag} -- python3 tools/run_q8192.py 8192 full > $o/${tag}_q8192f_trace.log 2>&1
cp "$(find $o/${tag}_q8192f_trace -name '*kernel_stats.csv' | head -1)" $o/${tag}_q8192_full_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $o/${tag}_q8192_pmc_$c -o ${tag} -- python3 tools/run_q8192.py 8192 full > /dev/null 2> $o/${tag}_q8192_pmc_$c.err
done
python3 tools/pmc_summary.py sympk_gemv_kernel $o/${tag}_q8192_pmc_sympk_gemv.json "$(find $o/${tag}_q8192_pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1)" "$(find $o/${tag}_q8192_pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)"

rm -rf $o/${tag}_q8192_trace $o/${tag}_q8192g_trace $o/${tag}_q8192f_trace $o/${tag}_q8192_pmc_FETCH_SIZE $o/${tag}_q8192_pmc_WRITE_SIZE
