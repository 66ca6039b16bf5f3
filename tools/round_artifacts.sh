#!/bin/bash
# Round artefacts on the GPU box: kernel-trace stats, PMC traffic passes (separate runs), config times, bench line.
# usage: bash tools/round_artifacts.sh r2 [quick]     (writes under gpurun_out/<tag>_*; copy what is to be judged into profiles/)
set -u
tag=${1:-r4}
quick=${2:-}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out
mkdir -p $o
B="--no-cpu-baseline --no-c5 --no-host --no-two-callers --no-rccl-check --no-live-pmc"   # the profiled command: the timed solves alone
# ---- config 1 (the bench command): per-kernel stats, then FETCH_SIZE / WRITE_SIZE in their own passes
rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_trace -o ${tag} -- python3 bench.py --steps 100 --warmup 10 $B > $o/${tag}_trace_bench.json 2> $o/${tag}_trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $o/${tag}_pmc_fetch -o ${tag} -- python3 bench.py --steps 5 --warmup 2 $B > /dev/null 2> $o/${tag}_pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $o/${tag}_pmc_write -o ${tag} -- python3 bench.py --steps 5 --warmup 2 $B > /dev/null 2> $o/${tag}_pmc_write.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $o/${tag}_pmc_mfma -o ${tag} -- python3 bench.py --steps 5 --warmup 2 $B > /dev/null 2> $o/${tag}_pmc_mfma.err
python3 tools/pmc_summary.py gram_ring_kernel $o/${tag}_pmc_gram_mfma.json "$(find $o/${tag}_pmc_mfma -name '*counter_collection.csv' | head -1)"
f=$(find $o/${tag}_pmc_fetch -name "*counter_collection.csv" | head -1)
w=$(find $o/${tag}_pmc_write -name "*counter_collection.csv" | head -1)
python3 tools/pmc_summary.py gram_ring_kernel $o/${tag}_pmc_gram.json "$f" "$w"
s=$(find $o/${tag}_trace -name "*kernel_stats.csv" | head -1)
cp "$s" $o/${tag}_kernel_stats.csv
if [ -z "$quick" ]; then
  # ---- configs 3 and 5 (the shared-slab Gram kernel) and config 4 (the fused GEMV iteration): stats + traffic
  for cfg in c3 c5; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_${cfg}_trace -o ${tag} -- python3 tools/run_c3.py $cfg 3 > $o/${tag}_${cfg}_trace.log 2>&1
    cp "$(find $o/${tag}_${cfg}_trace -name '*kernel_stats.csv' | head -1)" $o/${tag}_${cfg}_kernel_stats.csv
    for c in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --pmc $c --kernel-trace --output-format csv -d $o/${tag}_${cfg}_pmc_$c -o ${tag} -- python3 tools/run_c3.py $cfg 2 > /dev/null 2> $o/${tag}_${cfg}_pmc_$c.err
    done
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $o/${tag}_${cfg}_pmc_mfma -o ${tag} -- python3 tools/run_c3.py $cfg 2 > /dev/null 2> $o/${tag}_${cfg}_pmc_mfma.err
    k=gram_wd                                     # (p = 256 and p = 512 run on gram_wd.hip since round 5: one unit / two units + off-diagonal blocks)
    python3 tools/pmc_summary.py ${k}_kernel $o/${tag}_${cfg}_pmc_${k}_mfma.json "$(find $o/${tag}_${cfg}_pmc_mfma -name '*counter_collection.csv' | head -1)"
    python3 tools/pmc_summary.py ${k}_kernel $o/${tag}_${cfg}_pmc_${k}.json "$(find $o/${tag}_${cfg}_pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1)" "$(find $o/${tag}_${cfg}_pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)"
  done
  rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_c4_trace -o ${tag} -- python3 tools/run_c4.py > $o/${tag}_c4_trace.log 2>&1
  cp "$(find $o/${tag}_c4_trace -name '*kernel_stats.csv' | head -1)" $o/${tag}_c4_kernel_stats.csv
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $o/${tag}_c4_pmc_$c -o ${tag} -- python3 tools/run_c4.py > /dev/null 2> $o/${tag}_c4_pmc_$c.err
  done
  # (round 4: config 4 is ONE persistent launch, path_symcoop_kernel<3>; its FETCH_SIZE is the lower triangle once plus the polls)
  python3 tools/pmc_summary.py path_symcoop_kernel $o/${tag}_c4_pmc_symcoop.json "$(find $o/${tag}_c4_pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1)" "$(find $o/${tag}_c4_pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)"
  python3 tools/c4_time.py "" nosymcoop nosym 2>&1 | grep -v amdgpu.ids > $o/${tag}_c4_time.txt
  python3 tools/symcoop_check.py 1088 2048 3000 4096 2>&1 | grep -v amdgpu.ids > $o/${tag}_symcoop_sizes.txt
  SYMCOOP_CHECK_GROUPS=8 python3 tools/symcoop_check.py 1088 2048 3000 4096 2>&1 | grep -v amdgpu.ids > $o/${tag}_symcoop_sizes_group_penalties.txt
  if [ -f oem_amd/liboemgpu_diag.so ]; then
    OEMGPU_LIB=oem_amd/liboemgpu_diag.so python3 tools/symcoop_diag.py 4096 100 65536 2>&1 | grep -v "amdgpu.ids\|warn" > $o/${tag}_symcoop_stamped.txt
    OEMGPU_LIB=oem_amd/liboemgpu_diag.so python3 tools/symcoop_diag.py 4096 100 65536 0.3 2>&1 | grep -v "amdgpu.ids\|warn" >> $o/${tag}_symcoop_stamped.txt
    OEMGPU_LIB=oem_amd/liboemgpu_diag.so python3 tools/symcoop_diag.py 2048 2>&1 | grep -v "amdgpu.ids\|warn" >> $o/${tag}_symcoop_stamped.txt
    OEM_NO_ROWCOOP=1 OEMGPU_LIB=oem_amd/liboemgpu_diag.so python3 tools/symcoop_diag.py 2048 2>&1 | grep -v "amdgpu.ids\|warn" >> $o/${tag}_symcoop_stamped.txt
  fi
  python3 tools/pgen_rowcoop_ab.py 2>&1 | grep -v amdgpu.ids > $o/${tag}_pgen_rowcoop_ab.txt
  python3 tools/penalty_split_time.py 2>&1 | grep -v amdgpu.ids > $o/${tag}_penalty_split_times.txt
  python3 tools/wide_group_time.py 2>&1 | grep -v amdgpu.ids > $o/${tag}_wide_group_times.txt
  # ---- p >= n (n = 500, p = 20,000): the wide engine's column kernel
  rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_wide_trace -o ${tag} -- python3 tools/run_wide.py > $o/${tag}_wide_trace.log 2>&1
  cp "$(find $o/${tag}_wide_trace -name '*kernel_stats.csv' | head -1)" $o/${tag}_wide_n500_p20000_kernel_stats.csv
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $o/${tag}_wide_pmc_$c -o ${tag} -- python3 tools/run_wide.py > /dev/null 2> $o/${tag}_wide_pmc_$c.err
  done
  python3 tools/pmc_summary.py "wide_cols_kernel<8, 1>" $o/${tag}_wide_pmc_cols.json "$(find $o/${tag}_wide_pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1)" "$(find $o/${tag}_wide_pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)"
  python3 tools/wide_time.py > $o/${tag}_wide_times.txt 2>&1
  python3 tools/config_times.py > $o/${tag}_config_times.json 2> $o/${tag}_config_times.err
  # ---- the cooperating engines: p >= n in registers (element-wise and general form), 208 < p <= 1024, and what an exchange costs
  python3 tools/wcoop_time.py 2>&1 | grep -v amdgpu.ids > $o/${tag}_wcoop_times.txt
  python3 tools/wcoop_general_time.py 2>&1 | grep -v amdgpu.ids >> $o/${tag}_wcoop_times.txt
  if [ -f oem_amd/liboemgpu_diag.so ]; then
    OEMGPU_LIB=oem_amd/liboemgpu_diag.so python3 tools/wcoop_diag.py 500 2000 2>&1 | grep -v "amdgpu.ids\|warn" >> $o/${tag}_wcoop_times.txt
    OEMGPU_LIB=oem_amd/liboemgpu_diag.so python3 tools/wcoop_diag.py 500 2000 30 gen 2>&1 | grep -v "amdgpu.ids\|warn" >> $o/${tag}_wcoop_times.txt
  fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_wcoop_trace -o ${tag} -- python3 tools/run_wcoop.py > $o/${tag}_wcoop_trace.log 2>&1
  cp "$(find $o/${tag}_wcoop_trace -name '*kernel_stats.csv' | head -1)" $o/${tag}_wcoop_n500_p2000_kernel_stats.csv
  rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_wres_trace -o ${tag} -- python3 tools/run_wres.py > $o/${tag}_wres_trace.log 2>&1
  cp "$(find $o/${tag}_wres_trace -name '*kernel_stats.csv' | head -1)" $o/${tag}_wres_n500_p20000_kernel_stats.csv
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $o/${tag}_wres_pmc_$c -o ${tag} -- python3 tools/run_wres.py > /dev/null 2> $o/${tag}_wres_pmc_$c.err
  done
  python3 tools/pmc_summary.py "path_wres_kernel" $o/${tag}_wres_pmc.json "$(find $o/${tag}_wres_pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1)" "$(find $o/${tag}_wres_pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)"
  python3 tools/wres_time.py 2>&1 | grep -v amdgpu.ids > $o/${tag}_wres_times.txt
  if [ -f oem_amd/liboemgpu_diag.so ]; then
    OEMGPU_LIB=oem_amd/liboemgpu_diag.so python3 tools/wcoop_diag.py 500 20000 30 2>&1 | grep -v "amdgpu.ids\|warn" >> $o/${tag}_wres_times.txt
  fi
  OEM_NO_WRES=1 python3 tools/wstream_time.py 2>&1 | grep -v amdgpu.ids > $o/${tag}_wstream_times.txt
  python3 tools/coop_time.py 2>&1 | grep -v amdgpu.ids > $o/${tag}_coop_times.txt
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/hop_probe tools/hop_probe.hip 2> /dev/null && /tmp/hop_probe > $o/${tag}_exchange_probes.txt 2>&1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/xchg_probe tools/xchg_probe.hip 2> /dev/null && /tmp/xchg_probe >> $o/${tag}_exchange_probes.txt 2>&1
  # ---- round 5: the probes behind DESIGN.md section 3.3c's floor analysis; the bands between the configurations' sizes
  for pr in areg_fma_probe valu_probe; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/$pr tools/$pr.hip 2> /dev/null && /tmp/$pr > $o/${tag}_$pr.txt 2>&1
  done
  bash tools/gram_band.sh > $o/${tag}_gram_band_now.txt 2>&1
  bash tools/attic/gram_band_small.sh > $o/${tag}_gram_band_small.txt 2>&1
  for p in 128 256 300 512 1024; do python3 tools/gram_by_n.py $p 30000 60000 100000 200000 500000 2000000 2>&1 | grep -v amdgpu.ids; done > $o/${tag}_gram_by_n_now.txt
  python3 tools/large_q_time.py 2>&1 | grep -v amdgpu.ids > $o/${tag}_large_q_now.txt
  # ---- round 6: the HBM-bound GEMV loop (q = 8,192 on the packed lower triangle: kernel stats + FETCH_SIZE / WRITE_SIZE), what the register
  # engines do not take at 1024 < q <= 4096, config 4's stamps with the arrival spread of its gathers
  bash tools/prof_q8192.sh > /dev/null 2>&1
  python3 tools/scattered_groups_time.py 2>&1 | grep -v amdgpu.ids > $o/${tag}_scattered_groups_now.txt
  python3 tools/spk_check.py 4097 6145 8192 2>&1 | grep -v amdgpu.ids > $o/${tag}_spk_check.txt
  python3 tools/scale_factor_time.py 2>&1 | grep -v amdgpu.ids > $o/${tag}_scale_factor_now.txt
  python3 tools/rowcoop_accel_time.py 2>&1 | grep -v amdgpu.ids > $o/${tag}_rowcoop_accelerate_now.txt
fi
# the raw traces stay on the box: gpurun copies back at most 64 MiB
for d in $o/${tag}_trace $o/${tag}_pmc_fetch $o/${tag}_pmc_write $o/${tag}_pmc_mfma $o/${tag}_*_pmc_mfma $o/${tag}_*_trace $o/${tag}_*_pmc_FETCH_SIZE $o/${tag}_*_pmc_WRITE_SIZE; do [ -d "$d" ] && rm -rf "$d"; done
python3 bench.py > $o/${tag}_bench.json 2> $o/${tag}_bench.err
tail -c 300 $o/${tag}_bench.err
head -c 1500 $o/${tag}_kernel_stats.csv
[ -z "$quick" ] && head -c 4000 $o/${tag}_config_times.json
exit 0
