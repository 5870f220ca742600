#!/bin/bash
# Run ON THE GPU BOX: kernel times + SQ counters of the fused-vs-separate GroupNorm-apply experiment (tools/pmc_gn_fused.py).
set -e
out=gpurun_out/prof_gnfused
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 -L > $out/counters_available.txt 2>&1 || true
for mode in raw f32norm bf16norm; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t_$mode -- python3 tools/pmc_gn_fused.py $mode > $out/t_$mode.log 2>&1
  cp $(ls $out/t_$mode/*/*_kernel_stats.csv | head -1) $out/kernel_stats_$mode.csv
  for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_BANK_CONFLICT"; do
    tag=$(echo $pass | cut -d' ' -f1)
    timeout -k 10 200 rocprofv3 --pmc $pass --output-format csv -d $out/p_${mode}_$tag -- python3 tools/pmc_gn_fused.py $mode > $out/p_${mode}_$tag.log 2>&1 || echo "pass $tag failed for $mode"
    for c in $pass; do python3 tools/pmc_summary.py $out/p_${mode}_$tag $c $out/counters_$mode.json > /dev/null 2>&1 || true; done
  done
  rm -rf $out/t_$mode $out/p_${mode}_*
done
ls -la $out
