# HBM-side traffic of the decode projections at B=128: 64-row groups in one launch (key 13=1) vs one launch per group (13=0)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for mode in 1 0; do for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 280 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_groups/m${mode}_$c -- python3 $R/bench.py --no-cpu-baseline --batch 128 --prompt 128 --gen 3 --steps 1 --warmup 0 --tune 13=$mode > $R/gpurun_out/pmc_groups_m${mode}_$c.log 2>&1 || exit 1
done; done
cd $R
for mode in 1 0; do echo "== key 13 = $mode"; python3 scripts/pmc_kernel_bytes.py gpurun_out/pmc_groups/m${mode}_FETCH_SIZE gpurun_out/pmc_groups/m${mode}_WRITE_SIZE gpurun_out/pmc_groups_m$mode.json gemm_skinny; done
