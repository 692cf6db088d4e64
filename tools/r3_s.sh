cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py -q -x -k "from_sums or adam or conv_fwd_dgrad_wgrad" 2>&1 | tail -3
python -m pytest tests/test_gpu_e2e.py tests/test_gpu_fullsize_oracle.py -q -x 2>&1 | tail -3
for t in "fused_sums_finalize=0" "fused_sums_finalize=1" "fused_sums_finalize=0" "fused_sums_finalize=1"; do
python bench.py --steps 20 --warmup 5 --blocks 5 --no-cpu-baseline --no-extras --tune $t 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$t', d['ms_per_step'], d['timing']['block_ms_per_step'])"
done
