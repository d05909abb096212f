cd $GRAFT_REPO_ROOT
python __graft_entry__.py > /dev/null 2>&1
Q="python tools/archive_r5/quick.py"
for i in 1 2 3; do
  $Q base_$i --steps 20 --warmup 2
  PLONK_MSM_KSL=128 $Q ksl128_$i --steps 20 --warmup 2
done
PLONK_MSM_KSL=128 $Q ksl128_bench_like --steps 10 --warmup 2 --profile bench-like
$Q base_bench_like --steps 10 --warmup 2 --profile bench-like
