cd $GRAFT_REPO_ROOT
python __graft_entry__.py > /dev/null 2>&1
Q="python tools/archive_r5/quick.py"
$Q base --steps 10 --warmup 2
PLONK_MSM_KSL=32 $Q ksl32 --steps 10 --warmup 2
PLONK_MSM_KSL=128 $Q ksl128 --steps 10 --warmup 2
$Q base_again --steps 10 --warmup 2
