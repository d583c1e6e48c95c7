cd "$GRAFT_REPO_ROOT"
for b in 256 384 512; do for pers in 1 0; do echo "== blocks $b persistent $pers"; ICP_SIZE=1280x960 VH_ICP_BLOCKS=$b VH_ICP_PERSISTENT=$pers timeout 300 python tools/icp_only.py 100 2>&1 | grep "us per"; done; done
