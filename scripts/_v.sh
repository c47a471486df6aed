cd $GRAFT_REPO_ROOT
echo "== blas limited"; NOPROF=1 python scripts/profile_via_api.py 2>&1 | grep "call_ms\|per-iteration"
echo "== blas not limited"; JWAS_HOST_BLAS_THREADS=0 NOPROF=1 python scripts/profile_via_api.py 2>&1 | grep "call_ms"
cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null
