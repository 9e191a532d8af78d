cd $GRAFT_REPO_ROOT
for k in random descending blocky ascending; do timeout 300 python3 scripts/_asc.py $k 2>&1 | grep -v amdgpu.ids; done
