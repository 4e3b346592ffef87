cd $GRAFT_REPO_ROOT
cp gusto.jl_amd/libgusto_hip.so /tmp/libgusto_hip.keep
cp gusto.jl_amd/variants/w2f_m3.so gusto.jl_amd/libgusto_hip.so
for w in 0 4; do echo "== fine profile m3 B=256 W2=$w"; GUSTO_DEV_W2=$w timeout 200 python tools/gpu_prof.py 256 3 2>&1 | grep -v " 0.0%"; done
cp /tmp/libgusto_hip.keep gusto.jl_amd/libgusto_hip.so
