cd $GRAFT_REPO_ROOT
for cap in 4 6 8 12; do echo "cap $cap"; GUSTO_SHOOT_CAP=$cap python tools/shoot_ab.py 65536 | head -1; done
echo G16; GUSTO_SHOOT_G16=1 python tools/shoot_ab.py 65536 | head -1
echo G4; GUSTO_SHOOT_G4=1 python tools/shoot_ab.py 65536 | head -1
echo "G16 cap 5"; GUSTO_SHOOT_G16=1 GUSTO_SHOOT_CAP=5 python tools/shoot_ab.py 65536 | head -1
