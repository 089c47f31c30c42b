export S2P_LIB=$GRAFT_REPO_ROOT/s2p_amd/csrc/libs2p_hip_diag.so
for d in ${DIAGS:-0 1 2 3 4 5 6}; do S2P_DIAG=$d timeout -k 10 120 python tools/diag_plane.py 2>/dev/null | grep "DIAG\|loop"; done
S2P_NO_PLANE=1 timeout -k 10 120 python tools/diag_plane.py 2>/dev/null | grep DIAG
