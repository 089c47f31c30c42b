D=s2p_amd/csrc/libs2p_hip_diag.so
echo "== new forms"; S2P_LIB=$D python tools/bench_dhead.py 2>&1 | grep "^N"
echo "== old forms (switches 11, 12)"; S2P_LIB=$D S2P_DIAG_SET="11=1,12=1" python tools/bench_dhead.py 2>&1 | grep "^N"
echo "== conv_dma phase order / band XCD"
S2P_LIB=$D python tools/ab_step.py lib:10 3 0 1 2>&1 | tail -2
S2P_LIB=$D python tools/ab_step.py lib:15 3 0 1 2>&1 | tail -2
