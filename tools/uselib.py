"""tools/ and tests/tools/ only: `S2P_LIB=<path>` selects a second build of the library (normally the diagnostics build,
libs2p_hip_diag.so) for an A/B.  The product (`s2p_amd/_lib.py`) never reads the variable; a tool imports this module before
its first library call."""
import os
from s2p_amd import _lib
if os.environ.get("S2P_LIB"):
    _lib._SO = os.path.abspath(os.environ["S2P_LIB"])
# S2P_DIAG_SET="7=1,4=0": run-time switches of the diagnostics build (s2p_diag_set), for tools that do not flip them themselves
if os.environ.get("S2P_DIAG_SET"):
    import ctypes
    _L = ctypes.CDLL(_lib._SO)
    for _kv in os.environ["S2P_DIAG_SET"].split(","):
        _k, _v = _kv.split("=")
        assert _L.s2p_diag_set(int(_k), int(_v)) == 0
