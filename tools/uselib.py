"""tools/ and tests/tools/ only: `S2P_LIB=<path>` selects a second build of the library (normally the diagnostics build,
libs2p_hip_diag.so) for an A/B.  The product (`s2p_amd/_lib.py`) never reads the variable; a tool imports this module before
its first library call."""
import os
from s2p_amd import _lib
if os.environ.get("S2P_LIB"):
    _lib._SO = os.path.abspath(os.environ["S2P_LIB"])
