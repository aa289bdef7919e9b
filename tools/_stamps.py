"""Shared by the time-stamp tools: run a few propagates on a plan made with JTP_DEBUG=2 under a library built with
-DJT_STAMPS (python junction-tree_amd/build.py --out /tmp/stamps.so -DJT_STAMPS; JTPROP_LIB=/tmp/stamps.so) and read the
workgroups' time stamps back.  Slots (jtp_kernels.hip.h, JT_STAMP): 0 entry, 1 first element loads issued, 2 end of the
first staging attempt, 3 staged, 4 constants read, 5-8 after loop steps 0-3, 9 loop done, 10 epilogues done, 11 flush
stores issued, 12 flush stores retired, 13 staging attempts."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "junction-tree_amd"))
from junctiontree_amd import _capi      # noqa: E402

NSTAMP = 16


def read(plan):
    """(describe(), stamps in microseconds [blocks, 16] - slot 13 is the attempt count, untouched)."""
    if b"JT_STAMPS" not in plan._lib.jtp_version():
        raise SystemExit("time stamps need a library built with -DJT_STAMPS (and JTP_DEBUG=2): "
                         "python junction-tree_amd/build.py --out /tmp/stamps.so -DJT_STAMPS; JTPROP_LIB=/tmp/stamps.so JTP_DEBUG=2 ...")
    d = plan.describe()
    base, nb = d["dbg_base"], d["n_blocks"]
    if base < 0:
        raise SystemExit("the plan has no time-stamp region: set JTP_DEBUG=2")
    buf = np.empty(nb * NSTAMP)
    _capi.check(plan._lib.jtp_debug_read_msg(plan._handle, 0, base, nb * NSTAMP, buf.ctypes.data_as(C.POINTER(C.c_double))))
    st = buf.reshape(nb, NSTAMP).copy()
    att = st[:, 13].copy()
    st *= 0.01                      # 100 MHz ticks -> microseconds
    st[:, 13] = att
    return d, st


def coarse(st):
    """The six stage boundaries of rounds 1-2: entry, loads issued, staged, constants, loop done, flushed."""
    return st[:, [0, 1, 3, 4, 9, 12]]
