"""Developer aid: ask every thread of process PID (which has loaded bt_on_signal.so) for its native stack."""
import ctypes
import os
import sys
import time

pid = int(sys.argv[1])
libc = ctypes.CDLL(None, use_errno=True)
for tid in sorted(int(t) for t in os.listdir("/proc/%d/task" % pid)):
    libc.syscall(234, pid, tid, 12)          # tgkill(pid, tid, SIGUSR2)
    time.sleep(0.05)
