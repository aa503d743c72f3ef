import sys, ctypes as C
import numpy as np
sys.path.insert(0, "/root/repo")
from dspeed_amd import _lib
L = _lib.lib()
for n in (1 << 20, (1 << 24) + 12345, 1 << 26):
    a = np.zeros(n, dtype=np.uint8)
    p = a.ctypes.data
    r1 = L.dsp_host_register(p, a.nbytes)
    r2 = L.dsp_host_unregister(p)
    r3 = L.dsp_host_register(p, a.nbytes)
    r4 = L.dsp_host_register(p, a.nbytes)   # second time: must be refused
    r5 = L.dsp_host_unregister(p)
    r6 = L.dsp_host_unregister(p)           # second time: must fail
    print(n, hex(p & 0xfff), "register", r1, "unregister", r2, "re-register", r3, "double register", r4, "unregister", r5, "double unregister", r6)
    b = a[4096:]  # sub-range registration
    r7 = L.dsp_host_register(b.ctypes.data, b.nbytes); r8 = L.dsp_host_unregister(b.ctypes.data)
    print("   sub-range", r7, r8)
