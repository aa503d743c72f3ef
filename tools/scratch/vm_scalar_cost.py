import sys, json, numpy as np
sys.path.insert(0, '/root/repo')
from dspeed_amd import _lib
from dspeed_amd.chain import Chain, Program, Scalar
from dspeed_amd.device import DeviceArray, Event, Stream, sync
rows, n = 65536, 8192
st = Stream()
wf = DeviceArray.zeros((rows, n), np.float32)
res = []
for n_sc in (8, 40):
    for kind in ("reg", "const"):
        p = Program(); p.slots = [n]; p.n_sregs = 6
        io = p.add_io("wf", _lib.IO_WF_IN, np.float32, n, 0, n)
        col = p.add_io("col", _lib.IO_SCALAR_IN, np.float32)
        p.add_op(_lib.OP_LOAD, dst=0, io=io)
        p.add_op(_lib.OP_MIN_MAX, dst=0, src=0)
        for k in range(n_sc):
            a = Scalar.reg(3) if kind == "reg" else Scalar.const(2.0)
            p.add_op(_lib.OP_SCALAR_AFFINE, dst=4, sp=(a, Scalar.const(0.5), Scalar.const(1.0)))
        o = p.add_io("o", _lib.IO_SCALAR_OUT, np.float32)
        p.add_op(_lib.OP_STORE_SCALAR, io=o, ip=(4 if n_sc else 3,))
        ch = Chain(p, "t", np.float32)
        bufs = {"wf": wf, "col": DeviceArray.zeros((rows,), np.float32), "o": DeviceArray.zeros((rows,), np.float32)}
        for _ in range(2): ch.execute(bufs, rows, st)
        e0, e1 = Event(), Event(); e0.record(st)
        for _ in range(5): ch.execute(bufs, rows, st)
        e1.record(st); sync()
        res.append({"scalar_ops": n_sc, "operand": kind, "ms": e0.elapsed_ms(e1) / 5, "kernel": ch.kernel_name, **ch.geometry(rows)})
        print(json.dumps(res[-1]))
