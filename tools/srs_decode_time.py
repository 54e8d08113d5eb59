"""Time mh_bases_upload_serialized on 2^20 compressed points made from a device-generated SRS (GPU box)."""
import time, sys
import numpy as np
sys.path.insert(0, ".")
import marlin_amd as M
from marlin_amd import marlin as GM, _lib
from marlin_amd.api import Bases
M.init(0)
n = 1 << 20
srs = Bases.srs_powers(GM.fr_mont(0x123456789abcdef), n)
xy = srs.download()
L = _lib.FQ_LIMBS
P = _lib.Q_MOD if hasattr(_lib, "Q_MOD") else None
from oracle import fields as F
P = F.Q_MOD
rinv = pow(1 << (64 * L), -1, P)
t0 = time.perf_counter()
out = bytearray()
for row in xy:
    x = sum(int(v) << (64 * i) for i, v in enumerate(row[:L])) * rinv % P
    y = sum(int(v) << (64 * i) for i, v in enumerate(row[L:])) * rinv % P
    b = bytearray(x.to_bytes(8 * L, "little"))
    if y > P - y:
        b[-1] |= 0x80
    out += b
t1 = time.perf_counter()
b = Bases.from_serialized(bytes(out), n, True); M.synchronize()
t2 = time.perf_counter()
b2 = Bases.from_serialized(bytes(out), n, True); M.synchronize()
t3 = time.perf_counter()
assert (b2.download() == xy).all()
b3 = Bases(xy); M.synchronize()
t4 = time.perf_counter()
print("2^20 points: python serialisation %.1f s; mh_bases_upload_serialized (compressed, %d MB) %.1f ms, again %.1f ms; "
      "mh_bases_upload of the same points as limbs (%d MB) %.1f ms" % (t1 - t0, len(out) >> 20, (t2 - t1) * 1e3, (t3 - t2) * 1e3, xy.nbytes >> 20, (t4 - t3) * 1e3))
