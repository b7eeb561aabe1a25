"""The exit-time fault of round 2 (three SIGSEGVs after "tool finalization" under rocprofv3, gpurun_out/rocprof_segv.log), with
the library in the picture: a launch-bound problem swept through the persistent kernel, launched cooperatively
(RRI_ONCHIP_COOP=1: hipLaunchCooperativeKernel) or as the plain launch the library uses, the handle destroyed before exit
("release") or left to interpreter shutdown ("leak").  /proc/self/maps is written from an atexit hook -- before the C
exit handlers run, with every DSO still mapped -- so the frames the tool's signal handler prints can be resolved
(tools/exit_probe/resolve_frames.py).

    python3 tools/exit_probe/lib_exit_probe.py <maps-out> coop|plain release|leak
"""
import atexit
import os
import sys

maps_out, launch, owner = sys.argv[1], sys.argv[2], sys.argv[3]


def dump_maps():
    with open('/proc/self/maps') as f, open(maps_out, 'w') as g:
        g.write(f.read())


atexit.register(dump_maps)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ['RRI_ONCHIP_COOP'] = '1' if launch == 'coop' else '0'
import numpy as np  # noqa: E402
from rri_nmf_amd.engine import RRIEngine  # noqa: E402
from rri_nmf_amd.synthetic import planted_X, scaled_init  # noqa: E402

n, d, k = 10000, 1000, 20
X = planted_X(n, d, k, seed=0, dtype=np.float32)
W0, T0 = scaled_init(X, k, seed=1)
e = RRIEngine(n, d, k, dtype=np.float32)
e.upload_X(X), e.set_W(W0), e.set_T(T0), e.set_params()
e.sweep(3)
e.sweep(3)
print('%s launch: persistent launches %r, fallbacks %d, objective %.6e; handle %s' % (
    launch, e.onchip_info(), e.onchip_fallbacks(), e.objective(), 'destroyed before exit' if owner == 'release' else 'left to exit'), flush=True)
if owner == 'release':
    e.close()
else:
    KEEP = e            # alive until the interpreter tears the module down
