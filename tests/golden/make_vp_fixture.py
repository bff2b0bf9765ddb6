"""Extracts the first sensor-manager messages of the Victoria Park dataset shipped with the reference
(/root/reference/data/VictoriaPark: Sensors_manager.txt, inputs.dat, measurements.dat -- data files, not source) into a small
fixture: tests/golden/victoria_park_extract.npz (+ tests/golden/vp_extract/*.txt|dat, the first 900 messages in the dataset's text formats).  LASER.txt is missing from the reference (.MISSING_LARGE_BLOBS), so tests
use the synthetic constant-70 m scan of SURVEY §8d.   Run in the build container: python tests/golden/make_vp_fixture.py"""
import os

import numpy as np

SRC = "/root/reference/data/VictoriaPark"
N_MSG = 1500
here = os.path.dirname(os.path.abspath(__file__))
mgr = np.loadtxt(os.path.join(SRC, "Sensors_manager.txt"), max_rows=N_MSG)
t_end = mgr[-1, 0]
inputs = np.loadtxt(os.path.join(SRC, "inputs.dat"), max_rows=int(mgr[mgr[:, 1] == 2][:, 2].max()) + 1)
meas = np.loadtxt(os.path.join(SRC, "measurements.dat"), max_rows=20000)
meas = meas[meas[:, 0] <= t_end + 1e-9]
np.savez_compressed(os.path.join(here, "victoria_park_extract.npz"), manager=mgr, inputs=inputs, measurements=meas)
# the same extract in the dataset's own text formats (what the C++ driver host/rbphdslam_vp reads), first 900 messages
txt = os.path.join(here, "vp_extract")
os.makedirs(txt, exist_ok=True)
m9 = mgr[:900]
with open(os.path.join(txt, "Sensors_manager.txt"), "w") as fh:
    for t, typ, idx in m9:
        fh.write("%.3f\t%d\t%d\n" % (t, int(typ), int(idx)))
n_in = int(m9[m9[:, 1] == 2][:, 2].max())
with open(os.path.join(txt, "inputs.dat"), "w") as fh:
    for t, v, r in inputs[:n_in]:
        fh.write("%10.3f %10.3f %10.4f\n" % (t, v, r))
with open(os.path.join(txt, "measurements.dat"), "w") as fh:
    for t, r, b, d in meas[meas[:, 0] <= m9[-1, 0] + 1e-9]:
        fh.write("%10.3f %10.5f %10.5f %10.5f\n" % (t, r, b, d))
print("messages", len(mgr), "lidar", int((mgr[:, 1] == 3).sum()), "inputs", len(inputs), "detections", len(meas))
