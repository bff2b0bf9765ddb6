"""Extracts the first sensor-manager messages of the Victoria Park dataset shipped with the reference
(/root/reference/data/VictoriaPark: Sensors_manager.txt, inputs.dat, measurements.dat -- data files, not source) into a small
fixture: tests/golden/victoria_park_extract.npz.  LASER.txt is missing from the reference (.MISSING_LARGE_BLOBS), so tests
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
print("messages", len(mgr), "lidar", int((mgr[:, 1] == 3).sum()), "inputs", len(inputs), "detections", len(meas))
