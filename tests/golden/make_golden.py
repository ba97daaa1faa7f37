"""Regenerates tests/golden/voxel_errors_ref.npz from the reference's own run output.

Source: /root/reference/map_eval/scripts/voxel_errors.txt (7129 rows x 27 columns, written by
map_eval.cpp:292-302 in the run whose screenshot README.md:170 shows `VMD: 0.35303`, `SCS: 0.78121`)
and /root/reference/map_eval/scripts/voxel_wasserstein_cdf.txt (map_eval.cpp:337-340).
These are DATA produced by the reference (its only known-answer vectors), not source code.
Run here (the reference tree is not present on the GPU box): python tests/golden/make_golden.py
"""
import os

import numpy as np

REF = "/root/reference/map_eval/scripts"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "voxel_errors_ref.npz")

rows = np.loadtxt(os.path.join(REF, "voxel_errors.txt"))
cdf = np.loadtxt(os.path.join(REF, "voxel_wasserstein_cdf.txt"))
assert rows.shape == (7129, 27) and cdf.shape == (7129, 2)
np.savez_compressed(OUT, rows=rows, cdf=cdf, voxel_size=np.float64(3.0),
                    screenshot_vmd=np.float64(0.35303), screenshot_scs=np.float64(0.78121))
print("wrote", OUT, os.path.getsize(OUT), "bytes")
