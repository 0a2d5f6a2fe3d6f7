"""Generates tests/golden/realsense_reference.npz by IMPORTING the reference's own
/root/reference/utils/read_realsense_depth.py in this container (REFERENCE outputs, not self-oracle).

cv2 is not installed: it is stubbed in sys.modules and its ``remap`` records the two sampling maps the
reference computes (K_old K_new^-1 applied to the new pixel grid, read_realsense_depth.py:24-45); the
interpolation itself is OpenCV's (INTER_LINEAR, constant-0 border) and is restated in
touch_gs_amd.plumbing.remap_bilinear.  Only data is written; no reference source is copied.
Run from the repo root:  python tests/golden/make_realsense_golden.py
"""
import os
import sys
import types

import numpy as np

REF = "/root/reference/utils"


def main():
    rec = {}
    cv2 = types.ModuleType("cv2")
    cv2.INTER_LINEAR = 1

    def remap(img, map_x, map_y, interpolation=1):
        rec["map_x"], rec["map_y"] = map_x.copy(), map_y.copy()
        return img

    cv2.remap = remap
    sys.modules["cv2"] = cv2
    sys.path.insert(0, REF)
    import read_realsense_depth as rrd   # noqa: E402

    out = {}
    img = np.zeros((20, 30))
    # (a) the reference's default intrinsics on a reduced output grid
    rrd.convert_intrinsics(img, new_size=(48, 27))
    out["default/map_x"], out["default/map_y"] = rec["map_x"], rec["map_y"]
    # (b) explicit intrinsics
    old, new = (90.0, 92.0, 15.5, 9.25), (200.0, 210.0, 33.0, 19.5)
    rrd.convert_intrinsics(img, old_intrinsics=old, new_intrinsics=new, new_size=(64, 40))
    out["custom/old"], out["custom/new"] = np.array(old), np.array(new)
    out["custom/map_x"], out["custom/map_y"] = rec["map_x"], rec["map_y"]
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "realsense_reference.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
