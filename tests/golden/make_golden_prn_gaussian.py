#!/usr/bin/env python3
"""g12_prn_gaussian.npz: outputs of the REAL skimage.filters.gaussian (the function evaluate/tester.py:395-397 applies to
the one-hot PRN input maps via datasets/coco_data/prn_gaussian.py:2) on seeded 56x36 maps.

Run with an interpreter that has scikit-image (build image: /opt/conda/bin/python3.9 tests/golden/make_golden_prn_gaussian.py)."""
import os

import numpy as np
import skimage
from skimage.filters import gaussian

HERE = os.path.dirname(os.path.abspath(__file__))


def cases():
    rs = np.random.RandomState(12)
    maps = np.zeros((12, 56, 36))
    for i in range(12):
        n = [0, 1, 1, 2, 3, 5, 8, 1, 1, 1, 4, 30][i]
        ys, xs = rs.randint(0, 56, n), rs.randint(0, 36, n)
        if i == 7:
            ys, xs = np.array([0]), np.array([0])          # corner: 'nearest' edge handling
        if i == 8:
            ys, xs = np.array([55]), np.array([35])
        if i == 9:
            ys, xs = np.array([3]), np.array([35])
        maps[i, ys, xs] = 1
    return maps


def main():
    maps = cases()
    out = np.stack([gaussian(m) for m in maps])
    np.savez_compressed(os.path.join(HERE, "g12_prn_gaussian.npz"), maps=maps.astype(np.uint8), blurred=out,
                        skimage_version=np.array(skimage.__version__))
    print("skimage", skimage.__version__, out.dtype, out.max())


if __name__ == "__main__":
    main()
