#!/usr/bin/env python3
"""Writes tests/golden/g11_h5py_ckpt.h5 (+ g11_h5py_ckpt_module.h5) with the REAL h5py, exactly the way the reference's
save_net does (network/net_utils.py:30-35: create_dataset(k, data=v.cpu().numpy()) per key, attrs['epoch']).

Run with an interpreter that has h5py (in the build image: /opt/conda/bin/python3.9 tests/golden/make_golden_h5.py).
The fixtures pin multiposenet/pytorch_amd/network/hdf5min.py's reader to files produced by libhdf5; the expected values are
regenerated from the same seeded generator by tests/test_checkpoint.py."""
import os

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def tensors():
    """A miniature state_dict with the shapes/dtypes/names a poseNet checkpoint has (4-D conv weights, 1-D BN vectors,
    0-dim int64 counters, a 2-D Linear weight, dotted names) — enough entries to need more than one symbol-table node."""
    rs = np.random.RandomState(11)
    out = []
    out.append(("fpn.conv1.weight", rs.randn(8, 3, 7, 7).astype(np.float32)))
    for i in range(12):
        p = "fpn.layer%d.%d" % (1 + i % 4, i // 4)
        out.append((p + ".conv2.weight", rs.randn(4, 4, 3, 3).astype(np.float32)))
        out.append((p + ".bn2.weight", rs.rand(4).astype(np.float32)))
        out.append((p + ".bn2.bias", rs.randn(4).astype(np.float32)))
        out.append((p + ".bn2.running_mean", rs.randn(4).astype(np.float32)))
        out.append((p + ".bn2.running_var", rs.rand(4).astype(np.float32) + 0.5))
        out.append((p + ".bn2.num_batches_tracked", np.array(100 + i, dtype=np.int64)))
    out.append(("convfin.weight", rs.randn(18, 16, 1, 1).astype(np.float32)))
    out.append(("convfin.bias", rs.randn(18).astype(np.float32)))
    out.append(("prn.dens1.weight", rs.randn(5, 7).astype(np.float32)))
    out.append(("prn.dens1.bias", rs.randn(5).astype(np.float32)))
    return out


def main():
    for name, prefix in (("g11_h5py_ckpt.h5", ""), ("g11_h5py_ckpt_module.h5", "module.")):
        with h5py.File(os.path.join(HERE, name), mode="w") as h5f:
            for k, v in tensors():
                h5f.create_dataset(prefix + k, data=v)
            h5f.attrs["epoch"] = 37
    print("h5py", h5py.__version__, "hdf5", h5py.version.hdf5_version)


if __name__ == "__main__":
    main()
