"""Checkpoint I/O: the reference's save_net / load_net (network/net_utils.py:30-110) on the built-in
HDF5 subset.  Pinning: the READER against files written by the real h5py/libhdf5 exactly as the reference writes them
(tests/golden/g11_*.h5, made by tests/golden/make_golden_h5.py); the WRITER by reading its files back with libhdf5's own
tools when the image has them (h5dump, and h5py under /opt/conda/bin/python3.9) — and by round trip otherwise."""
import importlib.util
import json
import os
import pickle
import shutil
import subprocess

import numpy as np
import pytest
import torch

from helpers import GOLD

H5DUMP = shutil.which("h5dump") or ("/opt/conda/bin/h5dump" if os.path.exists("/opt/conda/bin/h5dump") else None)
PY_H5PY = "/opt/conda/bin/python3.9" if os.path.exists("/opt/conda/bin/python3.9") else None


def _golden_tensors():
    spec = importlib.util.spec_from_file_location("make_golden_h5", os.path.join(GOLD, "make_golden_h5.py"))
    src = open(spec.origin).read().replace("import h5py", "h5py = None")        # the generator itself needs h5py; its data does not
    ns = {"__file__": spec.origin}
    exec(compile(src, spec.origin, "exec"), ns)
    return ns["tensors"]()


def test_reader_on_files_written_by_real_h5py():
    from multiposenet.pytorch_amd.network import hdf5min
    want = _golden_tensors()
    for fname, prefix in (("g11_h5py_ckpt.h5", ""), ("g11_h5py_ckpt_module.h5", "module.")):
        with hdf5min.File(os.path.join(GOLD, fname)) as f:
            assert sorted(f.keys()) == sorted(prefix + k for k, _ in want)
            assert int(f.attrs["epoch"]) == 37
            for k, v in want:
                got = f[prefix + k]
                assert got.dtype == v.dtype and got.shape == v.shape and np.array_equal(got, v), k
            assert "nope" not in f
            with pytest.raises(KeyError):
                f["nope"]


def _tiny_state():
    rs = np.random.RandomState(5)
    d = [("w4", rs.randn(3, 2, 3, 3).astype(np.float32)), ("b", rs.randn(7).astype(np.float32)),
         ("count", np.array(12, dtype=np.int64)), ("empty", np.zeros((0, 4), np.float32)), ("f64", rs.randn(2, 2)),
         ("a.very.long.dotted.name.that.goes.on.and.on.weight", rs.randn(5).astype(np.float32))]
    d += [("layer.%04d.weight" % i, rs.randn(2, 3).astype(np.float32)) for i in range(1500)]        # > 1 symbol-table node
    return d


def test_writer_round_trip_and_libhdf5_tools(tmp_path):
    from multiposenet.pytorch_amd.network import hdf5min
    data = _tiny_state()
    path = str(tmp_path / "own.h5")
    hdf5min.write_file(path, data, attrs={"epoch": np.int64(5), "lr": np.float64(1e-4)})
    with hdf5min.File(path) as f:
        assert len(f) == len(data) and int(f.attrs["epoch"]) == 5 and float(f.attrs["lr"]) == 1e-4
        for k, v in data:
            got = f[k]
            assert got.dtype == v.dtype and got.shape == v.shape and np.array_equal(got, v), k
    if H5DUMP:            # libhdf5's own parser must accept the file and see the same names / values
        hdr = subprocess.run([H5DUMP, "-H", path], capture_output=True, text=True)
        assert hdr.returncode == 0, hdr.stderr
        assert hdr.stdout.count("DATASET ") == len(data) and 'ATTRIBUTE "epoch"' in hdr.stdout
        one = subprocess.run([H5DUMP, "-d", "/b", "-y", "-w", "0", path], capture_output=True, text=True)
        assert one.returncode == 0, one.stderr
        vals = one.stdout.split("DATA {")[1].split("}")[0].replace("\n", " ").split(",")
        assert np.allclose(np.array([float(x) for x in vals], np.float32), dict(data)["b"], rtol=1e-5, atol=0)     # h5dump prints 6 significant digits
    if PY_H5PY:           # and h5py reads every value bit-exactly
        code = ("import h5py, numpy as np, json, sys\n"
                "f = h5py.File(sys.argv[1], 'r')\n"
                "out = {k: [str(f[k].dtype), list(f[k].shape), float(np.asarray(f[k], dtype=np.float64).sum())] for k in f.keys()}\n"
                "print(json.dumps({'epoch': int(f.attrs['epoch']), 'lr': float(f.attrs['lr']), 'd': out}))\n")
        res = subprocess.run([PY_H5PY, "-c", code, path], capture_output=True, text=True)
        assert res.returncode == 0, res.stderr
        got = json.loads(res.stdout)
        assert got["epoch"] == 5 and got["lr"] == 1e-4 and len(got["d"]) == len(data)
        for k, v in data:
            dt, shape, s = got["d"][k]
            assert dt == str(v.dtype) and tuple(shape) == v.shape and s == float(v.astype(np.float64).sum()), k


def test_save_net_load_net_keep_the_reference_contract(tmp_path):
    """save_net -> load_net restores every state_dict entry bit-exactly (logical [Cout,Cin,R,S] layout on disk although the
    arena stores [Cout][R][S][Cin]); 'module.'-prefixed files load into a bare model; shape mismatches and missing layers
    only warn; the optimizer pickle restores FusedAdam; old optimizer states / checkpoints are pruned as the reference does."""
    from multiposenet.pytorch_amd.network import hdf5min, net_utils
    from multiposenet.pytorch_amd.network.posenet import poseNet
    from multiposenet.pytorch_amd.optim import FusedAdam
    torch.manual_seed(3)
    m = poseNet(50, prn_node_count=8, prn_coeff=1)
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn_like(p))
        m.fpn.bn1.running_var.uniform_(0.5, 2.0)
        m.fpn.bn1.num_batches_tracked.fill_(77)
    opt = FusedAdam(m, lr=2e-4)
    opt._bind()
    opt._m.normal_()
    opt._v.uniform_()
    opt._set_step(9)
    d = str(tmp_path)
    f1 = os.path.join(d, "ckpt_1.h5")
    net_utils.save_net(f1, m, epoch=1, optimizers=[opt], rm_prev_opt=True, max_n_ckpts=2)
    with hdf5min.File(f1) as f:
        assert sorted(f.keys()) == sorted(m.state_dict().keys())
        w = m.fpn.layer1[0].conv2.weight
        assert np.array_equal(f["fpn.layer1.0.conv2.weight"], w.detach().numpy()) and f["fpn.layer1.0.conv2.weight"].shape == tuple(w.shape)
        assert f["fpn.bn1.num_batches_tracked"].shape == () and int(f["fpn.bn1.num_batches_tracked"]) == 77
    m2 = poseNet(50, prn_node_count=8, prn_coeff=1)
    epoch, lr = net_utils.load_net(f1, m2)
    assert epoch == 1 and len(lr) == 0
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    assert m2._arena.consistent()                       # loaded in place: parameters still live in the arena
    epoch, sds = net_utils.load_net(f1, m2, load_state_dict=True)
    opt2 = FusedAdam(m2, lr=1e-3)
    opt2.load_state_dict(sds[0])
    assert opt2.step_count() == 9 and opt2.param_groups[0]["lr"] == 2e-4
    a, b = opt.state_dict()["state"], opt2.state_dict()["state"]
    assert len(a) == len(b) > 100 and all(torch.equal(a[i]["exp_avg"], b[i]["exp_avg"]) and torch.equal(a[i]["exp_avg_sq"], b[i]["exp_avg_sq"]) for i in a)
    # pruning (net_utils.py:49-66)
    for e in (2, 3):
        net_utils.save_net(os.path.join(d, "ckpt_%d.h5" % e), m, epoch=e, optimizers=[opt], rm_prev_opt=True, max_n_ckpts=2)
    names = sorted(os.listdir(d))
    assert names == ["ckpt_2.h5", "ckpt_3.h5", "ckpt_3.h5.optimizer_state.pk"], names
    # a DataParallel-style file ('module.' prefix) loads into a bare model; mismatches only warn
    sd = {("module." + k): v.numpy() for k, v in m.state_dict().items()}
    sd["module.convfin.bias"] = np.zeros(5, np.float32)            # wrong shape
    del sd["module.convfin.weight"]                                 # missing layer
    fm = os.path.join(d, "dp_9.h5")
    hdf5min.write_file(fm, sd, attrs={"epoch": np.int64(9)})
    m3 = poseNet(50, prn_node_count=8, prn_coeff=1)
    before = m3.convfin.bias.detach().clone(), m3.convfin.weight.detach().clone()
    epoch, _ = net_utils.load_net(fm, m3)
    assert epoch == 9
    assert torch.equal(m3.fpn.conv1.weight, m.fpn.conv1.weight)
    assert torch.equal(m3.convfin.bias, before[0]) and torch.equal(m3.convfin.weight, before[1])
