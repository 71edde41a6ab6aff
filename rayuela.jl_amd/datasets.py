"""Host mirror of src/read_datasets.jl (read_dataset) and demos/experiment_utils.jl:62-90 (load_experiment_data):
the name -> file table of the reference's demos, on top of the xvecs readers and the libhdf5 binding of this
package.  Arrays come back in the memory-image convention of the other mirrors: (n, d)
numpy == Julia's d-by-n matrix.

The reference hard-codes paths relative to the working directory ("./data/...") and a few absolute ones on its
authors' machines; `data_root` replaces the "./data" prefix, absolute entries are kept and can be overridden through
`paths={name: path}`."""
import os

import numpy as np

from . import h5results
from .xvecs import bvecs_read, fvecs_read, ivecs_read

# name -> (reader kind, path, hdf5 dataset name)            src/read_datasets.jl:11-200
TABLE = {
    "Deep1M_babenko": ("fvecs", "./data/deep_babenko/deep1M_learn.fvecs", None),
    "Deep1M_babenko_groundtruth": ("ivecs", "./data/deep_babenko/deep1M_groundtruth.ivecs", None),
    "Deep1M_babenko_query": ("fvecs", "./data/deep_babenko/deep1M_queries.fvecs", None),
    "Deep1M_babenko_base": ("fvecs", "./data/deep_babenko/deep1M_base.fvecs", None),
    "GIST1M": ("fvecs", "./data/gist/gist_learn.fvecs", None),
    "GIST1M_query": ("fvecs", "./data/gist/gist_query.fvecs", None),
    "GIST1M_groundtruth": ("ivecs", "./data/gist/gist_groundtruth.ivecs", None),
    "GIST1M_base": ("fvecs", "./data/gist/gist_base.fvecs", None),
    "SIFT1M": ("fvecs", "./data/sift/sift_learn.fvecs", None),
    "SIFT1M_query": ("fvecs", "./data/sift/sift_query.fvecs", None),
    "SIFT1M_groundtruth": ("ivecs", "./data/sift/sift_groundtruth.ivecs", None),
    "SIFT1M_base": ("fvecs", "./data/sift/sift_base.fvecs", None),
    "Convnet1M_base": ("h5", "./data/feats/feats_m_128.mat", "feats_m_128_base"),
    "Convnet1M": ("h5", "./data/feats/feats_m_128.mat", "feats_m_128_train"),
    "Convnet1M_query": ("h5", "./data/feats/feats_m_128.mat", "feats_m_128_test"),
    "Convnet1M_groundtruth": ("h5", "./data/feats/feats_m_128_gt.mat", "gt"),
    "Deep1M_base": ("h5", "./data/deep/deep.h5", "base"),
    "Deep1M": ("h5", "./data/deep/deep.h5", "train"),
    "Deep1M_query": ("h5", "./data/deep/deep.h5", "query"),
    "Deep1M_groundtruth": ("h5", "./data/deep/deep.h5", "gt"),
    "Deep1B": ("fvecs", "./data/deep1b/learn_00", None),
    "Deep1B_base": ("fvecs", "/scratch/julm/deep1b/base_all", None),
    "SIFT1B_query": ("bvecs_f32", "/hdd/sift1b/bigann_query.bvecs", None),
    "SIFT1B_base": ("bvecs_f32", "/hdd/sift1b/bigann_base.bvecs", None),
    "SIFT1B": ("bvecs_f32", "/hdd/sift1b/bigann_learn.bvecs", None),
    "SIFT10M": ("bvecs_f32", "/hdd/sift1b/bigann_learn.bvecs", None),
    "SIFT1B_groundtruth": ("ivecs", "/hdd/sift1b/gnd/idx_1000M.ivecs", None),
    "SIFT1B_groundtruth_10M": ("ivecs", "/hdd/sift1b/gnd/idx_10M.ivecs", None),
    "SIFT1B_groundtruth_1M": ("ivecs", "/hdd/sift1b/gnd/idx_1M.ivecs", None),
    "MNIST_query": ("h5_mnist", "./data/mnist/mnist.h5", "test"),
    "MNIST": ("h5_mnist", "./data/mnist/mnist.h5", "train"),
    "MNIST_base": ("h5_mnist", "./data/mnist/mnist.h5", "train"),
    "MNIST_groundtruth": ("h5", "./data/mnist/mnist.h5", "gt"),
    "labelme_query": ("h5_f32", "./data/labelme/label.h5", "query"),
    "labelme": ("h5_f32", "./data/labelme/label.h5", "train"),
    "labelme_base": ("h5_f32", "./data/labelme/label.h5", "train"),
    "labelme_groundtruth": ("h5", "./data/labelme/label.h5", "gt"),
}
EARLY_RETURN = {"Deep1M_groundtruth"}      # `return X` right after h5read (:118-122): not subset


def _resolve(path, data_root):
    if path.startswith("./data/"):
        return os.path.join(data_root, path[len("./data/"):])
    return path


def read_dataset(dname, nvectors, V=False, data_root="./data", paths=None):
    """read_dataset(dname, nvectors, V=false)      (src/read_datasets.jl:4-9)

    nvectors: n (the first n vectors) or a one-based inclusive (a, b) range, as the reference's Integer / UnitRange.
    HDF5-backed sets are read whole and then cut to the first n / the range (:231-242; asking for more vectors
    than the file holds is an error), except Deep1M_groundtruth, which the reference returns as stored."""
    if V:
        print("Loading %s... " % dname)
    if dname not in TABLE:
        raise KeyError("dataset %s unknown" % dname)
    kind, path, h5name = TABLE[dname]
    path = (paths or {}).get(dname, _resolve(path, data_root))
    if kind == "fvecs":
        return fvecs_read(nvectors, path)
    if kind == "ivecs":
        return ivecs_read(nvectors, path)
    if kind == "bvecs_f32":
        return bvecs_read(nvectors, path).astype(np.float32)          # convert(Matrix{Float32}, X)
    X = h5results.h5read(path, h5name)
    if kind == "h5_mnist":
        X = X.reshape(X.shape[0], -1).astype(np.float32)            # reshape(X, 28*28, N); convert(Matrix{Float32}, X)
    elif kind == "h5_f32":
        X = X.astype(np.float32)
    if dname in EARLY_RETURN:
        return X
    n = X.shape[0]                                                   # `_, n = size(X)`: our rows are Julia's columns
    if isinstance(nvectors, (int, np.integer)):
        if nvectors > n:
            raise ValueError("Asked to read %d vectors, but the datasets has only %d vectors." % (nvectors, n))
        return X[:nvectors]
    a, b = int(nvectors[0]), int(nvectors[1])
    return X[a - 1:b]


def load_experiment_data(dataset_name, ntrain, nbase, nquery, V=False, data_root="./data", paths=None):
    """load_experiment_data(dataset_name, ntrain, nbase, nquery, V=false) -> Xt, Xb, Xq, gt
    (demos/experiment_utils.jl:62-90).  gt comes back as the ONE-based uint32 id of the nearest neighbour of each
    query: SIFT1M / GIST1M ship zero-based ground truth (+1, :74-76); only the top neighbour is kept (:79-83)."""
    kw = dict(V=V, data_root=data_root, paths=paths)
    Xt = read_dataset(dataset_name, ntrain, **kw)
    Xb = read_dataset(dataset_name + "_base", nbase, **kw)
    Xq = read_dataset(dataset_name + "_query", nquery, **kw)[:nquery]
    gt = read_dataset(dataset_name + "_groundtruth", nquery, **kw)
    if dataset_name in ("SIFT1M", "GIST1M"):
        gt = gt + 1
    gt = np.asarray(gt)
    if dataset_name != "Deep1M":
        gt = gt[:nquery, 0]                 # Julia gt[1, 1:nquery]: first row = nearest neighbour, one column per query
    else:
        gt = gt.reshape(-1)[:nquery]        # Julia gt[1:nquery]
    return Xt, Xb, Xq, gt.astype(np.uint32)
