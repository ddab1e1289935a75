"""Dataset format of the reference: a directory with `filelist.txt`, one `true.png,noisy.png` per
line, gray PNGs scaled to [0, 1] Float64 and stacked as (M, N, K)
(/root/reference/src/Datasets.jl:54-65).  Returned here as (K, N, M) C-contiguous arrays == the same
column-major memory.  Names resolve by prefix like `full_datasetname` (:27-31).
"""
import os
import numpy as np

remotedatasets = ["cameraman_128_5", "cameraman_128_10", "faces_train_128_10", "faces_val_128_10",
                  "circle_128_10"]  # /root/reference/src/Datasets.jl:11-17


def full_datasetname(name):
    for d in remotedatasets:
        if d.startswith(name):
            return d
    raise ValueError('"%s" not found in remotedatasets %s' % (name, remotedatasets))


def _to_julia_batch(imgs_u8):
    a = np.asarray(imgs_u8, dtype=np.float64) / 255.0      # (K, H, W), PIL row major
    return np.ascontiguousarray(np.transpose(a, (0, 2, 1)))  # Julia A[row, col] column major


def load_filelist_dataset(directory):
    from PIL import Image
    with open(os.path.join(directory, "filelist.txt")) as fh:
        pairs = [l.strip().split(",") for l in fh.read().split("\n") if l.strip()]

    def rd(p):
        a = np.array(Image.open(os.path.join(directory, p)))
        if a.dtype == np.bool_:
            a = a.astype(np.uint8) * 255
        return a

    t = np.stack([rd(p[0]) for p in pairs])
    d = np.stack([rd(p[1]) for p in pairs])
    return _to_julia_batch(t), _to_julia_batch(d)


def testdataset(name, root=None, npz=None):
    """(true_images, data_images).  `root`: directory holding the dataset folders; `npz`: the packed
    fixture tests/golden/datasets.npz (uint8 pixels of the reference's MIT-licensed images)."""
    full = full_datasetname(name)
    if npz is not None:
        z = np.load(npz)
        return _to_julia_batch(z[full + "/true"]), _to_julia_batch(z[full + "/data"])
    if root is None:
        raise ValueError("give root= (dataset directory) or npz= (packed fixture)")
    return load_filelist_dataset(os.path.join(root, full))
