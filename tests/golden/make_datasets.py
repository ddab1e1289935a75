#!/usr/bin/env python3
"""Pack the reference's input images (MIT-licensed data, /root/reference/LICENSE) into one
small fixture file.  Data only: 8-bit / 1-bit gray PNG pixels + the `filelist.txt` pairing
(reference format: /root/reference/src/Datasets.jl:54-65, one `true.png,noisy.png` per line).

Run in the build container only (the GPU box has no /root/reference):
    python tests/golden/make_datasets.py
Writes tests/golden/datasets.npz with, per dataset <name>:
    <name>/true  uint8 (K, H, W)   ground-truth images, PIL row-major (H rows, W columns)
    <name>/data  uint8 (K, H, W)   noisy images
1-bit PNGs (circle true image) are stored as 0/255.
"""
import os, sys
import numpy as np
from PIL import Image

REF = "/root/reference/datasets"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "datasets.npz")

def load_gray_u8(path):
    im = Image.open(path)
    a = np.array(im)
    if a.dtype == np.bool_:
        a = a.astype(np.uint8) * 255
    assert a.dtype == np.uint8 and a.ndim == 2, (path, a.dtype, a.shape)
    return a

def main():
    out = {}
    for name in sorted(os.listdir(REF)):
        d = os.path.join(REF, name)
        with open(os.path.join(d, "filelist.txt")) as fh:
            lines = [l.strip() for l in fh.read().split("\n") if l.strip()]
        tr, da = [], []
        for l in lines:
            t, n = l.split(",")
            tr.append(load_gray_u8(os.path.join(d, t)))
            da.append(load_gray_u8(os.path.join(d, n)))
        out[name + "/true"] = np.stack(tr)
        out[name + "/data"] = np.stack(da)
        print(name, out[name + "/true"].shape)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")

if __name__ == "__main__":
    sys.exit(main())
