"""Where does dev.transpose differ from the oracle?  (development tool)  AB_LIB=... M4RI_HIP_TRANSPOSE_FLAGS=... python tools/transpose_check.py r c"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
if os.environ.get("AB_LIB"):
    import m4ri_rust_amd  # noqa
    from m4ri_rust_amd import _lib
    _lib.LIB_PATH = os.environ["AB_LIB"]
import numpy as np
import gf2util as g
from m4ri_rust_amd import device as dev
r, c = int(sys.argv[1]), int(sys.argv[2])
w = g.random_words(r, c, 23)
T = dev.transpose(dev.DMat.from_words(w, c)).to_words()
ref = g.o_transpose(w, r, c)
bad = np.argwhere(T != ref)
print("flags", os.environ.get("M4RI_HIP_TRANSPOSE_FLAGS"), "shape", r, c, "out", T.shape, "mismatching words", len(bad))
if len(bad):
    rows = np.unique(bad[:, 0]); cols = np.unique(bad[:, 1])
    print("rows", rows[:10], "...", rows[-5:], len(rows), "tile rows", np.unique(rows // 512)[:20])
    print("word cols", cols[:20], "...", cols[-5:], len(cols), "tile cols", np.unique(cols // 8)[:40])
    i, j = bad[0]
    print("first", i, j, hex(int(T[i, j])), hex(int(ref[i, j])))
