#!/usr/bin/env python3
"""Converts a reference dataset container (HDF5: video_N/{features,gtscore,user_summary,user_scores,change_points,
n_frames,picks}, reference data/dataset.py:64-136) into the .npz container `video-summarization_amd/data.py` reads when
h5py is not installed.  Run where h5py is available:   python tools/h5_to_npz.py <file.h5> [...]"""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(paths):
    import h5py
    data = importlib.import_module("video-summarization_amd.data")
    for path in paths:
        with h5py.File(path, "r") as f:
            videos = {k: {fld: np.array(f[k][fld]) for fld in f[k].keys()} for k in f.keys()}
        out = data.write_npz_container(path[:-3] + ".npz" if path.endswith(".h5") else path + ".npz", videos)
        print(path, "->", out, "(%d videos)" % len(videos))


if __name__ == "__main__":
    main(sys.argv[1:])
