#!/usr/bin/env python3
"""LETOR TSV files -> the per-query HDF5 files the `_trad` twins read -- the job of the reference's datasets_trad/convert_to_h5py.py,
usable where h5py is absent (files are written through lr2ppo_amd.h5lite = libhdf5 via ctypes; with h5py installed, through h5py).

Same command line and the same result: every `*.tsv` of --original_dir (tab-separated, no header: label, query id, features ...) is
grouped by query id (column 1, as integers); a query with fewer than 20 rows is resampled WITH replacement to 20, one with more
WITHOUT replacement to 20 (sklearn.utils.resample, random_state=0: datasets_trad/convert_to_h5py.py:17-23), and each query becomes
one float64 dataset named by its id in `<--target_dir>/<name>.h5`.  Host-side data preparation (pandas + scikit-learn, as upstream).

    python tools/convert_to_h5py.py --original_dir TSV_DIR --target_dir H5_DIR [--limit_rows N]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

DOCS = 20


def query_tables(tsv_path, limit_rows=None):
    """{query id: [20, columns] float array} of one TSV file."""
    import pandas as pd
    from sklearn.utils import resample
    table = pd.read_csv(tsv_path, sep="\t", header=None, nrows=limit_rows)
    table[1] = table[1].astype(int)
    out = {}
    for qid, rows in table.groupby(1):
        if len(rows) != DOCS:
            rows = resample(rows, replace=len(rows) < DOCS, n_samples=DOCS, random_state=0)
        out[qid] = rows.values
    return out


def convert(original_dir, target_dir, limit_rows=None):
    from lr2ppo_amd import h5lite
    os.makedirs(target_dir, exist_ok=True)
    done = []
    for name in sorted(os.listdir(original_dir)):
        if not name.endswith(".tsv"):
            continue
        tables = query_tables(os.path.join(original_dir, name), limit_rows)
        dst = os.path.join(target_dir, name[:-4] + ".h5")
        with h5lite.open_file(dst, "w") as hf:
            for qid, values in tables.items():
                hf.create_dataset(str(qid), data=values)
        print(f"Converted {name} to {os.path.basename(dst)} ({len(tables)} queries)")
        done.append(dst)
    return done


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description="Convert tsv files to h5 format.")
    ap.add_argument("--original_dir", required=True, help="Source directory containing tsv files")
    ap.add_argument("--target_dir", required=True, help="Target directory to save h5 files")
    ap.add_argument("--limit_rows", type=int, default=None, help="Limit processing to first N rows (optional)")
    a = ap.parse_args()
    convert(a.original_dir, a.target_dir, a.limit_rows)
