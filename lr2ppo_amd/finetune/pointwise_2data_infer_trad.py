"""The dimension-projection step of the `_trad` pipeline -- drop-in for the reference's finetune/pointwise_2data_infer_trad.py.

Upstream (:409-447) loads a trained pointwise_2data_trad.Classifier (`--dim_proj_ckpt_path`) and rewrites every `*.tsv` of `--input_dir`
(label, query id, 46 MQ2008 or 136 MSLR-WEB10K features per row) as label, query id and the row's 768 projected features --
`text_proj` / `text_proj3` = Mlp(46 | 136, 3072, 768), chosen by the row's width -- into `--output_dir`; those files, converted by
datasets_trad/convert_to_h5py.py, are what pointwise_trad / reward_trad / ppo_trad train on.  The reference projects one row per
forward; here a file's rows of one width go through `engine.feature_proj_forward` in slabs (the same two fused GEMMs as the
classifier's own forward).  Row order, the two leading columns (copied as text) and the float formatting of csv.writer on
`tensor.tolist()` are upstream's.  No CPU fallback.
"""
from __future__ import annotations

import csv
import os
from pathlib import Path

import torch

from .. import engine
from . import pointwise_2data_trad as p2
from .ppo import FEAT

SLAB = 16384      # rows per projection call


@torch.no_grad()
def project_rows(model: "p2.Classifier", rows: torch.Tensor) -> torch.Tensor:
    """rows [n, 46 | 136] fp32 on the HIP device -> [n, 768]: model.text_proj (46) or model.text_proj3 (136), pointwise_2data_infer_trad.py:437-443."""
    if rows.dtype != torch.float32 or not rows.is_cuda:
        raise TypeError("lr2ppo_amd: rows must be a float32 tensor on the HIP device (no CPU path)")
    if rows.dim() != 2 or rows.shape[1] not in p2.PROJ:
        raise ValueError("rows must be [n, 46] (MQ2008) or [n, 136] (MSLR-WEB10K)")
    if model._ws is None or model._ws.device != rows.device:
        model._ws = engine.Workspace(rows.device)
    out = torch.empty(rows.shape[0], FEAT, device=rows.device)
    P = model._P()
    for r0 in range(0, rows.shape[0], SLAB):
        part = rows[r0:r0 + SLAB].contiguous()
        out[r0:r0 + part.shape[0]].copy_(engine.feature_proj_forward(model._ws, P, p2.PROJ[part.shape[1]], part, part.shape[0],
                                                                     part.shape[1], FEAT, save=False))
    return out


def project_tsv(model, src: str, dst: str, device) -> int:
    """One file: every row's columns 2: projected, columns 0-1 copied (upstream :429-447).  -> rows written."""
    with open(src, "r") as f_in:
        rows = [row for row in csv.reader(f_in, delimiter="\t")]
    widths = {len(r) - 2 for r in rows}
    bad = widths - set(p2.PROJ)
    if bad:
        raise ValueError(f"{src}: rows with {sorted(bad)} features; the projections take 46 or 136 (pointwise_2data_infer_trad.py:437-443)")
    projected = [None] * len(rows)
    for width in sorted(widths):
        idx = [i for i, r in enumerate(rows) if len(r) - 2 == width]
        feats = torch.tensor([[float(v) for v in rows[i][2:]] for i in idx], dtype=torch.float32, device=device)
        out = project_rows(model, feats).cpu().tolist()
        for i, o in zip(idx, out):
            projected[i] = o
    with open(dst, "w") as f_out:
        writer = csv.writer(f_out, delimiter="\t")
        for row, feat in zip(rows, projected):
            writer.writerow(row[:2] + feat)
    return len(rows)


def main(argv=None):
    """python -m lr2ppo_amd.finetune.pointwise_2data_infer_trad --dim_proj_ckpt_path M.bin --input_dir TSV_IN --output_dir TSV_OUT"""
    import argparse
    from .pointwise_trad import build_parser
    parser = build_parser()
    parser.add_argument("--dim_proj_ckpt_path", type=str, required=True, help="checkpoint of pointwise_2data_trad.Classifier")
    parser.add_argument("--input_dir", type=str, required=True)
    parser.add_argument("--output_dir", type=str, required=True)
    args = parser.parse_args(argv)
    args.labels_num = 3
    model = p2.Classifier(args, argparse.Namespace(**vars(args)))
    model.load_state_dict(torch.load(args.dim_proj_ckpt_path, map_location="cpu"))        # strict, as upstream (:413)
    device = torch.device("cuda", 0)
    model = model.to(device).eval()
    out_dir = Path(args.output_dir)
    out_dir.mkdir(parents=True, exist_ok=True)
    total = 0
    for tsv in sorted(Path(args.input_dir).glob("*.tsv")):
        n = project_tsv(model, str(tsv), str(out_dir / tsv.name), device)
        print(f"{tsv.name}: {n} rows -> {FEAT} features")
        total += n
    return total


if __name__ == "__main__":
    main()
