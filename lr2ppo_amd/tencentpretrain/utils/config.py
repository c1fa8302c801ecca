"""Hyper-parameter merge with the reference's precedence: argparse defaults < JSON config < explicit CLI flags
(tencentpretrain/utils/config.py:6-23 of the reference; the CLI layer is recovered by scanning sys.argv)."""
import json
import sys
from argparse import Namespace


def load_hyperparam(default_args, argv=None):
    with open(default_args.config_path, mode="r", encoding="utf-8") as f:
        from_config = json.load(f)
    merged = dict(vars(default_args))
    argv = sys.argv if argv is None else argv
    explicit = [a[2:].split("=")[0] for a in argv if a.startswith("--") and "local_rank" not in a]
    cli = {k: merged[k] for k in explicit if k in merged}
    merged.update(from_config)
    merged.update(cli)
    return Namespace(**merged)
