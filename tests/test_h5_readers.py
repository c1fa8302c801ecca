"""A13 / SURVEY 8(f)-3: the readers on REAL HDF5 files.

`lr2ppo_amd.h5lite` binds the HDF5 C library the image ships (no h5py here); these tests write LRMovieNet- and LETOR-shaped
files through it, check them with the library's own `h5dump` where that tool exists, and run the product's readers against
fixtures the REFERENCE's reader classes produced on the same real files (tests/golden/letor_readers.json, readers.json;
oracle/gen_golden.py::gen_letor_readers, gen_readers).  The `gpu`-marked tests run the four `_trad` entry points on LETOR files.
"""
import argparse
import json
import os
import random
import shutil
import subprocess
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)

try:
    from lr2ppo_amd import h5lite
    h5lite.library()
    HAVE_HDF5 = True
except ImportError:
    HAVE_HDF5 = False
needs_hdf5 = pytest.mark.skipif(not HAVE_HDF5, reason="no HDF5 C library on this host")


@needs_hdf5
def test_h5lite_round_trip_and_h5py_surface(tmp_path):
    """File / Group / Dataset behave like the h5py calls the reference makes: name-ordered keys, len, `in`, [name], [:], [()],
    integer / slice indices, KeyError for a missing member, errors for a missing or foreign file, read-only protection."""
    path = str(tmp_path / "a.h5")
    rng = np.random.RandomState(0)
    arrays = {"f4": rng.standard_normal((3, 2, 4)).astype(np.float32), "f8": rng.standard_normal((20, 48)),
              "i8": rng.randint(-5, 5, size=(7,)).astype(np.int64), "u1": rng.randint(0, 255, size=(2, 3, 5)).astype(np.uint8),
              "i4": np.arange(6, dtype=np.int32).reshape(2, 3), "empty": np.zeros((0, 4), np.float32),
              "scalar": np.float64(2.5)}
    with h5lite.File(path, "w") as f:
        g = f.create_group("12")
        for k, v in arrays.items():
            g.create_dataset(k, data=v)
        for name in ("10002", "7", "345"):
            f.create_dataset(name, data=np.full((2, 2), float(name)))
        with pytest.raises(ValueError):
            f.create_dataset("7", data=np.zeros(1))                     # exists already
    f = h5lite.File(path)                                                # default mode 'r'
    assert f.keys() == ["10002", "12", "345", "7"] == list(f) and len(f) == 4
    assert "12" in f and "12/f4" in f and "nope" not in f and "12/nope" not in f and "nope/x" not in f
    g = f["12"]
    assert sorted(g.keys()) == sorted(arrays) and len(g) == len(arrays)
    for k, v in arrays.items():
        d = g[k]
        assert d.shape == np.shape(v) and d.dtype == np.asarray(v).dtype, k
        assert np.array_equal(d[()], v) and np.array_equal(np.asarray(d), v), k
        if d.shape:
            assert np.array_equal(d[:], v) and np.array_equal(d[...], v) and len(d) == v.shape[0], k
    assert np.array_equal(g["f4"][:][0], arrays["f4"][0]) and np.array_equal(g["f8"][3:5, 2:], arrays["f8"][3:5, 2:])
    assert np.array_equal(f["12/i8"][1], arrays["i8"][1]) and float(g["scalar"][()]) == 2.5
    assert float(f["7"][()][0, 0]) == 7.0
    with pytest.raises(KeyError):
        f["nope"]
    with pytest.raises(ValueError):
        f.create_dataset("new", data=np.zeros(2))                        # opened read-only
    f.close()
    with pytest.raises(FileNotFoundError):
        h5lite.File(str(tmp_path / "missing.h5"))
    (tmp_path / "text.h5").write_text("not an hdf5 file")
    with pytest.raises(OSError):
        h5lite.File(str(tmp_path / "text.h5"))
    with pytest.raises(TypeError):
        with h5lite.File(str(tmp_path / "s.h5"), "w") as w:
            w.create_dataset("s", data=np.array(["a", "b"]))             # numeric arrays only


@needs_hdf5
def test_h5lite_files_are_valid_hdf5_by_the_librarys_own_dump_tool(tmp_path):
    """The HDF5 distribution's `h5dump` (an independent reader of the format) lists the same tree, types, shapes and values."""
    tool = shutil.which("h5dump") or os.path.join(os.path.dirname(os.path.dirname(h5lite.library()[0])), "bin", "h5dump")
    if not os.path.exists(tool):
        pytest.skip("h5dump not installed")
    path = str(tmp_path / "clean_feat.h5")
    with h5lite.File(path, "w") as f:
        g = f.create_group("41")
        g.create_dataset("text_emb", data=np.arange(24, dtype=np.float32).reshape(3, 2, 4))
        g.create_dataset("img_emb", data=np.arange(10, dtype=np.float32).reshape(1, 5, 2))
        f.create_dataset("7", data=np.arange(6, dtype=np.float64).reshape(2, 3))
    head = subprocess.run([tool, "-H", path], capture_output=True, text=True, check=True).stdout
    for want in ('GROUP "41"', 'DATASET "text_emb"', "H5T_IEEE_F32LE", "( 3, 2, 4 )", 'DATASET "img_emb"', "( 1, 5, 2 )",
                 'DATASET "7"', "H5T_IEEE_F64LE", "( 2, 3 )"):
        assert want in head, (want, head)
    data = subprocess.run([tool, "-d", "/41/img_emb", path], capture_output=True, text=True, check=True).stdout
    assert "(0,0,0): 0, 1," in data and "(0,4,0): 8, 9" in data, data


@needs_hdf5
def test_h5lite_reads_chunked_and_gzip_compressed_datasets(tmp_path):
    """Files made elsewhere may store features chunked / deflated (h5py's compression="gzip"): the library decodes them on read;
    h5dump confirms the storage really is chunked + deflated."""
    path = str(tmp_path / "z.h5")
    rng = np.random.RandomState(1)
    text = np.round(rng.standard_normal((5, 196, 768)), 1).astype(np.float32)       # compressible
    table = rng.standard_normal((20, 138))
    with h5lite.File(path, "w") as f:
        g = f.create_group("3")
        g.create_dataset("text_emb", data=text, compression="gzip")
        g.create_dataset("img_emb", data=text[:1, :16], chunks=(1, 4, 768))
        f.create_dataset("77", data=table, chunks=True, compression="gzip", compression_opts=9)
    assert os.path.getsize(path) < 0.6 * text.nbytes
    with h5lite.File(path) as f:
        assert np.array_equal(f["3"]["text_emb"][:], text) and np.array_equal(f["3"]["img_emb"][:][0], text[0, :16])
        assert np.array_equal(f["77"][()], table)
    tool = shutil.which("h5dump") or os.path.join(os.path.dirname(os.path.dirname(h5lite.library()[0])), "bin", "h5dump")
    if os.path.exists(tool):
        props = subprocess.run([tool, "-p", "-H", path], capture_output=True, text=True, check=True).stdout
        assert "CHUNKED" in props and "COMPRESSION DEFLATE { LEVEL 9 }" in props and "COMPRESSION DEFLATE { LEVEL 4 }" in props, props


def _write_movienet(root):
    from oracle import lr2ppo_oracle as O
    items, h5 = O.fake_movienet()
    os.makedirs(os.path.join(root, "LRMovieNet"))
    with h5lite.File(os.path.join(root, "LRMovieNet", "clean_feat.h5"), "w") as f:
        for key, members in h5.items():
            g = f.create_group(key)
            for k, v in members.items():
                g.create_dataset(k, data=v)
    with open(os.path.join(root, "split.json"), "w") as f:
        json.dump(items, f)


@needs_hdf5
def test_movienet_readers_match_reference_on_a_real_hdf5_file(tmp_path, monkeypatch):
    """The three LRMovieNet readers (stage 1, 2, 3; both splits) open LRMovieNet/clean_feat.h5 -- a real HDF5 file, no stand-in
    for h5py -- and return the reference readers' items (tests/golden/readers.json; the generator re-ran the reference's own
    readers on the same real file and got the same items)."""
    from oracle import lr2ppo_oracle as O
    from lr2ppo_amd.finetune import pointwise as pw, ppo, reward_pair_dataloader as rp
    if "h5py" in sys.modules and not hasattr(sys.modules["h5py"], "__file__"):
        monkeypatch.delitem(sys.modules, "h5py")
    _write_movienet(str(tmp_path))
    monkeypatch.chdir(tmp_path)                                            # the readers open the file relative to the cwd, as upstream
    with open(os.path.join(GOLD, "readers.json")) as f:
        gold = json.load(f)
    for name, mod in (("ppo", ppo), ("pointwise", pw), ("reward_pair", rp)):
        for split in ("train", "val"):
            g = gold[f"{name}_{split}"]
            random.seed(11), np.random.seed(12), torch.manual_seed(13)
            ds = mod.MovieNet(argparse.Namespace(is_master=False, max_imgs=16, max_tags=g["max_tags"]), "split.json",
                              is_train=split == "train")
            assert type(ds.embed_data).__module__.split(".")[0] in ("lr2ppo_amd", "h5py")
            torch.manual_seed(14)
            assert len(ds) == g["len"], (name, split)
            for i, want in enumerate(g["items"]):
                assert O.describe_reader_item(ds[i]) == want, (name, split, i)
    # through the loader the launchers build (forked workers read the file): shapes of a stage-3 batch
    ds = ppo.MovieNet(argparse.Namespace(is_master=False, max_imgs=16, max_tags=2), "split.json", is_train=True)
    loader = torch.utils.data.DataLoader(ds, batch_size=2, num_workers=2)
    text, img, tgt = next(iter(loader))
    assert text.shape == (2, 2, 2, 4) and img.shape == (2, 16, 768) and tgt.shape == (2, 2) and text.dtype == torch.float32


@needs_hdf5
def test_letor_readers_match_the_references_ltrdataset_classes(tmp_path):
    """`LTRDataset` of pointwise_trad / pointwise_2data_trad / ppo_trad / reward_trad on train.h5 / test.h5 in the layout
    datasets_trad/convert_to_h5py.py writes, seeded like the generator: query order, drawn pairs, index layouts, label and
    feature rows, shapes and dtypes equal the reference classes' (tests/golden/letor_readers.json)."""
    from oracle import lr2ppo_oracle as O
    from lr2ppo_amd.finetune import letor, pointwise_2data_trad, pointwise_trad, ppo_trad, reward_trad
    root = str(tmp_path)
    letor.write_split(root, True, O.fake_letor(seed=5, n_queries=6))
    letor.write_split(root, False, O.fake_letor(seed=6, n_queries=4, feats=136))
    with open(os.path.join(GOLD, "letor_readers.json")) as f:
        gold = json.load(f)
    for name, mod in (("pointwise_trad", pointwise_trad), ("pointwise_2data_trad", pointwise_2data_trad), ("ppo_trad", ppo_trad),
                      ("reward_trad", reward_trad)):
        for split in ("train", "val"):
            g = gold[f"{name}_{split}"]
            random.seed(21), np.random.seed(22), torch.manual_seed(23)
            ds = mod.LTRDataset(argparse.Namespace(), root, is_train=split == "train", **g["kw"])
            assert len(ds) == g["len"], (name, split)
            for i, want in enumerate(g["items"]):
                assert O.describe_letor_item(ds[i]) == want, (name, split, i)
    assert gold["reward_trad_train"]["len"] < 6 * 4                        # the one-class query contributed no pair
    # collated like the launchers' loaders: ppo_trad training batches are [bs, 2] labels and [bs, 2, F] float64 features
    random.seed(1)
    loader = torch.utils.data.DataLoader(ppo_trad.LTRDataset(None, root, is_train=True, max_tags=2), batch_size=4, num_workers=2)
    gt, qid, feats = next(iter(loader))
    assert gt.shape == (4, 2) and feats.shape == (4, 2, 46) and feats.dtype == torch.float64 and len(qid) == 4
    gt, qid, feats = next(iter(torch.utils.data.DataLoader(pointwise_trad.LTRDataset(None, root, is_train=False), batch_size=2)))
    assert gt.shape == (2, 20) and feats.shape == (2, 20, 136) and list(qid) == ["10002", "18"]


@needs_hdf5
@pytest.mark.gpu
def test_ppo_trad_entry_point_trains_from_letor_h5_files(tmp_path):
    """`python -m lr2ppo_amd.finetune.ppo_trad --train_path DIR --dev_path DIR` (no --synthetic_items): one PPO cycle and one
    validation pass over real train.h5 / test.h5 files of 768-wide document features, checkpoint written, NDCG logged."""
    from oracle import lr2ppo_oracle as O
    from lr2ppo_amd.finetune import letor
    root = str(tmp_path)
    letor.write_split(root, True, O.fake_letor(seed=8, n_queries=3, feats=768))
    letor.write_split(root, False, O.fake_letor(seed=9, n_queries=3, feats=768))
    for split in ("train", "test"):                                        # labels of {0, 1, 2}, as the 3-class heads expect
        with h5lite.File(os.path.join(root, f"{split}.h5")) as f:
            tables = {k: f[k][()] for k in f.keys()}
        for t in tables.values():
            t[:, 0] = np.minimum(t[:, 0], 2)
        letor.write_split(root, split == "train", tables)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29683", PYTHONPATH=REPO)
    cmd = [sys.executable, "-m", "lr2ppo_amd.finetune.ppo_trad", "--config_path", "lr2ppo_amd/configs/roberta_base.json",
           "--vit_config_path", "lr2ppo_amd/configs/vit_base_16_224.json", "--train_path", root, "--dev_path", root, "--seq_length", "196",
           "--max_imgs", "16", "--visual_feat_dim", "768", "--learning_rate", "1e-4", "--batch_size", "4", "--mode", "reg", "--epochs_num", "2",
           "--critic_learning_rate", "1e-4", "--max_timesteps", "1", "--update_timesteps", "2", "--kl_div_loss_weight", "0.001",
           "--entropy_weight", "0.001", "--value_clip", "0.5", "--max_cycles", "1", "--output_model_path", os.path.join(root, "m.bin"),
           "--log_path", os.path.join(root, "log.txt")]
    r = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    log = open(os.path.join(root, "log.txt")).read()
    assert "The number of training instances: 60" in log and "NDCG@3" in log, log[-2000:]     # 3 queries x 20 pairs
    assert os.path.exists(os.path.join(root, "m.bin"))
    # finetune/ppo_eval_trad.py: the evaluation-only entry point loads that checkpoint (the whole ActorCritic, strict) and reports the
    # same validation NDCG the training run logged for it
    trained = [ln for ln in log.splitlines() if ln.startswith("NDCG@3=")][-1]
    cmd2 = [c if c != "lr2ppo_amd.finetune.ppo_trad" else "lr2ppo_amd.finetune.ppo_eval_trad" for c in cmd]
    cmd2[cmd2.index("--log_path") + 1] = os.path.join(root, "eval_log.txt")
    r2 = subprocess.run(cmd2 + ["--pretrained_model_path", os.path.join(root, "m.bin")], cwd=REPO, env=env, capture_output=True, text=True,
                        timeout=900)
    assert r2.returncode == 0, r2.stdout[-2000:] + r2.stderr[-4000:]
    elog = open(os.path.join(root, "eval_log.txt")).read()
    assert [ln for ln in elog.splitlines() if ln.startswith("NDCG@3=")][-1] == trained, (trained, elog[-1000:])


@needs_hdf5
def test_tsv_to_h5_conversion_resamples_every_query_to_20_rows(tmp_path):
    """tools/convert_to_h5py.py (datasets_trad/convert_to_h5py.py's job without h5py): queries of 7, 20 and 33 rows -> 20 rows each,
    drawn as sklearn.utils.resample(random_state=0) draws them, float64, named by query id; readable by LTRDataset."""
    import csv
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import convert_to_h5py as conv
    from sklearn.utils import resample
    rng = np.random.RandomState(7)
    src = tmp_path / "tsv"
    src.mkdir()
    sizes, rows, rid = {301: 7, 5: 20, 12000: 33}, [], 0
    for qid, n in sizes.items():
        for _ in range(n):
            rows.append([int(rng.randint(0, 3)), qid, float(rid)] + [round(float(v), 6) for v in rng.standard_normal(5)])
            rid += 1
    for name in ("train.tsv", "test.tsv"):
        with open(src / name, "w") as f:
            csv.writer(f, delimiter="\t").writerows(rows)
    (src / "notes.txt").write_text("ignored")
    out = conv.convert(str(src), str(tmp_path / "h5"))
    assert [os.path.basename(p) for p in out] == ["test.h5", "train.h5"]
    table = np.array(rows, dtype=np.float64)
    with h5lite.File(out[1]) as f:
        assert f.keys() == ["12000", "301", "5"]
        for qid, n in sizes.items():
            got, mine = f[str(qid)][()], table[table[:, 1] == qid]
            assert got.shape == (20, 8) and got.dtype == np.float64 and set(got[:, 1]) == {float(qid)}
            want = mine if n == 20 else resample(mine, replace=n < 20, n_samples=20, random_state=0)
            assert np.array_equal(got, want), qid
            if n > 20:
                assert len(set(got[:, 2])) == 20                         # without replacement
    from lr2ppo_amd.finetune import pointwise_trad
    ds = pointwise_trad.LTRDataset(None, str(tmp_path / "h5"), is_train=True)
    gt, qid, feats = ds[0]
    assert len(ds) == 3 and qid == "12000" and feats.shape == (20, 6)


def _letor_dirs(tmp_path, widths):
    from oracle import lr2ppo_oracle as O
    from lr2ppo_amd.finetune import letor
    dirs = []
    for k, width in enumerate(widths):
        d = str(tmp_path / f"set{k}")
        for is_train, seed in ((True, 30 + k), (False, 40 + k)):
            tables = O.fake_letor(seed=seed, n_queries=6, feats=width)
            for t in tables.values():
                t[:, 0] = np.minimum(t[:, 0], 2)
            letor.write_split(d, is_train, tables)
        dirs.append(d)
    return dirs


_TRAD_FLAGS = ["--config_path", "lr2ppo_amd/configs/roberta_base.json", "--vit_config_path", "lr2ppo_amd/configs/vit_base_16_224.json",
               "--seq_length", "196", "--max_imgs", "16", "--visual_feat_dim", "768", "--learning_rate", "1e-4", "--batch_size", "2",
               "--mode", "reg", "--epochs_num", "1", "--report_steps", "2"]


@needs_hdf5
@pytest.mark.gpu
@pytest.mark.parametrize("twin", ["pointwise_trad", "pointwise_2data_trad"])
def test_pointwise_trad_entry_points_train_from_letor_h5_files(tmp_path, twin):
    """BASELINE configs[0] end to end: `python -m lr2ppo_amd.finetune.pointwise_trad` on train.h5 / test.h5 of 768-wide document
    features, and the two-data-set twin on a 46-wide (MQ2008) + a 136-wide (MSLR-WEB10K) set through its two projections: 3 batches
    of 2 queries, validation NDCG over the 6 test queries after batch 2, best checkpoint written with the twin's own keys."""
    two = twin.endswith("2data_trad")
    dirs = _letor_dirs(tmp_path, (46, 136) if two else (768,))
    out, log = str(tmp_path / "m.bin"), str(tmp_path / "log.txt")
    cmd = [sys.executable, "-m", f"lr2ppo_amd.finetune.{twin}", *_TRAD_FLAGS, "--train_path", dirs[0], "--dev_path", dirs[0],
           "--output_model_path", out, "--log_path", log] + (["--train_path2", dirs[1]] if two else [])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29684", PYTHONPATH=REPO)
    r = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    text = open(log).read()
    assert "The number of training instances: 6" in text and "Training steps: 2" in text and "NDCG@3=" in text, text[-2000:]
    keys = set(torch.load(out, map_location="cpu").keys())
    assert "out_layer.fc1.weight" in keys and ("text_proj3.fc1.weight" in keys) == two


@needs_hdf5
@pytest.mark.gpu
def test_reward_trad_entry_point_trains_from_letor_h5_files(tmp_path):
    """`python -m lr2ppo_amd.finetune.reward_trad` (stage 2 at sequence length 1) on real train.h5 / test.h5: LTRDataset's
    label-stratified pairs, hinge steps, validation accuracy logged, best checkpoint written."""
    (d,) = _letor_dirs(tmp_path, (768,))
    out, log = str(tmp_path / "r.bin"), str(tmp_path / "log.txt")
    flags = [f if f != "2" or _TRAD_FLAGS[i - 1] != "--batch_size" else "8" for i, f in enumerate(_TRAD_FLAGS)]
    cmd = [sys.executable, "-m", "lr2ppo_amd.finetune.reward_trad", *flags, "--train_path", d, "--dev_path", d,
           "--output_model_path", out, "--log_path", log]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29685", PYTHONPATH=REPO)
    r = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    text = open(log).read()
    assert "The number of training instances: 100" in text and "Training steps: 2" in text and "val accuracy:" in text, text[-2000:]
    assert "pos_emb.weight" in torch.load(out, map_location="cpu")


@pytest.mark.gpu
def test_dimension_projection_entry_point_rewrites_tsv_files(tmp_path):
    """`python -m lr2ppo_amd.finetune.pointwise_2data_infer_trad` (the step between the 46- / 136-wide LETOR rows and the 768-wide
    files the other `_trad` twins train on; finetune/pointwise_2data_infer_trad.py:409-447): every row's features go through the
    checkpoint's text_proj / text_proj3, label and query id are copied as text, row order kept; values against the same Mlp in fp64."""
    import csv
    from lr2ppo_amd.finetune import pointwise_2data_trad as p2, ppo
    torch.manual_seed(3)
    model = p2.Classifier(argparse.Namespace(mode="reg", labels_num=3), None)
    ppo._init_normal(model)
    with torch.no_grad():
        for q in model.parameters():
            q.mul_(3.0)                                                    # activations of order 1: GELU is exercised off zero
    ckpt = str(tmp_path / "proj.bin")
    torch.save(model.state_dict(), ckpt)
    rng = np.random.RandomState(4)
    src, dst = tmp_path / "in", tmp_path / "out"
    src.mkdir()
    tables = {"train.tsv": rng.standard_normal((37, 46)), "test.tsv": rng.standard_normal((21, 136)),
              "mixed.tsv": None}                                          # one file with both widths, interleaved
    rows_of = {}
    for name, feats in tables.items():
        rows = []
        if feats is None:
            for i in range(10):
                w = 46 if i % 3 else 136
                rows.append([str(i % 3), str(900 + i)] + [repr(float(v)) for v in rng.standard_normal(w)])
        else:
            for i, f in enumerate(feats):
                rows.append([str(int(rng.randint(0, 3))), str(100 + i // 5)] + [repr(float(v)) for v in f])
        rows_of[name] = rows
        with open(src / name, "w") as f:
            csv.writer(f, delimiter="\t").writerows(rows)
    env = dict(os.environ, PYTHONPATH=REPO)
    r = subprocess.run([sys.executable, "-m", "lr2ppo_amd.finetune.pointwise_2data_infer_trad", "--dim_proj_ckpt_path", ckpt,
                        "--input_dir", str(src), "--output_dir", str(dst), "--train_path", "x", "--dev_path", "x"], cwd=REPO, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    sd = {k: v.double() for k, v in model.state_dict().items()}
    for name, rows in rows_of.items():
        with open(dst / name) as f:
            got = list(csv.reader(f, delimiter="\t"))
        assert len(got) == len(rows) and all(len(g) == 2 + 768 for g in got), name
        for g, row in zip(got, rows):
            assert g[:2] == row[:2]
            x = torch.tensor([float(v) for v in row[2:]], dtype=torch.float64)
            pre = "text_proj" if x.numel() == 46 else "text_proj3"
            h = torch.nn.functional.gelu(sd[pre + ".fc1.weight"] @ x + sd[pre + ".fc1.bias"])
            want = sd[pre + ".fc2.weight"] @ h + sd[pre + ".fc2.bias"]
            have = torch.tensor([float(v) for v in g[2:]], dtype=torch.float64)
            assert ((have - want).norm() / want.norm()).item() < 1e-4, (name, g[:2])     # split-bf16 x 3 products: ~1e-5


@needs_hdf5
@pytest.mark.gpu
def test_ppo_eval_scores_a_real_lrmovienet_split_and_dumps_the_cases(tmp_path, monkeypatch):
    """finetune/ppo_eval.py: `evaluate` over LRMovieNet/clean_feat.h5 (real HDF5) + a split json with the clips' records -- NDCG against
    the oracle's CPU scores with the same weights (north_star: +-0.002), and case/ppo_cases.json: the record fields as the DataLoader
    collates them, gold labels, the per-clip NDCG vector, tags in the order of the predicted scores."""
    from oracle import lr2ppo_oracle as O
    from lr2ppo_amd.finetune import ppo, ppo_eval
    rng = np.random.RandomState(2)
    root = tmp_path
    (root / "LRMovieNet").mkdir()
    clips = []
    with h5lite.File(str(root / "LRMovieNet" / "clean_feat.h5"), "w") as f:
        for i, n_tags in enumerate((4, 6, 3)):
            cid = f"tt{1000 + i}_{i:04d}"
            g = f.create_group(cid)
            g.create_dataset("text_emb", data=rng.standard_normal((n_tags, 196, 768)).astype(np.float32))
            frame = rng.standard_normal((1, 1, 768)).astype(np.float32)        # 5 equal frames: the reader's shuffle and cyclic padding
            g.create_dataset("img_emb", data=np.repeat(frame, 5, axis=1))      # (checked elsewhere) cannot change what the actor sees
            clips.append({"id": cid, "filename": f"clip_{i}.mp4", "description": f"scene {i}",
                          "tags": [{"tag": f"tag{i}_{t}", "target": int(rng.randint(0, 3))} for t in range(n_tags)]})
    (root / "dev.json").write_text(json.dumps(clips))
    monkeypatch.chdir(root)
    dev = torch.device("cuda", 0)
    args = argparse.Namespace(mode="reg", labels_num=3, seq_length=196, max_imgs=16, visual_feat_dim=768, is_master=True, device=dev,
                              max_tags=32, batch_size=1, num_workers=0)
    model = ppo.ActorCritic(args, None)
    P = O.seeded_params(O.head_param_spec("actor"), seed=7)
    model.actor.load_state_dict(P, strict=True)
    args.model = model.to(dev)
    ds = ppo_eval.MovieNet(args, "dev.json")
    val = ppo_eval.evaluate(args, ppo_eval.get_dataloader(args, ds, 1, 0), 0, split="val", num_tasks=1)
    cases = json.load(open(root / "case" / "ppo_cases.json"))
    assert len(cases) == 3
    rows = []
    for clip, case in zip(clips, cases):
        text, img, tgts, _ = ds[clips.index(clip)]
        n = len(clip["tags"])
        with torch.no_grad():
            want = O.actor_forward(P, text.unsqueeze(0), img.unsqueeze(0).unsqueeze(1).repeat(1, n, 1, 1), None).view(-1)
        assert case["filename"] == [clip["filename"]] and case["id"] == [clip["id"]] and case["description"] == [clip["description"]]
        assert case["tags"] == [{"tag": [t["tag"]], "target": t["target"]} for t in clip["tags"]]
        got_scores = torch.tensor([v for _, v in case["predict"]])
        assert all(a >= b for a, b in zip(got_scores[:-1], got_scores[1:]))                      # sorted by predicted score
        by_tag = {t["tag"][0]: v for t, v in case["predict"]}
        have = torch.tensor([by_tag[t["tag"]] for t in clip["tags"]])
        assert (have - want).abs().max().item() < 1e-3 * max(1.0, want.abs().max().item())     # the 1e-3 bar on the scores
        assert sorted(t["tag"][0] for t, _ in case["predict"]) == sorted(t["tag"] for t in clip["tags"])
        ref_row = O.ndcg_vector(want, tgts)
        rows.append(ref_row)
        assert max(abs(a - float(b)) for a, b in zip(case["ndcg"], ref_row)) <= 0.002 or (want.sort().values.diff().abs().min() < 1e-3)
    assert abs(float(val) - float(torch.stack(rows).mean(0)[5])) <= 0.002
