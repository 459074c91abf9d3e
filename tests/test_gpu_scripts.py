"""End-to-end runs of this build's two training scripts on a real MI355X with small synthetic
files of the reference's input schemas (embedding JSON, abalone CSV)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPTS = os.path.join(ROOT, "mui-deepautoencoder_amd", "script")


def _run(cmd, cwd):
    r = subprocess.run([sys.executable] + cmd, cwd=cwd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout


@pytest.mark.parametrize("precision,E,z", [("f32", 16, 48), ("bf16", 64, 192), ("bf16", 64, 64)])
def test_embedding_script_end_to_end(tmp_path, precision, E, z):
    import yaml
    rng = np.random.default_rng(1)
    cats = ["top", "bottom", "shoe"]
    centers = rng.standard_normal((5, 3 * E)).astype(np.float32)
    emb = {}
    for i in range(360):
        v = centers[i % 5] + 0.1 * rng.standard_normal(3 * E).astype(np.float32)
        emb["o%04d" % i] = {c: v[s * E:(s + 1) * E].tolist() for s, c in enumerate(cats)}
    (tmp_path / "emb.json").write_text(json.dumps(emb))
    cfg = {"MODEL": {"Z_SIZE": z, "BATCH_SIZE": 64, "NB_INPUT_LAYER": 2, "NB_OUTPUT_LAYER": 2, "STEEP_LAYER_SIZE": False,
                     "EPOCH": 4, "LEARNING_RATE": 1e-3, "WEIGHT_DECAY": 1e-4, "NB_CORRUPTED": 1, "TRUNK_GRAD": True},
           "DATASET": {"NAME": "EMBEDDING", "USED_CATEGORY": cats, "EMBEDDING_SIZE": E, "SHUFFLE": True, "SPLIT": [0.7, 0.3]},
           "SEED": 27493045}
    (tmp_path / "cfg.yaml").write_text(yaml.safe_dump(cfg))
    os.makedirs(tmp_path / "log")
    out = _run([os.path.join(SCRIPTS, "train_dae_on_embedding.py"), "--embedding_path", "emb.json", "--output_path", "out",
                "--config", "cfg.yaml", "--precision", precision], cwd=str(tmp_path))
    assert "TRAINING HAS ENDED." in out and "VALIDATION RANKING ERROR" in out
    runs = os.listdir(tmp_path / "out")
    book = json.load(open(tmp_path / "out" / runs[0] / "book.json"))
    assert len(book["ftl"]) == 4 and all(np.isfinite(book[k]).all() for k in book)
    assert book["ftl"][-1] < book["ftl"][0], book["ftl"]            # it learns
    assert 0 <= book["rl"][-1] <= 1
    assert os.path.exists(tmp_path / "out" / runs[0] / "full_RMSE.png")


def test_abalone_script_end_to_end(tmp_path):
    import yaml
    rng = np.random.default_rng(3)
    lines = []
    for i in range(241):
        fl = rng.random(7) * np.array([0.8, 0.65, 0.3, 2.8, 1.5, 0.76, 1.0]) + 0.01
        lines.append(",".join(["MFI"[int(rng.integers(0, 3))]] + ["%.4f" % v for v in fl] + [str(int(rng.integers(1, 30)))]))
    os.makedirs(tmp_path / "data"); os.makedirs(tmp_path / "log")
    (tmp_path / "data" / "abalone.data").write_text("\n".join(lines) + "\n")
    cfg = yaml.safe_load(open(os.path.join(ROOT, "mui-deepautoencoder_amd", "config", "abalone.yaml")))
    cfg["MODEL"]["EPOCH"] = 2
    cfg["MODEL"]["LEARNING_RATE"] = 1e-3
    cfg["PLOT"] = {k: False for k in cfg["PLOT"]}
    (tmp_path / "cfg.yaml").write_text(yaml.safe_dump(cfg))
    out = _run([os.path.join(SCRIPTS, "train_dae_on_abalone.py"), "--dataset_path", "data", "--output_path", "out",
                "--config", "cfg.yaml", "--nb_missing", "2"], cwd=str(tmp_path))
    assert "VALIDATION PARTIAL ERROR" in out and "[9, 36]" in out
    runs = os.listdir(tmp_path / "out")
    book = json.load(open(tmp_path / "out" / runs[0] / "book.json"))
    assert np.asarray(book["ptl_per_k"]).shape == (2, 2, 9) and np.isfinite(np.asarray(book["pvl_per_k"])).all()
    assert book["ftl"][1] < book["ftl"][0]
