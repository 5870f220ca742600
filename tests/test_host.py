"""CPU: host logic and the C-ABI surface (no compute calls -- there is no GPU here)."""
import ctypes
import os
import re
import subprocess
import sys

import pytest
import torch

from vae_tagger_amd import _lib, sharding, synth
from vae_tagger_amd.modules import AspectRatioBucketing, get_vae_latent_info

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_header_symbol():
    hdr = open(os.path.join(ROOT, "include", "vae_tagger_hip.h")).read()
    declared = set(re.findall(r"\b(vt_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    assert b"gfx950" in _lib.load().vt_version()


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libvae_tagger_hip.so")
    with pytest.raises(_lib.VTError, match="no CPU fallback"):
        _lib.load()


def test_models_refuse_to_run_on_cpu():
    from vae_tagger_amd.diffusers_vae_loader import DiffusersVAEWrapper, get_diffusers_vae_config, load_diffusers_vae_from_config
    cfg = dict(get_diffusers_vae_config(), block_out_channels=[64, 64], down_block_types=["DownEncoderBlock2D"] * 2)
    vae = DiffusersVAEWrapper(load_diffusers_vae_from_config(cfg))
    with pytest.raises(_lib.VTError, match="no CPU fallback"):
        vae.encode(torch.zeros(1, 3, 32, 32))


def test_state_dict_surface_matches_reference_keys():
    from vae_tagger_amd.autoencoder_kl import AutoencoderKL
    from vae_tagger_amd.modules import create_attention_decoder
    vae = AutoencoderKL(block_out_channels=(128, 256, 512, 512), latent_channels=16, use_quant_conv=False,
                        scaling_factor=0.3611, shift_factor=0.1159)
    assert set(vae.state_dict()) == set(synth.encoder_manifest())
    assert sum(p.numel() for p in vae.parameters()) == 34_274_208
    sd = synth.synth_state_dict(synth.encoder_manifest(), seed=5)
    sd["decoder.conv_in.weight"] = torch.zeros(1)           # real checkpoints carry decoder.* too
    del sd["encoder.conv_out.bias"]
    missing, unexpected = vae.load_state_dict(sd, strict=False)
    assert missing == ["encoder.conv_out.bias"] and unexpected == ["decoder.conv_in.weight"]
    assert hasattr(vae.config, "scaling_factor") and hasattr(vae.config, "shift_factor")
    dec = create_attention_decoder(16, 128, 128, 11, {"use_spatial_attention": True, "use_self_attention": True})
    assert set(dec.state_dict()) == set(synth.attention_decoder_manifest(11))
    assert sum(p.numel() for p in dec.parameters()) == 1_189_493


def test_bucketing_matches_survey():
    b = AspectRatioBucketing(512, 1024, 64)
    assert len(b.buckets) == 81
    assert b.bucket_for_ratio(1.0) == (512, 512)              # ties go to the smallest bucket
    reach = {b.bucket_for_ratio(w / h) for w in range(256, 2049, 8) for h in range(256, 2049, 8)}
    assert all(w % 64 == 0 and h % 64 == 0 for w, h in reach)
    assert get_vae_latent_info(1024)["latent_height"] == 128


def test_input_side_logic_reproduces_the_reference(tmp_path):
    """tests/golden/input_side.npz holds what the reference's own AspectRatioBucketing / SmartResize (modules.py:142-222)
    returned in the build container (oracle/make_goldens.py input_side): bucket list, assign_bucket over 784 image sizes,
    297 centre-crop boxes.  Integer / index work: bit-exact."""
    import numpy as np
    from PIL import Image
    from oracle import resize_ref
    from vae_tagger_amd.modules import SmartResize, smart_crop_box
    with np.load(os.path.join(ROOT, "tests", "golden", "input_side.npz")) as z:
        g = {k: z[k] for k in z.files}
    b = AspectRatioBucketing(512, 1024, 64)
    assert np.array_equal(np.array(b.buckets, dtype=np.int32), g["buckets"])
    got = np.array([b.bucket_for_ratio(int(w) / int(h)) for w, h in g["sizes"]], dtype=np.int32)
    assert np.array_equal(got, g["assigned"])
    assert len({tuple(x) for x in got}) == 67                  # 67 of the 81 buckets are reachable
    for w, h in ((300, 1000), (1920, 1080), (777, 778)):      # through the file-opening entry point as well
        f = tmp_path / f"{w}x{h}.png"
        Image.new("1", (w, h)).save(f)
        k = int(np.nonzero((g["sizes"] == (w, h)).all(1))[0][0]) if ((g["sizes"] == (w, h)).all(1)).any() else None
        assert b.assign_bucket(str(f)) == (tuple(g["assigned"][k]) if k is not None else b.bucket_for_ratio(w / h))
    assert b.assign_bucket(str(tmp_path / "missing.png")) == tuple(g["missing_default"]) == (512, 512)

    class Rec:                                                 # records what SmartResize asks of the image
        def __init__(self, size, log):
            self.size, self.log = size, log

        def crop(self, box):
            self.log["box"] = (box[0], box[1], box[2] - box[0], box[3] - box[1])
            return Rec((box[2] - box[0], box[3] - box[1]), self.log)

        def resize(self, size, resample=None):
            self.log["resize"] = (size, resample)
            return self
    for (ow, oh, tw, th), box in zip(g["crop_cases"].tolist(), g["crop_boxes"].tolist()):
        want = (box[0], box[1], box[2] - box[0], box[3] - box[1])           # reference boxes are (l, t, r, b)
        assert smart_crop_box(ow, oh, tw, th) == want
        assert resize_ref.smart_crop_box(ow, oh, tw, th) == want
        log = {}
        SmartResize(tw, th)(Rec((ow, oh), log))
        assert log.get("box", (0, 0, ow, oh)) == want and log["resize"] == ((tw, th), Image.LANCZOS)


def test_shard_range_and_cost_model():
    for n in (1, 7, 16, 129):
        for world in (1, 2, 3, 8):
            spans = [sharding.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert abs(sharding.image_cost(1024, 1024) - 4.8826) < 1e-3
    assert abs(sharding.image_cost(512, 512) - 1.1176) < 2e-3
    batches = [(1024, 1024, 8), (512, 512, 8), (960, 1024, 8), (512, 768, 8), (768, 512, 8), (640, 640, 8)]
    assign, loads = sharding.assign_batches(batches, 4)
    assert sorted(i for a in assign for i in a) == list(range(len(batches)))
    assert max(loads) <= 1.05 * max(sharding.image_cost(w, h) * n for w, h, n in batches)


_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from vae_tagger_amd import sharding
from oracle import decoder_ref
from vae_tagger_amd import synth
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + sys.argv[2], rank=int(sys.argv[3]), world_size=2)
sd = synth.synth_state_dict(synth.attention_decoder_manifest(11), seed=1)
g = torch.Generator().manual_seed(0)
lat = torch.randn(5, 16, 16, 16, generator=g)            # 5 images over 2 ranks: ragged 3 + 2
full = decoder_ref.attention_decoder_forward(sd, lat)
out = sharding.sharded_logits(lambda x: decoder_ref.attention_decoder_forward(sd, x), lat)
assert out.shape == full.shape and torch.allclose(out, full, atol=1e-6), (out - full).abs().max()
lat4 = lat[:4]                                           # even split: single all-gather path
out4 = sharding.sharded_logits(lambda x: decoder_ref.attention_decoder_forward(sd, x), lat4)
assert torch.allclose(out4, full[:4], atol=1e-6)
out1 = sharding.sharded_logits(lambda x: decoder_ref.attention_decoder_forward(sd, x), lat[:1])   # B < world: rank 1 holds no image
assert out1.shape == (1, 11) and torch.allclose(out1, full[:1], atol=1e-6)
# bucketed plan (bench.py --bucketed): whole same-shape batches placed by the cost model, ragged gather of the per-rank
# logits -- including a step whose single batch leaves rank 1 with nothing
rank = int(sys.argv[3])
for batches in ([(1024, 1024, 2), (512, 512, 3), (768, 512, 1)], [(640, 640, 2)]):
    assign, _ = sharding.assign_batches(batches, 2)
    lats = [torch.randn(n, 16, h // 64, w // 64, generator=torch.Generator().manual_seed(w * 7 + h)) for (w, h, n) in batches]
    outs = [decoder_ref.attention_decoder_forward(sd, lats[i]) for i in assign[rank]]
    local = torch.cat(outs, 0) if outs else torch.empty(0, 11)
    counts = [sum(batches[i][2] for i in assign[r]) for r in range(2)]
    got = sharding.all_gather_logits(local, counts)
    want = torch.cat([decoder_ref.attention_decoder_forward(sd, lats[i]) for r in range(2) for i in assign[r]], 0)
    assert got.shape == want.shape and torch.allclose(got, want, atol=1e-6)
    assert min(counts) == 0 or len(batches) > 1
dist.barrier(); dist.destroy_process_group()
print("rank", sys.argv[3], "ok")
'''


def test_sharded_logits_world2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    port = str(29500 + os.getpid() % 2000)
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, port, str(r)], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o


def test_evaluation_metrics_match_scikit_learn():
    """vae_tagger_amd.evaluation computes its metrics from confusion counts and a rank-based AP; scikit-learn (what the
    reference calls per class, evaluation.py:58-72) is the checker."""
    import numpy as np
    from sklearn.metrics import average_precision_score, f1_score, precision_score, recall_score
    from vae_tagger_amd.evaluation import MultiLabelEvaluator
    rng = np.random.default_rng(0)
    n, c = 300, 7
    y_true = (rng.random((n, c)) < np.array([0.5, 0.1, 0.9, 0.3, 0.02, 0.6, 0.4])).astype(np.float32)
    y_prob = np.clip(0.55 * y_true + 0.6 * rng.random((n, c)), 0, 1).astype(np.float32)
    y_prob[:, 3] = np.round(y_prob[:, 3], 1)                 # many tied scores
    y_pred = (y_prob > 0.5).astype(np.float32)
    ev = MultiLabelEvaluator([f"t{i}" for i in range(c)], device="cpu")
    for lo in range(0, n, 64):                                # several batches
        ev.update(torch.from_numpy(y_pred[lo:lo + 64]), torch.from_numpy(y_true[lo:lo + 64]), torch.from_numpy(y_prob[lo:lo + 64]))
    m = ev.compute_metrics(0.5)
    for avg in ("micro", "macro", "weighted"):
        assert abs(m[f"precision_{avg}"] - precision_score(y_true, y_pred, average=avg, zero_division=0)) < 1e-9
        assert abs(m[f"recall_{avg}"] - recall_score(y_true, y_pred, average=avg, zero_division=0)) < 1e-9
        assert abs(m[f"f1_{avg}"] - f1_score(y_true, y_pred, average=avg, zero_division=0)) < 1e-9
    assert abs(m["mAP"] - average_precision_score(y_true, y_prob, average="macro")) < 1e-6
    assert abs(m["mAP_micro"] - average_precision_score(y_true, y_prob, average="micro")) < 1e-6
    assert abs(m["mAP_weighted"] - average_precision_score(y_true, y_prob, average="weighted")) < 1e-6
    assert abs(m["accuracy"] - (y_true == y_pred).all(1).mean()) < 1e-12
    assert abs(m["hamming_loss"] - (y_true != y_pred).mean()) < 1e-12
    for i in range(c):
        pc = m["per_class"][f"t{i}"]
        assert abs(pc["ap"] - average_precision_score(y_true[:, i], y_prob[:, i])) < 1e-6
        assert abs(pc["f1"] - f1_score(y_true[:, i], y_pred[:, i], zero_division=0)) < 1e-9
        assert pc["support"] == int(y_true[:, i].sum())
    # the reference's conventions for degenerate classes: no positive sample -> zeros; every sample positive -> AP 1
    yt = y_true.copy(); yt[:, 0] = 0; yt[:, 1] = 1
    ev.reset_metrics(); ev.update(y_pred, yt, y_prob)
    m2 = ev.compute_metrics(0.5)
    assert m2["per_class"]["t0"] == {"precision": 0.0, "recall": 0.0, "f1": 0.0, "ap": 0.0, "support": 0}
    assert m2["per_class"]["t1"]["ap"] == 1.0 and m2["per_class"]["t1"]["recall"] == 1.0
    # a class without positives: scikit-learn (>= 1.2, the reference's pin) warns, scores it 0 and still averages
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for avg, key in (("macro", "mAP"), ("micro", "mAP_micro"), ("weighted", "mAP_weighted")):
            assert abs(m2[key] - average_precision_score(yt, y_prob, average=avg)) < 1e-6, key
    assert m2["mAP"] > 0.0


def test_resize_tables_match_the_oracle():
    """vt_resize_table (host code of the C ABI) builds Pillow's per-axis tables; the numpy oracle (pinned against Pillow in
    tests/test_oracle.py) is the checker.  No GPU involved."""
    import ctypes
    import numpy as np
    from oracle import resize_ref as R
    from vae_tagger_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(1)
    cases = [(4000, 1024, 1), (3000, 1024, 0), (300, 1024, 1), (1024, 1024, 0), (7, 3, 1), (1, 5, 0)]
    cases += [(int(rng.integers(1, 2500)), int(rng.integers(1, 1100)), int(rng.integers(0, 2))) for _ in range(30)]
    for n_in, n_out, kind in cases:
        ks = lib.vt_resize_table(n_in, n_out, kind, None, 0)
        buf = (ctypes.c_int * (n_out * (2 + ks)))()
        assert lib.vt_resize_table(n_in, n_out, kind, buf, len(buf)) == ks
        tab = np.frombuffer(buf, dtype=np.int32).reshape(n_out, 2 + ks)
        bounds, kk = R.coefficients(n_in, n_out, kind)
        assert kk.shape[1] == ks and np.array_equal(tab[:, :2], bounds) and np.array_equal(tab[:, 2:], kk), (n_in, n_out, kind)
    assert lib.vt_resize_table(0, 5, 0, None, 0) == -1 and lib.vt_resize_table(5, 5, 2, None, 0) == -1


# ---- bench.py's socket-power reader (hwmon): picks the card whose power rose, survives absent files ---------------------------------------------
def test_bench_socket_power_picks_the_loaded_card_and_tolerates_missing_files(tmp_path, monkeypatch):
    import glob as _glob
    import importlib.util
    import time
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    dirs = []
    for i, (idle, busy) in enumerate(((250_000_000, 251_000_000), (245_000_000, 1_360_000_000), (None, None))):
        d = tmp_path / f"card{i}" / "device" / "hwmon" / f"hwmon{i}"
        d.mkdir(parents=True)
        if idle is not None:
            (d / "power1_input").write_text(str(idle))
            (d / "freq1_input").write_text("150000000")
            (d / "power1_cap").write_text("1400000000")
        dirs.append((d, busy))
    monkeypatch.setattr(_glob, "glob", lambda pat: sorted(str(d) for d, _ in dirs))
    p = bench.SocketPower()
    assert p.idle == [250_000_000, 245_000_000, None]
    for d, busy in dirs:                                   # "the bench starts": card 1 goes to 1360 W at 1.9 GHz
        if busy is not None:
            (d / "power1_input").write_text(str(busy))
            (d / "freq1_input").write_text("1900000000" if busy > 1e9 else "150000000")
    with p:
        time.sleep(0.5)
    s = p.summary()
    assert s is not None and s["socket_w_median"] == 1360.0 and s["cap_w"] == 1400.0 and s["smu_sclk_mhz_median"] == 1900 and s["samples"] >= 3
    # no hwmon tree at all (a box that hides it): the bench line carries "power": null
    monkeypatch.setattr(_glob, "glob", lambda pat: [])
    q = bench.SocketPower()
    with q:
        time.sleep(0.2)
    assert q.summary() is None


_GATHER_WORKER = '''
import os, sys
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[2], RANK=sys.argv[3], WORLD_SIZE="2")
dist.init_process_group("gloo")
from vae_tagger_amd import infer_full, sharding
rank = int(sys.argv[3])
paths = [f"/data/img{i:02d}.png" for i in range(7)]
lo, hi = sharding.shard_range(len(paths), rank, 2)
items = [(p, {"max_confidence": float(i)}) for i, p in enumerate(paths)][lo:hi]
if rank == 1:
    items = items[:-1]                                   # rank 1 lost its last image
merged, processed, errors = infer_full.gather_results(items, len(items), 1 if rank == 1 else 0, 2, rank)
if rank == 0:
    assert [p for p, _ in merged] == paths[:6] and processed == 6 and errors == 1, (merged, processed, errors)
else:
    assert merged is None and processed == 0 and errors == 0
dist.barrier(); dist.destroy_process_group()
'''


def test_cli_result_gather_world2_gloo(tmp_path):
    """The sharded CLIs' only exchange: rank 0 receives every rank's finished entries and counters in rank (= path) order."""
    script = tmp_path / "gather_worker.py"
    script.write_text(_GATHER_WORKER)
    port = str(31500 + os.getpid() % 2000)
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, port, str(r)], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
