"""CPU-only checks: the C-ABI library loads and exports every symbol include/ofx.h declares, the
drop-in Python surface (import paths, configs, dispatch keys, state_dict key set, exceptions),
the collate processors against a reference-generated fixture, and the N>1 sharding logic on a
world_size-2 gloo group."""
import os
import pickle
import re
import socket
import subprocess
import sys
import warnings

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT, golden
from outfitx_amd import synth

warnings.simplefilter("ignore")


def test_library_exports_every_declared_symbol():
    from outfitx_amd import _lib
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "ofx.h")).read()
    declared = set(re.findall(r"\b(ofx_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"ofx_handle", "ofx_stream"}
    assert declared, "no declarations parsed"
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert lib.ofx_abi_version() == 5
    d = _lib.default_desc()
    assert (d.d_model, d.n_head, d.d_ffn, d.n_layers, d.vit_width, d.txt_width, d.proj_dim) == (1024, 16, 2024, 6, 768, 512, 512)


def test_no_silent_cpu_path():
    from outfitx_amd import _lib
    from outfitx_amd.engine import Engine, fitb_argmin
    with pytest.raises(_lib.OfxError):
        Engine(torch.device("cpu"))
    with pytest.raises(_lib.OfxError):
        fitb_argmin(torch.zeros(2, 8), torch.zeros(2, 4, 8))


def test_product_never_imports_the_oracle():
    for dp, _, fs in os.walk(os.path.join(ROOT, "outfitx_amd")):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b|import_module\(.oracle|np_oracle", src, re.M), f"{f} uses the oracle"


def test_configs_match_reference_fields_and_quirks():
    from src.models.configs import ItemEncoderConfig, OutfitXConfig
    c = OutfitXConfig()
    assert c.item_encoder.type == "slip" and c.d_embed == 1536 and c.model_name == "marqo-fashionSigLIP"
    c = OutfitXConfig(item_encoder=ItemEncoderConfig(type="clip"))
    assert (c.d_embed, c.max_length, c.padding, c.truncation, c.model_name) == (1024, 16, "max_length", True, "fashion-clip")
    t = c.transformer
    assert (t.n_head, t.d_ffn, t.n_layers, t.dropout) == (16, 2024, 6, 0.3)
    assert t.batch_first == (True,) and t.norm_first == (True,) and t.activation is torch.nn.functional.mish
    assert c.item_encoder.clip_model_name == "patrickjohncyh/fashion-clip" and c.item_encoder.dim_per_modality == 512
    with pytest.raises(ValueError):
        ItemEncoderConfig(type="nope")
    pickle.loads(pickle.dumps(c))


def test_model_surface_and_state_dict_keys():
    from src.models import OutfitX
    from src.models.configs import ItemEncoderConfig, OutfitXConfig
    from src.models.datatypes import (FashionItem, OutfitCompatibilityPredictionTask, OutfitComplementaryItemRetrievalTask,
                                      OutfitFillInTheBlankTask, OutfitPrecomputeEmbeddingTask)
    m = OutfitX(OutfitXConfig(item_encoder=ItemEncoderConfig(type="clip")))
    sd = m.state_dict()
    want = synth.outfit_transformer_shapes()
    want.update({synth.IMG_PREFIX + k: v for k, v in synth.vision_shapes().items()})
    want.update({synth.TXT_PREFIX + k: v for k, v in synth.text_shapes().items()})
    assert set(sd) == set(want) and len(sd) == 474
    assert all(tuple(sd[k].shape) == want[k] for k in want)
    assert sum(v.numel() for v in sd.values()) == 202_432_625
    assert sum(p.numel() for p in m.parameters() if p.requires_grad) == 51_155_313
    for attr in ("cfg", "device", "item_encoder", "transformer_encoder", "outfit_token", "cp_ffn", "cir_ffn", "target_item_image_emb", "forward_"):
        assert hasattr(m, attr)
    assert m.item_encoder.d_embed == 1024 and m.item_encoder.image_size == (224, 224)
    assert set(m.forward_) == {OutfitCompatibilityPredictionTask, OutfitComplementaryItemRetrievalTask, OutfitFillInTheBlankTask, OutfitPrecomputeEmbeddingTask}
    assert m.forward_[OutfitFillInTheBlankTask] == m.forward_[OutfitComplementaryItemRetrievalTask]
    with pytest.raises(KeyError):
        m(task=FashionItem)
    with pytest.raises(NotImplementedError):
        OutfitX()                                    # default type='slip' is outside the built path
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.outfit_transformer_weights(3).items()}, strict=False)
    pickle.loads(pickle.dumps(m.cfg))


def test_same_seed_gives_torch_default_init_of_the_reference_modules():
    """The outfit transformer's parameters come from torch's own module constructors, in the
    reference's construction order -> same RNG consumption as the reference for those modules."""
    from outfitx_amd.outfit_x import OutfitX
    from src.models.configs import ItemEncoderConfig, OutfitXConfig
    a = OutfitX(OutfitXConfig(item_encoder=ItemEncoderConfig(type="clip")))
    l0, l5 = a.transformer_encoder.layers[0], a.transformer_encoder.layers[5]
    assert torch.equal(l0.linear1.weight, l5.linear1.weight)          # nn.TransformerEncoder deep-copies one layer
    assert float(a.outfit_token.std()) < 0.03 and a.cp_ffn[1].weight.shape == (1, 1024)


def test_collate_processors_match_reference_fixture():
    from src.models.configs import ItemEncoderConfig, OutfitXConfig
    from src.models.datatypes import FashionItem, OutfitCompatibilityPredictionTask as CP, OutfitComplementaryItemRetrievalTask as CIR, OutfitFillInTheBlankTask as FITB, OutfitPrecomputeEmbeddingTask as PE
    from src.models.processor import OutfitXProcessorFactory as F
    cfg = OutfitXConfig(item_encoder=ItemEncoderConfig(type="clip"))
    g = golden("aux")
    lens = g["proc_lens"]
    rows = [synth.item_embeddings(1243, f"o{i}", int(n)) for i, n in enumerate(lens)]
    batch = [(CP(outfit=[FashionItem(item_id=j, embedding=r[j]) for j in range(len(r))]), float(i % 2)) for i, r in enumerate(rows)]
    proc = F.get_processor(CP, cfg)
    bd = pickle.loads(pickle.dumps(proc))(batch)                       # picklable like the reference's
    assert bd["input_dict"]["task"] is CP
    assert synth.checksum(bd["input_dict"]["outfit_embedding"].numpy()) == str(g["proc_emb_crc"])
    assert np.array_equal(bd["input_dict"]["outfit_mask"].numpy(), g["proc_mask"])
    assert np.array_equal(bd["label"].numpy(), g["proc_label"])
    # FITB / CIR / PE shapes and keys
    q = FITB(outfit=[FashionItem(embedding=rows[0][0])], target_item=FashionItem(item_id=9, embedding=rows[1][0], text_embedding=rows[1][0][512:]))
    fb = F.get_processor(FITB, cfg)([(q, torch.zeros(4, 1024), 2)])
    assert fb["input_dict"]["task"] is CIR and fb["candidate_item_embedding"].shape == (1, 4, 1024) and fb["answer_index"].tolist() == [2]
    assert fb["input_dict"]["target_item_text_embedding"].shape == (1, 512)
    with pytest.raises(ValueError):
        F.get_processor(CIR, cfg)
    cq = CIR(outfit=q.outfit, target_item=q.target_item)
    tr = F.get_processor(CIR, cfg, run_mode="train")([(cq, [rows[2][0], rows[2][1]])])
    assert tr["neg_items_embedding"].shape == (1, 16, 1024) and tr["neg_items_mask"].sum() == 14 and tr["pos_item_embedding"].shape == (1, 1024)
    te = F.get_processor(CIR, cfg, run_mode="test")([(cq, None)])
    assert te["pos_item_id"] == [9]
    pe = F.get_processor(PE, cfg)([PE(fashion_item=FashionItem(item_id=3, category="tops"))])
    assert pe["input_dict"]["texts"] == [["tops"]] and pe["item_id"] == [3]
    for t in (CP, FITB, PE):
        pickle.dumps(F.get_processor(t, cfg))


def test_clip_preprocess_matches_hf_processor():
    tf = pytest.importorskip("transformers")
    from PIL import Image
    from outfitx_amd.encoders import clip_preprocess
    g = np.random.default_rng(0)
    ims = [Image.fromarray(g.integers(0, 256, (h, w, 3), dtype=np.uint8)) for h, w in ((224, 224), (300, 260), (231, 517))]
    try:
        proc = tf.CLIPImageProcessor(do_convert_rgb=False)
        want = proc(images=ims, return_tensors="pt")["pixel_values"].numpy()
    except Exception as e:                                   # pragma: no cover
        pytest.skip(f"HF image processor unavailable: {e}")
    got = clip_preprocess(ims).numpy()
    assert got.shape == want.shape == (3, 3, 224, 224)
    assert np.abs(got - want).max() < 1e-5


def test_ragged_inputs_raise_value_error():
    from outfitx_amd.encoders import CLIPImageEncoder, CLIPTextEncoder, aggregate_embeddings
    enc = CLIPImageEncoder()
    with pytest.raises(ValueError):
        enc._pixels([[np.zeros((8, 8, 3), np.uint8)], []])
    with pytest.raises(ValueError):
        CLIPTextEncoder()._ids([["a"], []])
    with pytest.raises(ValueError):
        aggregate_embeddings(torch.zeros(1, 2), torch.zeros(1, 2), "sum")
    with pytest.raises(ValueError):
        aggregate_embeddings()


def test_shard_range_is_a_partition():
    from outfitx_amd.parallel import shard_range
    for n in (0, 1, 7, 256, 1000, 100_000):
        for w in (1, 2, 3, 8):
            cuts = [shard_range(n, r, w) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in cuts) - min(h - l for l, h in cuts) <= 1


_WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["OFX_ROOT"])
from outfitx_amd import synth
from outfitx_amd.parallel import shard_range, sharded_topk, gather_scores
from oracle import np_oracle as O
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
Q = (synth.item_embeddings(1241, "queries", 100) * 3.0).astype(np.float32)
P = synth.item_embeddings(1241, "pool", 5000)
lo, hi = shard_range(len(P), rank, world)
def local(Qt, Pt, k, base):          # CPU stand-ins for the two GPU kernels (checker code, test only)
    i, d = O.l2_topk(Qt.numpy(), Pt.numpy(), k)
    return torch.from_numpy(i + base), torch.from_numpy(d)
def merge(ii, dd):
    W, nq, k = ii.shape
    i = ii.permute(1, 0, 2).reshape(nq, W * k).numpy(); d = dd.permute(1, 0, 2).reshape(nq, W * k).numpy()
    order = np.lexsort((i, d), axis=-1)[:, :k]
    return torch.from_numpy(np.take_along_axis(i, order, -1)), torch.from_numpy(np.take_along_axis(d, order, -1))
idx, dst = sharded_topk(torch.from_numpy(Q), torch.from_numpy(P[lo:hi]), 50, lo, local, merge)
g = np.load(os.path.join(os.environ["OFX_GOLDEN"], "scoring.npz"))
assert np.array_equal(idx.numpy(), g["topk_idx"]), "sharded top-k differs from the reference's unsharded result"
# batch sharding + score gather (CP path): every rank scores its slice, scores come back in batch order
B = 37
s_lo, s_hi = shard_range(B, rank, world)
scores = torch.arange(B, dtype=torch.float32)[s_lo:s_hi].unsqueeze(1) * 2
allv = gather_scores(scores, B)
assert torch.equal(allv.flatten(), torch.arange(B, dtype=torch.float32) * 2)
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_pool_sharded_topk_and_score_gather_world2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, OFX_ROOT=ROOT, OFX_GOLDEN=GOLDEN, OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("ok") == 2


def test_indexed_processor_reproduces_the_reference_collate():
    """N3: the index-emitting collate + a row gather equals the reference processor's padded output (fixture from the
    reference itself: lengths 3, 8, 20 -> truncated to 16, 1)."""
    from outfitx_amd.configs import ItemEncoderConfig, OutfitXConfig
    from outfitx_amd.datatypes import FashionItem, OutfitCompatibilityPredictionTask as CP
    from outfitx_amd.embedding_store import EmbeddingTable
    from outfitx_amd.processor import OutfitXIndexedProcessor
    g = golden("aux")
    lens = g["proc_lens"].tolist()
    rows = [synth.item_embeddings(1243, f"o{i}", n) for i, n in enumerate(lens)]
    ids = np.arange(1000, 1000 + sum(lens))[::-1].copy()                 # arbitrary, non-sorted item ids
    table = EmbeddingTable(ids, np.concatenate(rows))
    off = np.concatenate([[0], np.cumsum(lens)])
    batch = [(CP(outfit=[FashionItem(item_id=int(ids[off[i] + j])) for j in range(n)]), float(i % 2)) for i, n in enumerate(lens)]
    proc = OutfitXIndexedProcessor(CP, OutfitXConfig(item_encoder=ItemEncoderConfig(type="clip")), id_to_row=table.index())
    proc = pickle.loads(pickle.dumps(proc))                              # collate_fn objects travel to DataLoader workers
    out = proc(batch)
    idx, cu = out["input_dict"]["item_index"].numpy(), out["input_dict"]["cu_seqlens"].numpy()
    assert idx.dtype == np.int32 and cu.tolist() == [0, 3, 11, 27, 28]
    emb = np.zeros((len(lens), 16, 1024), np.float32); mask = np.ones((len(lens), 16), bool)
    for b in range(len(lens)):
        n = cu[b + 1] - cu[b]
        emb[b, :n] = table.embeddings[idx[cu[b]:cu[b + 1]]]; mask[b, :n] = False
    assert synth.checksum(emb) == str(g["proc_emb_crc"]) and np.array_equal(mask, g["proc_mask"])
    assert np.array_equal(out["label"].numpy(), g["proc_label"])


def test_embedding_store_roundtrip(tmp_path):
    from outfitx_amd import embedding_store as S
    g = np.random.default_rng(3)
    ids0, ids1 = [5, 9, 2], [100, 7]
    e0, e1 = g.standard_normal((3, 1024)).astype(np.float32), g.standard_normal((2, 1024)).astype(np.float32)
    S.save_pickle_shard(str(tmp_path), "fashion-clip", 1, ids1, e1)
    p0 = S.save_pickle_shard(str(tmp_path), "fashion-clip", 0, ids0, e0)
    assert os.path.basename(p0) == "fashion-clip_embedding_subset_0.pkl"          # the reference's file name
    with open(p0, "rb") as f:
        d = pickle.load(f)
    assert set(d) == {"ids", "embeddings"} and d["embeddings"].dtype == np.float32      # and its layout
    ids, emb = S.load_pickle_shards(str(tmp_path), "fashion-clip")
    assert ids.tolist() == ids0 + ids1 and np.array_equal(emb, np.concatenate([e0, e1]))
    S.convert_to_mmap(str(tmp_path), "fashion-clip")
    t = S.EmbeddingTable.open(str(tmp_path), "fashion-clip")
    assert isinstance(t.embeddings, np.memmap)
    assert np.array_equal(t.gather([7, 5, 7, 2]), np.stack([e1[1], e0[0], e1[1], e0[2]]))
    with pytest.raises(KeyError):
        t.gather([12345])
    with pytest.raises(FileNotFoundError):
        S.load_pickle_shards(str(tmp_path), "other-model")
