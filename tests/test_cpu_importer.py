"""The `.npy` parameter-tree importer against the reference's on-disk format (python/save.py,
src/model/**/load.rs): a tree written in that format from seeded weights must come back as exactly the flat buffer the C ABI
takes, with the configs recovered from the tree itself.  Host-only."""
import importlib
import os

import numpy as np
import pytest

from oracle import clip as OCL, config as OC
from util import to_pkg_cfg, to_pkg_vcfg

KIND = {"Conv": 0, "Res": 1, "Down": 2, "ResT": 3, "ResTU": 4, "ResU": 5}


@pytest.fixture(scope="module")
def imp(pkg):
    if not os.path.exists(pkg.LIB_PATH):
        import __graft_entry__ as ge
        ge.build()
    return importlib.import_module(pkg.__name__ + ".importer")


def _dump_unet(imp, ocfg, root, seed):
    specs = OC.unet_param_specs(ocfg)
    W = OC.synth_weights(specs, seed)
    imp.export_tree(specs, W, root, "unet")
    inp, mid, out = OC.unet_block_plan(ocfg)
    blocks = lambda bl: [(KIND[b["kind"]], b.get("depth", 0), b.get("n_head", 0)) for b in bl]   # noqa: E731
    imp.export_unet_structure(root, ocfg.model_channels, blocks(inp), blocks(out), mid["depth"], mid["n_head"])
    return specs, W


@pytest.mark.parametrize("which", ["tiny", "tiny_refiner"])
def test_unet_tree_round_trip(pkg, imp, tmp_path, which):
    ocfg = OC.tiny_config() if which == "tiny" else OC.tiny_refiner_config()
    root = str(tmp_path / "diffuser")
    specs, W = _dump_unet(imp, ocfg, root, 3)
    cfg, flat = imp.load_unet(pkg, root, is_refiner=ocfg.is_refiner)
    want = to_pkg_cfg(pkg, ocfg)
    assert cfg.model_channels == want.model_channels and list(cfg.channel_mults) == list(want.channel_mults)
    assert cfg.adm_in_channels == want.adm_in_channels and cfg.context_dim == want.context_dim
    assert cfg.n_head_channels == want.n_head_channels and (cfg.in_channels, cfg.out_channels) == (want.in_channels, want.out_channels)
    # levels without transformers carry depth 0 in the tree; what matters is that the parameter list is the same
    assert [(p.name, tuple(p.shape)) for p in pkg.unet_param_specs(cfg)] == [(p.name, tuple(p.shape)) for p in specs]
    assert np.array_equal(flat, pkg.flatten_weights(pkg.unet_param_specs(want), W))


def test_reference_file_format(imp, tmp_path):
    # save.py:12-18 -- dims first, then the values, all float32; scalars are [1.0, v]
    a = np.arange(24, dtype=np.float32).reshape(2, 3, 4) - 5
    f = str(tmp_path / "x" / "weight.npy")
    imp.write_tensor(a, f)
    raw = np.load(f)
    assert raw.dtype == np.float32 and raw.shape == (27,) and list(raw[:3]) == [2, 3, 4] and np.array_equal(raw[3:], a.reshape(-1))
    assert np.array_equal(imp.read_tensor(f, (2, 3, 4)), a) and np.array_equal(imp.read_tensor(f), a)
    imp.write_scalar(6, str(tmp_path / "n.npy"))
    assert list(np.load(str(tmp_path / "n.npy"))) == [1.0, 6.0] and imp.read_scalar(str(tmp_path / "n.npy")) == 6.0


def test_importer_errors(pkg, imp, tmp_path):
    ocfg = OC.tiny_config()
    root = str(tmp_path / "d")
    _dump_unet(imp, ocfg, root, 1)
    f = os.path.join(root, "middle_block", "res1", "conv_in", "weight.npy")
    raw = np.load(f)
    os.remove(f)
    with pytest.raises(imp.ImportError_, match="missing parameter file"):
        imp.load_unet(pkg, root)
    bad = raw.copy(); bad[0] += 1
    np.save(f, bad)
    with pytest.raises(imp.ImportError_, match="expected shape"):
        imp.load_unet(pkg, root)
    np.save(f, raw.astype(np.float64))
    with pytest.raises(imp.ImportError_, match="float32"):
        imp.load_unet(pkg, root)
    with open(os.path.join(root, "input_blocks", "1", "type.txt"), "w") as fh:
        fh.write("resnet_v2")
    np.save(f, raw)
    with pytest.raises(imp.ImportError_, match="unknown block kind"):
        imp.load_unet(pkg, root)


@pytest.mark.parametrize("encoder", [False, True])
def test_vae_tree_round_trip(pkg, imp, tmp_path, encoder):
    v = OC.tiny_vae_config()
    specs = OC.vae_encoder_param_specs(v) if encoder else OC.vae_decoder_param_specs(v)
    W = OC.synth_weights(specs, 4)
    root = str(tmp_path / "autoencoder")
    imp.export_tree(specs, W, root, "vae")
    if encoder:   # PaddedConv2d: the Conv2d of a downsampler sits one directory down (autoencoder/load.rs:74)
        assert os.path.exists(os.path.join(root, "encoder", "blocks", "0", "downsampler", "conv", "weight.npy"))
    pv = to_pkg_vcfg(pkg, v)
    flat = imp.load_vae(pkg, root, pv, encoder)
    assert np.array_equal(flat, pkg.flatten_weights(pkg.vae_param_specs(pv, encoder), W))


@pytest.mark.parametrize("with_projection", [True, False])
def test_clip_tree_round_trip(pkg, imp, tmp_path, with_projection):
    ocfg = OCL.tiny_open_clip_config() if with_projection else OCL.tiny_clip_config()
    specs = OCL.clip_param_specs(ocfg)
    W = OC.synth_weights(specs, 5)
    root = str(tmp_path / "clip")
    imp.export_tree(specs, W, root, "clip")
    imp.write_scalar(ocfg.n_layer, os.path.join(root, "n_layer.npy"))
    imp.write_scalar(ocfg.n_head, os.path.join(root, "blocks", "0", "attn", "n_head.npy"))
    if not with_projection:
        os.remove(os.path.join(root, "text_projection.npy"))      # the CLIP-L dump has none (python/clip.py:45-46)
        W = dict(W); W["text_projection"] = np.zeros_like(W["text_projection"])
    cfg, flat = imp.load_clip(pkg, root, quick_gelu=ocfg.quick_gelu)
    assert cfg == pkg.CLIPConfig(**ocfg.__dict__)
    assert np.array_equal(flat, pkg.flatten_weights(pkg.clip_param_specs(cfg), W))


def test_alphas_cumprod_file(imp, tmp_path):
    a = OC.alphas_cumprod().astype(np.float32)
    imp.write_tensor(a, str(tmp_path / "alphas_cumprod.npy"))
    got = imp.load_alphas_cumprod(str(tmp_path))
    assert got.dtype == np.float64 and np.array_equal(got.astype(np.float32), a)
