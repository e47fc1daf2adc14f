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


# ---- burn .mpk records: the reader is UNVALIDATED against a real file (none on this box); these tests only pin that it reads
# the layout documented in importer.py, including the enum-variant level, PaddedConv2d's inner conv and Option::None
def test_mpk_record_round_trip(pkg, imp, tmp_path):
    ocfg = OC.tiny_config()
    specs = OC.unet_param_specs(ocfg)
    W = OC.synth_weights(specs, 8)
    item = {"n_steps": None, "alpha_cumulative_products": imp.mpk_param(OC.alphas_cumprod()), "is_refiner": None, "diffusion": {}}
    inp, mid, out = OC.unet_block_plan(ocfg)
    variant = {"input_blocks": [b["kind"] for b in inp], "output_blocks": [b["kind"] for b in out]}
    for p in specs:
        keys = p.name.split(".")
        node = item["diffusion"]
        i = 0
        while i < len(keys) - 1:
            k = keys[i]
            if keys[i + 1].isdigit():                       # Vec<Module>
                lst = node.setdefault(k, [])
                idx = int(keys[i + 1])
                while len(lst) <= idx:
                    lst.append(None)
                if lst[idx] is None:
                    lst[idx] = {variant[k][idx]: {}} if k in variant else {}
                node = lst[idx][variant[k][idx]] if k in variant else lst[idx]
                i += 2
            else:
                node = node.setdefault(k, {})
                i += 1
        if keys[-1] == "eps":
            node["eps"] = None                              # module constant: not a tensor of the record
            continue
        node[keys[-1]] = imp.mpk_param(W[p.name], pid=p.name, as_bytes=W[p.name].size > 4096)   # big tensors as raw f16 bytes (speed)
    path = str(tmp_path / "diffuser.mpk")
    imp.write_mpk(path, item)
    tree = imp.read_mpk(path)
    flat = imp.mpk_flat(pkg.unet_param_specs(to_pkg_cfg(pkg, ocfg)), tree, "diffusion")
    want = pkg.flatten_weights(pkg.unet_param_specs(to_pkg_cfg(pkg, ocfg)),
                               {k: (np.asarray(v, np.float32) if k.endswith(".eps") else
                                    np.asarray(v, np.float32).astype(np.float16).astype(np.float32)) for k, v in W.items()})
    assert np.array_equal(flat, want)                      # the record holds f16: equal to the f16-rounded weights
    a = imp._mpk_tensor(tree["alpha_cumulative_products"], "alphas")
    assert a.shape == (1000,) and abs(float(a[0]) - float(OC.alphas_cumprod()[0])) < 1e-3
    with pytest.raises(imp.ImportError_, match="missing in the record"):
        imp.mpk_flat(pkg.unet_param_specs(to_pkg_cfg(pkg, ocfg)), tree, "alpha_cumulative_products")


def test_mpk_padded_conv_and_optional(pkg, imp, tmp_path):
    from oracle import clip as OCL
    ocfg = OCL.tiny_clip_config()
    specs = OCL.clip_param_specs(ocfg)
    W = OC.synth_weights(specs, 9)
    clip = {"blocks": [{} for _ in range(ocfg.n_layer)]}
    for p in specs:
        if p.name == "text_projection":
            clip["text_projection"] = None                 # Option::None (the CLIP-L record has no projection)
            continue
        keys = p.name.split(".")
        node = clip
        i = 0
        while i < len(keys) - 1:
            if keys[i + 1].isdigit():
                node = node[keys[i]][int(keys[i + 1])]; i += 2
            else:
                node = node.setdefault(keys[i], {}); i += 1
        if keys[-1] == "eps":
            continue                                       # module constant: absent from the record -> Config default
        # tensor record wrapped once more in { data: ... }: tolerated
        node[keys[-1]] = {"id": "x", "param": {"data": imp.mpk_param(W[p.name])["param"]}}
    path = str(tmp_path / "embedder.mpk")
    imp.write_mpk(path, {"clip": clip, "clip_tokenizer": None})
    flat = imp.mpk_flat(pkg.clip_param_specs(pkg.CLIPConfig(**ocfg.__dict__)), imp.read_mpk(path), "clip", optional=("text_projection",))
    W16 = {k: (np.asarray(v, np.float32) if k.endswith(".eps") else np.asarray(v, np.float32).astype(np.float16).astype(np.float32))
           for k, v in W.items()}
    W16["text_projection"] = np.zeros_like(W16["text_projection"])
    assert np.array_equal(flat, pkg.flatten_weights(pkg.clip_param_specs(pkg.CLIPConfig(**ocfg.__dict__)), W16))
    # PaddedConv2d { conv: Conv2d { weight, bias }, ... }: the inner level is skipped
    node = {"downsampler": {"conv": {"weight": imp.mpk_param(np.ones((2, 2, 3, 3))), "bias": None}, "kernel_size": None}}
    t = imp._mpk_tensor(imp._mpk_descend(imp._mpk_descend(node, "downsampler", "x"), "weight", "x"), "x")
    assert t.shape == (2, 2, 3, 3)


def test_per_norm_eps_travels_through_the_npy_tree(pkg, imp, tmp_path):
    # the reference reads eps per norm (groupnorm/load.rs:19, layernorm/load.rs:17; written by save.py:31,39): a dump whose
    # norms carry eps = 1e-6 must reach the flat buffer in the `.eps` slots, everything else untouched
    ocfg = OC.tiny_config()
    specs = OC.unet_param_specs(ocfg)
    W = OC.synth_weights(specs, 3)
    assert all(float(W[p.name][0]) == np.float32(1e-5) for p in specs if p.name.endswith(".eps"))     # synthetic default
    changed = [p.name for p in specs if p.name.endswith(".eps") and (".transformer.norm" in p.name or p.name == "norm_out.eps")]
    assert changed
    for n in changed:
        W[n] = np.asarray([1e-6], dtype=np.float32)
    root = str(tmp_path / "diffuser_base")
    pspecs = pkg.unet_param_specs(to_pkg_cfg(pkg, ocfg))
    imp.export_tree(pspecs, W, root, "unet")
    assert imp.read_scalar(os.path.join(root, "norm_out", "eps.npy")) == float(np.float32(1e-6))
    flat = imp.load_flat(pspecs, root, "unet")
    assert np.array_equal(flat, pkg.flatten_weights(pspecs, W))
    off = 0
    for p in pspecs:
        if p.name.endswith(".eps"):
            assert flat[off] == np.float32(1e-6 if p.name in changed else 1e-5), p.name
        off += int(np.prod(p.shape))
