"""Reader (and writer) of the reference's intermediate `.npy` parameter tree -- SURVEY section 8f row 2.

The reference dumps SDXL with python/dump.py + python/save.py into `params/{diffuser_base,diffuser_refiner,autoencoder,clip,
open_clip}` and reads it back with src/model/**/load.rs: one file per tensor, a 1-D float32 array `[dims..., values...]`
(save.py:12-18, model/load.rs:15-25; scalars are `[1.0, v]`, save.py:7-10), Linear weights already transposed to [in, out]
(save.py:23), conv weights [out, in, kh, kw], the kind of every UNet block in `type.txt` (unet/load.rs:295-308).  This module
maps that tree onto the C ABI's flat fp32 buffer (`sdxl_*_param_spec()` order), so `UNet(ctx, cfg, weights=load_unet(...))`
replaces `load_unet(path, device)`; the packing into MFMA layouts stays on the GPU.  Host-side file I/O only: no GPU needed.

`.mpk` (burn NamedMpkFileRecorder, HalfPrecisionSettings): a FIRST reader is at the bottom of this file (read_mpk / mpk_flat).
Its field layout is defined by burn 0.13's record derive, whose source is not available on this box, and no
record file exists here, so that reader is UNVALIDATED against real data -- it is written tolerantly around the points that
could not be checked, and its tests only pin self-consistency.  The `.npy` path above is the checked one.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np


class ImportError_(RuntimeError):
    pass


DEFAULT_EPS = 1e-5   # GroupNormConfig / LayerNormConfig default


# ------------------------------------------------------------------------------------------------ one tensor <-> one file
def read_tensor(path: str, shape: Optional[Sequence[int]] = None) -> np.ndarray:
    """`[dims..., values...]` float32 -> ndarray.  With `shape` the header must match it exactly; without, the rank is
    inferred as the unique D with prod(header[:D]) == len - D (model/load.rs:15-25 knows D from the call site)."""
    if not os.path.exists(path):
        raise ImportError_(f"missing parameter file {path}")
    raw = np.load(path)
    if raw.ndim != 1 or raw.dtype != np.float32:
        raise ImportError_(f"{path}: expected a 1-D float32 array [dims..., values...], got {raw.dtype} {raw.shape}")
    if shape is not None:
        d = len(shape)
        if raw.size != d + int(np.prod(shape)) or tuple(int(v) for v in raw[:d]) != tuple(int(s) for s in shape):
            raise ImportError_(f"{path}: header {tuple(int(v) for v in raw[:min(raw.size, d)])} / {raw.size - d} values, "
                               f"expected shape {tuple(shape)}")
        return raw[d:].reshape(shape)
    cands = [d for d in range(1, 5) if raw.size > d and all(v >= 1 and v == int(v) for v in raw[:d])
             and int(np.prod(raw[:d].astype(np.int64))) == raw.size - d]
    if len(cands) != 1:
        raise ImportError_(f"{path}: cannot infer the tensor rank (candidates {cands})")
    d = cands[0]
    return raw[d:].reshape([int(v) for v in raw[:d]])


def read_scalar(path: str) -> float:
    """save_scalar (save.py:7-10): `[1.0, v]`"""
    return float(read_tensor(path, (1,))[0])


def write_tensor(arr, path: str):
    """save_tensor (save.py:12-18)"""
    arr = np.asarray(arr, dtype=np.float32)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    np.save(path, np.concatenate([np.asarray(arr.shape, dtype=np.float32), arr.reshape(-1)]).astype(np.float32))


def write_scalar(v, path: str):
    write_tensor(np.asarray([float(v)], dtype=np.float32), path)


# ------------------------------------------------------------------------------------------------ spec name -> file
def spec_path(model: str, name: str) -> str:
    """relative file of one `sdxl_*_param_spec()` entry inside its model directory.
    model: 'unet' | 'vae' | 'clip'.  Field names are the reference's struct fields, so the path is the dotted name with
    `/`, except: norm gamma/beta are stored as weight/bias (groupnorm/load.rs:14-21, layernorm/load.rs:11-12); transformer
    blocks live in `transformer_{j}` (unet/load.rs:122-126); the encoder's downsampler is a PaddedConv2d whose Conv2d sits
    in `conv/` (autoencoder/load.rs:74); CLIP's position table is `position_embedding/weight` and the projection a bare
    `text_projection.npy` (clip/load.rs:84-102)."""
    parts = name.split(".")
    leaf = {"gamma": "weight", "beta": "bias"}.get(parts[-1], parts[-1])
    parts = parts[:-1]
    if model == "clip":
        if name == "position_embedding":
            return "position_embedding/weight.npy"
        if name == "text_projection":
            return "text_projection.npy"
    out: List[str] = []
    i = 0
    while i < len(parts):
        if model == "unet" and parts[i] == "blocks" and i > 0 and parts[i - 1] == "transformer":
            out.append(f"transformer_{parts[i + 1]}")
            i += 2
            continue
        out.append(parts[i])
        i += 1
    if model == "vae" and len(out) >= 1 and out[-1] == "downsampler":
        out.append("conv")
    return "/".join(out + [leaf + ".npy"])


def load_flat(specs, root: str, model: str, optional: Sequence[str] = ()) -> np.ndarray:
    """every spec entry from its file, concatenated in spec order (the `weights_flat` argument of sdxl_*_create)"""
    parts = []
    for p in specs:
        f = os.path.join(root, spec_path(model, p.name))
        if p.name in optional and not os.path.exists(f):
            parts.append(np.zeros(int(np.prod(p.shape)), dtype=np.float32))
            continue
        parts.append(read_tensor(f, tuple(p.shape)).reshape(-1))
    return np.ascontiguousarray(np.concatenate(parts), dtype=np.float32)


def export_tree(specs, weights: Dict[str, np.ndarray], root: str, model: str):
    """the inverse (tests, and converting other checkpoints INTO the reference's tree): tensors only"""
    for p in specs:
        write_tensor(np.asarray(weights[p.name], dtype=np.float32).reshape(p.shape), os.path.join(root, spec_path(model, p.name)))


_TYPE = {0: "conv", 1: "resnet", 2: "downsample", 3: "resnet_transformer", 4: "resnet_transformer_upsample", 5: "resnet_upsample"}


def export_unet_structure(root: str, model_channels: int, input_blocks, output_blocks, mid_depth: int, mid_heads: int):
    """the non-tensor files load_unet needs beside the tensors (unet/load.rs:286-308,365-367,116-121,48): block counts,
    `type.txt`, `model_channels`, transformer `n_blocks`, attention `n_head`.  Blocks are (kind, depth, n_head) with kind
    in BlockKind order (conv, resnet, downsample, resnet_transformer, resnet_transformer_upsample, resnet_upsample)."""
    write_scalar(model_channels, os.path.join(root, "model_channels.npy"))

    def transformer(path, depth, heads):
        write_scalar(depth, os.path.join(path, "n_blocks.npy"))
        for j in range(depth):
            for a in ("attn1", "attn2"):
                write_scalar(heads, os.path.join(path, f"transformer_{j}", a, "n_head.npy"))

    for group, blocks in (("input_blocks", input_blocks), ("output_blocks", output_blocks)):
        write_scalar(len(blocks), os.path.join(root, group, "n_blocks.npy"))
        for i, (kind, depth, heads) in enumerate(blocks):
            d = os.path.join(root, group, str(i))
            os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, "type.txt"), "w") as fh:
                fh.write(_TYPE[kind])
            if kind in (3, 4):
                transformer(os.path.join(d, "transformer"), depth, heads)
    transformer(os.path.join(root, "middle_block", "transformer"), mid_depth, mid_heads)


# ------------------------------------------------------------------------------------------------ configs from the tree
_KINDS = {"conv", "resnet", "downsample", "resnet_transformer", "resnet_transformer_upsample", "resnet_upsample"}


def _header(path: str) -> Tuple[int, ...]:
    raw = np.load(path, mmap_mode="r")
    return tuple(int(v) for v in raw[:4])


def infer_unet_config(root: str):
    """UNetConfig fields from the dumped tree itself (the reference rebuilds them from `.cfg` JSON, sample/main.rs:29-43):
    returns a dict of the package's UNetConfig constructor arguments."""
    mc = int(read_scalar(os.path.join(root, "model_channels.npy")))
    n_in = int(read_scalar(os.path.join(root, "input_blocks", "n_blocks.npy")))
    kinds = []
    for i in range(n_in):
        k = open(os.path.join(root, "input_blocks", str(i), "type.txt")).read().strip()
        if k not in _KINDS:
            raise ImportError_(f"input_blocks/{i}/type.txt: unknown block kind {k!r}")
        kinds.append(k)
    mults, depths = [], []
    cur = None
    for i, k in enumerate(kinds):
        if k in ("resnet", "resnet_transformer"):
            sub = "" if k == "resnet" else "res"
            cout = _header(os.path.join(root, "input_blocks", str(i), sub, "conv_in", "weight.npy"))[0]
            depth = 0
            if k == "resnet_transformer":
                depth = int(read_scalar(os.path.join(root, "input_blocks", str(i), "transformer", "n_blocks.npy")))
            if cur is None:
                cur = (cout // mc, depth)
        elif k == "downsample" and cur is not None:
            mults.append(cur[0]); depths.append(cur[1]); cur = None
    if cur is not None:
        mults.append(cur[0]); depths.append(cur[1])
    mid_n = os.path.join(root, "middle_block", "transformer", "n_blocks.npy")
    if depths and os.path.exists(mid_n):
        depths[-1] = int(read_scalar(mid_n))     # the middle block takes transformer_depths[-1] (unet/mod.rs:238-248)
    in_w = _header(os.path.join(root, "input_blocks", "0", "weight.npy"))
    out_w = _header(os.path.join(root, "conv_out", "weight.npy"))
    adm = _header(os.path.join(root, "lin1_label_embed", "weight.npy"))[0]
    ctx_dim, heads_c = 0, 64
    mid = os.path.join(root, "middle_block", "transformer", "transformer_0", "attn2")
    if os.path.exists(os.path.join(mid, "key", "weight.npy")):
        kd = _header(os.path.join(mid, "key", "weight.npy"))
        ctx_dim = kd[0]
        heads_c = kd[1] // int(read_scalar(os.path.join(mid, "n_head.npy")))
    return dict(adm_in_channels=adm, model_channels=mc, channel_mults=mults, n_head_channels=heads_c,
                transformer_depths=depths, context_dim=ctx_dim, in_channels=in_w[1], out_channels=out_w[0])


def infer_clip_config(root: str, quick_gelu: bool):
    """CLIPConfig fields from the tree (quick_gelu is not stored: true for `clip`, false for `open_clip`, clip/load.rs:13-30)"""
    tok = _header(os.path.join(root, "token_embedding", "weight.npy"))
    pos = _header(os.path.join(root, "position_embedding", "weight.npy"))
    n_layer = int(read_scalar(os.path.join(root, "n_layer.npy")))
    n_head = int(read_scalar(os.path.join(root, "blocks", "0", "attn", "n_head.npy")))
    proj = os.path.join(root, "text_projection.npy")
    embed = _header(proj)[1] if os.path.exists(proj) else tok[1]
    return dict(n_vocab=tok[0], n_state=tok[1], embed_dim=embed, n_head=n_head, n_ctx=pos[0], n_layer=n_layer,
                quick_gelu=quick_gelu)


# ------------------------------------------------------------------------------------------------ whole models
def load_unet(pkg, root: str, is_refiner: bool = False):
    """load_unet (unet/load.rs:365-401) -> (UNetConfig, flat weights).  `root` = params/diffuser_base | diffuser_refiner"""
    cfg = pkg.UNetConfig(**infer_unet_config(root), is_refiner=is_refiner)
    return cfg, load_flat(pkg.unet_param_specs(cfg), root, "unet")


def load_vae(pkg, root: str, cfg=None, encoder: bool = False):
    """load_autoencoder (autoencoder/load.rs:186-201) -> flat decoder (or encoder) weights.  `root` = params/autoencoder"""
    cfg = cfg or pkg.VAEConfig()
    return load_flat(pkg.vae_param_specs(cfg, encoder), root, "vae")


def load_clip(pkg, root: str, quick_gelu: bool):
    """load_clip_text_transformer (clip/load.rs:79-115) -> (CLIPConfig, flat weights).  `root` = params/clip | open_clip;
    a missing text_projection (the CLIP-L dump has none, python/clip.py:45-46) becomes zeros -- only forward_hidden is
    ever called on that encoder (stablediffusion/mod.rs:759-770)"""
    cfg = pkg.CLIPConfig(**infer_clip_config(root, quick_gelu))
    return cfg, load_flat(pkg.clip_param_specs(cfg), root, "clip", optional=("text_projection",))


def load_alphas_cumprod(params_root: str) -> np.ndarray:
    """params/alphas_cumprod.npy (python/dump.py:33-35; stablediffusion/load.rs:56-60)"""
    return read_tensor(os.path.join(params_root, "alphas_cumprod.npy")).astype(np.float64)


# ------------------------------------------------------------------------------------------------ burn `.mpk` records
# UNVALIDATED against a real file: no `.mpk` exists on this box and burn 0.13's sources are not available.
# The reader below follows the record layout burn documents for `NamedMpkFileRecorder<HalfPrecisionSettings>` as far as it is
# known here, and is written to tolerate the points that could not be checked:
#   file        = rmp_serde "named" MessagePack of BurnRecord { metadata: {...}, item: <module record> }
#   module      = map of its struct fields by name; Vec<Module> = array; enum module = { "<Variant>": record } (externally tagged);
#                 Option::None = nil; non-parameter fields (usize, f64, ...) = constant records (nil / empty) -- ignored
#   Param<T>    = { id: str, param: <tensor record> };  tensor record = { value: [...], shape: [...] }, possibly wrapped once
#                 more ({ data: {...} }); f16 elements arrive as their 16 raw bits (the `half` crate's serde form) or as floats
# Field names are the reference's struct fields, which is what the spec names of the C ABI already are; the enum variant level
# (UNetBlocks::{Conv, Res, ...}, unet/mod.rs:509-516) and PaddedConv2d's inner `conv` are skipped when walking a name.
def _mpk_tensor(node, where: str) -> np.ndarray:
    for _ in range(3):                                   # { id, param: ... } / { data: ... } wrappers
        if isinstance(node, dict) and "value" not in node:
            inner = node.get("param", node.get("data"))
            if inner is None:
                break
            node = inner
    if not (isinstance(node, dict) and "value" in node and "shape" in node):
        raise ImportError_(f"{where}: not a tensor record (keys {list(node) if isinstance(node, dict) else type(node).__name__})")
    shape = [int(v) for v in node["shape"]]
    val = node["value"]
    if isinstance(val, (bytes, bytearray)):
        arr = np.frombuffer(val, dtype=np.float16).astype(np.float32)
    else:
        arr = np.asarray(val)
        arr = arr.astype(np.uint16).view(np.float16).astype(np.float32) if arr.dtype.kind in "iu" else arr.astype(np.float32)
    if arr.size != int(np.prod(shape)):
        raise ImportError_(f"{where}: {arr.size} values for shape {shape}")
    return arr.reshape(shape)


def _mpk_descend(node, key: str, where: str):
    while True:
        if isinstance(node, (list, tuple)):
            if not key.isdigit() or int(key) >= len(node):
                raise ImportError_(f"{where}: index {key!r} into an array of {len(node)}")
            return node[int(key)]
        if not isinstance(node, dict):
            raise ImportError_(f"{where}: cannot take field {key!r} of {type(node).__name__}")
        if key in node:
            return node[key]
        if len(node) == 1 and next(iter(node))[:1].isupper():     # enum variant wrapper
            node = next(iter(node.values()))
            continue
        if "conv" in node and key in ("weight", "bias"):           # PaddedConv2d { conv: Conv2d, ... }
            node = node["conv"]
            continue
        raise ImportError_(f"{where}: no field {key!r} (have {sorted(node)[:12]})")


def read_mpk(path: str):
    """the `item` tree of a burn NamedMpk record file (see the layout note above)"""
    import msgpack
    with open(path, "rb") as fh:
        rec = msgpack.unpackb(fh.read(), raw=False, strict_map_key=False)
    if not (isinstance(rec, dict) and "item" in rec):
        raise ImportError_(f"{path}: not a burn record (top-level keys {list(rec) if isinstance(rec, dict) else type(rec).__name__})")
    return rec["item"]


def mpk_flat(specs, item, prefix: str = "", optional: Sequence[str] = (), dtype=np.float32) -> np.ndarray:
    """spec entries looked up by field path under `prefix` (e.g. 'diffusion' for a Diffuser record, 'autoencoder' for a
    LatentDecoder record, 'clip' / 'open_clip' for an Embedder record), concatenated in spec order"""
    root = item
    for k in [p for p in prefix.split(".") if p]:
        root = _mpk_descend(root, k, prefix)
    parts = []
    for p in specs:
        node, ok = root, True
        try:
            for k in p.name.split("."):
                node = _mpk_descend(node, k, p.name)
        except ImportError_:
            ok = False
        if p.name.endswith(".eps"):
            # module constants (eps, n_group, ...) are not tensors of a burn record: the loaded module keeps what its
            # Config::init set, i.e. the 1e-5 default (groupnorm/mod.rs:13-14, layernorm/mod.rs:12-13); a numeric field is honoured
            parts.append(np.asarray([float(node) if ok and isinstance(node, (int, float)) else DEFAULT_EPS], dtype=np.float32))
            continue
        if (not ok or node is None) and p.name in optional:
            parts.append(np.zeros(int(np.prod(p.shape)), dtype=np.float32))
            continue
        if not ok or node is None:
            raise ImportError_(f"{p.name}: missing in the record")
        t = _mpk_tensor(node, p.name)
        if tuple(t.shape) != tuple(p.shape):
            raise ImportError_(f"{p.name}: record has shape {tuple(t.shape)}, expected {tuple(p.shape)}")
        parts.append(t.reshape(-1))
    # dtype=np.float16: the record's own precision, handed to sdxl_*_create_f16 without the fp32 expansion (exact: every
    # value of a HalfPrecisionSettings record is an f16; the 1e-5 eps default rounds to 1.0014e-5)
    return np.ascontiguousarray(np.concatenate(parts), dtype=dtype)


def write_mpk(path: str, item) -> None:
    """writes `item` in the layout read_mpk expects (tests; f16 values as raw 16-bit integers)"""
    import msgpack
    with open(path, "wb") as fh:
        fh.write(msgpack.packb({"metadata": {"float": "f16", "int": "i32", "format": "burn_core::record::file::NamedMpkFileRecorder",
                                             "version": "0.13.0", "settings": "HalfPrecisionSettings"}, "item": item}))


def mpk_param(arr, pid: str = "0", as_bytes: bool = False) -> dict:
    a = np.asarray(arr, dtype=np.float32).astype(np.float16)
    val = a.tobytes() if as_bytes else a.reshape(-1).view(np.uint16).tolist()
    return {"id": pid, "param": {"value": val, "shape": list(a.shape)}}
