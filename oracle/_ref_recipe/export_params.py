"""Writes the tiny-config probe parameters in the reference's .npy tree format (see README.md).  TEST INFRASTRUCTURE.

    python -m oracle._ref_recipe.export_params <out_dir>
"""
import importlib.util
import json
import os
import sys

from .. import config as OC

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _importer():
    spec = importlib.util.spec_from_file_location("sdxl_importer", os.path.join(ROOT, "stable-diffusion-xl-burn_amd", "importer.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _structure(imp, specs, root, model, ucfg=None):
    """the non-tensor files the reference's loaders read next to every tensor (python/save.py:27-107 conventions; load_conv2d
    src/model/load.rs:119-155, load_group_norm groupnorm/load.rs:11-38, load_padded_conv2d autoencoder/load.rs:69-98)"""
    import numpy as np
    down = set()
    if ucfg is not None:
        inp, _, _ = OC.unet_block_plan(ucfg)
        down = {f"input_blocks.{i}" for i, b in enumerate(inp) if b["kind"] == "Down"}
    for p in specs:
        d = os.path.join(root, os.path.dirname(imp.spec_path(model, p.name)))
        if p.name.endswith(".weight") and len(p.shape) == 4:
            cout, cin, k, _ = p.shape
            padded = model == "vae" and ".downsampler" in p.name          # PaddedConv2d: its Conv2d sits in conv/ with padding (0, 0)
            stride = 2 if (padded or any(p.name.startswith(x + ".") for x in down)) else 1
            pad = 0 if (padded or k == 1) else k // 2
            for name, v in (("stride", [stride, stride]), ("padding", [pad, pad]), ("dilation", [1, 1]), ("kernel_size", [k, k])):
                imp.write_tensor(np.asarray(v, dtype=np.float32), os.path.join(d, name + ".npy"))
            for name, v in (("n_group", 1), ("n_channels_in", cin), ("n_channels_out", cout)):
                imp.write_scalar(v, os.path.join(d, name + ".npy"))
            if padded:
                up = os.path.dirname(d)                                       # .../downsampler
                imp.write_tensor(np.asarray([cin, cout], dtype=np.float32), os.path.join(up, "channels.npy"))
                imp.write_scalar(k, os.path.join(up, "kernel_size.npy"))
                imp.write_scalar(2, os.path.join(up, "stride.npy"))
                imp.write_tensor(np.asarray([0, 1, 0, 1], dtype=np.float32), os.path.join(up, "padding.npy"))   # left, right, top, bottom
        if p.name.endswith(".gamma") and (model == "vae" or ".norm" in p.name or p.name.startswith("norm")):
            is_group = model == "vae" or not any(t in p.name for t in (".norm1.", ".norm2.", ".norm3."))
            if is_group:
                imp.write_scalar(32, os.path.join(d, "n_group.npy"))
                imp.write_scalar(p.shape[0], os.path.join(d, "n_channel.npy"))


def main(out):
    imp = _importer()
    ucfg, vcfg = OC.tiny_config(), OC.tiny_vae_config()
    specs = OC.unet_param_specs(ucfg)
    root = os.path.join(out, "params_unet")
    imp.export_tree(specs, OC.synth_weights(specs, 0), root, "unet")
    kind = {"Conv": 0, "Res": 1, "Down": 2, "ResT": 3, "ResTU": 4, "ResU": 5}
    inp, mid, outb = OC.unet_block_plan(ucfg)
    blocks = lambda bl: [(kind[b["kind"]], b.get("depth", 0), b.get("n_head", 0)) for b in bl]   # noqa: E731
    imp.export_unet_structure(root, ucfg.model_channels, blocks(inp), blocks(outb), mid["depth"], mid["n_head"])
    _structure(imp, specs, root, "unet", ucfg)
    vroot = os.path.join(out, "params_vae")          # an Autoencoder tree: encoder/, decoder/, quant_conv/, post_quant_conv/
    for sp, sub, n in ((OC.vae_encoder_param_specs(vcfg), "encoder", len(vcfg.enc_channels)), (OC.vae_decoder_param_specs(vcfg), "decoder", len(vcfg.dec_channels))):
        imp.export_tree(sp, OC.synth_weights(sp, 0), vroot, "vae")
        _structure(imp, sp, vroot, "vae")
        imp.write_scalar(n, os.path.join(vroot, sub, "n_block.npy"))
    with open(os.path.join(out, "inputs.json"), "w") as fh:
        json.dump({"recipe": "arb_tensor = sin(arange(n)).reshape(shape)", "unet": {"x": [1, 4, 8, 8], "context": [1, 1, 20], "y": [1, 8], "t": [1]},
                   "encoder": {"x": [1, 3, 16, 16]}, "decoder": {"x": [1, 4, 4, 4]}, "weights": "oracle/config.py synth_weights(seed 0), tiny configs"}, fh, indent=1)
    print(f"probe parameters written under {out}")


if __name__ == "__main__":
    main(sys.argv[1])
