"""Paper model of Winograd F(2x2, 3x3) for the 3x3 convolutions of the SDXL step and of the VAE decode (verdict r3 item 4), plus the
numerical price of f16 products measured with numpy against the fp32 convolution.

    python tools/winograd_model.py [profiles/r03_step_launches_final.csv] > profiles/r04_winograd_model.txt

Model (every number it uses is a MEASURED rate of this engine on MI355X; sources in brackets):
  direct   = the implicit-GEMM time of the launch as measured inside the step            [profiles/r03_step_launches_final.csv]
  winograd = 16 point-GEMMs [M/4 tiles x Cin] x [Cin x Cout] (2.25x fewer MACs per output, k-loop 9x shorter per point)
             + input transform  (read M Cin, write 4 M Cin elements: every 2x2 output tile needs a 4x4 input patch = 16 values)
             + output transform (read 16 (M/4) Cout fp32 accumulators, write M Cout, + bias / residual / time-embedding epilogue)
  GEMM part: k-loop at the launch's own measured steady-state rate (direct time minus the 7.5 us fixed cost of a launch
             [profiles/r04_wreg_knockout.txt: 16.4 / 42.7 / 147.8 us at K = 1280 / 5120 / 20480 -> 7.6 us intercept]), divided by 2.25,
             plus the same fixed cost once (one batched launch over the 16 points).
  transform passes: an HBM / L2-bound elementwise kernel moves 1.8 TB/s at these sizes and has an 8 us floor
             [DESIGN 3.3: gn_apply 1.8 TB/s, 7 - 10 us floor per launch]; fusing them is priced separately below:
             * the input transform cannot ride on the operand staging of these kernels -- the activation tile goes HBM -> LDS by DMA
               (global_load_lds), never through registers -- but it CAN ride on the GroupNorm-apply kernel that writes the conv's
               input (it then writes 4x the bytes instead of 1x: + 3 M Cin elements of traffic, no extra launch);
             * the output transform needs the 16 point results of a tile in one workgroup: 16 accumulator tiles per wave tile
               (256 registers for ONE 32x32 tile) -- not available next to the operand fragments -> it stays a pass.
Both variants are listed: `unfused` (two extra passes) and `gn-fused` (input transform inside gn_apply, output pass kept).
"""
import collections
import csv
import sys

import numpy as np

FIXED_US = 7.5
PASS_TBS = 1.8
PASS_FLOOR_US = 8.0


def pass_us(nbytes):
    return max(PASS_FLOOR_US, nbytes / (PASS_TBS * 1e12) * 1e6)


def model(M, Cin, Cout, direct_us, act_bytes, acc_bytes=4):
    kloop = max(direct_us - FIXED_US, 0.0)
    gemm = FIXED_US + kloop / 2.25
    t_in = pass_us(M * Cin * act_bytes * (1 + 4))                    # read x, write V (16 values per 4 outputs)
    t_in_fused = (3 * M * Cin * act_bytes) / (PASS_TBS * 1e12) * 1e6    # extra bytes written by gn_apply, no launch
    t_out = pass_us(4 * M * Cout * acc_bytes + 2 * M * Cout * act_bytes)   # read 16 x (M/4) accumulators, residual in, y out
    return gemm, t_in, t_in_fused, t_out


def f16(x):
    return x.astype(np.float16).astype(np.float32)


def winograd_error(C=64, H=32, W=32, Co=64, seed=0):
    """max-abs error relative to max|ref| of (a) the direct conv with f16-rounded operands, fp32 accumulate (what the engine's f16
    GEMM does) and (b) F(2x2,3x3) with fp32 transforms, f16-rounded U / V, fp32 accumulate, fp32 output transform"""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((C, H + 2, W + 2)).astype(np.float32)
    x[:, 0] = x[:, -1] = 0
    x[:, :, 0] = x[:, :, -1] = 0
    w = (rng.standard_normal((Co, C, 3, 3)) / np.sqrt(9 * C)).astype(np.float32)

    def conv(xx, ww):
        out = np.zeros((Co, H, W), np.float64)
        for dy in range(3):
            for dx in range(3):
                out += np.einsum("oc,chw->ohw", ww[:, :, dy, dx].astype(np.float64), xx[:, dy:dy + H, dx:dx + W].astype(np.float64))
        return out
    ref = conv(x, w)
    direct = conv(f16(x), f16(w))
    Bt = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float32)
    G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float32)
    At = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float32)
    U = f16(np.einsum("ij,ocjk,lk->ocil", G, w, G))                  # [Co][C][4][4]
    out = np.zeros((Co, H, W), np.float64)
    for ty in range(H // 2):
        for tx in range(W // 2):
            d = x[:, 2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4]
            V = f16(np.einsum("ij,cjk,lk->cil", Bt, d, Bt))
            Mm = np.einsum("ocil,cil->oil", U.astype(np.float64), V.astype(np.float64))
            out[:, 2 * ty:2 * ty + 2, 2 * tx:2 * tx + 2] = np.einsum("ij,ojk,lk->oil", At, Mm, At)
    s = np.abs(ref).max()
    return np.abs(direct - ref).max() / s, np.abs(out - ref).max() / s


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else "profiles/r03_step_launches_final.csv"
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["class"] == "0" and r["ksize"] == "3" and int(r["K"]) >= 2880 and int(r["N"]) >= 320:
            k = (int(r["M"]), int(r["N"]), int(r["K"]))
            agg[k][0] += 1
            agg[k][1] += float(r["ms"]) * 1e3
    print(__doc__)
    print("UNet step (CFG pair, f16), 3x3 convolutions with stride 1 (eager per-launch times include ~3 us of event overhead):")
    print(f"{'M x Cout, Cin':>24} {'n':>3} {'direct':>8} | {'gemm':>7} {'in':>6} {'out':>6} {'unfused':>8} | {'in(gn)':>6} {'gn-fused':>8} | gain per step (unfused / gn-fused), us")
    tot = [0.0, 0.0, 0.0]
    for (M, N, K), (n, us) in sorted(agg.items()):
        Cin = K // 9
        d = us / n
        gemm, t_in, t_inf, t_out = model(M, Cin, N, d, 2)
        unf, fus = gemm + t_in + t_out, gemm + t_inf + t_out
        tot[0] += n * d; tot[1] += n * min(d, unf); tot[2] += n * min(d, fus)
        print(f"{M:>8} x {N:>4}, {Cin:>5} {n:>3} {d:8.1f} | {gemm:7.1f} {t_in:6.1f} {t_out:6.1f} {unf:8.1f} | {t_inf:6.1f} {fus:8.1f} | {n * (d - unf):+8.1f} {n * (d - fus):+8.1f}")
    print(f"3x3 class of the step: direct {tot[0] / 1e3:.2f} ms; taking Winograd only where the model says it wins: unfused {tot[1] / 1e3:.2f} ms, "
          f"gn-fused {tot[2] / 1e3:.2f} ms  (gain {(tot[0] - tot[1]) / 1e3:.2f} / {(tot[0] - tot[2]) / 1e3:.2f} ms per step)")
    print()
    print("VAE decode (f32_split: HL16 operands, 4 bytes per element, 3 MFMAs per product; decode = 37.9 ms, 271 TFLOP/s of conv work):")
    vae = [("128^2 x 512", 128 * 128, 512, 512, 11), ("256^2 x 512", 256 * 256, 512, 512, 8), ("512^2 x 256", 512 * 512, 256, 256, 6),
           ("1024^2 x 128", 1024 * 1024, 128, 128, 6)]
    vt = [0.0, 0.0]
    for name, M, Cin, Cout, n in vae:
        d = 2.0 * M * Cin * 9 * Cout / 271e12 * 1e6 + FIXED_US
        gemm, t_in, t_inf, t_out = model(M, Cin, Cout, d, 4)
        fus = gemm + t_inf + t_out
        vt[0] += n * d; vt[1] += n * min(d, fus)
        print(f"{name:>14} x{n:>2}: direct {d:8.1f} us | gemm {gemm:8.1f} + in(gn) {t_inf:7.1f} + out {t_out:7.1f} = {fus:8.1f} us  ({n * (d - fus) / 1e3:+.2f} ms per decode)")
    print(f"VAE 3x3 class: direct {vt[0] / 1e3:.1f} ms -> {vt[1] / 1e3:.1f} ms where it wins  (gain {(vt[0] - vt[1]) / 1e3:.2f} ms per decode)")
    print()
    e = [winograd_error(seed=s) for s in range(3)]
    print("numerics (64 -> 64 channels, 32 x 32, N(0,1) inputs, N(0, 1/fan_in) weights, three seeds; max-abs error / max|ref|):")
    print("   direct, f16 operands, fp32 accumulate : " + "  ".join(f"{a:.2e}" for a, _ in e))
    print("   F(2x2,3x3), f16 U / V, fp32 accumulate: " + "  ".join(f"{b:.2e}" for _, b in e) + f"   ({np.mean([b / a for a, b in e]):.1f}x the direct error)")


if __name__ == "__main__":
    main()
