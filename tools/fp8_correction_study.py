"""Would the two CORRECTION products of the fp16x2 split survive fp8?  (CPU, numpy; NOTEBOOK.md section 8)

The irrep GEMMs evaluate every product a * w as a_h w_h + a_h w_l + a_l w_h (a = a_h + a_l, two fp16 planes; 3 MFMAs per term at the fp16
rate).  The two correction terms are 2^-11 of the main one; on gfx950 an fp8 MFMA runs at twice the fp16 rate, so evaluating them in fp8
would cost (1 + 2 x 0.5) / 3 = 2 / 3 of today's matrix time - IF the result stays inside the 1e-4 tolerance of BASELINE.json with margin.
This script emulates PartI (direct 13-tap form: the Fourier form applies orthogonal transforms to the same sums) on seeded weights and inputs
with each layer's products evaluated in: fp32 (the oracle's arithmetic), the shipped 3-product fp16 split, main product in fp16 + corrections
with BOTH operands rounded to fp8 e4m3 / e5m2 (per-tensor power-of-two scales, as an MFMA needs them), 2 products, 1 product - and prints the
worst relative error of the descriptor against an fp64 evaluation.

    python tools/fp8_correction_study.py [keypoints = 48]
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from yoho_amd import synth, weights as W           # noqa: E402
from yoho_amd.tables import default_tables          # noqa: E402

G, NTAP = 60, 13


def f16(x):
    return x.astype(np.float16).astype(np.float64)


def q8(x, mant, emin):
    """round to an fp8-like format: `mant` mantissa bits, smallest normal 2^emin (subnormals below), after a per-tensor power-of-two scale
    that puts the largest magnitude just under the format's maximum (e4m3: 448 -> scale to <= 256; e5m2: 57344 -> <= 32768)"""
    top = 256.0 if mant == 3 else 32768.0
    amax = np.abs(x).max()
    if amax == 0:
        return x
    s = 2.0 ** np.floor(np.log2(top / amax))
    y = x * s
    m, e = np.frexp(y)                                   # y = m * 2^e, 0.5 <= |m| < 1
    e = np.maximum(e, emin + 1)                          # subnormal range: fixed quantum
    q = 2.0 ** (e - (mant + 1))
    return np.round(y / q) * q / s


def product(A, Wm, mode):
    """(rows, K) @ (K, O) with the operand arithmetic of `mode`; accumulation in fp32 as the MFMA does"""
    mm = lambda a, b: (a.astype(np.float32) @ b.astype(np.float32)).astype(np.float64)
    if mode == "f64":
        return A @ Wm
    if mode == "f32":
        return mm(A, Wm)
    Ah, Wh = f16(A), f16(Wm)
    Al, Wl = f16(A - Ah), f16(Wm - Wh)
    main = mm(Ah, Wh)
    if mode == "fp16x2 (3 products, shipped)":
        return main + mm(Ah, Wl) + mm(Al, Wh)
    if mode == "2 products (w_l dropped)":
        return main + mm(Al, Wh)
    if mode == "1 product":
        return main
    mant, emin = (3, -6) if "e4m3" in mode else (2, -14)
    q = lambda v: q8(v, mant, emin)
    return main + mm(q(Ah), q(Wl)) + mm(q(Al), q(Wh))


def partI(x, sd, N, mode):
    p = "PartI_net."

    def gather(v):                                       # (B,C,60) -> (B*60, C*13), column = c*13 + k
        B, C, _ = v.shape
        return v[:, :, N].transpose(0, 2, 1, 3).reshape(B * G, C * NTAP)

    def conv(v, pre):
        w = sd[pre + ".weight"].astype(np.float64)       # (O,C,1,13)
        O, C = w.shape[0], w.shape[1]
        y = product(gather(v), w.reshape(O, C * NTAP).T, mode) + sd[pre + ".bias"].astype(np.float64)
        return y.reshape(v.shape[0], G, O).transpose(0, 2, 1)

    def bnrelu(v, pre):
        s = sd[pre + ".weight"].astype(np.float64) / np.sqrt(sd[pre + ".running_var"].astype(np.float64) + 1e-5)
        t = sd[pre + ".bias"].astype(np.float64) - sd[pre + ".running_mean"].astype(np.float64) * s
        return np.maximum(v * s[None, :, None] + t[None, :, None], 0.0)

    h0 = conv(x, p + "Conv_in.0")
    r = p + "SO3_Conv_layers.0."
    m = conv(bnrelu(h0, r + "comb_layer_in.0"), r + "comb_layer_in.2")
    h2 = conv(bnrelu(m, r + "comb_layer_out.0"), r + "comb_layer_out.2") + h0
    y = conv(bnrelu(h2, p + "Conv_out.comb_layer.0"), p + "Conv_out.comb_layer.2")
    eqv = y + x
    return eqv / np.maximum(np.sqrt((eqv * eqv).sum(1, keepdims=True)), 1e-4)


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    N = default_tables().N.astype(np.int64)
    sd = W.synth_state_dict(W.PARTI_SPEC, 7)
    x = synth.unit_features(B, seed=10).astype(np.float64)
    ref = partI(x, sd, N, "f64")
    print("| arithmetic of every product | MFMA time vs shipped | worst relative error of eqv vs fp64 |")
    print("|---|---|---|")
    for mode, cost in (("f32", "-"), ("fp16x2 (3 products, shipped)", "1"), ("main fp16 + corrections in fp8 e4m3", "2/3"),
                       ("main fp16 + corrections in fp8 e5m2", "2/3"), ("2 products (w_l dropped)", "2/3"), ("1 product", "1/3")):
        got = partI(x, sd, N, mode)
        print(f"| {mode} | {cost} | {np.abs(got - ref).max() / np.abs(ref).max():.2e} |")


if __name__ == "__main__":
    main()
