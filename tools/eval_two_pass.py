"""Verdict r1 item 4: would a 2-pass split-operand product do for the heads' 1x1 layers (256 -> 256 hidden, 256 -> n_out)?
Pure numpy emulation on a layer of that shape (post-ReLU inputs, Kaiming-scale weights), every variant against float64:
   fp32      : what the reference computes (torch fp32 convolution = fp32 products, fp32 accumulation)
   bf16x3    : wh*xh + wh*xl + wl*xh, exact products, fp32 accumulation      (what cf_head_fused runs)
   bf16x2    : the same without wl*xh (weights rounded to 8 bits)  /  without wh*xl (activations rounded)
   f16x2     : 11-bit halves, one cross term dropped
Prints max-norm and RMS error relative to max|y|.  No GPU, no oracle.  (The fp16 + FP6 arm of round 5 uses the oracle's quantiser:
tests/tools/eval_two_pass_mx.py.)"""
import numpy as np

def split(x, bits):
    """hi = x rounded to `bits` significant bits (round to nearest even via float arithmetic), lo = the next `bits`."""
    m, e = np.frexp(x.astype(np.float64))
    hi = np.ldexp(np.round(m * 2.0 ** bits), e - bits)
    r = x.astype(np.float64) - hi
    m2, e2 = np.frexp(r)
    lo = np.ldexp(np.round(m2 * 2.0 ** bits), e2 - bits)
    return hi, lo

def main():
    rng = np.random.default_rng(0)
    K, N, P = 256, 256, 4096
    x = np.maximum(rng.standard_normal((K, P)), 0).astype(np.float32)           # post-ReLU hidden activations
    w = (rng.standard_normal((N, K)) * np.sqrt(2.0 / K)).astype(np.float32)
    y64 = w.astype(np.float64) @ x.astype(np.float64)
    scale = np.abs(y64).max()
    def rep(name, y):
        d = (y.astype(np.float64) - y64) / scale
        print(f"{name:34s} max {np.abs(d).max():.3e}   rms {np.sqrt((d * d).mean()):.3e}")
    acc = np.zeros((N, P), np.float32)                                          # fp32 products, sequential fp32 accumulation
    for k in range(K):
        acc += w[:, k:k + 1] * x[k:k + 1, :]
    rep("fp32 FMA chain (reference-like)", acc)
    rep("fp32 BLAS (numpy sgemm)", w @ x)
    for bits, tag in ((8, "bf16"), (11, "f16")):
        wh, wl = split(w, bits)
        xh, xl = split(x, bits)
        f32 = lambda a: a.astype(np.float32).astype(np.float64)                 # one fp32 rounding of each pass's sum
        rep(f"{tag}x3  wh*xh + wh*xl + wl*xh", f32(wh @ xh) + f32(wh @ xl) + f32(wl @ xh))
        rep(f"{tag}x2  drop wl*xh", f32(wh @ xh) + f32(wh @ xl))
        rep(f"{tag}x2  drop wh*xl", f32(wh @ xh) + f32(wl @ xh))
        rep(f"{tag}x1", f32(wh @ xh))

if __name__ == "__main__":
    main()
