"""Round 5, the arm of tools/eval_two_pass.py that needs the oracle's quantiser (hence under tests/): fp16 main
term + BOTH cross terms on block-scaled FP6 e2m3 (oracle/mx_emul.py), same layer, same yardstick.
    python tests/tools/eval_two_pass_mx.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import mx_emul


def main_mx():
    """Round 5: fp16 main term + BOTH cross terms on block-scaled FP6 e2m3 (oracle/mx_emul.py's quantiser):
    the same layer, the same yardstick."""
    rng = np.random.default_rng(0)
    K, N, P = 256, 256, 4096
    x = np.maximum(rng.standard_normal((K, P)), 0).astype(np.float32)
    w = (rng.standard_normal((N, K)) * np.sqrt(2.0 / K)).astype(np.float32)
    y64 = w.astype(np.float64) @ x.astype(np.float64)
    scale = np.abs(y64).max()
    wh, wl = mx_emul.split_f16(w * np.float32(2.0 ** 13))
    xh, xl = mx_emul.split_f16(x.T * np.float32(16.0))            # (P, K): blocks of 32 along K
    q = lambda v: mx_emul.quant_blocks(v)[2]
    main_t = wh.astype(np.float64) @ xh.astype(np.float64).T
    for name, cross in (("f16 + FP6 cross terms (1.5 passes)", q(wh) @ q(xl).T + q(wl) @ q(xh).T),
                        ("... weights refined by a 2nd FP6 term (2.0)", (q(wh) + q(wh - q(wh))) @ q(xl).T + (q(wl) + q(wl - q(wl))) @ q(xh).T),
                        ("f16x3 (exact cross terms, 3 passes)", wh.astype(np.float64) @ xl.astype(np.float64).T + wl.astype(np.float64) @ xh.astype(np.float64).T)):
        d = ((main_t + cross) * 2.0 ** -17 - y64) / scale
        print(f"{name:44s} max {np.abs(d).max():.3e}   rms {np.sqrt((d * d).mean()):.3e}")



if __name__ == "__main__":
    main_mx()
