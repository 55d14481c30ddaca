"""Dev tool (CPU): gate (a) of the heads' FP6 scheme - the oracle's head chain under each candidate arithmetic (bf16x3; fp16 main
term + block-scaled FP6 cross terms on the first layer / on every layer / with refined weights) against the float64-anchored
bounds of tests/test_gpu_model.py.  Raw output of the round: docs/experiments/r5_heads_mx_numerics.txt.
    python tests/tools/eval_head_mx.py /tmp/e2e_2_448_800.pt [variant names ...]"""
import sys, time, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
import torch.nn.functional as F
from oracle import model_ref, mx_emul
from tests.golden import cases
torch.set_num_threads(8)
d = torch.load(sys.argv[1])
feat32, r32, r64, hm = d["feat32"], d["r32"], d["r64"], d["hm"]
sd = cases.tuned_state_dict(radar=True, seed=0)
heads, head_conv = model_ref.head_spec(True)
hp = "detectHead_0"

def mm_bf16x3(w, x):
    def sp(v):
        hi = v.to(torch.bfloat16).float(); lo = (v - hi).to(torch.bfloat16).float(); return hi.double(), lo.double()
    wh, wl = sp(w); xh, xl = sp(x)
    return ((wh @ xh).float() + ((wh @ xl).float() + (wl @ xh).float()))

def mm_mx(w, x, refine_w=False, own_lo=True):
    """(N, K) x (K, P) with oracle/mx_emul.py's quantiser (numpy): fp16 main term + FP6 cross terms; own_lo: the lo block takes
    its exponent from its own maximum (what the kernels do) instead of the hi block's exponent - 11"""
    import numpy as np
    s = mx_emul.weight_scale_exp(w)
    wh, wl = mx_emul.split_f16((w.double() * 2.0 ** s).float().numpy())
    xh, xl = mx_emul.split_f16((x.float() * 16.0).numpy().T)             # (P, K): blocks of 32 along K
    q = lambda v: mx_emul.quant_blocks(v)[2]
    wh6, wl6 = q(wh), q(wl)
    if refine_w:
        wh6 = wh6 + q(wh - wh6)
        wl6 = wl6 + q(wl - wl6)
    xh6 = q(xh)
    if own_lo:
        xl6 = q(xl)
    else:
        codes, e, _ = mx_emul.quant_blocks(xh)
        b_ = xl.astype(np.float64).reshape(xl.shape[0], -1, 32)
        sc = np.ldexp(1.0, e - 11)[..., None]
        xl6 = (mx_emul.e2m3_values(mx_emul.e2m3_codes(b_ / sc)) * sc).reshape(xl.shape)
    main = wh.astype(np.float64) @ xh.astype(np.float64).T
    cross = wh6 @ xl6.T + wl6 @ xh6.T
    return torch.from_numpy(((main.astype(np.float32) + cross.astype(np.float32)) * np.float32(2.0 ** -(s + 4))))

def conv(x, weight, bias, pad, mm):
    B, C, H, W = x.shape
    co, ci, kh, kw = weight.shape
    cp = (ci + 31) // 32 * 32
    cols = F.unfold(x.float(), (kh, kw), padding=pad).view(B, C, kh * kw, H * W)
    cols = F.pad(cols.permute(0, 2, 1, 3), (0, 0, 0, cp - ci)).reshape(B, kh * kw * cp, H * W)
    w2 = F.pad(weight.float().permute(0, 2, 3, 1), (0, cp - ci)).reshape(co, kh * kw * cp)
    out = torch.stack([mm(w2, cols[b]) for b in range(B)], 0).view(B, co, H, W)
    return out + bias.float().view(1, -1, 1, 1)

def head(p, x, n_hidden, mms):
    x = torch.relu(conv(x, sd[p + ".0.weight"], sd[p + ".0.bias"], 1, mms[0]))
    idx = 2
    for _ in range(n_hidden - 1):
        x = torch.relu(conv(x, sd[f"{p}.{idx}.weight"], sd[f"{p}.{idx}.bias"], 0, mms[1]))
        idx += 2
    return conv(x, sd[f"{p}.{idx}.weight"], sd[f"{p}.{idx}.bias"], 0, mms[2])

mx = lambda w, x: mm_mx(w, x, own_lo=False)
mxo = lambda w, x: mm_mx(w, x, own_lo=True)
mxr = lambda w, x: mm_mx(w, x, refine_w=True, own_lo=True)
variants = {
  "bf16x3 everywhere (shipped)": (mm_bf16x3, mm_bf16x3, mm_bf16x3),
  "mx first, bf16x3 rest": (mxo, mm_bf16x3, mm_bf16x3),
  "mx first+hidden, bf16x3 out": (mxo, mxo, mm_bf16x3),
  "mx all, own lo max": (mxo, mxo, mxo),
  "mx all, weights refined (2.0 passes)": (mxr, mxr, mxr),
  "mxr first, bf16x3 rest": (mxr, mm_bf16x3, mm_bf16x3),
  "mx first, mxr hidden, bf16x3 out": (mxo, mxr, mm_bf16x3),
  "mxr first, mxr hidden, bf16x3 out": (mxr, mxr, mm_bf16x3),
}
sel = sys.argv[2:] or list(variants)
with torch.no_grad():
    sec = torch.cat([feat32, hm.float()], 1)
    for name in sel:
        mms = variants[name]
        t0 = time.time(); y = {}
        for h in heads:
            src = sec if h in model_ref.SECONDARY_HEADS else feat32
            y[h] = head(f"{hp}.{h}", src, len(head_conv[h]), mms)
        y["heatmap"] = torch.clamp(torch.sigmoid(y["heatmap"]), min=1e-4, max=1 - 1e-4)
        y["depthMap"] = y["depth2"]
        y["depth"] = model_ref.sigmoid_depth(y["depth"]); y["depth2"] = model_ref.sigmoid_depth(y["depth2"])
        print(f"== {name}  ({time.time()-t0:.0f}s)")
        for k, t in r64.items():
            if k == "calib" or k not in y: continue
            g, c = y[k].double(), r32[k].double()
            scale = float(t.abs().max()) + 1e-300
            rms = float(t.pow(2).mean().sqrt()) + 1e-300
            e_gpu, e_cpu = float((g - t).abs().max()) / scale, float((c - t).abs().max()) / scale
            r_gpu, r_cpu = float((g - t).pow(2).mean().sqrt()) / rms, float((c - t).pow(2).mean().sqrt()) / rms
            g1 = r_gpu <= 1.25 * r_cpu + 2e-6; g2 = e_gpu <= 2.0 * e_cpu + 2e-5
            print(f"{k:>16s}: rms {r_gpu:.2e} vs fp32 {r_cpu:.2e} ratio {r_gpu/r_cpu:.3f} | max {e_gpu:.2e} vs {e_cpu:.2e} {'ok' if g1 and g2 else 'FAIL'}", flush=True)
