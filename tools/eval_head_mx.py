import sys, time, torch
sys.path.insert(0, '/root/repo')
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

def mm_mx(w, x, refine_w=False, own_lo=False):
    s = mx_emul.weight_scale_exp(w)
    wh, wl = mx_emul.split_f16(w.double().mul(2.0 ** s).float())
    xh, xl = mx_emul.split_f16(x.float() * 16.0)
    q = mx_emul.q6_blocks
    wh6, wl6 = q(wh, 1), q(wl, 1)
    if refine_w:
        wh6 = wh6 + q(wh - wh6, 1); wl6 = wl6 + q(wl - wl6, 1)
    xb = xh.movedim(0, -1).reshape(x.shape[1], -1, 32).double()
    e_hi = mx_emul.block_exponent(xb.abs().amax(-1, keepdim=True))
    xh6 = q(xh, 0, exponent=e_hi)
    xl6 = q(xl, 0, exponent=None if own_lo else e_hi - 11)
    main = (wh.double() @ xh.double()).float()
    cross = (wh6.double() @ xl6.double() + wl6.double() @ xh6.double()).float()
    return (main + cross) * (2.0 ** -(s + 4))

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

mx = lambda w, x: mm_mx(w, x)
mxo = lambda w, x: mm_mx(w, x, own_lo=True)
mxr = lambda w, x: mm_mx(w, x, refine_w=True, own_lo=True)
variants = {
  "bf16x3 everywhere (shipped)": (mm_bf16x3, mm_bf16x3, mm_bf16x3),
  "mx first, bf16x3 rest": (mxo, mm_bf16x3, mm_bf16x3),
  "mx first+hidden, bf16x3 out": (mxo, mxo, mm_bf16x3),
  "mx all, own lo max": (mxo, mxo, mxo),
  "mx all, weights refined (2.0 passes)": (mxr, mxr, mxr),
  "mxr first, bf16x3 rest": (mxr, mm_bf16x3, mm_bf16x3),
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
