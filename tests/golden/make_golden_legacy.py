"""DEV-ONLY: legacy parameter names for every key of the CenterFusion / CenterNet state_dict, produced by
the REFERENCE's own `toggleWeightName` (/root/reference/src/lib/model/model.py:169-250), and a check that the
reference's `elasticLoadStateDict` loads a legacy-keyed checkpoint into OUR module unchanged.
    python tests/golden/make_golden_legacy.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def main():
    from tests.golden import make_golden
    make_golden._install_inert_modules()
    sys.path[:0] = ["/root/reference/src", "/root/reference/src/lib"]
    from model.model import toggleWeightName, elasticLoadStateDict
    from centerfusiondetect3d_amd import getModel, centerfusion_middle_config, centernet_config
    out = {}
    for tag, cfg in (("centerfusion", centerfusion_middle_config((64, 64))), ("centernet", centernet_config((64, 64)))):
        m = getModel(cfg)
        keys = list(m.state_dict().keys())
        old = [toggleWeightName(k, "old") for k in keys]
        old2 = [toggleWeightName(k, "oldv2") for k in keys]
        for k, a, b in zip(keys, old, old2):
            assert toggleWeightName(a, "new") == k and toggleWeightName(b, "new") == k, (k, a, b)
        out[f"{tag}_new"], out[f"{tag}_old"], out[f"{tag}_oldv2"] = np.array(keys), np.array(old), np.array(old2)
        # the reference's own loader on OUR module, from a legacy-keyed DataParallel checkpoint
        g = torch.Generator().manual_seed(0)
        src = {k: (torch.randn(v.shape, generator=g) if v.is_floating_point() else v.clone())
               for k, v in m.state_dict().items()}
        legacy = {"module." + o: src[k] for k, o in zip(keys, old)}
        m2 = elasticLoadStateDict(getModel(cfg), legacy)
        for k, v in m2.state_dict().items():
            assert torch.equal(v, src[k]), k
        print(f"{tag}: {len(keys)} keys, {sum(a != k for a, k in zip(old, keys))} renamed (v1), "
              f"{sum(a != k for a, k in zip(old2, keys))} renamed (v2); reference elasticLoadStateDict -> our DLASeg ok")
    np.savez_compressed(os.path.join(HERE, "legacy_keys.npz"), **out)


if __name__ == "__main__":
    main()
