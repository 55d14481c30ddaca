"""GPU: the steps around the model as they compose - fused decode + postProcess, nuScenes result
serialisation (vs the reference-generated fixture), the Detector.run-shaped chain
(uint8 frames + raw radar sweeps -> final boxes) against the oracle chain, and the RCCL all-gather."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import (decode_ref, model_ref, pillar_ref, postprocess_ref, preprocess_ref, radar_ref,
                    serialize_ref)
from tests.golden import cases
from tests.golden import cases_dataset as cd

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


# ------------------------------------------------------------------------- decode + postProcess fused
@pytest.mark.parametrize("seed,norm2d", [(0, False), (2, True)])
def test_decode_post_fused_equals_separate_launches(dev, seed, norm2d):
    from centerfusiondetect3d_amd import decode_packed, decode_post_packed, post_process_packed
    from centerfusiondetect3d_amd.postprocess import inverse_affine_device
    out = cases.decode_case(seed, radar=True)
    out["depth2"] = out["depth2"].abs() * 20 + 2
    out["dimension"] = out["dimension"].abs() + 0.1
    out["dimension"][:, 1] -= 0.4                                  # some non-positive dimensions
    calibs = cases.model_inputs(2, 448, 800, seed=0)[2].to(dev)
    center, scale = np.array([800.0, 450.0], np.float32), 1600.0
    d1 = {k: v.to(dev) for k, v in out.items()}
    d2 = {k: v.to(dev) for k, v in out.items()}
    det, _ = decode_packed([d1], (112, 200), 100, norm2d)
    post_sep = post_process_packed(det, calibs, center, scale, 112, 200)
    tinv = inverse_affine_device(center, scale, (200, 112), dev)
    post, det2 = decode_post_packed([d2], calibs, tinv, (112, 200), 100, norm2d, want_det=True)
    assert torch.equal(det2, det)
    assert torch.equal(post, post_sep)                            # bit for bit (NaN-free inputs)
    post_only = decode_post_packed([{k: v.to(dev) for k, v in out.items()}], calibs, tinv, (112, 200), 100, norm2d)
    assert torch.equal(post_only, post)


# -------------------------------------------------------------------------------- serialisation
def _post_rows_from_case(case, K):
    """Build (B,K,54) post rows whose fields carry the fixture's per-detection values; unused slots get a
    non-positive dimension so the merge filter drops them."""
    ids = list(case["images"].keys())
    B = len(ids)
    post = np.zeros((B, K, 54), np.float32)
    post[:, :, 10:13] = -1.0
    for b, iid in enumerate(ids):
        for j, d in enumerate(case["results"].get(iid, [])):
            r = post[b, j]
            r[0], r[1] = d["score"], d["class"]
            r[10:13] = d["dimension"]
            r[15:23] = d["nuscenes_att"]
            r[23:26] = d["velocity"]
            r[26:29] = d["location"]
            r[29] = d["yaw"]
    return ids, post


def test_serialize_matches_reference_golden(dev):
    from centerfusiondetect3d_amd import convert_eval_format, NuScenesResults
    name, case = cd.serialize_cases()[0]
    g = np.load(os.path.join(GOLDEN, f"serialize_{name}.npz"))
    ids, post = _post_rows_from_case(case, K=100)
    have = [i for i in ids if i in case["results"]]                # image 99 has no results: never handed over
    sel = [ids.index(i) for i in have]
    infos = [case["images"][i] for i in have]
    ret = convert_eval_format(torch.from_numpy(post[sel]).to(dev), infos, use_radar=True)
    assert ret["meta"] == {"use_camera": True, "use_lidar": False, "use_radar": True, "use_map": False,
                           "use_external": False}
    assert sorted(ret["results"].keys()) == [str(t) for t in g["tokens"]]
    for t, rows in ret["results"].items():
        assert len(rows) == len(g[f"{t}_score"])
        if not rows:
            continue
        assert np.array_equal(np.array([r["translation"] for r in rows]), g[f"{t}_translation"])   # fp32 exact
        assert np.array_equal(np.array([r["size"] for r in rows]), g[f"{t}_size"])
        assert np.array_equal(np.array([r["velocity"] for r in rows]), g[f"{t}_velocity"])
        assert np.array_equal(np.array([r["detection_score"] for r in rows]), g[f"{t}_score"])     # order + top-500 cut
        assert [r["detection_name"] for r in rows] == list(g[f"{t}_name"])
        assert [r["attribute_name"] for r in rows] == list(g[f"{t}_attribute"])
        assert np.array_equal(np.array([r["sensor_id"] for r in rows]), g[f"{t}_sensor_id"])
        for r in rows:
            assert r["tracking_name"] == r["detection_name"] and r["tracking_id"] == 1 and r["det_id"] == -1
            assert r["sample_token"] == t and r["tracking_score"] == r["detection_score"]
    # accumulated in two batches: same result
    acc = NuScenesResults(use_radar=True)
    h = len(sel) // 2
    acc.add(torch.from_numpy(post[sel[:h]]).to(dev), infos[:h])
    acc.add(torch.from_numpy(post[sel[h:]]).to(dev), infos[h:])
    assert acc.convert_eval_format() == ret


def test_serialize_rotation_and_filter(dev):
    from centerfusiondetect3d_amd import convert_eval_format
    rs = np.random.RandomState(3)
    B, K = 3, 20
    post = np.zeros((B, K, 54), np.float32)
    post[..., 0] = rs.uniform(0.05, 0.9, (B, K))
    post[..., 1] = rs.randint(1, 11, (B, K))
    post[..., 10:13] = rs.uniform(0.5, 4, (B, K, 3))
    post[0, 3, 11] = 0.0                                           # non-positive dimension -> dropped
    post[1, 5, 0] = -1.0                                           # score -1 -> dropped
    post[..., 15:23] = rs.standard_normal((B, K, 8))
    post[..., 23:29] = rs.standard_normal((B, K, 6)) * 10
    post[..., 29] = rs.uniform(-np.pi, np.pi, (B, K))
    infos = []
    for b in range(B):
        q1, q2 = rs.standard_normal(4), rs.standard_normal(4)
        infos.append(dict(sample_token=f"s{b % 2}", trans_matrix=np.eye(4).tolist(),
                          velocity_trans_matrix=np.eye(4).tolist(), sensor_id=b + 1,
                          cs_record_rot=(q1 / np.linalg.norm(q1)).tolist(),
                          pose_record_rot=(q2 / np.linalg.norm(q2)).tolist()))
    ret = convert_eval_format(torch.from_numpy(post).to(dev), infos)
    n = sum(len(v) for v in ret["results"].values())
    assert n == B * K - 2
    # oracle for every surviving row
    for t, rows in ret["results"].items():
        frames = [b for b in range(B) if infos[b]["sample_token"] == t]
        exp = []
        for b in frames:
            for j in range(K):
                p = post[b, j]
                if p[0] > -1 and (p[10:13] > 0).all():
                    item = {"class": p[1], "score": p[0], "dimension": p[10:13], "location": p[26:29],
                            "nuscenes_att": p[15:23], "velocity": p[23:26]}
                    e = serialize_ref.eval_format_item(item, np.eye(4, dtype=np.float32), np.eye(4, dtype=np.float32))
                    e["rotation"] = serialize_ref.box_rotation(p[29], infos[b]["cs_record_rot"], infos[b]["pose_record_rot"])
                    exp.append(e)
        order = sorted((-e["score"], i) for i, e in enumerate(exp))
        exp = [exp[i] for _, i in order]
        assert len(exp) == len(rows)
        for e, r in zip(exp, rows):
            assert np.array_equal(e["translation"], np.array(r["translation"]))
            np.testing.assert_allclose(r["rotation"], e["rotation"], rtol=0, atol=1e-15)
            assert serialize_ref.ID_TO_ATTRIBUTE[e["attribute"]] == r["attribute_name"]


# ------------------------------------------------------------------ Detector.run-shaped composite
def test_detector_chain_matches_oracle_chain(dev):
    """uint8 frames + raw radar sweeps -> final boxes, every step on the device, against the oracle chain
    preprocess_ref -> radar_ref -> pillar_ref -> model_ref -> decode_ref -> postprocess_ref."""
    from centerfusiondetect3d_amd import Detector, centerfusion_middle_config, getModel
    inH, inW = 128, 224                                            # network input; maps 32 x 56
    Hs, Ws = 450, 800                                              # camera frame (uint8, HWC)
    B = 2
    rs = np.random.RandomState(7)
    frames = [rs.randint(0, 256, (Hs, Ws, 3)).astype(np.uint8) for _ in range(B)]
    K3 = cd.NUSC_K * 0.5
    K3[2, 2] = 1.0
    calib = np.concatenate([K3, np.zeros((3, 1))], axis=1)
    sweeps = []
    for b in range(B):
        pc = cd._sweep(np.random.RandomState(50 + b), 150, max_z=70.0, lateral=0.7)
        pc[2, ::13] *= -1
        sweeps.append(pc)
    infos = [dict(calib=calib.tolist(), camera_intrinsic=K3.tolist(), width=Ws, height=Hs) for _ in range(B)]
    cfg = centerfusion_middle_config((inH, inW))
    sd = cases.tuned_state_dict(radar=True, seed=0)
    model = getModel(cfg)
    model.load_state_dict(sd)
    det = Detector(cfg, model=model, device=dev)
    ret = det.run(frames, infos, sweeps)

    # ---- oracle chain
    center, scale = np.array([Ws / 2.0, Hs / 2.0], np.float32), float(max(Hs, Ws))
    m_in = pillar_ref.affine_transform_matrix(center, scale, (inW, inH))
    m_out = pillar_ref.affine_transform_matrix(center, scale, (inW // 4, inH // 4))
    x = preprocess_ref.pre_process_images(frames, m_in, (inH, inW), det.mean, det.std)
    pc_dep = []
    for pc in sweeps:
        p2, p3 = radar_ref.ingest_radar(pc, K3, (Ws, Hs), 60.0, 0.0)
        pc_dep.append(pillar_ref.process_point_cloud(p2, p3, calib, m_out, (inH // 4, inW // 4))[2])
    pc_dep = torch.from_numpy(np.stack(pc_dep))
    assert int((pc_dep[:, 0] != 0).sum()) > 50
    calibs = torch.from_numpy(np.stack([calib.astype(np.float32)] * B))
    with torch.no_grad():
        y = model_ref.forward(sd, torch.from_numpy(x), pc_dep=pc_dep, calib=calibs)
    dets = decode_ref.fusion_decode(y, (inH // 4, inW // 4), 100)
    ref = postprocess_ref.post_process(dets, center, scale, inH // 4, inW // 4, calibs)

    # ---- discrete path identical, floating point within the model tolerance
    got = ret["detects"]
    assert torch.equal(ret["outputs"][0]["pc_hm_in"].cpu(), pc_dep[:, :1])            # pre-processing: bit-exact pc_dep
    assert np.array_equal(got["classIds"].numpy(), ref["classIds"].numpy())
    for k in ("scores", "centers", "bboxes", "depth", "alpha", "dimension", "locations", "yaws", "velocity",
              "bboxes3d", "nuscenes_att"):
        a, b = got[k].double().numpy(), ref[k].double().numpy()
        scale_k = np.abs(b).max() + 1e-12
        if k in ("alpha", "yaws"):                                 # angles wrap: compare on the circle
            dlt = np.abs(np.angle(np.exp(1j * (a - b))))
        else:
            dlt = np.abs(a - b)
        assert (dlt <= 2e-3 * np.abs(b) + 1e-3 * scale_k).all(), (k, float(dlt.max() / scale_k))
    boxes = ret["predictBoxes"]
    assert len(boxes) == B and all(len(bx) > 0 for bx in boxes)
    assert set(boxes[0][0]) >= {"class", "score", "dimension", "location", "yaw", "bboxes", "bboxes3d",
                                "nuscenes_att", "velocity"}


# ------------------------------------------------------------------------------------- RCCL path
def test_rccl_all_gather_runs_on_device(dev):
    """The collective of SURVEY §8(e) executed through RCCL (backend "nccl") on this GPU: world size 1 is all
    a one-GPU box offers, but it is the same all_gather_into_tensor call, on a side stream, that N ranks issue."""
    import torch.distributed as dist
    from centerfusiondetect3d_amd.distributed import DetectionGatherer, gather_detections
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        det = torch.randn(16, 100, 54, device=dev)
        out = gather_detections(det, force_collective=True)
        assert out.shape == det.shape and torch.equal(out, det) and out.data_ptr() != det.data_ptr()
        # overlapped form: the gather of step i runs on its own stream while step i+1 computes
        g = DetectionGatherer(device=dev, force_collective=True)
        pending = []
        for i in range(4):
            d = det + float(i)
            pending.append((g.submit(d), d.clone()))
            _ = (torch.randn(1024, 1024, device=dev) @ torch.randn(1024, 1024, device=dev))   # next step's compute
        for h, d in pending:
            assert torch.equal(h.wait(), d)
    finally:
        if created:
            dist.destroy_process_group()


def test_trunk_streams_overlap_after_rccl_init(dev):
    """What a rank of an N-GPU run does (bench.py under torchrun): the RCCL process group exists - and has taken its
    streams - BEFORE the first model.  The two trunk sub-batches of `model.streams = 2` must still run concurrently
    (the side-stream pair is chosen by a probe of spin kernels, model._pick_streams, not by creation order) and the
    result must be the single-stream one bit for bit."""
    import torch.distributed as dist
    from centerfusiondetect3d_amd import getModel, centerfusion_middle_config
    from centerfusiondetect3d_amd import model as M
    from centerfusiondetect3d_amd.distributed import gather_detections
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29534")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        gather_detections(torch.randn(16, 100, 54, device=dev), force_collective=True)   # the communicator is live
        M._SIDE_STREAMS.clear()                          # as in a fresh process: nothing probed yet
        H, W, B = 448, 800, 16
        m = getModel(centerfusion_middle_config((H, W)))
        m.load_state_dict(cases.tuned_state_dict(radar=True, seed=0), strict=True)
        m = m.to(dev).eval()
        x, pc_dep, calib = cases.model_inputs(B, H, W, seed=47, radar=True, n_points=(50, 200))
        xd, pd, cdv = x.to(dev), pc_dep.to(dev), calib.to(dev)
        with torch.no_grad():
            m.streams = 1
            one = m(xd, pc_dep=pd, calib=cdv)[0]
            m.streams, m.record_spans = 2, True
            for _ in range(3):
                two = m(xd, pc_dep=pd, calib=cdv)[0]
            spans, overlap = m.trunk_overlap()
        for k in one:
            assert torch.equal(one[k], two[k]), k
        probe = M._PROBE_LOG[-1]
        print(f"trunk spans {spans[0]:.2f} / {spans[1]:.2f} ms, overlap {overlap:.2f} ms; probe {probe}")
        assert not probe[3], f"no concurrent stream pair found: {probe}"
        assert overlap >= 0.6 * min(spans), (spans, overlap, probe)
    finally:
        if created:
            dist.destroy_process_group()


def test_bench_line_carries_every_ranks_diagnostics(dev):
    """bench.py as a rank of an N-GPU run, on this one GPU (`--force-collective`: RCCL group + the per-step all-gather at
    world size 1): the line's `ranks` block holds this rank's step time, host-enqueue time per step (before any wait)
    and the compute stream's wait on the exchange (HIP events either side of work.wait())."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--force-collective", "--steps", "4", "--warmup",
                        "2", "--no-cpu-baseline", "--batch", "2", "--height", "128", "--width", "160"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and len(d["ranks"]) == 1
    r = d["ranks"][0]
    assert r["rank"] == 0 and 0 < r["host_enqueue_ms"] and 0 < r["step_ms"] == pytest.approx(d["ms_per_step"], rel=1e-3)
    assert 0 <= r["gather_wait_ms"] < r["step_ms"] and 0 <= r["gather_wait_host_ms"] < r["step_ms"]
    assert d["roofline"]["launches_timed"] == 8


def test_stream_probe_tells_shared_queue_from_concurrent(dev):
    """The probe's two answers, on this device: a stream against ITSELF is the serialised case (two spins take 2x),
    and some pair among a handful of fresh streams is concurrent (1x)."""
    from centerfusiondetect3d_amd import _lib
    from centerfusiondetect3d_amd import model as M
    lib = _lib.load()
    s = [torch.cuda.Stream(dev) for _ in range(4)]
    torch.cuda.synchronize()
    same, ms_same = M._concurrent(lib, s[0], s[0])
    assert not same and ms_same > 1.8e-3 * M._PROBE_US, ms_same
    found = [(i, j) for i in range(4) for j in range(i + 1, 4) if M._concurrent(lib, s[i], s[j])[0]]
    assert found, "no two of four fresh streams run concurrently"
    assert lib.cf_spin_us(0, None) != 0 and lib.cf_spin_us(200000, None) != 0        # bounded by contract


def test_run_pipelined_equals_run(dev):
    """Detector.run_pipelined (batch i+1's copy / warp / radar ingest on a feed stream beside batch i's forward, results
    fetched one batch behind) yields, in order, exactly what Detector.run returns batch by batch - also with batches of
    different sizes and with the host-side merge on."""
    from centerfusiondetect3d_amd import Detector, centerfusion_middle_config
    H, W = 128, 160
    cfg = centerfusion_middle_config((H, W))
    det = Detector(cfg, device=dev)
    det.model.load_state_dict(cases.tuned_state_dict(radar=True, seed=0), strict=True)
    calib = np.concatenate([cd.NUSC_K, np.zeros((3, 1))], axis=1)
    batches = []
    for i, B in enumerate((2, 3, 1, 2)):
        rs = np.random.RandomState(40 + i)
        frames = torch.from_numpy(rs.randint(0, 256, (B, 900, 1600, 3)).astype(np.uint8)).pin_memory()
        infos = [dict(calib=calib.tolist(), camera_intrinsic=cd.NUSC_K.tolist(), width=1600, height=900)] * B
        sweeps = [cd._sweep(np.random.RandomState(400 + 10 * i + b), 60 + 20 * b) for b in range(B)]
        batches.append((frames, infos, sweeps))
    with torch.no_grad():
        ref = [det.run(*b) for b in batches]
        got = list(det.run_pipelined(iter(batches)))
        assert len(got) == len(ref)
        for r, g in zip(ref, got):
            assert torch.equal(r["post"], g["post"])
            for k in r["outputs"][0]:
                assert torch.equal(r["outputs"][0][k], g["outputs"][0][k]), k
            assert len(r["predictBoxes"]) == len(g["predictBoxes"])
            for k in r["detects"]:
                assert torch.equal(r["detects"][k], g["detects"][k])
        assert list(det.run_pipelined(iter([]))) == []
        one = list(det.run_pipelined(iter(batches[:1]), merge=False))
        assert len(one) == 1 and torch.equal(one[0]["post"], ref[0]["post"])


def test_detector_loads_the_configured_checkpoint(dev, tmp_path):
    """detector.py:28-33: `Detector(config)` with `MODEL.LOAD_DIR` set (configs/Centerfusion_Middle.yaml:43) runs the
    CHECKPOINT's weights - here a DataParallel-prefixed file under the original CenterFusion release's parameter names
    (tests/golden/legacy_keys.npz, from the reference's toggleWeightName) - bit for bit what a module filled by
    load_state_dict gives; and not what a random-init module gives."""
    from centerfusiondetect3d_amd import Detector, centerfusion_middle_config, getModel
    H, W = 128, 160
    g = np.load(os.path.join(GOLDEN, "legacy_keys.npz"))
    new, old = list(g["centerfusion_new"]), list(g["centerfusion_old"])
    sd = cases.tuned_state_dict(radar=True, seed=0)
    assert set(sd.keys()) == set(new)
    legacy_name = dict(zip(new, old))
    assert sum(k != v for k, v in legacy_name.items()) > 150
    path = tmp_path / "centerfusion_e60.pth"
    torch.save({"epoch": 60, "state_dict": {"module." + legacy_name[k]: v for k, v in sd.items()}}, path)
    cfg = centerfusion_middle_config((H, W))
    cfg.MODEL.LOAD_DIR = str(path)
    det = Detector(cfg)                                              # the reference's call, nothing else
    assert next(det.model.parameters()).device.type == "cuda" and not det.model.training
    model = getModel(centerfusion_middle_config((H, W)))
    model.load_state_dict(sd, strict=True)
    det_ref = Detector(centerfusion_middle_config((H, W)), model=model, device=dev)
    det_rand = Detector(centerfusion_middle_config((H, W)), device=dev)
    calib = np.concatenate([cd.NUSC_K, np.zeros((3, 1))], axis=1)
    B = 2
    frames = torch.from_numpy(np.random.RandomState(11).randint(0, 256, (B, 900, 1600, 3)).astype(np.uint8))
    infos = [dict(calib=calib.tolist(), camera_intrinsic=cd.NUSC_K.tolist(), width=1600, height=900)] * B
    sweeps = [cd._sweep(np.random.RandomState(110 + b), 90) for b in range(B)]
    with torch.no_grad():
        got, ref, rnd = (d.run(frames, infos, sweeps, merge=False) for d in (det, det_ref, det_rand))
    assert torch.equal(got["post"], ref["post"])
    for k in ref["outputs"][0]:
        assert torch.equal(got["outputs"][0][k], ref["outputs"][0][k]), k
    assert not torch.equal(got["post"], rnd["post"])


def test_run_stage_times_carry_the_reference_keys(dev):
    """Detector.run(stage_times=True): the per-stage seconds of the reference's `@return_time` hook under the reference's own
    keys (detector.py:140-155), measured with HIP events - same results as an untimed run, stages positive and adding up to
    no more than the total."""
    from centerfusiondetect3d_amd import Detector, centerfusion_middle_config
    H, W = 128, 160
    det = Detector(centerfusion_middle_config((H, W)), device=dev)
    det.model.load_state_dict(cases.tuned_state_dict(radar=True, seed=0), strict=True)
    calib = np.concatenate([cd.NUSC_K, np.zeros((3, 1))], axis=1)
    B = 2
    frames = torch.from_numpy(np.random.RandomState(7).randint(0, 256, (B, 900, 1600, 3)).astype(np.uint8))
    infos = [dict(calib=calib.tolist(), camera_intrinsic=cd.NUSC_K.tolist(), width=1600, height=900)] * B
    sweeps = [cd._sweep(np.random.RandomState(70 + b), 80) for b in range(B)]
    with torch.no_grad():
        plain = det.run(frames, infos, sweeps)
        det.run(frames, infos, sweeps, stage_times=True)                  # (plans exist, clocks up)
        timed = det.run(frames, infos, sweeps, stage_times=True)
    assert torch.equal(plain["post"], timed["post"])
    assert not any(k in plain for k in ("net", "tot"))
    for k in ("load", "preprocess", "net", "decode", "postprocess", "merge", "display", "tot"):
        assert isinstance(timed[k], float) and timed[k] >= 0.0, k
    for k in ("preprocess", "net", "decode", "merge"):
        assert 0.0 < timed[k] < 5.0, (k, timed[k])
    assert timed["preprocess"] + timed["net"] + timed["decode"] <= timed["tot"] * 1.05
    assert timed["load"] == timed["display"] == timed["postprocess"] == 0.0
